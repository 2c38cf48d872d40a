cd $GRAFT_REPO_ROOT
L="viprs_amd/lib/libviprs_hip.so build/libviprs_hip_tcpl.so build/libviprs_hip_late.so build/libviprs_hip_d8.so build/libviprs_hip_lated8.so"
for a in "" "upper" "int8 upper" "mix" "fast"; do echo "== $a"; timeout 900 python tools/multi_ab.py $L -- $a; done > gpurun_out/r04f_ab.txt 2>&1
VIPRS_HIP_LIB=build/libviprs_hip_lateprof.so timeout 300 python tools/panel_profile.py cfg3 > gpurun_out/r04f_pprof_late.txt 2>&1
