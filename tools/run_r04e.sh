cd $GRAFT_REPO_ROOT
VIPRS_HIP_LIB=build/libviprs_hip_tcplprof.so timeout 300 python tools/panel_profile.py cfg3 > gpurun_out/r04e_pprof_tcpl.txt 2>&1
