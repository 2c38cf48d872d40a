cd $GRAFT_REPO_ROOT
L="viprs_amd/lib/libviprs_hip.so build/libviprs_hip_base2.so build/libviprs_hip_spec40.so build/libviprs_hip_spec54.so"
for a in "" "int8 upper" "upper" "fast"; do echo "== $a"; timeout 900 python tools/multi_ab.py $L -- $a; done > gpurun_out/r04g_ab.txt 2>&1
VIPRS_HIP_LIB=build/libviprs_hip_spec40prof.so timeout 300 python tools/panel_profile.py cfg3 > gpurun_out/r04g_pprof_spec40.txt 2>&1
VIPRS_HIP_LIB=build/libviprs_hip_spec40.so timeout 900 python -m pytest tests/test_gpu_farfield.py tests/test_gpu_fullsize_parity.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r04g_pytest.txt
