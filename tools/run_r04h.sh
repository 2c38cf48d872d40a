cd $GRAFT_REPO_ROOT
L="build/libviprs_hip_base2.so build/libviprs_hip_base3.so build/libviprs_hip_hf.so build/libviprs_hip_hfspec.so"
for a in "" "int8 upper" "fast"; do echo "== $a"; timeout 900 python tools/multi_ab.py $L -- $a; done > gpurun_out/r04h_ab.txt 2>&1
VIPRS_HIP_LIB=build/libviprs_hip_hfspecprof.so timeout 300 python tools/panel_profile.py cfg3 > gpurun_out/r04h_pprof.txt 2>&1
VIPRS_HIP_LIB=build/libviprs_hip_hfspec.so timeout 900 python -m pytest tests/test_gpu_farfield.py tests/test_gpu_fullsize_parity.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r04h_pytest.txt
