set -x
cd $GRAFT_REPO_ROOT
for a in "" "upper" "int8 upper" "int8" "mix"; do
  timeout 600 python tools/ab_bench.py viprs_amd/lib/libviprs_hip.so build/libviprs_hip_tcpl.so $a
done > gpurun_out/r04d_ab.txt 2>&1
VIPRS_HIP_LIB=build/libviprs_hip_tcpl.so timeout 900 python -m pytest tests/test_gpu_farfield.py tests/test_gpu_fullsize_parity.py tests/test_gpu_edge_cases.py tests/test_fit.py -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04d_pytest_tcpl.txt
VIPRS_HIP_LIB=build/libviprs_hip_tcpltrace.so timeout 300 python tools/sweep_trace.py > gpurun_out/r04d_trace_sym.txt 2>&1
VIPRS_HIP_LIB=build/libviprs_hip_tcpl.so timeout 300 python bench.py --config cfg3max --no-secondary --cpu-seconds 0 > gpurun_out/r04d_cfg3max.json 2>&1
timeout 300 python -m pytest "tests/test_gpu_fast_math.py::test_fast_math_well_conditioned" -x -q 2>&1 | tail -30 > gpurun_out/r04d_wc.txt
tail -3 gpurun_out/r04d_pytest_tcpl.txt
