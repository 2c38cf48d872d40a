set -x
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04c_pytest.txt
VIPRS_HIP_LIB=build/libviprs_hip_trace.so timeout 300 python tools/sweep_trace.py > gpurun_out/r04c_trace_sym.txt 2>&1
VIPRS_HIP_LIB=build/libviprs_hip_trace.so timeout 300 python tools/sweep_trace.py upper > gpurun_out/r04c_trace_upper.txt 2>&1
VIPRS_HIP_LIB=build/libviprs_hip_prof.so timeout 300 python tools/panel_profile.py cfg3 > gpurun_out/r04c_pprof_sym.txt 2>&1
for w in 10 20; do for lm in "" "--low-memory"; do timeout 300 python bench.py --model mixture --width $w $lm --no-secondary --cpu-seconds 0; done; done > gpurun_out/r04c_mixwide.jsonl 2> gpurun_out/r04c_mixwide.err
timeout 900 python bench.py > gpurun_out/r04c_bench.json 2> gpurun_out/r04c_bench.err
tail -3 gpurun_out/r04c_pytest.txt
