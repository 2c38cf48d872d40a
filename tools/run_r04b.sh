set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fast_math.py -x -q -s 2>&1 | tail -60 > gpurun_out/r04b_fast_tests.txt
timeout 900 python tools/fit_fast_explore.py > gpurun_out/r04b_fit_fast.txt 2>&1
timeout 900 python bench.py > gpurun_out/r04b_bench.json 2> gpurun_out/r04b_bench.err
tail -3 gpurun_out/r04b_fast_tests.txt; tail -3 gpurun_out/r04b_bench.err
