set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04a_pytest.txt
timeout 600 python tools/fast_math_error.py > gpurun_out/r04a_fast_err.txt 2>&1
for m in exact fast; do MATH=$m timeout 300 python tools/one_block_bench.py; done > gpurun_out/r04a_oneblock.txt 2>&1
timeout 600 python bench.py > gpurun_out/r04a_bench_exact.json 2> gpurun_out/r04a_bench_exact.err
for extra in "" "--low-memory" "--low-memory --ld-dtype int8" "--ld-dtype int8" "--model mixture" "--model grid" "--config cfg3max"; do
  timeout 300 python bench.py --math fast --no-secondary --cpu-seconds 0 $extra
done > gpurun_out/r04a_bench_fast.jsonl 2> gpurun_out/r04a_bench_fast.err
tail -5 gpurun_out/r04a_pytest.txt
