cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04i_pytest.txt
timeout 900 python bench.py > gpurun_out/r04i_bench.json 2> gpurun_out/r04i_bench.err
for extra in "--model grid" "--model grid --ld-dtype int8" "--model grid --ld-dtype int8 --low-memory" "--model grid --low-memory"; do
  timeout 300 python bench.py --no-secondary --cpu-seconds 0 $extra
done > gpurun_out/r04i_grid.jsonl 2> gpurun_out/r04i_grid.err
tail -3 gpurun_out/r04i_pytest.txt
