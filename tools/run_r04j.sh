cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_farfield.py tests/test_gpu_fast_math.py -m gpu -x -q -k "grid or well" 2>&1 | tail -5 > gpurun_out/r04j_pytest.txt
for extra in "--model grid" "--model grid --ld-dtype int8" "--model grid --ld-dtype int16" "--model grid --ld-dtype int8 --low-memory" "--model grid --low-memory"; do
  timeout 300 python bench.py --no-secondary --cpu-seconds 0 $extra
done > gpurun_out/r04j_grid.jsonl 2> gpurun_out/r04j_grid.err
