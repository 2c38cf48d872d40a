cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_farfield.py tests/test_gpu_fullsize_parity.py tests/test_gpu_fast_math.py -m gpu -x -q -k "grid or well" 2>&1 | tail -4
for extra in "--model grid --ld-dtype int8" "--model grid --ld-dtype int16"; do timeout 300 python bench.py --no-secondary --cpu-seconds 0 $extra | cut -c1-210; done
