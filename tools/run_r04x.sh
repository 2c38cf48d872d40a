cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_farfield.py tests/test_gpu_fullsize_parity.py tests/test_grid.py -m gpu -x -q -k "grid" 2>&1 | tail -3
for r in 1 2; do for extra in "--model grid" "--model grid --ld-dtype int8"; do timeout 300 python bench.py --no-secondary --cpu-seconds 0 $extra | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$extra', 'kernel %.4f ms/step %.4f'%(d['roofline']['kernel_ms_avg'], d['ms_per_step']))"; done; done
