cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_float64.py tests/test_golden.py tests/test_fit.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2; do for l in build/libviprs_hip_f64old.so viprs_amd/lib/libviprs_hip.so; do VIPRS_HIP_LIB=$l timeout 300 python bench.py --no-secondary --cpu-seconds 0 --precision float64 --low-memory --ld-dtype int8 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$l'.split('/')[-1], 'kernel %.4f sweep %.4f ms/step %.4f'%(d['roofline']['kernel_ms_avg'], d['roofline']['sweep_ms_avg'], d['ms_per_step']))"; done; done
