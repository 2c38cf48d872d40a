cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for l in viprs_amd/lib/libviprs_hip.so build/libviprs_hip_wide.so; do
  for extra in "--config cfg3max" "--config cfg3max --low-memory" "--config cfg3max --math fast"; do
    VIPRS_HIP_LIB=$l timeout 300 python bench.py --no-secondary --cpu-seconds 0 $extra | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$l'.split('/')[-1], '$extra', 'kernel %.4f'%d['roofline']['kernel_ms_avg'])"
  done; done; done
