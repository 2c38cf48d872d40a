for cfg in "" "upper" "int8"; do echo "== $cfg"; for lib in viprs_amd/lib/libviprs_hip.so build/libviprs_hip_nw3.so build/libviprs_hip_nw2.so; do VIPRS_HIP_LIB=$PWD/$lib timeout 200 python - $cfg <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from viprs_amd.plan import DeviceState, LDPlan
from viprs_amd.utils import synthetic as syn
upper = "upper" in sys.argv; dt = np.int8 if "int8" in sys.argv else np.float32
ld, ss, inp = syn.make_problem("cfg3", low_memory=upper, ld_dtype=dt)
for env in ({}, {"VIPRS_TEAM0": "16", "VIPRS_TEAM1": "8"}, {"VIPRS_TEAM0": "24", "VIPRS_TEAM1": "6", "VIPRS_MEDIUM_BLOCK": "1280"}):
    for k in ("VIPRS_TEAM0", "VIPRS_TEAM1", "VIPRS_MEDIUM_BLOCK"): os.environ.pop(k, None)
    os.environ.update(env)
    plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, upper); ds = DeviceState(plan)
    ds.upload("std_beta", inp.std_beta)
    for k in ("u_logs", "sqrt_half_var_tau", "mu_mult"): ds.upload(k, getattr(inp, k))
    for _ in range(5): ds.reset(inp.pi); ds.e_step(ld.dq_scale, sync=False)
    ds.synchronize(); plan.timing_reset()
    for _ in range(30): ds.reset(inp.pi); ds.e_step(ld.dq_scale, sync=False)
    ds.synchronize(); t = np.array(plan.timing_history(which=1))
    print(os.path.basename(os.environ["VIPRS_HIP_LIB"]), env, "p50 %.4f" % np.median(t), flush=True)
    ds.close(); plan.close()
PY
done; done
