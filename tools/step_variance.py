"""Per-sweep panel-kernel times (HIP events) of 3 x 30 sweeps: python tools/step_variance.py [cfg3|cfg2|cfg1] [upper]."""
import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from viprs_amd.plan import DeviceState, LDPlan
from viprs_amd.utils import synthetic as syn
CFG = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
LOWMEM = len(sys.argv) > 2 and sys.argv[2] == "upper"
ld, ss, inp = syn.make_problem(CFG, low_memory=LOWMEM)
plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, LOWMEM)
ds = DeviceState(plan)
ds.upload("std_beta", inp.std_beta)
for k in ("u_logs", "sqrt_half_var_tau", "mu_mult"):
    ds.upload(k, getattr(inp, k))
for rep in range(3):
    plan.timing_reset()
    for _ in range(30):
        ds.reset(0.01); ds.e_step(1.0, sync=False)
    ds.synchronize()
    h = np.array(plan.timing_history(1, 64))
    print("kernel ms:", " ".join(f"{x:.2f}" for x in h[-30:]), "| mean %.3f" % h[-30:].mean())
