#!/usr/bin/env python3
"""Does WHERE the allocator puts a plan's LD buffer move the sweep?  Several identical cfg3 plans alive at once in one
process (different allocations), one state each (no state probe), kernel ms p50 of 30 sweeps, two rounds.
    python tools/plan_placement_probe.py [n_plans=5] [sym]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from viprs_amd.plan import DeviceState, LDPlan
from viprs_amd.utils import synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
upper = "sym" not in sys.argv
ld = syn.make_ld(syn.block_sizes("cfg3"), low_memory=upper, kind="longrange", data=False)
inp = syn.make_inputs(syn.make_sumstats(ld))
plans, states = [], []
for i in range(n):
    p = LDPlan.synthetic(ld)
    s = DeviceState(p, "float32", "spike_slab", placement="off")
    for k in ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult"):
        s.upload(k, getattr(inp, k))
    plans.append(p); states.append(s)
def med(p, s, k=30):
    for _ in range(5):
        s.reset(inp.pi); s.e_step(ld.dq_scale, sync=False)
    s.synchronize(); p.timing_reset()
    for _ in range(k):
        s.reset(inp.pi); s.e_step(ld.dq_scale, sync=False)
    s.synchronize()
    return float(np.median(p.timing_history(which=1)))
med(plans[0], states[0], 60)
for r in range(3):
    print("round", r, " ".join(f"{med(p, s):.4f}" for p, s in zip(plans, states)), flush=True)
# the same plan with fresh states (state placement only)
p = plans[0]
extra = []
for i in range(4):
    s = DeviceState(p, "float32", "spike_slab", placement="off")
    for k in ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult"):
        s.upload(k, getattr(inp, k))
    extra.append(s)
print("plan 0, four more states:", " ".join(f"{med(p, s):.4f}" for s in extra))
