"""Chain speed of the dense panel kernels on isolated blocks (ns per serial SNP step)."""
import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from viprs_amd.plan import DeviceState, LDPlan
from viprs_amd.utils import synthetic as syn
import os
CONFIGS = ([1200], [2200], [6000], [1200]*256, [6000]*32) if not os.environ.get('MANY') else ([600]*1700, [300]*3400, [1200]*850)
for sizes in CONFIGS:
    ld, ss, inp = syn.make_problem(sizes=sizes, low_memory=False, seed=3)
    plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, False, math_mode=os.environ.get("MATH", "exact"))
    ds = DeviceState(plan)
    ds.upload("std_beta", inp.std_beta)
    for k in ("u_logs", "sqrt_half_var_tau", "mu_mult"):
        ds.upload(k, getattr(inp, k))
    ds.reset(0.01)
    for _ in range(3): ds.e_step(1.0)
    t0 = time.perf_counter(); n = 20
    for _ in range(n): ds.e_step(1.0, sync=False)
    ds.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"[{os.environ.get('MATH', 'exact')}] {len(sizes)} x {sizes[0]}: sweep {dt*1e6:.0f} us = {dt*1e9/sizes[0]:.0f} ns per chain step")
    del ds
    plan.close()
