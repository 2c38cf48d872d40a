"""ns per serial chain step of every model on ONE isolated LD block (the calibration of bench.py's CHAIN_NS: the time a
sweep cannot go below is its largest block's chain).
    python tools/chain_ns_models.py [block size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from viprs_amd.plan import DeviceState, LDPlan
from viprs_amd.utils import synthetic as syn

size = int(sys.argv[1]) if len(sys.argv) > 1 else 3619
for prec, model, width in (("float32", "spike_slab", 1), ("float32", "mixture", 4), ("float32", "grid", 32), ("float64", "spike_slab", 1)):
    for math in ("exact", "fast"):
        if prec == "float64" and math == "fast":
            continue
        T = np.float64 if prec == "float64" else np.float32
        ld, ss, inp = syn.make_problem(sizes=[size], low_memory=False, seed=3, kind="longrange", float_precision=T)
        plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, False, math_mode=math)
        ds = DeviceState(plan, prec, model, width)
        ds.upload("std_beta", inp.std_beta)
        active, pi0 = None, inp.pi
        if model == "spike_slab":
            for k in ("u_logs", "sqrt_half_var_tau", "mu_mult"):
                ds.upload(k, getattr(inp, k))
        else:
            extra = syn.make_mixture_inputs(ss, width) if model == "mixture" else syn.make_grid_inputs(ss, width)
            pi0 = extra.pop("pi")
            for k, a in extra.items():
                ds.upload(k, a)
            if model == "grid":
                active = np.arange(width, dtype=np.int32)
        for _ in range(3):
            ds.reset(pi0); ds.e_step(ld.dq_scale, active, sync=False)
        ds.synchronize(); plan.timing_reset()
        for _ in range(20):
            ds.reset(pi0); ds.e_step(ld.dq_scale, active, sync=False)
        ds.synchronize()
        t = np.median(plan.timing_history(which=1))
        print(f"{prec} {model}{width} {math}: one block of {size}: kernel {t * 1e3:.0f} us = {t * 1e6 / size:.0f} ns per chain step", flush=True)
        ds.close(); plan.close()
