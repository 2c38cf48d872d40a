"""float_precision='float64' (VIPRS.py:72): the state is double, every block takes the panel-walking kernels of
estep_tile.h (VIPRS_F64_ROW_BY_ROW=1: the row-by-row generic kernels).  Sweep time and SNP-updates/s on cfg2 / cfg3
through the device-resident state API, next to the fp32 state on the same plan.
    python tools/fp64_bench.py [cfg2|cfg3] [int8|float32|float64] [upper|sym] [K of a sparse mixture prior]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from viprs_amd.plan import DeviceState, LDPlan
from viprs_amd.utils import synthetic as syn

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
dt = np.dtype(sys.argv[2] if len(sys.argv) > 2 else "int8")
upper = not (len(sys.argv) > 3 and sys.argv[3] == "sym")
K = int(sys.argv[4]) if len(sys.argv) > 4 else 0
form = "upper-triangular" if upper else "symmetric"
prior = f", mixture K = {K}" if K else ""
ld, ss, inp = syn.make_problem(cfg, low_memory=upper, ld_dtype=dt, kind="longrange", float_precision=np.float64)
plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, upper)
for prec in ("float64", "float32"):
    if prec == "float32":
        ld32, ss32, inp = syn.make_problem(cfg, low_memory=upper, ld_dtype=dt, kind="longrange")
    pi0 = inp.pi
    if K:
        ds = DeviceState(plan, prec, "mixture", K)
        extra = syn.make_mixture_inputs(ss, K, float_precision=np.dtype(prec))
        pi0 = extra.pop("pi")
        ds.upload("std_beta", inp.std_beta)
        for k, a in extra.items():
            ds.upload(k, a)
    else:
        ds = DeviceState(plan, prec, "spike_slab", 1)
        for k in ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult"):
            ds.upload(k, getattr(inp, k))
    for _ in range(2):
        ds.reset(pi0); ds.e_step(ld.dq_scale, sync=False)
    ds.synchronize()
    n = 5 if prec == "float64" else 20
    t0 = time.perf_counter()
    for _ in range(n):
        ds.reset(pi0); ds.e_step(ld.dq_scale, sync=False)
    ds.synchronize()
    dt_s = (time.perf_counter() - t0) / n
    print(f"{cfg} LD {dt.name} {form}{prior}, state {prec}: {dt_s * 1e3:.2f} ms per sweep = {ld.m / dt_s / 1e6:.1f} M SNP-updates/s", flush=True)
    ds.close()
