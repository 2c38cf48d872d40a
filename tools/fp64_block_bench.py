"""float64 state, isolated blocks: sweep time vs block size (one workgroup per block, estep_tile.h) -- the per-panel
cost splits into the chain (64 steps, constant) and the rows-onto-columns pass (proportional to the block size).
    python tools/fp64_block_bench.py [int8|float32] [upper|sym]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from viprs_amd.plan import DeviceState, LDPlan
from viprs_amd.utils import synthetic as syn

dt = np.dtype(sys.argv[1] if len(sys.argv) > 1 else "int8")
upper = len(sys.argv) > 2 and sys.argv[2] == "upper"
for sizes in ([64], [640], [1280], [2560], [3648], [640] * 512, [640] * 2048):
    ld, ss, inp = syn.make_problem(sizes=sizes, low_memory=upper, ld_dtype=dt, seed=3, float_precision=np.float64)
    plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, upper)
    ds = DeviceState(plan, "float64", "spike_slab", 1)
    for k in ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult"):
        ds.upload(k, getattr(inp, k))
    for _ in range(3):
        ds.reset(inp.pi); ds.e_step(ld.dq_scale)
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        ds.reset(inp.pi); ds.e_step(ld.dq_scale, sync=False)
    ds.synchronize()
    t = (time.perf_counter() - t0) / n
    panels = (sizes[0] + 63) // 64
    print(f"{len(sizes)} x {sizes[0]} ({dt.name}, {'upper' if upper else 'sym'}): sweep {t * 1e6:.0f} us = {t * 1e6 / panels:.1f} us per panel "
          f"= {t * 1e9 / sizes[0]:.0f} ns per SNP", flush=True)
    ds.close(); plan.close()
