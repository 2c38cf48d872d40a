"""Sweep time on workloads whose block-size distribution differs from cfg3's (team classes populated differently):
looks for cliffs in the schedule.    python tools/mixed_blocks_bench.py [upper] [int8] [mix|grid|f64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from viprs_amd.plan import DeviceState, LDPlan
from viprs_amd.utils import synthetic as syn

upper = "upper" in sys.argv
dt = np.int8 if "int8" in sys.argv else np.float32
model = "mixture" if "mix" in sys.argv else "grid" if "grid" in sys.argv else "spike_slab"
width = {"mixture": 4, "grid": 32, "spike_slab": 1}[model]
prec = "float64" if "f64" in sys.argv else "float32"
WORKLOADS = {
    "10000 x 100": [100] * 10000,
    "60000 x 12": [12] * 60000,
    "100 x 2500 + 200 x 2000": [2500] * 100 + [2000] * 200,
    "30 x 3000 + 300 x 1700 + 500 x 400": [3000] * 30 + [1700] * 300 + [400] * 500,
    "300 x 2400 + 1000 x 300": [2400] * 300 + [300] * 1000,
    "20 x 6000 + 100 x 2000 + 1000 x 500": [6000] * 20 + [2000] * 100 + [500] * 1000,
    "400 x 1650 + 400 x 1550": [1650] * 400 + [1550] * 400,
}
for name, sizes in WORKLOADS.items():
    ld, ss, inp = syn.make_problem(sizes=sizes, low_memory=upper, ld_dtype=dt, seed=5, float_precision=np.dtype(prec))
    plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, upper)
    ds = DeviceState(plan, prec, model, width)
    active, pi0 = None, inp.pi
    ds.upload("std_beta", inp.std_beta)
    if model == "spike_slab":
        for k in ("u_logs", "sqrt_half_var_tau", "mu_mult"):
            ds.upload(k, getattr(inp, k))
    else:
        extra = syn.make_mixture_inputs(ss, width, float_precision=np.dtype(prec)) if model == "mixture" else syn.make_grid_inputs(ss, width, float_precision=np.dtype(prec))
        pi0 = extra.pop("pi")
        for k, a in extra.items():
            ds.upload(k, a)
        if model == "grid":
            active = np.arange(width, dtype=np.int32)
    try:
        ds.reset(pi0); ds.e_step(ld.dq_scale, active)
    except Exception as e:                                  # noqa: BLE001 -- report and go on with the next workload
        print(f"{name:40s} FAILED: {e}", flush=True)
        continue
    for _ in range(2):
        ds.reset(pi0); ds.e_step(ld.dq_scale, active, sync=False)
    ds.synchronize(); plan.timing_reset()
    for _ in range(10):
        ds.reset(pi0); ds.e_step(ld.dq_scale, active, sync=False)
    ds.synchronize()
    ms = float(np.median(plan.timing_history(which=1)))
    nbytes = int(ld.ld_indptr[-1]) * np.dtype(dt).itemsize * (2 if upper else 1)
    print(f"{name:40s} {ld.m:8d} SNPs  {ms:7.3f} ms  {nbytes / ms / 1e6:7.0f} GB/s", flush=True)
    ds.close(); plan.close()
