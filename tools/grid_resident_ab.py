"""Batched grid E-step on cfg3 (G = 32): resident form (default) vs the streaming form for every block
(VIPRS_GRID_RESIDENT=0), same process, alternating; kernel ms (main kernel + lower pass / second pass) p50 of 20 sweeps.
    python tools/grid_resident_ab.py [upper] [int8] [uniform]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from viprs_amd.plan import DeviceState, LDPlan
from viprs_amd.utils import synthetic as syn

upper = "upper" in sys.argv
dt = np.int8 if "int8" in sys.argv else np.float32
sizes = np.full(1700, 650) if "uniform" in sys.argv else None
ld, ss, inp = syn.make_problem("cfg3", low_memory=upper, ld_dtype=dt, sizes=sizes, kind="longrange")
plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, upper)
ds = DeviceState(plan, "float32", "grid", 32)
extra = syn.make_grid_inputs(ss, 32)
pi0 = extra.pop("pi")
ds.upload("std_beta", inp.std_beta)
for k, a in extra.items():
    ds.upload(k, a)
active = np.arange(32, dtype=np.int32)
out = {}
for rnd in range(3):
    for mode in ("1", "0"):
        os.environ["VIPRS_GRID_RESIDENT"] = mode
        for _ in range(3):
            ds.reset(pi0); ds.e_step(ld.dq_scale, active, sync=False)
        ds.synchronize(); plan.timing_reset()
        for _ in range(20):
            ds.reset(pi0); ds.e_step(ld.dq_scale, active, sync=False)
        ds.synchronize()
        t = np.array(plan.timing_history(which=1))
        print(f"round {rnd} resident={mode}: kernel ms p50 {np.median(t):.3f} p10 {np.percentile(t, 10):.3f} p90 {np.percentile(t, 90):.3f}", flush=True)
        if rnd == 0:
            out[mode] = {k: ds.download(k) for k in ("var_gamma", "var_mu", "eta", "q", "eta_diff")}
same = all(np.array_equal(out["0"][k], out["1"][k]) for k in out["0"])
print("resident == streaming, all five state arrays, bit for bit:", same)
