"""Schedule-parameter sweep on the headline workload (cfg3, fp32, symmetric): team sizes, class limits, resident
workgroups per CU, admission factor.  One process, one host copy of the workload; every configuration builds its own
plan (the environment switches are read at plan creation / launch).  Prints median / p10 / p90 kernel ms of 30 sweeps.
    python tools/sched_sweep.py [upper] [int8]"""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from viprs_amd.plan import DeviceState, LDPlan
from viprs_amd.utils import synthetic as syn

upper = "upper" in sys.argv
dt = np.int8 if "int8" in sys.argv else np.float32
ld, ss, inp = syn.make_problem("cfg3", low_memory=upper, ld_dtype=dt)
CONFIGS = [dict()]
for lg, md in (("1792", "1600"), ("2304", "1280"), ("2304", "1408"), ("1792", "1280"), ("2816", "1600"), ("2304", "1024")):
    CONFIGS.append(dict(VIPRS_LARGE_BLOCK=lg, VIPRS_MEDIUM_BLOCK=md))
for t0, t1 in (("8", "4"), ("16", "4"), ("12", "6"), ("12", "8"), ("16", "8"), ("12", "2")):
    CONFIGS.append(dict(VIPRS_TEAM0=t0, VIPRS_TEAM1=t1))
CONFIGS.append(dict(VIPRS_LARGE_BLOCK="1792", VIPRS_MEDIUM_BLOCK="1280", VIPRS_TEAM0="12", VIPRS_TEAM1="6"))
CONFIGS.append(dict())                        # the default again (box drift over the run)
if "teams" in sys.argv:
    # team sizes only, three alternations (a box drifts by several % within a minute)
    one = [dict(VIPRS_TEAM0=a, VIPRS_TEAM1=b) for a, b in (("12", "4"), ("16", "4"), ("14", "4"), ("16", "5"), ("16", "6"), ("12", "6"), ("12", "3"), ("16", "3"))]
    CONFIGS = one * 3
KEYS = ("VIPRS_BOTTOM_MOD", "VIPRS_MAX_WG_PER_CU", "VIPRS_TEAM0", "VIPRS_TEAM1", "VIPRS_LARGE_BLOCK", "VIPRS_MEDIUM_BLOCK")
for cfg in CONFIGS:
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(cfg)
    plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, upper)
    ds = DeviceState(plan)
    ds.upload("std_beta", inp.std_beta)
    for k in ("u_logs", "sqrt_half_var_tau", "mu_mult"):
        ds.upload(k, getattr(inp, k))
    for _ in range(5):
        ds.reset(inp.pi); ds.e_step(ld.dq_scale, sync=False)
    ds.synchronize(); plan.timing_reset()
    for _ in range(30):
        ds.reset(inp.pi); ds.e_step(ld.dq_scale, sync=False)
    ds.synchronize()
    t = np.array(plan.timing_history(which=1))
    print(f"{str(cfg):110s} kernel ms p50 {np.median(t):.3f}  p10 {np.percentile(t, 10):.3f}  p90 {np.percentile(t, 90):.3f}", flush=True)
    ds.close(); plan.close()
