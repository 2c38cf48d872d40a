#!/usr/bin/env python3
"""Hunts the one-off slow sweep the round-5 driver run showed in bench.py's `math_mode=fast` int8 secondary (one sweep of
~1.9 ms among ten of 0.38): repeats that secondary's exact sequence -- exact-mode sweeps on a plan, set_math_mode("fast"),
0.1 s of untimed sweeps, 3 warm-up steps, 10 timed steps -- N times and logs every timed sweep's kernel time (HIP events)
with the host's wall clock around it.  python tools/stall_hunt.py [N=40] [int8|float32]
Prints the sequences that contain a sweep > 1.5 x the sequence's median, and a summary."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from viprs_amd.plan import DeviceState, LDPlan
from viprs_amd.utils import synthetic as syn

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 40
dt = np.float32 if "float32" in sys.argv else np.int8
sizes = syn.block_sizes("cfg3")
ld = syn.make_ld(sizes, low_memory=True, ld_dtype=dt, kind="longrange", data=False)
ss = syn.make_sumstats(ld)
inp = syn.make_inputs(ss)
plan = LDPlan.synthetic(ld)
st = DeviceState(plan, "float32", "spike_slab")
for k in ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult"):
    st.upload(k, getattr(inp, k))
dq = ld.dq_scale


def steps(n):
    for _ in range(n):
        st.reset(inp.pi)
        st.e_step(dq, sync=False)


bad, all_t, t_begin = [], [], time.time()
for it in range(N):
    plan.set_math_mode("exact")
    steps(13)                                           # the exact-mode secondary in front of it
    st.synchronize()
    plan.set_math_mode("fast")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        steps(1)
    st.synchronize()
    steps(3)
    st.synchronize()
    plan.timing_reset()
    w0 = time.perf_counter()
    steps(10)
    st.synchronize()
    wall = (time.perf_counter() - w0) * 1e3
    k = plan.timing_history(which=1)
    all_t += list(k)
    med = float(np.median(k))
    if max(k) > 1.5 * med:
        bad.append((it, [round(x, 3) for x in k], round(wall, 3)))
        print(f"sequence {it}: kernel ms {bad[-1][1]}  wall of the 10 steps {wall:.3f} ms  (t = {time.time() - t_begin:.1f} s)", flush=True)
a = np.array(all_t)
print(f"{N} sequences x 10 timed sweeps ({np.dtype(dt).name} upper-triangular LD, math_mode=fast): median {np.median(a):.4f} ms, p99 {np.percentile(a, 99):.4f}, "
      f"max {a.max():.4f}; sequences with a sweep > 1.5 x their median: {len(bad)}")
