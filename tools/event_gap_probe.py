#!/usr/bin/env python3
"""Does a sweep's HIP-event kernel time depend on how full the stream is when it is submitted?  The events bracket the
launch; if the start event reaches an EMPTY queue before the host has enqueued the kernel, the host's enqueue latency
(occupancy query, team split, launch gate -- or a scheduler hiccup) is counted as kernel time.
    python tools/event_gap_probe.py [int8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from viprs_amd.plan import DeviceState, LDPlan
from viprs_amd.utils import synthetic as syn
dt = np.int8 if "int8" in sys.argv else np.float32
ld = syn.make_ld(syn.block_sizes("cfg3"), low_memory=True, ld_dtype=dt, kind="longrange", data=False)
inp = syn.make_inputs(syn.make_sumstats(ld))
plan = LDPlan.synthetic(ld)
st = DeviceState(plan, "float32", "spike_slab", placement="off")
for k in ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult"):
    st.upload(k, getattr(inp, k))
for _ in range(40):
    st.reset(inp.pi); st.e_step(ld.dq_scale, sync=False)
st.synchronize()
plan.timing_reset()
for _ in range(40):
    st.reset(inp.pi); st.e_step(ld.dq_scale, sync=False)
st.synchronize()
full = np.array(plan.timing_history(which=1))
empty = []
for _ in range(40):
    st.reset(inp.pi)
    st.synchronize()                       # the queue is empty when the sweep is submitted
    st.e_step(ld.dq_scale, sync=True)
    empty.append(plan.last_kernel_ms(1))
empty = np.array(empty)
print(f"back to back: median {np.median(full[5:]):.4f} ms (first of the burst {full[0]:.4f}); into an empty queue: median {np.median(empty):.4f}, "
      f"max {empty.max():.4f} ms; ratio of medians {np.median(empty) / np.median(full[5:]):.4f}")
