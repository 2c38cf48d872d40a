#!/usr/bin/env python3
"""Stress: N sweeps of the E-step on cfg3 from the same start must be bit-identical (development tool; reports the
blocks whose state differs from the first sweep).  Two consecutive sweeps WITHOUT a reset in between are part of every
iteration, so that the hand-off tags / arrival counters / queue heads of one launch meet the next launch unzeroed.
    python tools/stress_repro.py N [--low-memory] [--int8] [--mixture] [--longrange] [--populous]
(--populous: 45 x 2 400 + 60 x 2 000 + 30 x 500 SNPs instead of cfg3 -- more team blocks than teams)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viprs_amd.plan import DeviceState, LDPlan          # noqa: E402
from viprs_amd.utils import synthetic as syn            # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
low_memory = "--low-memory" in sys.argv
mixture = "--mixture" in sys.argv
sizes = [2400] * 45 + [2000] * 60 + [500] * 30 if "--populous" in sys.argv else None
ld, ss, inp = syn.make_problem("cfg3", sizes=sizes, low_memory=low_memory, ld_dtype=np.int8 if "--int8" in sys.argv else np.float32,
                               kind="longrange" if "--longrange" in sys.argv else "ar1")
plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, low_memory)
if mixture:
    st = DeviceState(plan, "float32", "mixture", 4)
    extra = syn.make_mixture_inputs(ss, 4)
    inp.pi = extra.pop("pi")
    st.upload("std_beta", inp.std_beta)
    for k, a in extra.items():
        st.upload(k, a)
else:
    st = DeviceState(plan)
    for k in ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult"):
        st.upload(k, getattr(inp, k))
ref = None
bad = 0
for i in range(n):
    st.reset(inp.pi)
    st.e_step(ld.dq_scale, sync=False)
    st.e_step(ld.dq_scale)                     # second sweep straight behind the first
    out = {k: st.download(k) for k in ("var_gamma", "var_mu", "eta", "q", "eta_diff")}
    if ref is None:
        ref = out
        continue
    diff = np.zeros(ld.m, bool)
    for k in out:
        d = out[k] != ref[k]
        diff |= d.any(axis=1) if d.ndim == 2 else d
    if diff.any():
        bad += 1
        blocks = np.unique(np.searchsorted(ld.block_start, np.nonzero(diff)[0], side="right") - 1)
        sizes = np.diff(ld.block_start)[blocks]
        first = np.nonzero(diff)[0][:5]
        print(f"sweep {i}: {int(diff.sum())} SNPs differ in blocks {blocks[:8]} (sizes {sizes[:8]}); first SNPs {first}; "
              f"q there {out['q'][first]} vs {ref['q'][first]}", flush=True)
print(f"{bad} of {n - 1} sweeps differ from the first")
