#!/usr/bin/env python3
"""Time the batched grid E-step (G models per sweep) on one MI355X (development tool).

    python tools/grid_bench.py cfg2 32          # workload, number of grid models
    VIPRS_GRID_MFMA=0 python tools/grid_bench.py cfg3 32    # per-(block, model) item path
Options: --low-memory, --ld-dtype float32|int8|int16"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viprs_amd.plan import DeviceState, LDPlan          # noqa: E402
from viprs_amd.utils import synthetic as syn            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("n_models", type=int)
    ap.add_argument("--low-memory", action="store_true")
    ap.add_argument("--ld-dtype", default="float32")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    G = args.n_models
    if "x" in args.workload:                      # equal-size blocks: SIZExCOUNT
        bsz, cnt = (int(x) for x in args.workload.split("x"))
        sizes = np.full(cnt, bsz)
    else:
        sizes = syn.block_sizes(args.workload)
    ld = syn.make_ld(sizes, low_memory=args.low_memory, ld_dtype=np.dtype(args.ld_dtype), seed=5)
    ss = syn.make_sumstats(ld, seed=5)
    inp = syn.make_inputs(ss)
    m = ld.m
    plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, ld.low_memory)
    st = DeviceState(plan, model="grid", width=G)
    rng = np.random.default_rng(3)
    scale = np.exp(rng.uniform(-0.3, 0.3, size=G)).astype(np.float32)
    st.upload("std_beta", inp.std_beta)
    st.upload("u_logs", np.asfortranarray(inp.u_logs[:, None] + np.log(scale)[None, :]).astype(np.float32, order="F"))
    st.upload("sqrt_half_var_tau", np.asfortranarray((inp.sqrt_half_var_tau ** 2)[:, None] * scale[None, :]).astype(np.float32, order="F"))
    st.upload("mu_mult", np.asfortranarray(inp.mu_mult[:, None] * np.ones((1, G), np.float32)).astype(np.float32, order="F"))
    active = np.arange(G, dtype=np.int32)
    for _ in range(2):
        st.reset(inp.pi)
        st.e_step(ld.dq_scale, active)
    plan.timing_reset()
    for _ in range(args.reps):
        st.reset(inp.pi)
        st.e_step(ld.dq_scale, active)
    ms = float(np.median(plan.timing_history(1)))
    nbytes = int(ld.ld_indptr[-1]) * np.dtype(args.ld_dtype).itemsize * (2 if ld.low_memory else 1)
    print(f"{args.workload} G={G} low_memory={args.low_memory} {args.ld_dtype}: {ms:8.3f} ms/sweep  "
          f"{m * G / ms / 1e6:8.2f} G SNP-model-updates/s  LD once = {nbytes / ms / 1e6:7.1f} GB/s")
    plan.close()


if __name__ == "__main__":
    main()
