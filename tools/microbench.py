#!/usr/bin/env python3
"""Kernel micro-benchmarks on one MI355X (development tool; not part of the test-suite).

    python tools/microbench.py sizes 650x1536 2048x256 3619x8      # equal-size blocks (size x count)
    python tools/microbench.py classes                              # cfg3 split by block-size class
Options: --math exact|fast, --low-memory, --ld-dtype float32|int8|int16
Working sets below 256 MiB stay in the Infinity Cache between sweeps: use enough blocks to exceed it
when the number is meant to be an HBM number."""
import argparse
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viprs_amd.plan import DeviceState, LDPlan          # noqa: E402
from viprs_amd.utils import synthetic as syn            # noqa: E402


def run(sizes, tag, args, reps=5):
    sizes = np.asarray(sizes)
    ld = syn.make_ld(sizes, low_memory=args.low_memory, ld_dtype=np.dtype(args.ld_dtype), seed=5)
    ss = syn.make_sumstats(ld, seed=5)
    inp = syn.make_inputs(ss)
    plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, ld.low_memory, math_mode=args.math)
    st = DeviceState(plan)
    for n in ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult"):
        st.upload(n, getattr(inp, n))
    for _ in range(2):
        st.reset(inp.pi)
        st.e_step(ld.dq_scale)
    plan.timing_reset()
    for _ in range(reps):
        st.reset(inp.pi)
        st.e_step(ld.dq_scale)
    ms = float(np.median(plan.timing_history(1)))
    nbytes = int(ld.ld_indptr[-1]) * np.dtype(args.ld_dtype).itemsize * (2 if ld.low_memory else 1)
    print(f"{tag:24s} blocks={len(sizes):5d} snps={ld.m:8d} {ms * 1e3:9.1f} us {ms * 1e6 / ld.m:8.2f} ns/SNP "
          f"{nbytes / ms / 1e6:8.1f} GB/s")
    plan.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["sizes", "classes"])
    ap.add_argument("specs", nargs="*")
    ap.add_argument("--math", default="exact")
    ap.add_argument("--low-memory", action="store_true")
    ap.add_argument("--ld-dtype", default="float32")
    args = ap.parse_args()
    if args.mode == "sizes":
        for spec in args.specs:
            b, n = (int(x) for x in spec.split("x"))
            run([b] * n, spec, args)
    else:
        s = syn.block_sizes("cfg3")
        run(s[s >= 2304], ">= 2304 (teams of 8)", args)
        run(s[(s >= 1280) & (s < 2304)], "1280..2303 (teams of 2)", args)
        run(s[s < 1280], "< 1280", args)
        run(s, "cfg3", args)


if __name__ == "__main__":
    main()
