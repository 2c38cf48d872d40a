"""Sweep time on BANDED (windowed-estimator) LD -- one ragged component, served by the generic kernel --
next to the CPU reference build.  Usage: python tools/banded_bench.py [--m 100000] [--w 250] [--upper]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viprs_amd.plan import DeviceState, LDPlan          # noqa: E402
from viprs_amd.utils import synthetic as syn            # noqa: E402


def banded(m, w, upper, rho=0.9, seed=0):
    j = np.arange(m)
    lo = j + 1 if upper else np.maximum(j - w, 0)
    hi = np.minimum(j + w + 1, m)
    length = np.maximum(hi - lo, 0)
    ip = np.concatenate([[0], np.cumsum(length)]).astype(np.int64)
    data = np.empty(int(ip[-1]), np.float32)
    pw = np.power(rho, np.arange(w + 1)).astype(np.float32)
    for r in range(m):
        cols = np.arange(lo[r], hi[r])
        data[ip[r]:ip[r + 1]] = pw[np.abs(cols - r)]
    return lo.astype(np.int32), ip, data


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=100000)
    ap.add_argument("--w", type=int, default=250)
    ap.add_argument("--upper", action="store_true")
    ap.add_argument("--cpu", action="store_true", help="also time the reference build (oracle/_ref)")
    a = ap.parse_args()
    lb, ip, data = banded(a.m, a.w, a.upper)
    ld = syn.SyntheticLD(lb, ip, data, np.array([0, a.m]), np.zeros(1), a.upper, 1.0)
    rng = np.random.default_rng(1)
    std_beta = (rng.standard_normal(a.m) * 0.003).astype(np.float32)
    inp = syn.make_inputs(syn.SyntheticSumstats(std_beta, np.full(a.m, 1e5), np.zeros(a.m, np.float32), 1e5))
    plan = LDPlan(lb, ip, data, a.upper)
    ds = DeviceState(plan)
    ds.upload("std_beta", inp.std_beta)
    for k in ("u_logs", "sqrt_half_var_tau", "mu_mult"):
        ds.upload(k, getattr(inp, k))
    ds.reset(0.01)
    for _ in range(2):
        ds.e_step(1.0)
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        ds.e_step(1.0, sync=False)
    ds.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"banded m={a.m} w={a.w} {'upper' if a.upper else 'symmetric'}: nnz={ip[-1] / 1e6:.1f} M, blocks={plan.n_blocks}, "
          f"HIP sweep {dt * 1e3:.2f} ms = {a.m / dt / 1e6:.2f} M SNP-updates/s")
    if a.cpu:
        from oracle import oracle as O
        st = inp.state_copy()
        t0 = time.perf_counter()
        O.cpp_e_step(lb, ip, data, inp.std_beta, st["var_gamma"], st["var_mu"], st["eta"], st["q"], st["eta_diff"],
                     inp.u_logs, inp.sqrt_half_var_tau, inp.mu_mult, 1.0, 1, a.upper, kind="reference")
        dc = time.perf_counter() - t0
        print(f"reference build, 1 core: {dc * 1e3:.1f} ms = {a.m / dc / 1e6:.2f} M SNP-updates/s")


if __name__ == "__main__":
    main()
