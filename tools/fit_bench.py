#!/usr/bin/env python3
"""Time `VIPRS.fit()` iterations on a synthetic genome-scale data set (development tool).

    python tools/fit_bench.py [--chroms 1] [--iters 30] [--host-mirrored]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viprs_amd.data import ArrayDataLoader          # noqa: E402
from viprs_amd.model.VIPRS import VIPRS             # noqa: E402
from viprs_amd.utils import synthetic as syn        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--chroms", type=int, default=1, help="split the blocks over this many chromosomes")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--host-mirrored", action="store_true")
    ap.add_argument("--no-merge", action="store_true", help="one device plan per chromosome")
    ap.add_argument("--mixture", type=int, default=0, help="VIPRSMix with this many components")
    ap.add_argument("--grid", type=int, default=0, help="batched grid fit (VIPRSGrid) with this many (pi, sigma_epsilon) points")
    args = ap.parse_args()
    sizes = syn.block_sizes(args.config)
    parts = np.array_split(np.arange(len(sizes)), args.chroms)
    gdl = ArrayDataLoader.synthetic({c + 1: sizes[p] for c, p in enumerate(parts)}, forms=("symmetric",))
    if args.grid:
        from viprs_amd.model.gridsearch.HyperparameterGrid import HyperparameterGrid
        from viprs_amd.model.gridsearch.VIPRSGrid import VIPRSGrid
        side = max(2, int(round(args.grid ** 0.5)))
        grid = HyperparameterGrid(n_snps=gdl.m)
        grid.generate_pi_grid(steps=(args.grid + side - 1) // side)
        grid.generate_sigma_epsilon_grid(steps=side)
        model = VIPRSGrid(gdl, grid, low_memory=False)
        t0 = time.perf_counter()
        model.fit(max_iter=args.iters, min_iter=args.iters, batched=True)
        t1 = time.perf_counter()
        print(f"{args.config} x{args.chroms} chromosomes, batched grid fit of {model.n_models} models: "
              f"{(t1 - t0) / args.iters * 1e3:.3f} ms per EM iteration incl. set-up ({args.iters} iterations)")
        return
    if args.mixture:
        from viprs_amd.model.VIPRSMix import VIPRSMix
        model = VIPRSMix(gdl, K=args.mixture, low_memory=False, device_resident=not args.host_mirrored,
                         merge_chromosomes=not args.no_merge)
    else:
        model = VIPRS(gdl, low_memory=False, device_resident=not args.host_mirrored, merge_chromosomes=not args.no_merge)
    t0 = time.perf_counter()
    model.fit(max_iter=3, min_iter=3)                       # warm-up (plans, first launches)
    t1 = time.perf_counter()
    model.fit(max_iter=args.iters, min_iter=args.iters, continued=True)
    t2 = time.perf_counter()
    n_it = len(model.history["ELBO"]) - 3
    print(f"{args.config} x{args.chroms} chromosomes, {'mixture K=%d, ' % args.mixture if args.mixture else ''}device_resident={model._resident}: warm-up {t1 - t0:.2f} s, "
          f"{(t2 - t1) / max(1, n_it) * 1e3:.3f} ms per EM iteration ({n_it} iterations), ELBO {model.history['ELBO'][-1]:.6g}; {model.optim_result.message}")


if __name__ == "__main__":
    main()
