"""Development: a few EM iterations of one model family on cfg3, to be run under `rocprofv3 --kernel-trace --stats` for the
per-kernel times of an ITERATION (prep, sweep, sums) rather than of the sweep alone:
    python3 tools/fit_kernels.py [grid|mix|ss|chr|chrmix] [iterations]"""
import os
import sys


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench as B                                     # noqa: E402
from viprs_amd.utils import synthetic as syn          # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "grid"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ld, ss, inp = syn.make_problem("cfg3", low_memory=True)
name = {"grid": "VIPRSGrid(32 models, batched)", "mix": "VIPRSMix(K=4)", "ss": "VIPRS"}.get(kind)
if name:
    out = B.measure_fit_iteration(name, ld, ss, 0, iters=iters)
    print({k: out.get(k) for k in ("name", "ms_per_iteration", "sweep_kernels_ms_avg", "split_ms")})
else:
    out = B.measure_per_chromosome(ld, ss, 0, iters=iters, mixture_k=4 if kind == "chrmix" else 0)
    print(out["name"], out["batched"], out["sequential"]["ms_per_round"])
