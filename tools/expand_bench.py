"""Plan-creation cost: symmetric LD uploaded from host arrays (what the reference's
`load(return_symmetric=True)` feeds) vs the upper-triangular store expanded on the device
(`LDPlan.from_upper`).  Usage: python tools/expand_bench.py [--config cfg3] [--ld-dtype float32|int8]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viprs_amd.plan import LDPlan                     # noqa: E402
from viprs_amd.utils import synthetic as syn          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--ld-dtype", default="float32")
    a = ap.parse_args()
    dt = np.dtype(a.ld_dtype)
    sizes = syn.block_sizes(a.config)
    up = syn.make_ld(sizes, low_memory=True, ld_dtype=dt)
    t0 = time.perf_counter()
    sym = syn.make_ld(sizes, low_memory=False, ld_dtype=dt)
    t_gen = time.perf_counter() - t0
    LDPlan(sym.ld_left_bound[:64].copy() * 0, np.arange(65, dtype=np.int64) * 64, sym.ld_data[:4096].copy(), False).close()   # warm the runtime
    for rep in range(2):
        t0 = time.perf_counter()
        p = LDPlan(sym.ld_left_bound, sym.ld_indptr, sym.ld_data, False)
        t_sym = time.perf_counter() - t0
        nnz = p.nnz
        p.close()
        t0 = time.perf_counter()
        p = LDPlan.from_upper(up.ld_indptr, up.ld_data)
        t_exp = time.perf_counter() - t0
        assert p.nnz == nnz
        p.close()
        print(f"rep {rep}: symmetric host arrays {sym.ld_data.nbytes / 1e9:.2f} GB -> plan in {t_sym * 1e3:.0f} ms;  "
              f"upper store {up.ld_data.nbytes / 1e9:.2f} GB + device mirror -> plan in {t_exp * 1e3:.0f} ms")
    print(f"(building the symmetric arrays on the host took {t_gen:.1f} s with the synthetic generator)")


if __name__ == "__main__":
    main()
