#!/usr/bin/env python3
"""Development check: are the upper-triangular (low_memory=True) results bit-identical to the oracle too?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O                                  # noqa: E402
from tests import helpers as H                                  # noqa: E402
from tests.test_oracle_vs_ref import _grid_inputs, _mixture_inputs   # noqa: E402
from viprs_amd.utils import synthetic as syn                    # noqa: E402
from viprs_amd.vi import e_step_hip as S                        # noqa: E402

STATE = H.STATE


def report(tag, got, ref):
    bad = [k for k in STATE if not np.array_equal(got[k], ref[k])]
    print(tag, "bit-identical" if not bad else f"DIFFERS in {bad}: " + ", ".join(
        f"{k} max|d|={np.abs(got[k].astype(np.float64) - ref[k].astype(np.float64)).max():.2e}" for k in bad))


for dt in (np.float32, np.int8, np.int16):
    for sizes in ([130, 1300, 64], [2500, 333]):
        ld, ss, inp = syn.make_problem(sizes=sizes, low_memory=True, seed=5, ld_dtype=dt)
        st0 = inp.state_copy()
        report(f"spike-slab {np.dtype(dt).name} {sizes}", H.run_hip(ld, inp, st0, sweeps=2), H.run_oracle(ld, inp, st0, sweeps=2))

ld, ss, inp = syn.make_problem(sizes=[70, 1400, 333], low_memory=True, seed=31)
for K in (1, 4, 8):
    mix, st0 = _mixture_inputs(ld, ss, K)
    out = {}
    for name, mod in (("ref", O), ("hip", S)):
        st = {k: v.copy() for k, v in st0.items()}
        for _ in range(2):
            mod.cpp_e_step_mixture(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, inp.std_beta, st["var_gamma"], st["var_mu"],
                                   st["eta"], st["q"], st["eta_diff"], mix["log_null_pi"], mix["u_logs"], mix["shvt"],
                                   mix["mu_mult"], ld.dq_scale, 1, True)
        out[name] = st
    report(f"mixture K={K}", out["hip"], out["ref"])

for mfma in ("0", "1"):
    os.environ["VIPRS_GRID_MFMA"] = mfma
    ld, ss, inp = syn.make_problem(sizes=[130, 1300, 64], low_memory=True, seed=33 + int(mfma))
    g, st0 = _grid_inputs(ld, ss, 32)
    active = np.arange(32, dtype=np.int32)
    out = {}
    for name, mod in (("ref", O), ("hip", S)):
        st = {k: v.copy(order="F") for k, v in st0.items()}
        for _ in range(2):
            mod.cpp_e_step_grid(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, inp.std_beta, st["var_gamma"], st["var_mu"],
                                st["eta"], st["q"], st["eta_diff"], g["u_logs"], g["hvt"], g["mu_mult"], ld.dq_scale,
                                active, 1, True)
        out[name] = st
    report(f"grid G=32 mfma={mfma}", out["hip"], out["ref"])
