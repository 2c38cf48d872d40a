"""Development tool: error of math_mode='fast' against the oracle (exact arithmetic) on far-field-sensitive LD.
Per state array: worst relative error (floor 1e-7 max|ref|, tests/helpers.py), entries beyond 1e-5, skip-branch flips,
and the same with the error taken relative to the block-level scale max(|ref|, rms(ref))."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from tests import helpers as H
from tests.test_oracle_vs_ref import _grid_inputs, _mixture_inputs
from tests.test_gpu_farfield import _run_mix, _run_grid
from viprs_amd.utils import synthetic as syn
from viprs_amd.vi import e_step_hip as S


def stats(got, ref, tag):
    same = (got["eta_diff"] == 0) == (ref["eta_diff"] == 0)
    out = [f"{tag}: flips {int((~same).sum())}"]
    for k in H.STATE:
        g, r = got[k].astype(np.float64).ravel(), ref[k].astype(np.float64).ravel()
        scale = np.max(np.abs(r))
        rel = np.abs(g - r) / np.maximum(np.abs(r), 1e-7 * scale + 1e-300)
        rms = np.sqrt(np.mean(r * r))
        rel2 = np.abs(g - r) / np.maximum(np.abs(r), rms)
        out.append(f"{k}: max {rel.max():.2e} n>1e-5 {int((rel > 1e-5).sum())}/{r.size} p99.9 {np.percentile(rel, 99.9):.1e} | rms-floored max {rel2.max():.2e}")
    print("  ".join(out[:1]) + "\n    " + "\n    ".join(out[1:]), flush=True)


S.set_default_math_mode("fast")
for sizes, kind, dt, sweeps in (([700, 1400, 90], "longrange", np.float32, 2), ([1700, 2400, 65], "longrange", np.float32, 2),
                                ([3619, 650, 1536], "longrange", np.float32, 2), ([6000, 77], "longrange", np.float32, 1),
                                ([1700, 650], "sample", np.float32, 2), ([2400, 3619], "longrange", np.int8, 2),
                                ([700, 300], "ar1", np.float32, 3)):
    for low in (False, True):
        ld, ss, inp = syn.make_problem(sizes=sizes, low_memory=low, ld_dtype=dt, seed=61, kind=kind)
        st0 = inp.state_copy()
        stats(H.run_hip(ld, inp, st0, sweeps=sweeps), H.run_oracle(ld, inp, st0, sweeps=sweeps),
              f"ss {sizes} {kind} {np.dtype(dt).name} {'upper' if low else 'sym'} x{sweeps}")
for low in (False, True):
    ld, ss, inp = syn.make_problem(sizes=[700, 1400, 1700, 2400], low_memory=low, seed=62, kind="longrange")
    mix, st0 = _mixture_inputs(ld, ss, 4)
    stats(_run_mix(S, ld, inp, mix, st0, 2), _run_mix(O, ld, inp, mix, st0, 2), f"mixture K=4 {'upper' if low else 'sym'} x2")
    for mf in ("1", "0"):
        os.environ["VIPRS_GRID_MFMA"] = mf
        ld, ss, inp = syn.make_problem(sizes=[700, 1400, 90], low_memory=low, seed=63, kind="longrange")
        g, st0 = _grid_inputs(ld, ss, 32)
        act = np.arange(0, 32, 5, dtype=np.int32)
        got, ref = _run_grid(S, ld, inp, g, st0, act, 2), _run_grid(O, ld, inp, g, st0, act, 2)
        stats({k: v[:, act] for k, v in got.items()}, {k: v[:, act] for k, v in ref.items()},
              f"grid G=32 mfma={mf} {'upper' if low else 'sym'} x2")
