"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np

from oracle import oracle as O
from viprs_amd.utils import synthetic as syn

STATE = ("var_gamma", "var_mu", "eta", "q", "eta_diff")

# Parity tolerance stated by BASELINE.json's north_star: 1e-5 relative in fp32.  `floor` keeps the
# relative test meaningful for entries that are ~0: 1e-7 * max|ref|.
RTOL_F32 = 1e-5


def run_oracle(ld, inp, state, kind="restated", sweeps=1):
    st = {k: v.copy() for k, v in state.items()}
    for _ in range(sweeps):
        O.cpp_e_step(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, inp.std_beta, st["var_gamma"], st["var_mu"],
                     st["eta"], st["q"], st["eta_diff"], inp.u_logs, inp.sqrt_half_var_tau, inp.mu_mult,
                     ld.dq_scale, 1, ld.low_memory, kind=kind)
    return st


def run_hip(ld, inp, state, sweeps=1):
    from viprs_amd.vi import e_step_hip as H
    st = {k: v.copy() for k, v in state.items()}
    for _ in range(sweeps):
        H.cpp_e_step(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, inp.std_beta, st["var_gamma"], st["var_mu"],
                     st["eta"], st["q"], st["eta_diff"], inp.u_logs, inp.sqrt_half_var_tau, inp.mu_mult,
                     ld.dq_scale, 1, ld.low_memory)
    return st


def cut_far_field(ld, width=128):
    """A copy of `ld` (block LD of `synthetic.make_ld`, either form) with every entry more than `width`
    columns off the diagonal set to 0 -- the probe for "does this input exercise the far field at all"."""
    import copy
    out = copy.copy(ld)
    data = ld.ld_data.copy()
    for bi in range(len(ld.block_start) - 1):
        s, e = int(ld.block_start[bi]), int(ld.block_start[bi + 1])
        b = e - s
        if b <= width + 1:
            continue
        o = int(ld.ld_indptr[s])
        if ld.low_memory:
            for r in range(b - 1 - width):           # row r holds columns r+1 .. b-1
                n = b - 1 - r
                data[o + width:o + n] = 0
                o += n
        else:
            M = data[o:o + b * b].reshape(b, b)
            i = np.arange(b)
            M[np.abs(i[:, None] - i[None, :]) > width] = 0
    out.ld_data = data
    return out


def assert_close(got, ref, rtol=RTOL_F32, what=""):
    """|got - ref| <= rtol * max(|ref|, floor), floor = 1e-7 * max|ref| (SURVEY.md 8c parity metric)."""
    scale = float(np.max(np.abs(ref))) if ref.size else 0.0
    floor = 1e-7 * scale + np.finfo(ref.dtype).tiny
    tol = rtol * np.maximum(np.abs(ref), floor) + 4 * np.finfo(ref.dtype).eps * rtol * scale
    bad = np.abs(got.astype(np.float64) - ref.astype(np.float64)) > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{ref.size} entries beyond rtol={rtol}; worst "
                           f"{np.max(np.abs(got - ref))} at {int(np.argmax(np.abs(got - ref)))}")


def branch_flips(got, ref):
    """SNPs where exactly one side took the skip branch (eta_diff == 0, e_step.hpp:410-413)."""
    return int(np.sum((got["eta_diff"] == 0) != (ref["eta_diff"] == 0)))


def assert_state_close(got, ref, rtol=RTOL_F32, max_flips=0):
    same = (got["eta_diff"] == 0) == (ref["eta_diff"] == 0)
    assert int((~same).sum()) <= max_flips, f"{int((~same).sum())} skip-branch flips"
    for k in ("eta", "q", "eta_diff"):
        assert_close(got[k], ref[k], rtol, k)
    for k in ("var_gamma", "var_mu"):
        assert_close(got[k][same], ref[k][same], rtol, k)


def assert_state_equal(got, ref):
    for k in STATE:
        assert np.array_equal(got[k], ref[k]), f"{k}: {int((got[k] != ref[k]).sum())} entries differ bitwise"
