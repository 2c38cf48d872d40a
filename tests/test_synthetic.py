"""CPU: the synthetic LD generator's three kinds of block, and the property the parity suite relies on --
the far field of "longrange" / "sample" blocks changes an E-step result, that of AR(1) blocks does not."""
import numpy as np
import pytest

from tests import helpers as H
from viprs_amd.utils import synthetic as syn


@pytest.mark.parametrize("kind", syn.LD_KINDS)
@pytest.mark.parametrize("ld_dtype", [np.float32, np.int8])
def test_blocks_are_correlation_matrices_and_forms_agree(kind, ld_dtype):
    a = syn.make_ld([300, 70], low_memory=False, kind=kind, seed=5, ld_dtype=ld_dtype)
    b = syn.make_ld([300, 70], low_memory=True, kind=kind, seed=5, ld_dtype=ld_dtype)
    assert a.ld_indptr[-1] == 300 * 300 + 70 * 70 and b.ld_indptr[-1] == (300 * 299 + 70 * 69) // 2
    for bi in range(2):
        R, Ru = syn.dense_block(a, bi), syn.dense_block(b, bi)
        assert np.array_equal(R, Ru)                       # the two forms hold the same numbers
        assert np.array_equal(R, R.T) and np.all(np.abs(R.diagonal() - 1) < 1e-6)
        assert np.linalg.eigvalsh(R).min() > (0.02 if ld_dtype == np.int8 else 0.05)
    if kind != "ar1":
        R = syn.dense_block(a, 0)
        far = np.abs(R[np.abs(np.subtract.outer(np.arange(300), np.arange(300))) > 128])
        assert far.mean() > (0.05 if kind == "longrange" else 0.01)
        assert not np.allclose(R[0, 1:20], R[1, 2:21])     # not Toeplitz


@pytest.mark.parametrize("kind", syn.LD_KINDS)
def test_sumstats_follow_the_block_model(kind):
    """std_beta - R beta has covariance R / N: its whitened norm is chi-square distributed."""
    ld = syn.make_ld([400, 250], kind=kind, seed=9)
    ss = syn.make_sumstats(ld, n=1e4, seed=9, float_precision=np.float64)
    ss2 = syn.make_sumstats(syn.make_ld([400, 250], kind=kind, seed=9, low_memory=True), n=1e4, seed=9,
                            float_precision=np.float64)
    assert np.array_equal(ss.std_beta, ss2.std_beta)
    for bi in range(2):
        s, e = int(ld.block_start[bi]), int(ld.block_start[bi + 1])
        R = syn.dense_block(ld, bi)
        resid = (ss.std_beta[s:e] - R @ ss.beta_true[s:e]) * np.sqrt(ss.n)
        chi2 = resid @ np.linalg.solve(R, resid)
        assert 0.6 * (e - s) < chi2 < 1.5 * (e - s)


@pytest.mark.parametrize("low_memory", [False, True])
@pytest.mark.parametrize("kind, blind", [("ar1", True), ("longrange", False), ("sample", False)])
def test_far_field_sensitivity(kind, blind, low_memory):
    ld, ss, inp = syn.make_problem(sizes=[900, 400], kind=kind, seed=13, low_memory=low_memory)
    ref = H.run_oracle(ld, inp, inp.state_copy())
    cut = H.run_oracle(H.cut_far_field(ld, 128), inp, inp.state_copy())
    changed = int((cut["q"] != ref["q"]).sum())
    if blind:
        assert changed == 0          # the reason AR(1) inputs alone are not a parity test
    else:
        assert changed > ld.m // 2
