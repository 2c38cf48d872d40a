"""GPU: windowed (banded / ragged) LD components -- the band kernel (estep_band.h) against the oracle.
The reference walks such matrices row by row through (ld_left_bound, ld_indptr), e_step.hpp:387-433."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H
from tests.test_oracle_vs_ref import _grid_inputs
from viprs_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu


def banded_ld(m, w_left, w_right, low_memory, ld_dtype=np.float32, seed=0, jitter=0):
    """Rows with windows [j - wl_j, j + wr_j] (symmetric form, diagonal stored) or [j + 1, j + wr_j] (upper
    form); `jitter` makes the reach vary from row to row (ragged windows)."""
    rng = np.random.default_rng(seed)
    j = np.arange(m)
    wl = np.full(m, w_left) - (rng.integers(0, jitter + 1, m) if jitter else 0)
    wr = np.full(m, w_right) - (rng.integers(0, jitter + 1, m) if jitter else 0)
    lo = j + 1 if low_memory else np.maximum(j - np.maximum(wl, 0), 0)
    hi = np.minimum(j + np.maximum(wr, 0) + 1, m)
    length = np.maximum(hi - lo, 0)
    ip = np.concatenate([[0], np.cumsum(length)]).astype(np.int64)
    n = int(ip[-1])
    integer = np.issubdtype(np.dtype(ld_dtype), np.integer)
    qmax = np.iinfo(ld_dtype).max if integer else None
    data = np.empty(n, dtype=ld_dtype)
    sign = np.where(rng.random(m) < 0.3, -1.0, 1.0)
    for r in range(m):
        cols = np.arange(lo[r], hi[r])
        v = np.power(0.7, np.abs(cols - r)) * sign[cols] * sign[r]        # D R D of a truncated AR(1): well conditioned
        data[ip[r]:ip[r + 1]] = np.round(v * qmax) if integer else v
    dq = 1.0 / qmax if integer else 1.0
    lb = np.where(length > 0, lo, np.minimum(lo, m - 1)).astype(np.int32)
    return syn.SyntheticLD(lb, ip, data, np.array([0, m]), np.zeros(1), bool(low_memory), dq)


def _inputs(m, seed=1):
    rng = np.random.default_rng(seed)
    beta = (rng.standard_normal(m) * 0.004).astype(np.float32)
    beta[rng.integers(0, m, max(1, m // 50))] += 0.05
    ss = syn.SyntheticSumstats(beta, np.full(m, 1e5), np.zeros(m, np.float32), 1e5)
    return ss, syn.make_inputs(ss)


@pytest.mark.parametrize("ld_dtype", [np.float32, np.int8, np.int16])
@pytest.mark.parametrize("low_memory", [False, True])
@pytest.mark.parametrize("m, wl, wr, jitter", [(700, 23, 23, 0), (1500, 130, 70, 40), (64, 5, 9, 3), (333, 400, 400, 0),
                                               (2100, 64, 63, 0)])
def test_banded_components_bit_exact(gpu, m, wl, wr, jitter, low_memory, ld_dtype):
    ld = banded_ld(m, wl, wr, low_memory, ld_dtype, seed=m, jitter=jitter)
    ss, inp = _inputs(m)
    st0 = inp.state_copy()
    ref = H.run_oracle(ld, inp, st0, sweeps=2)
    got = H.run_hip(ld, inp, st0, sweeps=2)
    H.assert_state_equal(got, ref)


@pytest.mark.parametrize("low_memory", [False, True])
def test_band_kernel_equals_generic_kernel(gpu, low_memory, monkeypatch):
    ld = banded_ld(2500, 90, 140, low_memory, np.float32, seed=5, jitter=60)
    ss, inp = _inputs(2500, seed=4)
    st0 = inp.state_copy()
    band = H.run_hip(ld, inp, st0, sweeps=3)
    monkeypatch.setenv("VIPRS_BAND", "0")
    generic = H.run_hip(ld, inp, st0, sweeps=3)
    H.assert_state_equal(band, generic)


@pytest.mark.parametrize("low_memory", [False, True])
def test_banded_next_to_dense_blocks(gpu, low_memory):
    """One plan with dense blocks (panel kernels) and windowed components (band kernel)."""
    dense = syn.make_ld([300, 70, 1400], low_memory=low_memory, seed=3)
    band = banded_ld(900, 40, 40, low_memory, np.float32, seed=8)
    m0 = dense.m
    lb = np.concatenate([dense.ld_left_bound, band.ld_left_bound + m0]).astype(np.int32)
    ip = np.concatenate([dense.ld_indptr, band.ld_indptr[1:] + dense.ld_indptr[-1]]).astype(np.int64)
    data = np.concatenate([dense.ld_data, band.ld_data])
    ld = syn.SyntheticLD(lb, ip, data, np.array([0, m0 + 900]), np.zeros(1), bool(low_memory), 1.0)
    ss, inp = _inputs(m0 + 900, seed=6)
    st0 = inp.state_copy()
    H.assert_state_equal(H.run_hip(ld, inp, st0, sweeps=2), H.run_oracle(ld, inp, st0, sweeps=2))


@pytest.mark.parametrize("low_memory", [False, True])
def test_banded_grid_models(gpu, low_memory):
    from viprs_amd.vi import e_step_hip as S
    ld = banded_ld(800, 50, 50, low_memory, np.float32, seed=2)
    ss, inp = _inputs(800, seed=9)
    g, st0 = _grid_inputs(ld, ss, 6)
    active = np.array([5, 0, 3], dtype=np.int32)
    out = {}
    for name, mod in (("ref", O), ("hip", S)):
        st = {k: v.copy(order="F") for k, v in st0.items()}
        for _ in range(2):
            mod.cpp_e_step_grid(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, inp.std_beta, st["var_gamma"], st["var_mu"],
                                st["eta"], st["q"], st["eta_diff"], g["u_logs"], g["hvt"], g["mu_mult"], ld.dq_scale,
                                active, 1, low_memory)
        out[name] = st
    H.assert_state_equal(out["hip"], out["ref"])


@pytest.mark.parametrize("ld_dtype", [np.float32, np.int8])
@pytest.mark.parametrize("low_memory", [False, True])
def test_band_kernel_on_dense_blocks(gpu, low_memory, ld_dtype, monkeypatch):
    """VIPRS_NO_DENSE=1 classifies every LD block as a windowed component: the band kernel then serves
    LDetect-style dense blocks (window = whole block) and must reproduce the oracle bit for bit."""
    monkeypatch.setenv("VIPRS_NO_DENSE", "1")
    ld, ss, inp = syn.make_problem(sizes=[700, 64, 1, 333, 1300, 65], low_memory=low_memory, ld_dtype=ld_dtype, seed=13)
    st0 = inp.state_copy()
    H.assert_state_equal(H.run_hip(ld, inp, st0, sweeps=2), H.run_oracle(ld, inp, st0, sweeps=2))


@pytest.mark.parametrize("low_memory", [False, True])
def test_fit_on_windowed_ld(gpu, low_memory):
    """VIPRS.fit with the whole EM iteration on the device, two chromosomes of banded LD in one merged plan
    (band kernel), against the same fit driven by the oracle through the host logic."""
    from viprs_amd.data import ArrayDataLoader, LDArrays, SumstatsArrays
    from viprs_amd.model import VIPRS
    ld, ss = {}, {}
    for chrom, m in ((21, 900), (22, 1500)):
        band = banded_ld(m, 45, 45, low_memory, np.float32, seed=chrom)
        form = (band.ld_left_bound, band.ld_indptr, band.ld_data)
        ld[chrom] = LDArrays(symmetric=None if low_memory else form, upper=form if low_memory else None,
                             stored_dtype=np.float32)
        s, _ = _inputs(m, seed=chrom)
        ss[chrom] = SumstatsArrays(s.std_beta, s.n_per_snp)
    gdl = ArrayDataLoader(ld, ss, n=1e5)
    theta = {"pi": 0.02, "sigma_epsilon": 0.9}
    hip = VIPRS(gdl, low_memory=low_memory)
    hip.fit(max_iter=25, theta_0=dict(theta))
    assert hip._merged and hip._plans["*"].info(__import__("viprs_amd._lib", fromlist=["x"]).INFO_N_RAGGED) == 2
    ref = VIPRS(gdl, low_memory=low_memory, e_step_fn=O.cpp_e_step)
    ref.fit(max_iter=25, theta_0=dict(theta))
    assert hip.optim_result.nit == ref.optim_result.nit
    np.testing.assert_allclose(hip.history["ELBO"], ref.history["ELBO"], rtol=1e-7, atol=0.02)
    for c in hip.chromosomes:
        np.testing.assert_allclose(hip.pip[c], ref.pip[c], rtol=1e-3, atol=1e-6)
        np.testing.assert_allclose(hip.post_mean_beta[c], ref.post_mean_beta[c], rtol=1e-3, atol=1e-7)


@pytest.mark.parametrize("K", [1, 3, 8, 10])          # K <= 8: band kernel (components serial per lane); K = 10: generic kernel
@pytest.mark.parametrize("low_memory", [False, True])
def test_banded_mixture(gpu, low_memory, K, monkeypatch):
    """The mixture model on windowed components: same bits as the oracle and as the row-by-row generic kernel."""
    from tests.test_gpu_models import _run_mix
    from tests.test_oracle_vs_ref import _mixture_inputs
    from viprs_amd.vi import e_step_hip as S
    ld = banded_ld(600, 30, 30, low_memory, np.float32, seed=12)
    ss, inp = _inputs(600, seed=14)
    mix, st0 = _mixture_inputs(ld, ss, K)
    got = _run_mix(S, ld, inp, mix, st0, 2)
    H.assert_state_equal(got, _run_mix(O, ld, inp, mix, st0, 2))
    monkeypatch.setenv("VIPRS_BAND", "0")
    H.assert_state_equal(got, _run_mix(S, ld, inp, mix, st0, 2))
