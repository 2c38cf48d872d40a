"""Symmetric plans built from the compact upper-triangular LD store (`viprs_plan_create_expanded`,
`LDPlan.from_upper`): what the reference does on the host with `ld_mat.load(return_symmetric=True)`
(viprs/model/VIPRS.py:167-172) happens on the device."""
import numpy as np
import pytest

from tests import helpers as H
from viprs_amd.data import mirror_upper_ld
from viprs_amd.utils import synthetic as syn


def _banded_upper(m, w, dtype, seed=0):
    """Upper-triangular store of a banded matrix: row j holds columns j+1 .. min(j+w, m-1)."""
    rng = np.random.default_rng(seed)
    length = np.minimum(np.arange(m) + w, m - 1) - np.arange(m)
    ip = np.concatenate([[0], np.cumsum(length)]).astype(np.int64)
    if np.issubdtype(np.dtype(dtype), np.integer):
        data = rng.integers(-40, 41, int(ip[-1])).astype(dtype)
    else:
        data = (rng.uniform(-0.3, 0.3, int(ip[-1]))).astype(dtype)
    return ip, data


def _dense_from_upper(ip, data, diag):
    m = ip.shape[0] - 1
    R = np.zeros((m, m), dtype=data.dtype)
    for j in range(m):
        n = int(ip[j + 1] - ip[j])
        R[j, j + 1:j + 1 + n] = data[ip[j]:ip[j + 1]]
    R = R + R.T
    R[np.arange(m), np.arange(m)] = diag
    return R


@pytest.mark.parametrize("ld_dtype", [np.float32, np.int8, np.int16])
def test_host_model_reproduces_the_symmetric_generator(ld_dtype):
    sizes = [1, 2, 63, 64, 65, 500, 3, 129, 1]
    up = syn.make_ld(sizes, low_memory=True, ld_dtype=ld_dtype)
    sym = syn.make_ld(sizes, low_memory=False, ld_dtype=ld_dtype)
    lb, ip, data = mirror_upper_ld(up.ld_indptr, up.ld_data)
    assert np.array_equal(lb, sym.ld_left_bound)
    assert np.array_equal(ip, sym.ld_indptr)
    assert np.array_equal(data, sym.ld_data)


def test_host_model_banded_windows_match_the_dense_matrix():
    ip_u, data_u = _banded_upper(90, 7, np.float32)
    lb, ip, data = mirror_upper_ld(ip_u, data_u)
    R = _dense_from_upper(ip_u, data_u, 1.0)
    for j in range(90):
        n = int(ip[j + 1] - ip[j])
        assert np.array_equal(data[ip[j]:ip[j + 1]], R[j, lb[j]:lb[j] + n])
        assert not R[j, :lb[j]].any() and not R[j, lb[j] + n:].any()


def test_host_model_rejects_windows_that_do_not_mirror():
    ip_u = np.array([0, 3, 3, 4, 4, 4], np.int64)      # row 0 reaches SNP 3, row 1 reaches nothing
    with pytest.raises(ValueError):
        mirror_upper_ld(ip_u, np.zeros(4, np.float32))


# ---- device ------------------------------------------------------------------------------------------
def _sweeps_on(plan, ld_sym, inp, st0, sweeps=2):
    st = {k: v.copy() for k, v in st0.items()}
    for _ in range(sweeps):
        plan.e_step(inp.std_beta, st["var_gamma"], st["var_mu"], st["eta"], st["q"], st["eta_diff"], inp.u_logs,
                    inp.sqrt_half_var_tau, inp.mu_mult, ld_sym.dq_scale)
    return st


@pytest.mark.gpu
@pytest.mark.parametrize("ld_dtype", [np.float32, np.int8, np.int16, np.float64])
def test_expanded_plan_equals_symmetric_plan_and_oracle(gpu, ld_dtype):
    """Dense blocks of every size class (single workgroup, 2-CU and 8-CU teams)."""
    from viprs_amd.plan import LDPlan
    sizes = [2500, 1300, 70, 1, 64, 333, 2]
    sym, ss, inp = syn.make_problem(sizes=sizes, low_memory=False, ld_dtype=ld_dtype, seed=21)
    up = syn.make_ld(sizes, low_memory=True, ld_dtype=ld_dtype, seed=21)
    plan = LDPlan.from_upper(up.ld_indptr, up.ld_data)
    try:
        lb, ip = plan.windows()
        assert np.array_equal(lb, sym.ld_left_bound) and np.array_equal(ip, sym.ld_indptr)
        assert plan.nnz == sym.ld_data.shape[0]
        st0 = inp.state_copy()
        got = _sweeps_on(plan, sym, inp, st0)
    finally:
        plan.close()
    ref = H.run_oracle(sym, inp, st0, sweeps=2)
    H.assert_state_equal(got, ref)
    H.assert_state_equal(got, H.run_hip(sym, inp, st0, sweeps=2))


@pytest.mark.gpu
@pytest.mark.parametrize("ld_dtype", [np.float32, np.int8])
def test_expanded_banded_windows(gpu, ld_dtype):
    """Ragged (banded) windows: the mirrored rows feed the generic kernel."""
    from oracle import oracle as O
    from viprs_amd.plan import LDPlan
    m = 700
    ip_u, data_u = _banded_upper(m, 23, ld_dtype, seed=4)
    lb, ip, data = mirror_upper_ld(ip_u, data_u)
    dq = 1.0 / 127 if ld_dtype == np.int8 else 1.0
    ld = syn.SyntheticLD(lb, ip, data, np.array([0, m]), np.zeros(1), False, dq)
    ss = syn.make_sumstats(syn.make_ld([m], low_memory=False, seed=2), seed=8)
    inp = syn.make_inputs(ss)
    st0 = inp.state_copy()
    plan = LDPlan.from_upper(ip_u, data_u)
    try:
        wl, wi = plan.windows()
        assert np.array_equal(wl, lb) and np.array_equal(wi, ip)
        got = _sweeps_on(plan, ld, inp, st0)
    finally:
        plan.close()
    H.assert_state_equal(got, H.run_oracle(ld, inp, st0, sweeps=2))


@pytest.mark.gpu
def test_expanded_plan_edge_cases(gpu):
    from viprs_amd.plan import LDPlan
    plan = LDPlan.from_upper(np.zeros(1, np.int64), np.zeros(0, np.float32))      # empty chromosome
    assert plan.m == 0 and plan.nnz == 0
    plan.close()
    plan = LDPlan.from_upper(np.zeros(6, np.int32), np.zeros(0, np.int8))         # 5 unlinked SNPs: diagonal only
    lb, ip = plan.windows()
    assert np.array_equal(lb, np.arange(5)) and np.array_equal(ip, np.arange(6)) and plan.nnz == 5
    plan.close()
    with pytest.raises(ValueError):                                                # rows that do not mirror
        LDPlan.from_upper(np.array([0, 3, 3, 4, 4, 4], np.int64), np.zeros(4, np.float32))
    with pytest.raises(ValueError):                                                # window past the last SNP
        LDPlan.from_upper(np.array([0, 1, 3], np.int64), np.zeros(3, np.float32))
    with pytest.raises(ValueError):
        LDPlan.from_upper(np.array([0, 2, 3, 3], np.int64), np.zeros(2, np.float32))   # indptr[-1] != len(data)


# ---- VIPRS.fit on an LD source that only has the upper-triangular store -----------------------------
_CHROMS = {1: [300, 90, 1, 64], 2: [129, 500], 3: [70]}


def _fit_pair(ld_dtype, **kw):
    from viprs_amd.data import ArrayDataLoader
    from viprs_amd.model.VIPRS import VIPRS
    both = ArrayDataLoader.synthetic(_CHROMS, ld_dtype=ld_dtype, seed=31)
    upper_only = ArrayDataLoader.synthetic(_CHROMS, ld_dtype=ld_dtype, seed=31, forms=("upper",))
    common = dict(low_memory=False, dequantize_on_the_fly=np.issubdtype(np.dtype(ld_dtype), np.integer), **kw)
    ref = VIPRS(both, **common)
    exp = VIPRS(upper_only, **common)
    with pytest.raises(ValueError):
        VIPRS(upper_only, expand_ld_on_device=False, **common)
    for model in (ref, exp):
        model.fit(max_iter=12, theta_0={"pi": 0.02, "sigma_epsilon": 0.85})
    return ref, exp


def _assert_same_fit(ref, exp):
    assert np.array_equal(ref.history["ELBO"], exp.history["ELBO"])
    for c in ref.chromosomes:
        for name in ("var_gamma", "var_mu", "q", "eta"):
            assert np.array_equal(getattr(ref, name)[c], getattr(exp, name)[c]), (c, name)


@pytest.mark.parametrize("ld_dtype", [np.float32, np.int8])
def test_fit_from_upper_store_host_logic(ld_dtype):
    from oracle import oracle as O
    ref, exp = _fit_pair(ld_dtype, e_step_fn=O.cpp_e_step)
    _assert_same_fit(ref, exp)


@pytest.mark.gpu
@pytest.mark.parametrize("merge", [True, False])
@pytest.mark.parametrize("ld_dtype", [np.float32, np.int8])
def test_fit_from_upper_store_hip(gpu, ld_dtype, merge):
    ref, exp = _fit_pair(ld_dtype, merge_chromosomes=merge)
    assert exp._expanded and not ref._expanded
    _assert_same_fit(ref, exp)
