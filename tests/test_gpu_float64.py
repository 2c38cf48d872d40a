"""GPU: float64 state (float_precision='float64', VIPRS.py:72) -- the panel-walking kernels of estep_tile.h (and the row-by-row
kernels of estep_generic.h) against the oracle run in double: BIT-IDENTICAL (`==` on all five state arrays, round 5), both LD
forms, every LD dtype, all three models.  The device evaluates glibc's own double exp (device_math.h: exp_glibc_f64_*,
constants read from the host's libm; the model is pinned against exp() on the CPU, tests/test_oracle_vs_ref.py), IEEE
divides, every fma where the reference has one, and the second pass of the upper-triangular form (update_q_factor,
e_step.hpp:331-337) sums a row's products in the reference's column order (one lane per row).
`math_mode="fast"` changes nothing for a float64 state (round 6): it has no fast kernels, every result stays `==` and the plan
reports exact arithmetic."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H
from tests.test_band import banded_ld
from tests.test_oracle_vs_ref import _grid_inputs
from viprs_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu
RTOL_F64 = 1e-10
T = np.float64


def assert_state_close_f64(got, ref, rtol=RTOL_F64):
    """1e-10 relative, entries below 1e-4 of a vector's largest magnitude held to that floor: var_mu = mu_mult (beta -
    q) and eta_diff = gamma mu - eta are differences of O(scale) quantities, their small entries carry the absolute
    rounding noise of q (a 2 500-term fma chain seeded by an exp that is not glibc's) -- ~1e-15 absolute here."""
    assert H.branch_flips(got, ref) == 0
    for k in H.STATE:
        assert_close_f64(got[k], ref[k], k, rtol)


def check_state_f64(got, ref, low_memory):
    """`==`, both LD forms (module docstring)."""
    H.assert_state_equal(got, ref)


def assert_close_f64(got, ref, what, rtol=RTOL_F64):
    scale = float(np.max(np.abs(ref)))
    tol = rtol * np.maximum(np.abs(ref), 1e-4 * scale)
    bad = np.abs(got - ref) > tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.size} beyond {rtol}; worst {np.max(np.abs(got - ref))}"


@pytest.mark.parametrize("ld_dtype", [np.int8, np.int16, np.float32, np.float64, np.int32])
@pytest.mark.parametrize("low_memory", [False, True])
def test_float64_spike_slab_matches_oracle_on_far_field_ld(gpu, low_memory, ld_dtype):
    """Blocks of 1 .. 40 panels (ragged last panels, more than 1 024 columns: several column passes per panel) on
    long-range LD; every stored entry matters (the probe with the far field cut away must differ)."""
    ld, ss, inp = syn.make_problem(sizes=[70, 1400, 333, 2500, 64, 1], low_memory=low_memory, ld_dtype=ld_dtype, seed=41,
                                   kind="longrange", float_precision=T)
    st0 = inp.state_copy()
    ref = H.run_oracle(ld, inp, st0, sweeps=3)
    got = H.run_hip(ld, inp, st0, sweeps=3)
    assert got["q"].dtype == np.float64
    check_state_f64(got, ref, low_memory)
    cut = H.run_oracle(H.cut_far_field(ld), inp, st0, sweeps=3)
    assert np.max(np.abs(cut["q"] - ref["q"])) > 1e-4 * np.max(np.abs(ref["q"]))


@pytest.mark.parametrize("low_memory", [False, True])
def test_float64_tile_kernel_equals_row_by_row_kernel(gpu, low_memory, monkeypatch):
    ld, ss, inp = syn.make_problem(sizes=[700, 130, 1100], low_memory=low_memory, ld_dtype=np.int8, seed=42,
                                   kind="longrange", float_precision=T)
    st0 = inp.state_copy()
    tile = H.run_hip(ld, inp, st0, sweeps=2)
    monkeypatch.setenv("VIPRS_F64_ROW_BY_ROW", "1")
    rows = H.run_hip(ld, inp, st0, sweeps=2)
    check_state_f64(tile, rows, low_memory)


@pytest.mark.parametrize("ld_dtype", [np.float32, np.int8])
@pytest.mark.parametrize("low_memory", [False, True])
@pytest.mark.parametrize("m, wl, wr, jitter", [(700, 23, 23, 0), (1500, 130, 70, 40), (64, 5, 9, 3), (333, 400, 400, 0)])
def test_float64_windowed_components(gpu, m, wl, wr, jitter, low_memory, ld_dtype):
    ld = banded_ld(m, wl, wr, low_memory, ld_dtype, seed=m, jitter=jitter)
    rng = np.random.default_rng(1)
    beta = rng.standard_normal(m) * 0.004
    beta[rng.integers(0, m, max(1, m // 50))] += 0.05
    ss = syn.SyntheticSumstats(beta.astype(T), np.full(m, 1e5), np.zeros(m, T), 1e5)
    inp = syn.make_inputs(ss, float_precision=T)
    st0 = inp.state_copy()
    ref = H.run_oracle(ld, inp, st0, sweeps=2)
    got = H.run_hip(ld, inp, st0, sweeps=2)
    check_state_f64(got, ref, low_memory)


@pytest.mark.parametrize("low_memory", [False, True])
def test_float64_grid_matches_oracle(gpu, low_memory):
    from viprs_amd.vi import e_step_hip as S
    ld, ss, inp = syn.make_problem(sizes=[130, 1300, 64, 1900], low_memory=low_memory, seed=33, kind="longrange",
                                   float_precision=T)           # 1 900: the 8-wave class beside the 4-wave one
    g, st0 = _grid_inputs(ld, ss, 12, T=T)
    active = np.array([11, 0, 7, 8, 3], dtype=np.int32)
    out = {}
    for name, mod in (("ref", O), ("hip", S)):
        st = {k: v.copy(order="F") for k, v in st0.items()}
        for _ in range(2):
            mod.cpp_e_step_grid(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, inp.std_beta, st["var_gamma"], st["var_mu"],
                                st["eta"], st["q"], st["eta_diff"], g["u_logs"], g["hvt"], g["mu_mult"], ld.dq_scale,
                                active, 1, low_memory)
        out[name] = st
    check_state_f64(out["hip"], out["ref"], low_memory)
    untouched = [c for c in range(12) if c not in active]
    assert np.all(out["hip"]["eta"][:, untouched] == 0)


def test_float64_skip_branch_counts(gpu):
    """Ten sweeps in, about half of the SNPs take the skip branch (|eta_diff| < 1e-8, e_step.hpp:410): they keep
    var_mu / var_gamma / eta and get eta_diff = 0, the same SNPs as in the oracle."""
    ld, ss, inp = syn.make_problem(sizes=[300, 90], low_memory=False, seed=43, kind="longrange", float_precision=T)
    st0 = inp.state_copy()
    ref = H.run_oracle(ld, inp, st0, sweeps=10)
    got = H.run_hip(ld, inp, st0, sweeps=10)
    skipped = ref["eta_diff"] == 0
    assert 0.3 < np.mean(skipped) < 0.9                    # both branches are exercised
    assert np.array_equal(got["eta_diff"] == 0, skipped)
    H.assert_state_equal(got, ref)


def test_float64_chain_sigmoid_equals_the_host_and_an_80_bit_reference(gpu):
    """The chain's own exp / divide (estep_tile.h: glibc's double exp with the table across the lanes, IEEE divide): 1-SNP
    blocks with mu_mult = std_beta = 1, sqrt_half_var_tau = 0 make var_gamma = sigmoid(u_logs) and eta_diff = var_gamma --
    `==` the oracle (the host's exp and `/`) over the whole argument range, subnormal results of exp (x < -708) and the
    special ranges included, and within 1 ulp of numpy's long double."""
    if np.finfo(np.longdouble).eps > 1e-18:
        pytest.skip("no extended-precision long double on this host")
    rng = np.random.default_rng(9)
    x = np.concatenate([np.linspace(-760, 760, 40001), rng.standard_normal(40000) * 30, rng.standard_normal(20000),
                        [0.0, -0.0, 1e-300, -1e-300, 708.0, -708.0, -708.5, 709.0, -745.2, 1e6, -1e6]])
    m = x.size
    ld = syn.SyntheticLD(np.arange(m, dtype=np.int32), np.arange(m + 1, dtype=np.int64), np.ones(m, np.float32),
                         np.arange(m + 1), np.zeros(m), False, 1.0)
    ss = syn.SyntheticSumstats(np.ones(m), np.full(m, 1e5), np.zeros(m), 1e5)
    inp = syn.make_inputs(ss, float_precision=T)
    inp.mu_mult[:] = 1.0
    inp.sqrt_half_var_tau[:] = 0.0
    inp.u_logs[:] = x
    got = H.run_hip(ld, inp, inp.state_copy(), sweeps=1)
    H.assert_state_equal(got, H.run_oracle(ld, inp, inp.state_copy(), sweeps=1))
    xl = x.astype(np.longdouble)
    e = np.exp(-np.abs(xl))
    ref = np.where(xl < 0, e, np.longdouble(1)) / (1 + e)
    applied = got["eta_diff"] != 0
    assert np.array_equal(applied, np.abs(ref.astype(T)) >= 1e-8)          # the skip branch (|eta_diff| < 1e-8)
    g = got["var_gamma"][applied].astype(np.longdouble)
    err = np.abs(g - ref[applied]) / ref[applied]
    assert float(err.max()) < 2 * np.finfo(T).eps, float(err.max())       # exp: < 1 ulp, the divide: 0.5 ulp
    np.testing.assert_array_equal(got["eta_diff"][applied], got["var_gamma"][applied])
    assert np.all(got["var_mu"][applied] == 1.0) and np.all(got["q"] == 0)


@pytest.mark.parametrize("low_memory", [False, True])
def test_float64_cfg3_full_size(gpu, low_memory):
    """BASELINE configs[2] (1.1 M SNPs, 1 700 blocks, int8 long-range LD) with a float64 state: the two block classes of
    the launch (8-wave workgroups for the largest blocks beside 4-wave ones, two streams) leave a state that is
    reproducible bit for bit from run to run, whose skip count agrees with it, and that equals the oracle run in double on
    a sample of blocks from both classes (largest, smallest, around the class limit, the median)."""
    from viprs_amd.plan import DeviceState, LDPlan
    ld, ss, inp = syn.make_problem("cfg3", low_memory=low_memory, ld_dtype=np.int8, kind="longrange", float_precision=T)
    plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, low_memory)
    state = DeviceState(plan, "float64", "spike_slab", 1)
    for name in ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult"):
        state.upload(name, getattr(inp, name))
    outs = []
    for _ in range(2):
        state.reset(inp.pi)
        state.e_step(ld.dq_scale)
        state.e_step(ld.dq_scale)                       # two sweeps: the second starts from a state with history
        outs.append({k: state.download(k) for k in H.STATE})
    H.assert_state_equal(outs[0], outs[1])
    got = outs[0]
    assert plan.last_skipped() == int((got["eta_diff"] == 0).sum())
    sizes = np.diff(ld.block_start)
    order = np.argsort(sizes)
    limit = int(np.searchsorted(sizes[order], 1792))
    sample = list(order[:2]) + list(order[-2:]) + list(order[limit - 2:limit + 2]) + [order[len(order) // 2]]
    for bi in sample:
        s, e = int(ld.block_start[bi]), int(ld.block_start[bi + 1])
        b = e - s
        lo, hi = int(ld.ld_indptr[s]), int(ld.ld_indptr[e])
        sub = syn.SyntheticLD(ld.ld_left_bound[s:e] - s, ld.ld_indptr[s:e + 1] - lo, ld.ld_data[lo:hi], np.array([0, b]),
                              np.zeros(1), low_memory, ld.dq_scale)
        st = {k: v[s:e].copy() for k, v in inp.state_copy().items()}
        for _ in range(2):
            O.cpp_e_step(sub.ld_left_bound, sub.ld_indptr, sub.ld_data, inp.std_beta[s:e].copy(), st["var_gamma"],
                         st["var_mu"], st["eta"], st["q"], st["eta_diff"], inp.u_logs[s:e].copy(),
                         inp.sqrt_half_var_tau[s:e].copy(), inp.mu_mult[s:e].copy(), ld.dq_scale, 1, low_memory)
        check_state_f64({k: got[k][s:e] for k in H.STATE}, st, low_memory)
    state.close()
    plan.close()


@pytest.mark.parametrize("low_memory", [False, True])
# K <= 4 and 5 .. 10: the panel-walking kernel (estep_tile.h; 10 = the reference's own test, tests/test_basic.py:60); 12: row by row
@pytest.mark.parametrize("K", [1, 3, 4, 6, 10, 12])
def test_float64_mixture_matches_oracle(gpu, K, low_memory):
    from tests.test_gpu_models import _run_mix
    from tests.test_oracle_vs_ref import _mixture_inputs
    from viprs_amd.vi import e_step_hip as S
    ld, ss, inp = syn.make_problem(sizes=[70, 1400, 333, 2000], low_memory=low_memory, ld_dtype=np.int8, seed=31,
                                   kind="longrange", float_precision=T)
    mix, st0 = _mixture_inputs(ld, ss, K, T=T)
    ref = _run_mix(O, ld, inp, mix, st0, 2)
    got = _run_mix(S, ld, inp, mix, st0, 2)
    assert got["var_gamma"].dtype == np.float64 and got["var_gamma"].shape == (ld.m, K)
    for k in H.STATE:
        assert np.array_equal(got[k], ref[k]), f"{k}: {int((got[k] != ref[k]).sum())} entries differ"
    cut = _run_mix(O, H.cut_far_field(ld), inp, mix, st0, 2)
    assert np.max(np.abs(cut["q"] - ref["q"])) > 1e-4 * np.max(np.abs(ref["q"]))


@pytest.mark.parametrize("low_memory", [False, True])
@pytest.mark.parametrize("sizes", [[2000, 1900], [100, 64, 65, 1, 300], [1792, 1791]],
                         ids=["big-class-only", "small-class-only", "at-the-class-limit"])
def test_float64_block_classes(gpu, sizes, low_memory):
    """The float64 launch runs blocks >= 1 792 SNPs on 8-wave workgroups beside the rest on a second stream; with one
    class empty it is a single launch."""
    ld, ss, inp = syn.make_problem(sizes=sizes, low_memory=low_memory, ld_dtype=np.int8, seed=44, kind="longrange", float_precision=T)
    st0 = inp.state_copy()
    check_state_f64(H.run_hip(ld, inp, st0, sweeps=3), H.run_oracle(ld, inp, st0, sweeps=3), low_memory)


@pytest.mark.parametrize("ld_dtype", [np.int8, np.float32])
def test_float64_ignores_fast_mode(gpu, ld_dtype):
    """math_mode='fast' with a float64 state: there are no fast float64 kernels -- the sweep AND the upper-triangular form's
    second pass are the exact ones (rounds 3-5 swapped in a lane-order second pass, 1e-10 from the reference, which made the
    warning "this model runs in EXACT arithmetic" untrue): `==`, and the plan reports exact arithmetic."""
    from viprs_amd.vi import e_step_hip as S
    ld, ss, inp = syn.make_problem(sizes=[70, 1400, 333, 2000], low_memory=True, ld_dtype=ld_dtype, seed=47, kind="longrange",
                                   float_precision=T)
    st0 = inp.state_copy()
    ref = H.run_oracle(ld, inp, st0, sweeps=2)
    S.clear_plan_cache()
    S.set_default_math_mode("fast")
    try:
        got = H.run_hip(ld, inp, st0, sweeps=2)
        plan = S.plan_for(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, True)
        assert plan.effective_math_mode() == "exact"
    finally:
        S.set_default_math_mode("exact")
        S.clear_plan_cache()
    H.assert_state_equal(got, ref)
