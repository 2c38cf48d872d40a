"""GPU: `math_mode="fast"` (v_exp_f32 / v_rcp_f32 instead of the reference's glibc expf and double divide,
e_step.hpp:245-261, :222-241) as a first-class mode -- on LD whose far field matters, team blocks, both LD forms,
int8 LD, the K = 4 mixture and the grid (batched matrix-core kernel and item schedule).

What can be asserted.  The north-star tolerance -- 1e-5 relative per entry -- is asserted where the problem is
well conditioned (analytic AR(1) blocks: `test_fast_math_well_conditioned`).  On far-field LD it cannot hold for ANY
arithmetic other than the bit-identical one, and the tests say so with a measurement instead of a looser constant:
the REFERENCE ITSELF (the oracle, exact arithmetic) moves by up to 7e-4 relative in var_gamma and 4e-3 in eta when its
own input `std_beta` is changed by +-1 ulp (the sigmoid's argument is u^2 + ulog with u^2 up to 10^3, and q is ~10 x
beta - q: a relative change of 1e-7 in q becomes 1e-4 in the logit).  So every far-field case computes that yardstick
with the oracle -- error quantiles of "oracle on inputs perturbed by one ulp" against the oracle -- and asserts that
the fast mode's error quantiles stay within a small factor of it: fast mode is as close to the reference as the
reference is to itself under the smallest possible change of its inputs.  Skip-branch flips (e_step.hpp:410-413) are
counted and bounded the same way.  The exact mode stays the default and stays `==` (tests/test_gpu_farfield.py).
"""
import copy

import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H
from tests.test_gpu_farfield import _run_grid, _run_mix, assert_far_field_matters
from tests.test_oracle_vs_ref import _grid_inputs, _mixture_inputs
from viprs_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu

QUANTILES = (50.0, 99.0, 99.9)
FACTOR = 3.0          # fast-mode error quantiles vs the one-ulp yardstick's
FACTOR_MAX = 10.0     # the maximum is a single entry (whichever value sits closest to 0): a looser factor
FLOOR = 1e-5          # the north-star tolerance: errors below it pass whatever the yardstick says


def rel_err(got, ref):
    g, r = got.astype(np.float64).ravel(), ref.astype(np.float64).ravel()
    scale = float(np.max(np.abs(r))) if r.size else 0.0
    return np.abs(g - r) / np.maximum(np.abs(r), 1e-7 * scale + 1e-300)


def one_ulp(inp, seed=1):
    """The inputs with std_beta moved by one ulp up or down (a random sign per SNP)."""
    out = copy.copy(inp)
    sgn = np.random.default_rng(seed).integers(0, 2, inp.std_beta.shape[0]) * 2 - 1
    up = np.nextafter(inp.std_beta, np.float32(np.inf))
    dn = np.nextafter(inp.std_beta, np.float32(-np.inf))
    out.std_beta = np.where(sgn > 0, up, dn).astype(np.float32)
    return out


def assert_within_one_ulp_yardstick(fast, ref, ulp, what):
    """`fast` (HIP, math_mode=fast) against `ref` (oracle), judged by `ulp` (oracle on one-ulp-perturbed inputs)."""
    report = []
    flips_fast, flips_ulp = H.branch_flips(fast, ref), H.branch_flips(ulp, ref)
    assert flips_fast <= 2 * flips_ulp + 3, f"{what}: {flips_fast} skip-branch flips (one-ulp yardstick: {flips_ulp})"
    for k in H.STATE:
        ef, eu = rel_err(fast[k], ref[k]), rel_err(ulp[k], ref[k])
        for qt in QUANTILES:
            a, b = np.percentile(ef, qt), np.percentile(eu, qt)
            assert a <= max(FACTOR * b, FLOOR), f"{what}: {k} p{qt} error {a:.2e}, one-ulp yardstick {b:.2e}"
        assert ef.max() <= max(FACTOR_MAX * eu.max(), FLOOR), f"{what}: {k} max error {ef.max():.2e}, yardstick {eu.max():.2e}"
        report.append(f"{k} p99.9 {np.percentile(ef, 99.9):.1e}/{np.percentile(eu, 99.9):.1e} n>1e-5 {int((ef > 1e-5).sum())}/{int((eu > 1e-5).sum())}")
    print(f"[fast math] {what}: flips {flips_fast} (yardstick {flips_ulp}); fast/yardstick " + "; ".join(report))


@pytest.fixture
def fast_mode():
    from viprs_amd.vi import e_step_hip as S
    S.set_default_math_mode("fast")
    yield S
    S.set_default_math_mode("exact")


SS_CASES = [
    ([700, 1400, 90], "longrange", np.float32, 2),           # single workgroups
    ([1700, 2400, 65], "longrange", np.float32, 2),          # teams
    ([3619, 650, 1536], "longrange", np.float32, 2),         # cfg3's largest block
    ([6000, 77], "longrange", np.float32, 1),                # BASELINE's clip limit
    ([1700, 650], "sample", np.float32, 2),
    ([2400, 3619], "longrange", np.int8, 2),
]


@pytest.mark.parametrize("low_memory", [False, True], ids=["symmetric", "upper"])
@pytest.mark.parametrize("sizes, kind, ld_dtype, sweeps", SS_CASES,
                         ids=[f"{'-'.join(map(str, c[0]))}_{c[1]}_{np.dtype(c[2]).name}" for c in SS_CASES])
def test_spike_slab_fast_far_field(gpu, fast_mode, sizes, kind, ld_dtype, sweeps, low_memory):
    ld, ss, inp = syn.make_problem(sizes=sizes, low_memory=low_memory, ld_dtype=ld_dtype, seed=61, kind=kind)
    assert_far_field_matters(ld, inp)
    st0 = inp.state_copy()
    ref = H.run_oracle(ld, inp, st0, sweeps=sweeps)
    ulp = H.run_oracle(ld, one_ulp(inp), st0, sweeps=sweeps)
    fast = H.run_hip(ld, inp, st0, sweeps=sweeps)
    assert not all(np.array_equal(fast[k], ref[k]) for k in H.STATE), "fast mode did not run (results are bit-identical)"
    assert_within_one_ulp_yardstick(fast, ref, ulp, f"spike-and-slab {sizes} {kind} {np.dtype(ld_dtype).name} "
                                                    f"{'upper' if low_memory else 'symmetric'}")


@pytest.mark.parametrize("low_memory", [False, True], ids=["symmetric", "upper"])
@pytest.mark.parametrize("sizes, ld_dtype", [([700, 1400, 1700, 2400], np.float32), ([1400, 2400], np.int8)],
                         ids=["f32", "int8"])
def test_mixture_k4_fast_far_field(gpu, fast_mode, sizes, ld_dtype, low_memory):
    ld, ss, inp = syn.make_problem(sizes=sizes, low_memory=low_memory, ld_dtype=ld_dtype, seed=62, kind="longrange")
    mix, st0 = _mixture_inputs(ld, ss, 4)
    ref = _run_mix(O, ld, inp, mix, st0, 2)
    ulp = _run_mix(O, ld, one_ulp(inp), mix, st0, 2)
    fast = _run_mix(fast_mode, ld, inp, mix, st0, 2)
    assert not all(np.array_equal(fast[k], ref[k]) for k in H.STATE)
    assert_within_one_ulp_yardstick(fast, ref, ulp, f"mixture K=4 {sizes} {'upper' if low_memory else 'symmetric'}")


@pytest.mark.parametrize("mfma", ["1", "0"], ids=["mfma", "items"])
@pytest.mark.parametrize("low_memory", [False, True], ids=["symmetric", "upper"])
def test_grid_fast_far_field(gpu, fast_mode, low_memory, mfma, monkeypatch):
    monkeypatch.setenv("VIPRS_GRID_MFMA", mfma)
    ld, ss, inp = syn.make_problem(sizes=[700, 1400, 1700], low_memory=low_memory, seed=63, kind="longrange")
    g, st0 = _grid_inputs(ld, ss, 32)
    active = np.arange(0, 32, 5, dtype=np.int32)
    take = lambda st: {k: v[:, active] for k, v in st.items()}
    ref = take(_run_grid(O, ld, inp, g, st0, active, 2))
    ulp = take(_run_grid(O, ld, one_ulp(inp), g, st0, active, 2))
    fast = take(_run_grid(fast_mode, ld, inp, g, st0, active, 2))
    assert not all(np.array_equal(fast[k], ref[k]) for k in H.STATE)
    assert_within_one_ulp_yardstick(fast, ref, ulp, f"grid G=32 ({'batched' if mfma == '1' else 'items'}) "
                                                    f"{'upper' if low_memory else 'symmetric'}")


@pytest.mark.parametrize("low_memory", [False, True], ids=["symmetric", "upper"])
def test_fast_math_well_conditioned(gpu, fast_mode, low_memory):
    """AR(1) blocks (|q| << |beta|), three sweeps: 99 % of the entries of the posterior arrays (PIPs, posterior means) are
    within the north-star tolerance 1e-5; the tail (a few SNPs with u^2 in the hundreds, whose logit amplifies any
    rounding difference) is judged like the far-field cases, against the one-ulp yardstick."""
    ld, ss, inp = syn.make_problem(sizes=[700, 300, 1700], low_memory=low_memory, seed=12)
    st0 = inp.state_copy()
    ref = H.run_oracle(ld, inp, st0, sweeps=3)
    ulp = H.run_oracle(ld, one_ulp(inp), st0, sweeps=3)
    fast = H.run_hip(ld, inp, st0, sweeps=3)
    for k in ("var_gamma", "var_mu", "eta", "q"):
        e = rel_err(fast[k], ref[k])
        assert np.percentile(e, 99.0) <= 1e-5, f"{k}: p99 {np.percentile(e, 99.0):.2e}"
    assert_within_one_ulp_yardstick(fast, ref, ulp, f"AR(1) {'upper' if low_memory else 'symmetric'}")


def test_fit_fixtures_in_fast_mode(gpu):
    """The fit trajectories captured from the reference (tests/golden/fit_*.npz, fp32 state) reproduced with
    math_mode="fast" at the SAME tolerances as the exact mode: iteration counts, stopping messages, ELBO history
    (2e-7 relative), hyper-parameters, posterior (2e-3) and pseudo-R2."""
    from tests.test_fit import FIT, build_model, check_against_fixture
    n = 0
    for path in FIT:
        fx = np.load(path)
        if "float_precision" in fx and str(fx["float_precision"]) != "float32":
            continue                                   # a float64 state has one arithmetic (estep_tile.h)
        model, theta = build_model(fx, e_step="hip", math_mode="fast")
        assert all(p.math_mode == "fast" for p in model._plans.values())
        model.fit(max_iter=60, theta_0=theta)
        check_against_fixture(model, fx, pi_rtol=2e-3 if int(fx["K"]) else 2e-4)
        n += 1
    assert n >= 7


def test_grid_and_per_chromosome_fits_in_fast_mode(gpu):
    """math_mode="fast" as a fit option for the other two drivers (round 6): the serial and the batched grid fit against
    the `fitgrid_*` fixtures and the lock-step per-chromosome fit against the `fitchr_*` fixtures: `nit`, stopping messages,
    ELBO and hyper-parameters at the exact mode's tolerances, the posterior at them but for isolated skip-branch flips."""
    import os
    from tests import test_grid as TG
    from tests import test_per_chromosome as TP
    from tests.test_fit import loader_from_fixture
    from viprs_amd.model import VIPRSGrid
    for name in ("fitgrid_pathwise", "fitgrid_independent"):
        fx = np.load(os.path.join(TG.HERE, "golden", name + ".npz"))
        gdl = loader_from_fixture(fx)
        model = VIPRSGrid(gdl, TG._grid(fx, gdl.m), low_memory=True, math_mode="fast")
        model.fit(pathwise=bool(fx["pathwise"]), max_iter=80)
        assert all(p.effective_math_mode() == "fast" for p in model._plans.values())
        # Iteration counts, messages, ELBO and hyper-parameters at the exact mode's tolerances.  The posterior: e_step's skip
        # branch (e_step.hpp:410-413) freezes a SNP whose update falls below float epsilon, and a sigmoid that differs in
        # the last place can take the other branch for a single SNP on a single sweep -- one of 5 400 PIPs of
        # `fitgrid_independent` ends 1.2 % off (3e-5 absolute).  So: at most 0.1 % of the entries beyond the exact mode's
        # 5e-3, none beyond 2e-2.
        TG._check(model, fx, rtol_post=2e-2)
        for name_, atol in (("pip", 5e-6), ("post_mean_beta", 5e-7)):
            got, ref = getattr(model, name_)[22], fx[f"{name_}_22"]
            bad = np.abs(got - ref) > atol + 5e-3 * np.abs(ref)
            assert bad.mean() <= 1e-3, f"{name}: {int(bad.sum())} of {bad.size} entries of {name_} beyond the exact mode's tolerance"
    fx = np.load(os.path.join(TG.HERE, "golden", "fitgrid_independent.npz"))
    gdl = loader_from_fixture(fx)
    exact = VIPRSGrid(gdl, TG._grid(fx, gdl.m), low_memory=True).fit(batched=True, max_iter=80)
    fast = VIPRSGrid(gdl, TG._grid(fx, gdl.m), low_memory=True, math_mode="fast").fit(batched=True, max_iter=80)
    assert all(p.effective_math_mode() == "fast" for p in fast._plans.values())
    assert [r.nit for r in fast.optim_results] == [r.nit for r in exact.optim_results]
    # (which of two success rules fires first on the stopping iteration -- ELBO within 1e-6 or max |eta_diff| < 1e-6 -- can
    #  differ: the ELBOs agree to ~5e-9 relative, i.e. ~6e-4 absolute)
    assert [r.success for r in fast.optim_results] == [r.success for r in exact.optim_results]
    np.testing.assert_allclose(fast.model_elbos, exact.model_elbos, rtol=2e-7)
    np.testing.assert_allclose(fast.post_mean_beta[22], exact.post_mean_beta[22], rtol=2e-3, atol=2e-7)
    np.testing.assert_allclose(fast.pip[22], exact.pip[22], rtol=2e-3, atol=2e-6)
    for path in TP.FITCHR:
        fx = np.load(path)
        model = TP.build(fx, e_step="hip", math_mode="fast").fit(max_iter=100, theta_0=TP.theta_of(fx))
        assert model._plans["*"].effective_math_mode() == "fast"
        TP.check_against_fixture(model, fx, device_sums=True, fast=True)


@pytest.mark.parametrize("low_memory", [False, True], ids=["symmetric", "upper"])
@pytest.mark.parametrize("model", ["spike_slab", "mixture", "grid"])
def test_fast_math_windowed_components(gpu, fast_mode, model, low_memory):
    """The band kernel (windowed / banded LD components, estep_band.h) has the same fast policies: spike-and-slab, the
    K = 4 mixture (components evaluated serially in the SNP's lane) and grid columns on a +-130 / 70 ragged band."""
    from tests.test_band import _inputs, banded_ld
    ld = banded_ld(1500, 130, 70, low_memory, np.float32, seed=1500, jitter=40)
    ss, inp = _inputs(ld.m)
    if model == "spike_slab":
        st0 = inp.state_copy()
        ref = H.run_oracle(ld, inp, st0, sweeps=2)
        ulp = H.run_oracle(ld, one_ulp(inp), st0, sweeps=2)
        fast = H.run_hip(ld, inp, st0, sweeps=2)
    elif model == "mixture":
        mix, st0 = _mixture_inputs(ld, ss, 4)
        ref = _run_mix(O, ld, inp, mix, st0, 2)
        ulp = _run_mix(O, ld, one_ulp(inp), mix, st0, 2)
        fast = _run_mix(fast_mode, ld, inp, mix, st0, 2)
    else:
        g, st0 = _grid_inputs(ld, ss, 6)
        active = np.arange(6, dtype=np.int32)
        ref = _run_grid(O, ld, inp, g, st0, active, 2)
        ulp = _run_grid(O, ld, one_ulp(inp), g, st0, active, 2)
        fast = _run_grid(fast_mode, ld, inp, g, st0, active, 2)
    assert not all(np.array_equal(fast[k], ref[k]) for k in H.STATE), "fast mode did not run (results are bit-identical)"
    assert_within_one_ulp_yardstick(fast, ref, ulp, f"band kernel, {model}, {'upper' if low_memory else 'symmetric'}")


def test_effective_math_mode_is_reported(gpu):
    """math_mode='fast' exists for spike-and-slab / grid / K <= 8 mixtures on an fp32 state only; wider mixtures
    and float64 states run exact kernels whatever was asked for, and the plan says so (viprs_plan_last_math_modes)."""
    import warnings
    from viprs_amd.plan import DeviceState, LDPlan
    ld, ss, inp = syn.make_problem(sizes=[300, 90], seed=3)
    plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, False, math_mode="fast")
    assert plan.effective_math_mode() is None
    st = DeviceState(plan, "float32", "spike_slab", 1)
    for name in ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult"):
        st.upload(name, getattr(inp, name))
    st.reset(inp.pi)
    st.e_step(ld.dq_scale, None, sync=True)
    assert plan.effective_math_mode() == "fast"
    plan.set_math_mode("exact")
    st.e_step(ld.dq_scale, None, sync=True)
    assert plan.effective_math_mode() == "exact"
    plan.set_math_mode("fast")
    st.close()
    for K, want in ((4, "fast"), (10, "exact")):
        x = syn.make_mixture_inputs(ss, K)
        pi0 = x.pop("pi")
        sm = DeviceState(plan, "float32", "mixture", K)
        sm.upload("std_beta", inp.std_beta)
        for name, arr in x.items():
            sm.upload(name, arr)
        sm.reset(pi0)
        sm.e_step(ld.dq_scale, None, sync=True)
        assert plan.effective_math_mode() == want, (K, plan.effective_math_mode())
        sm.close()
    sd = DeviceState(plan, "float64", "spike_slab", 1)
    for name in ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult"):
        sd.upload(name, getattr(inp, name).astype(np.float64))
    sd.reset(inp.pi)
    sd.e_step(ld.dq_scale, None, sync=True)
    assert plan.effective_math_mode() == "exact"
    sd.close()
    plan.close()
    # the model classes say it up front
    from viprs_amd.data import ArrayDataLoader, LDArrays, SumstatsArrays
    from viprs_amd.model import VIPRS
    gdl = ArrayDataLoader({1: LDArrays(symmetric=(ld.ld_left_bound, ld.ld_indptr, ld.ld_data), dq_scale=ld.dq_scale)},
                          {1: SumstatsArrays(ss.std_beta, ss.n_per_snp)}, n=float(ss.n))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        VIPRS(gdl, low_memory=False, math_mode="fast", float_precision="float64")
    assert any("EXACT arithmetic" in str(x.message) for x in w)
