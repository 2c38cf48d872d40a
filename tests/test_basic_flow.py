"""The reference's own smoke test (tests/test_basic.py: TestVIPRS / TestVIPRSMix / TestVIPRSGrid), replayed on
this implementation: same initialisation checks, `fit(max_iter=10)`, the same reporting calls, model selection
and averaging on the grid -- on a synthetic data set (the reference downloads 1000G / UKB data through magenpy).
CPU: the E-step goes through the oracle (`e_step_fn`); `-m gpu`: through the HIP kernels."""
from functools import partial

import numpy as np
import pytest

from oracle import oracle as O
from viprs_amd.data import ArrayDataLoader
from viprs_amd.model import VIPRS, VIPRSGrid, VIPRSMix
from viprs_amd.model.gridsearch import HyperparameterGrid, bayesian_model_average, select_best_model

CHROM = 22


@pytest.fixture(scope="module")
def gdl_object():
    return ArrayDataLoader.synthetic({CHROM: [180, 75, 130]}, seed=123)


def _grid():
    grid = HyperparameterGrid()
    grid.generate_pi_grid(steps=4)
    return grid


def _check_init(model, gdl, wide):
    assert model.m == gdl.m
    model.initialize()
    assert model.std_beta[CHROM].shape == (model.m,)
    assert model.ld_indptr[CHROM].shape == (model.m + 1,)
    assert model.ld_left_bound[CHROM].shape == (model.m,)
    assert model.ld_data[CHROM].shape == (model.ld_indptr[CHROM][-1],)
    assert np.all((0.0 < np.asarray(model.pi)) & (np.asarray(model.pi) < 1.0)) and 0.0 < np.sum(model.pi) < 1.0
    assert 0.0 < model.sigma_epsilon < 1.0 and np.all(np.asarray(model.tau_beta) > 0.0)
    shape = model.shapes[CHROM] if not wide else (model.m, wide)
    for p in (model.var_gamma, model.var_mu, model.var_tau):
        assert p[CHROM].shape == (shape if wide else (model.m,))
    for p in (model.q, model.eta):
        assert p[CHROM].shape == (model.m,)


def _check_fitted(model, width=None):
    shape = (model.m,) if width is None else (model.m, width)
    for p in (model.pip, model.post_mean_beta, model.post_var_beta):
        assert p[CHROM].shape == shape
    model.to_table()
    model.to_theta_table()
    model.to_history_table()
    for v in (model.mse(), model.log_prior(), model.loglikelihood(), model.entropy()):
        assert np.all(np.isfinite(v))


def _run_all(gdl, kw_ss, kw_mix):
    m = VIPRS(gdl, **kw_ss)
    _check_init(m, gdl, None)
    m.fit(max_iter=10)
    _check_fitted(m)

    mix = VIPRSMix(gdl, K=10, **kw_mix)
    assert mix.n_per_snp[CHROM].shape == (mix.m, 1)
    _check_init(mix, gdl, 10)
    mix.fit(max_iter=10)
    for p in (mix.var_gamma, mix.var_mu, mix.var_tau):
        assert p[CHROM].shape == (mix.m, 10)
    _check_fitted(mix)

    vb = {CHROM: (gdl.sumstats_table[CHROM].standardized_beta if hasattr(gdl.sumstats_table[CHROM], "standardized_beta")
                  else m.std_beta[CHROM]).astype(np.float32)}
    for criterion in (partial(select_best_model, criterion="ELBO"),
                      partial(select_best_model, criterion="pseudo_validation"), bayesian_model_average):
        g = VIPRSGrid(gdl, _grid(), **kw_ss)
        g._reset_search()
        g.validation_std_beta = vb
        g.fit(max_iter=10)
        _check_fitted(g, g.n_models)
        assert np.all(np.isfinite(g.pseudo_validate()))
        criterion(g)
        g.fit(max_iter=10)                       # the selected / averaged model keeps fitting as a single model
        for p in (g.pip, g.post_mean_beta, g.post_var_beta):
            assert p[CHROM].shape == (g.m,)
        g.to_table()
        g.to_theta_table()
        g.to_history_table()
        for v in (g.mse(), g.log_prior(), g.loglikelihood(), g.entropy(), g.pseudo_validate()):
            assert np.all(np.isfinite(v))


def test_reference_basic_flow_cpu(gdl_object):
    _run_all(gdl_object, dict(e_step_fn=O.cpp_e_step), dict(e_step_fn=O.cpp_e_step_mixture))


@pytest.mark.gpu
def test_reference_basic_flow_hip(gpu, gdl_object):
    _run_all(gdl_object, {}, {})


def test_check_support_flags():
    """e_step_cpp.pyx:71-88: the reference reports whether it was built with BLAS / OpenMP."""
    from viprs_amd.vi import e_step_hip as S
    assert S.check_blas_support() in (True, False) and S.check_omp_support() in (True, False)
