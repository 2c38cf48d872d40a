"""Model selection / averaging over a fitted ``VIPRSGrid`` (viprs/model/gridsearch/grid_utils.py).

``select_best_model`` supports the ELBO criterion; the validation criteria of the reference need its
evaluation / prediction stack (out of scope).  Deviation, on purpose: models are ranked by
their own final ELBOs (``VIPRSGrid.model_elbos`` = the ``ELBO`` column of ``validation_result``).  The
reference calls ``VIPRS.elbo()`` on the (m, n_models) arrays (grid_utils.py:38), which sums the
variational terms over ALL models and therefore only differs between models through the
sigma_epsilon term.
"""
import copy

import numpy as np


def select_best_model(viprs_grid_model, validation_gdl=None, criterion="ELBO"):
    if criterion != "ELBO":
        raise NotImplementedError("only the ELBO criterion is available (validation metrics are out of scope)")
    m = viprs_grid_model
    ok = m.valid_terminated_models
    if np.sum(ok) < 2:
        raise ValueError("Less than two models converged successfully. Cannot perform model selection.")
    elbo = np.array(m.model_elbos, dtype=np.float64)
    elbo[~ok] = -np.inf
    best = int(np.argmax(elbo))
    for param in (m.pip, m.post_mean_beta, m.post_var_beta, m.var_gamma, m.var_mu, m.var_tau, m.eta, m.zeta, m.q,
                  m._log_var_tau):
        for c in param:
            param[c] = param[c][:, best]
    for c in m.eta_diff:
        if m.eta_diff[c].ndim == 2:
            m.eta_diff[c] = m.eta_diff[c][:, best]
    m.sigma_epsilon, m._sigma_g = m.sigma_epsilon[best], m._sigma_g[best]
    m.tau_beta, m.pi = m.tau_beta[best], m.pi[best]
    m.n_models = 1
    m.best_model_idx = best
    m.set_fixed_params(m.grid_table.iloc[best].to_dict())
    return m


def bayesian_model_average(viprs_grid_model, normalization="softmax"):
    m = viprs_grid_model
    if m.n_models < 2:
        return m
    if np.sum(m.valid_terminated_models) < 1:
        raise ValueError("No models converged successfully. Cannot average models.")
    keep = np.where(m.valid_terminated_models)[0]
    elbos = np.array(m.model_elbos, dtype=np.float64)[keep]
    if normalization == "softmax":
        w = np.exp(elbos - elbos.max())
        w /= w.sum()
    elif normalization == "sum":
        w = elbos - elbos.min() + 1.0
        w /= w.sum()
    else:
        raise KeyError(f"Normalization scheme not recognized. Valid options are: `softmax`, `sum`. Got: {normalization}")
    for param in (m.var_gamma, m.var_mu, m.var_tau, m.q):
        for c in param:
            param[c] = (param[c][:, keep] * w).sum(axis=1).astype(param[c].dtype)
    m.eta = m.compute_eta()
    m.zeta = m.compute_zeta()
    m.update_posterior_moments()
    m._log_var_tau = {c: np.log(m.var_tau[c]) for c in m.var_tau}
    m.eta_diff = {c: np.zeros_like(e) for c, e in m.eta.items()}
    m.model_weights = w
    # hyper-parameters implied by the averaged posterior (grid_utils.py:176-183)
    fixed = copy.deepcopy(m.fix_params)
    m.fix_params = {}
    m._host_stale = False
    m._sums_valid = False
    m.pi = m.sigma_epsilon = m.tau_beta = None
    m.lambda_min = np.float32(0.0) if not np.isscalar(m.lambda_min) else m.lambda_min
    m.m_step()
    m.fix_params = fixed
    m.n_models = 1
    return m
