"""Model selection / averaging over a fitted ``VIPRSGrid`` (viprs/model/gridsearch/grid_utils.py).

``select_best_model`` supports the ELBO and the pseudo-validation criteria; the `validation` criterion of the
reference needs its genotype prediction stack (out of scope).  Deviation, on purpose: models are ranked by
their own final ELBOs (``VIPRSGrid.model_elbos`` = the ``ELBO`` column of ``validation_result``).  The
reference calls ``VIPRS.elbo()`` on the (m, n_models) arrays (grid_utils.py:38), which sums the
variational terms over ALL models and therefore only differs between models through the
sigma_epsilon term.
"""
import copy

import numpy as np


def select_best_model(viprs_grid_model, validation_gdl=None, criterion="ELBO"):
    """grid_utils.py:8-100.  `pseudo_validation`: the model with the highest summary-statistics pseudo-R^2 on
    held-out standardized betas -- `validation_gdl` may be a `{chromosome: std_beta}` dict (or an object with
    that dict as `.std_beta`); otherwise `viprs_grid_model.validation_std_beta` is used."""
    if criterion not in ("ELBO", "validation", "pseudo_validation"):
        raise AssertionError(f"unknown criterion {criterion!r}")
    if criterion == "validation":
        raise NotImplementedError("the genotype-based validation criterion needs the reference's prediction stack")
    m = viprs_grid_model
    ok = m.valid_terminated_models
    if np.sum(ok) < 2:
        raise ValueError("Less than two models converged successfully. Cannot perform model selection.")
    if criterion == "ELBO":
        score = np.array(m.model_elbos, dtype=np.float64)
    else:
        vb = validation_gdl if isinstance(validation_gdl, dict) else getattr(validation_gdl, "std_beta", None)
        if vb is None:
            vb = getattr(m, "validation_std_beta", None)
        if vb is None:
            raise ValueError("Validation GWADataLoader or standardized betas from a validation set must be "
                             "initialized for the pseudo_validation criterion.")
        score = np.nan_to_num(np.asarray(m.pseudo_validate(vb), dtype=np.float64), nan=0.0, neginf=0.0, posinf=0.0)
        m.validation_result["Pseudo_Validation_R2"] = score
    score = score.copy()
    score[~ok] = -np.inf
    best = int(np.argmax(score))
    for param in (m.pip, m.post_mean_beta, m.post_var_beta, m.var_gamma, m.var_mu, m.var_tau, m.eta, m.zeta, m.q,
                  m._log_var_tau):
        for c in param:
            param[c] = np.ascontiguousarray(param[c][:, best])      # (a strided view is no valid download target)
    for c in m.eta_diff:
        if m.eta_diff[c].ndim == 2:
            m.eta_diff[c] = np.ascontiguousarray(m.eta_diff[c][:, best])
    m.sigma_epsilon, m._sigma_g = m.sigma_epsilon[best], m._sigma_g[best]
    m.tau_beta, m.pi = m.tau_beta[best], m.pi[best]
    m.n_models = 1
    m.best_model_idx = best
    m.set_fixed_params(m.grid_table.iloc[best].to_dict())
    # the cached partial sums and the device state still belong to the last fitted grid point: drop the sums and
    # put the selected model's state on the device, so that elbo() / to_theta_table() / fit(continued=True) see it
    m._sums, m._sums_valid, m._host_stale = None, False, False
    if m._dstate:
        m._push_state()
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
