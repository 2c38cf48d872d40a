"""``VIPRSGrid`` -- the spike-and-slab model over a grid of fixed hyper-parameters.

Two ways to fit the grid:

* ``fit(pathwise=True|False)`` -- the reference's scheme (viprs/model/gridsearch/VIPRSGrid.py:128-258):
  the grid points are fitted one after the other through ``VIPRS.fit`` (warm-started from the
  previous point when ``pathwise``); every fit runs the spike-and-slab E-step kernels.
* ``fit(batched=True)`` -- all grid points at once: the (m, n_models) state lives in
  one device-resident grid state, every EM iteration runs ``e_step_grid`` (e_step.hpp:555-647, the
  kernel the reference ships but no longer calls) over the models that are still active, the per-model
  M-step / ELBO sums come back as a few scalars per model, and converged models drop out of
  ``active_model_idx``.  Models are independent fits from the standard initialisation.

Either way the outputs have the reference's layout: ``var_gamma / var_mu / var_tau / q`` of shape
``(m, n_models)`` per chromosome, vector-valued ``pi / sigma_epsilon / tau_beta / _sigma_g``, and the
``validation_result`` table (grid columns + ``ELBO``, ``Converged``, ``Optimization_message``).
"""
import copy

import numpy as np
import pandas as pd

from ..VIPRS import VIPRS
from .._lockstep import LockstepEM


class VIPRSGrid(VIPRS):

    def __init__(self, gdl, grid, **kwargs):
        self.grid_table = grid.to_table()
        self.n_models = len(self.grid_table)
        assert self.n_models > 1, "Grid search requires at least 2 models."
        self.validation_result = None
        self.optim_results = []
        super().__init__(gdl, **kwargs)
        self._grid_state = {}

    # (VIPRS puts all local chromosomes into one device plan -- key "*" of self._plans; the batched fit
    #  below builds its (m, G) grid state on whatever plans there are)

    # ---- bookkeeping (VIPRSGrid.py:65-126) --------------------------------------------------------
    @property
    def converged_models(self):
        return np.array([r.success for r in self.optim_results])

    @property
    def terminated_models(self):
        return np.array([r.stop_iteration for r in self.optim_results])

    @property
    def valid_terminated_models(self):
        return np.array([r.valid_optim_result for r in self.optim_results])

    @property
    def models_to_keep(self):
        return np.logical_or(~self.terminated_models, self.converged_models)

    def to_validation_table(self):
        if self.validation_result is None or len(self.validation_result) < 1:
            raise ValueError("Validation result is not set!")
        return pd.DataFrame(self.validation_result)

    def write_validation_result(self, v_filename, sep="\t"):
        self.to_validation_table().to_csv(v_filename, index=False, sep=sep)

    def _reset_search(self):
        """Start the grid search over (VIPRSGrid.py:56-65), e.g. after model selection / averaging."""
        self.n_models = len(self.grid_table)
        assert self.n_models > 1, "Grid search requires at least 2 models."
        self.validation_result = None
        self.optim_results = []

    def init_optim_meta(self):
        super().init_optim_meta()
        self.optim_results = []

    # ---- per-model ELBO parts / summaries after a grid fit: vectors of length n_models -------------------
    def _model_sums(self):
        """Per-model sums over the published (m, n_models) arrays (float64), VIPRS._partial_sums layout."""
        res = np.finfo(np.float64).resolution
        G = self.n_models
        s = np.zeros((11, G))
        for c in self.chromosomes:
            g = np.asarray(self.var_gamma[c], dtype=np.float64)
            mu, vt = np.asarray(self.var_mu[c], dtype=np.float64), np.asarray(self.var_tau[c], dtype=np.float64)
            eta, q = np.asarray(self.eta[c], dtype=np.float64), np.asarray(self.q[c], dtype=np.float64)
            zeta = g * (mu ** 2 + 1.0 / vt)
            gc, ng = np.clip(g, res, 1 - res), np.clip(1.0 - g, res, 1 - res)
            s[1] += zeta.sum(axis=0)
            s[2] += ((1.0 + self.lambda_min) * zeta + q * eta).sum(axis=0)
            s[3] += np.asarray(self.std_beta[c], dtype=np.float64) @ eta
            s[4] += (eta ** 2).sum(axis=0)
            s[5] += (gc * np.log(gc)).sum(axis=0)
            s[6] += (ng * np.log(ng)).sum(axis=0)
            s[7] += gc.sum(axis=0)
            s[8] += ng.sum(axis=0)
            s[9] += (gc * np.log(vt)).sum(axis=0)
        return s

    def _grid_params(self):
        f = lambda x: np.asarray(x, dtype=np.float64) * np.ones(self.n_models)
        return f(self.pi), f(self.tau_beta), f(self.sigma_epsilon), f(self._sigma_g)

    def entropy(self, sum_axis=0):
        if np.ndim(self.pi) == 0:
            return super().entropy()
        s = self._model_sums()
        return 0.5 * self.n_snps * (np.log(2.0 * np.pi) + 1.0) - s[5] - s[6] - 0.5 * s[9]

    def loglikelihood(self):
        if np.ndim(self.pi) == 0:
            return super().loglikelihood()
        s = self._model_sums()
        _, _, sig, sg = self._grid_params()
        return -0.5 * self.n * (np.log(2.0 * np.pi * sig) + (1.0 / sig) * (1.0 - 2.0 * s[3] + sg))

    def log_prior(self, sum_axis=0):
        if np.ndim(self.pi) == 0:
            return super().log_prior()
        s = self._model_sums()
        pi, tau, _, _ = self._grid_params()
        return (0.5 * np.log(tau) * s[7] + np.log(pi) * s[7] + np.log(1.0 - pi) * s[8] - 0.5 * tau * s[1]
                - 0.5 * self.n_snps * np.log(2.0 * np.pi))

    def mse(self, sum_axis=0):
        if np.ndim(self.pi) == 0:
            return super().mse()
        s = self._model_sums()
        _, _, _, sg = self._grid_params()
        return 1.0 - 2.0 * s[3] + (sg - s[1] + s[4])

    def to_theta_table(self):
        """Long table: one block of hyper-parameter rows per grid model."""
        if np.ndim(self.pi) == 0:
            return super().to_theta_table()
        pi, tau, sig, sg = self._grid_params()
        rows = []
        for g in range(self.n_models):
            for k, v in (("ELBO", float(self.model_elbos[g])), ("Residual_variance", sig[g]),
                         ("Heritability", sg[g] / (sg[g] + sig[g])), ("Proportion_causal", pi[g]),
                         ("Average_effect_variance", pi[g] / tau[g]), ("tau_beta", tau[g])):
                rows.append({"Model": g, "Parameter": k, "Value": v})
        return pd.DataFrame(rows)

    # ---- fitting -------------------------------------------------------------------------------------
    def fit(self, pathwise=True, batched=False, **fit_kwargs):
        fit_kwargs.pop("disable_pbar", None)
        if self.n_models == 1:                   # after model selection / averaging: an ordinary VIPRS model
            return super().fit(**fit_kwargs)
        if batched:
            return self._fit_batched(**fit_kwargs)
        return self._fit_serial(pathwise, **fit_kwargs)

    def _collect(self, store, i):
        for c in self.shapes:
            store["var_gamma"][c][:, i] = self.var_gamma[c]
            store["var_mu"][c][:, i] = self.var_mu[c]
            store["var_tau"][c][:, i] = self.var_tau[c]
            store["q"][c][:, i] = self.q[c]
        store["sigma_epsilon"][i], store["pi"][i] = self.sigma_epsilon, self.pi
        store["sigma_g"][i], store["tau_beta"][i] = self._sigma_g, self.tau_beta

    def _new_store(self):
        T = self.float_precision
        mk = lambda: {c: np.empty((s, self.n_models), dtype=T) for c, s in self.shapes.items()}
        return dict(var_gamma=mk(), var_mu=mk(), var_tau=mk(), q=mk(),
                    sigma_epsilon=np.empty(self.n_models, T), pi=np.empty(self.n_models, T),
                    sigma_g=np.empty(self.n_models, T), tau_beta=np.empty(self.n_models, T),
                    elbo=np.empty(self.n_models, T))

    def _publish(self, store, optim_results):
        self.optim_result.nit = int(np.sum([r.nit for r in optim_results]))
        self.optim_results = optim_results
        self.var_gamma, self.var_mu, self.var_tau, self.q = (store[k] for k in ("var_gamma", "var_mu", "var_tau", "q"))
        self.eta = self.compute_eta()
        self.zeta = self.compute_zeta()
        self._log_var_tau = {c: np.log(self.var_tau[c]) for c in self.var_tau}
        self._host_stale = False
        self.update_posterior_moments()
        self.sigma_epsilon, self.pi = store["sigma_epsilon"], store["pi"]
        self._sigma_g, self.tau_beta = store["sigma_g"], store["tau_beta"]
        self.model_elbos = store["elbo"].astype(np.float64)
        self.validation_result = self.grid_table.copy()
        self.validation_result["ELBO"] = store["elbo"]
        self.validation_result["Converged"] = self.converged_models
        self.validation_result["Optimization_message"] = [r.message for r in self.optim_results]
        return self

    def _fit_serial(self, pathwise, **fit_kwargs):
        """One VIPRS.fit per grid point (VIPRSGrid.py:194-225)."""
        store, results = self._new_store(), []
        params = self.grid_table.to_dict(orient="records")
        for i in range(self.n_models):
            self.set_fixed_params(params[i])
            super().fit(continued=(i > 0 and pathwise), **fit_kwargs)
            results.append(copy.deepcopy(self.optim_result))
            self.optim_result.reset()
            store["elbo"][i] = self.history["ELBO"][-1]
            self._collect(store, i)
        return self._publish(store, results)

    # ---- all grid points at once -------------------------------------------------------------------------
    def _fit_batched(self, max_iter=1000, theta_0=None, min_iter=3, f_abs_tol=1e-6, x_abs_tol=1e-6, patience=10,
                     on_iteration=None, **kwargs):
        if self._e_step_fn is not None or self.comm.world_size != 1:
            raise NotImplementedError("the batched grid fit runs on one GPU through the device-resident grid state")
        from ...plan import DeviceState
        G, T = self.n_models, self._T
        params = self.grid_table.to_dict(orient="records")
        # per-model hyper-parameters: grid values are fixed, the rest follows VIPRS.initialize_theta
        # (the constructor's own fix_params stay fixed for every model, as in the serial fit where set_fixed_params
        # only ADDS the grid point's values; a lambda_min grid sets each model's own ridge penalty)
        th = []
        base_fixed, base_lambda = dict(self.fix_params), self.lambda_min
        for g in range(G):
            self.fix_params = {**base_fixed, **params[g]}
            self.initialize_theta(dict(theta_0) if theta_0 else None)
            lam = T.type(params[g]["lambda_min"]) if "lambda_min" in params[g] else base_lambda
            th.append(dict(pi=self.pi, sigma_epsilon=self.sigma_epsilon, tau_beta=self.tau_beta,
                           lam=lam, fixed=set(self.fix_params)))
        self.fix_params = base_fixed
        states = {}
        merged = getattr(self, "_merged", False)
        chroms = self.chromosomes
        for key, plan in self._plans.items():       # one plan per chromosome, or "*" = all of them concatenated
            st = self._grid_state.get(key)
            if st is None:
                st = self._grid_state[key] = DeviceState(plan, self.float_precision, "grid", G)
                if merged:
                    st.upload("std_beta", np.concatenate([self.std_beta[c] for c in chroms]))
                    st.set_n_per_snp(np.concatenate([np.asarray(self.n_per_snp[c], dtype=np.float64).ravel() for c in chroms]))
                    st.set_snp_weights(np.concatenate([np.full(self.shapes[c], 1.0 / self.shapes[c]) for c in chroms]))
                else:
                    st.upload("std_beta", self.std_beta[key])
                    st.set_n_per_snp(self.n_per_snp[key])
            for g in range(G):
                st.reset_column(g, float(th[g]["pi"]))
            states[key] = st

        # ---- per-model hyper-parameters as ARRAYS (`LockstepEM`: M-step, ELBO and stopping rules of all active models in
        # a handful of NumPy calls, in the serial fit's dtypes)
        em = LockstepEM(T, th, self.n_snps, self.n, n_chroms_total=self._n_chroms_total, min_iter=min_iter,
                        f_abs_tol=f_abs_tol, x_abs_tol=x_abs_tol, patience=patience)
        lam1_v = em.lam1

        def all_sums(models):
            """(len(models), 11) sums: one batched reduction per chromosome, all in flight at once"""
            for st in states.values():
                st.sums_columns_begin(models, lam1_v[models])
            tot = np.zeros((len(models), 11))
            for key, st in states.items():
                v = st.sums_columns_end()
                tot[:, 0] += v[:, 0] if merged else v[:, 0] / self.shapes[key]      # merged: weights 1 / m_c on the device
                tot[:, 1:10] += v[:, 1:10]
                tot[:, 10] = np.maximum(tot[:, 10], v[:, 10])
            return tot

        active = np.arange(G, dtype=np.int32)
        for st in states.values():               # initial ELBO needs var_tau of the initial hyper-parameters
            st.prep_columns(em.prep_rows(active))

        for i in range(1, max_iter + 1):
            if active.size == 0:
                break
            a = active
            em.mark_e_step(a)                                                   # what var_tau is built from
            rows = em.prep_rows(a)
            for st in states.values():               # one prep launch, one sweep and one reduction per plan
                st.prep_columns(rows)
                st.e_step(self.dequantize_scale, active_model_idx=a, sync=False)
            code = em.update(a, all_sums(a), i)
            active = a[code == 0]
            if on_iteration is not None:
                on_iteration(i)
        em.finish()
        self._lockstep = (em, states, all_sums)        # (measurement hook: bench.py takes an iteration apart phase by phase)
        results, sigma_g, elbos = em.results, em.sigma_g, em.elbos
        # back into the per-model records the publishing code reads (in the serial fit's dtypes)
        for g in range(G):
            p = th[g]
            p["pi"], p["sigma_epsilon"], p["tau_beta"] = em.theta(g)
            p["sigma_epsilon_e"], p["tau_beta_e"] = em.sig_e[g], em.tau_e[g]

        # ---- read the (m, G) state back in the reference's layout ---------------------------------------
        store = self._new_store()
        seg = self._seg if merged else None

        def pull(name):
            if merged:
                full = states["*"].download(name)
                return {c: full[seg[c][0]:seg[c][1]] for c in chroms}
            return {c: states[c].download(name) for c in chroms}

        pulled = {name: pull(name) for name in ("var_gamma", "var_mu", "q")}
        for c in self.chromosomes:
            for name in ("var_gamma", "var_mu", "q"):
                store[name][c][...] = pulled[name][c]
            for g in range(G):
                p = th[g]
                store["var_tau"][c][:, g] = (self.n_per_snp[c] * (1.0 + p["lam"]) / p["sigma_epsilon_e"]) + p["tau_beta_e"]
        for g in range(G):
            p = th[g]
            store["sigma_epsilon"][g], store["pi"][g] = p["sigma_epsilon"], p["pi"]
            store["tau_beta"][g], store["sigma_g"][g], store["elbo"][g] = p["tau_beta"], sigma_g[g], elbos[g]
        self.eta_diff = {c: np.asfortranarray(v) for c, v in pull("eta_diff").items()}
        return self._publish(store, results)
