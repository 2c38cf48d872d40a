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
from ...utils.optim import ConditionStreak, OptimizeResult


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

        # ---- per-model hyper-parameters as ARRAYS: the host side of an iteration is a handful of NumPy calls over the active
        # models instead of ~40 Python statements (several np.log / np.isclose on scalars) per model -- for 32 models that
        # loop cost as much as a third of the batched sweep itself.  The arithmetic follows the serial fit's DTYPES, which
        # decide roundings: a hyper-parameter that is fixed stays a scalar of the state precision (VIPRS.set_fixed_params /
        # _cast_theta), one that the M-step updates becomes a float64 (VIPRS.m_step) -- `*_is32` track which is which,
        # and `in_dtype` evaluates an expression in float32 for the former.  (Array ufuncs give the same bits as the
        # scalar calls they replace.)
        f32, f64 = np.float32, np.float64
        is32 = lambda v: isinstance(v, np.floating) and v.dtype == np.float32
        pi_v = np.array([p["pi"] for p in th], dtype=T)                       # always of the state precision (m_step casts)
        sig_v = np.array([p["sigma_epsilon"] for p in th], dtype=f64)
        tau_v = np.array([p["tau_beta"] for p in th], dtype=f64)
        sig_is32 = np.array([is32(p["sigma_epsilon"]) for p in th])
        tau_is32 = np.array([is32(p["tau_beta"]) for p in th])
        fx_pi = np.array(["pi" in p["fixed"] for p in th])
        fx_tau = np.array(["tau_beta" in p["fixed"] for p in th])
        fx_sig = np.array(["sigma_epsilon" in p["fixed"] for p in th])
        lam1_v = np.array([float(1.0 + p["lam"]) for p in th], dtype=f64)
        sig_e, tau_e = sig_v.copy(), tau_v.copy()                             # what var_tau of the last E-step was built from

        def in_dtype(v, m32, fn):
            """fn evaluated in float32 where the serial fit holds a float32 scalar (mask m32), in float64 elsewhere."""
            out = fn(v)
            if m32.any():
                out = np.where(m32, fn(v.astype(f32)).astype(f64), out)
            return out

        def prep_rows(a):
            pa = pi_v[a]
            logit = (np.log(pa) - np.log(1.0 - pa)).astype(f64)
            return np.column_stack([a.astype(f64), logit, in_dtype(tau_v[a], tau_is32[a], np.log), sig_v[a], tau_v[a], lam1_v[a]])

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

        results = [OptimizeResult() for _ in range(G)]
        sigma_g = np.zeros(G)
        prev_elbo = np.full(G, -np.inf)
        prev_sigma_g = np.zeros(G)
        plateau_n, dropping_n = np.zeros(G, dtype=np.int64), np.zeros(G, dtype=np.int64)     # ConditionStreak counters
        elbos = np.zeros(G)
        active = np.arange(G, dtype=np.int32)
        for st in states.values():               # initial ELBO needs var_tau of the initial hyper-parameters
            st.prep_columns(prep_rows(active))
        MESSAGES = (None, "The MSE is negative ({:.6f}).", "Objective (ELBO) is undefined.", "Residual variance estimate is negative.",
                    "Estimated heritability is out of bounds.", "Objective (ELBO) converged successfully.",
                    "Variational parameters converged successfully.", "LD-weighted variational parameters converged successfully.",
                    "The objective (ELBO) is decreasing.")
        SUCCESS = (False, False, False, False, False, True, True, True, False)

        for i in range(1, max_iter + 1):
            if active.size == 0:
                break
            a = active
            sig_e[a], tau_e[a] = sig_v[a], tau_v[a]                             # what var_tau is built from
            rows = prep_rows(a)
            for st in states.values():               # one prep launch, one sweep and one reduction per plan
                st.prep_columns(rows)
                st.e_step(self.dequantize_scale, active_model_idx=a, sync=False)
            s = all_sums(a)
            # ---- VIPRS.m_step, per model (VIPRS.py:426-484) ----
            pi_v[a] = np.where(fx_pi[a], pi_v[a], (s[:, 0] / self._n_chroms_total).astype(T))
            upd = ~fx_tau[a]
            tau_v[a] = np.where(upd, pi_v[a] * self.n_snps / s[:, 1], tau_v[a])
            tau_is32[a] &= ~upd
            sigma_g[a] = s[:, 2]
            upd = ~fx_sig[a]
            sig_v[a] = np.where(upd, (1.0 + (-2.0 * s[:, 3]).astype(T)) + sigma_g[a], sig_v[a])
            sig_is32[a] &= ~upd
            # ---- ELBO (VIPRS.py:497-581) in the serial fit's dtypes ----
            sg, sa, ta, pa = sigma_g[a], sig_v[a], tau_v[a], pi_v[a]
            e = in_dtype(sa, sig_is32[a], lambda v: -np.log(2.0 * np.pi * v))
            e = np.where(fx_sig[a], e - in_dtype(sa, sig_is32[a], lambda v: 1.0 / v) * (1.0 - 2.0 * s[:, 3] + sg), e - 1.0)
            e = e * (0.5 * self.n)
            e = e - (s[:, 5] - np.log(pa) * s[:, 7])
            e = e - (s[:, 6] - np.log(1.0 - pa) * s[:, 8])
            e = e + 0.5 * (in_dtype(ta, tau_is32[a], lambda v: 1.0 + np.log(v)) * s[:, 7] - s[:, 9])
            e = e - 0.5 * ta * s[:, 1]
            elbos[a] = e
            mse = 1.0 - 2.0 * s[:, 3] + (sg - s[:, 1] + s[:, 4])
            h2 = sg / (sg + sa)
            # ---- VIPRS.fit stopping rules (VIPRS.py:1003-1080), first match wins ----
            pl = (i > min_iter) & np.isclose(sg, prev_sigma_g[a], atol=x_abs_tol, rtol=0.0) & (s[:, 10] < x_abs_tol * 10)
            dr = (e < prev_elbo[a]) & ~np.isclose(e, prev_elbo[a], atol=1e3 * f_abs_tol, rtol=1e-4)
            plateau_n[a] = np.where(pl, plateau_n[a] + 1, 0)
            dropping_n[a] = np.where(dr, dropping_n[a] + 1, 0)
            code = np.select(
                [mse < 0.0, ~np.isfinite(e), sa < 0.0, (h2 > 1.0) | (h2 < 0.0),
                 (i > min_iter) & np.isclose(prev_elbo[a], e, atol=f_abs_tol, rtol=0.0),
                 (i > min_iter) & (s[:, 10] < x_abs_tol), plateau_n[a] > patience, dropping_n[a] > patience],
                [1, 2, 3, 4, 5, 6, 7, 8], default=0)
            for k, g in enumerate(a):
                c = int(code[k])
                if c == 0:
                    results[g].update(float(e[k]))
                else:
                    msg = MESSAGES[c].format(float(mse[k])) if c == 1 else MESSAGES[c]
                    results[g].update(float(e[k]), stop_iteration=True, success=SUCCESS[c], message=msg)
            prev_elbo[a], prev_sigma_g[a] = e, sg
            active = a[code == 0]
            if on_iteration is not None:
                on_iteration(i)
        # back into the per-model records the publishing code reads (in the serial fit's dtypes)
        for g in range(G):
            p = th[g]
            p["pi"] = pi_v[g]
            p["sigma_epsilon"] = f32(sig_v[g]) if sig_is32[g] else sig_v[g]
            p["tau_beta"] = f32(tau_v[g]) if tau_is32[g] else tau_v[g]
            p["sigma_epsilon_e"], p["tau_beta_e"] = sig_e[g], tau_e[g]
        for g in range(G):
            if not results[g].stop_iteration:
                results[g].update(elbos[g], stop_iteration=True, success=False, increment=False,
                                  message="Maximum iterations reached without convergence.\\n"
                                          "You may need to run the model for more iterations.")

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
