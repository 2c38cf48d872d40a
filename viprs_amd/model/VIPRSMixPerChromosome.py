"""``VIPRSMixPerChromosome`` -- one independent ``VIPRSMix`` model PER CHROMOSOME, fitted in lock step on one device plan.

The mixture counterpart of ``VIPRSPerChromosome`` (``viprs_fit`` fits whatever model it is given once per chromosome unless
``--genomewide``: bin/viprs_fit:232-238, :1079-1086).  The device side is the same idea -- the chromosomes are SNP groups
of one mixture state (``viprs_state_set_groups``), ``viprs_state_prep_mixture_groups`` writes every group's E-step inputs
from its own ``(pi_k, tau_beta_k, sigma_epsilon)``, ONE sweep updates all groups that still iterate,
``viprs_state_sums_mixture_groups_*`` returns the 6 + 6 K sums and max |eta_diff| per group, converged chromosomes leave
the sweep (``viprs_plan_set_active_blocks``).

The host side of an iteration exists in two forms that are held `==` (tests/test_per_chromosome_mix.py):

* ``host="scalar"``: each chromosome's model is the scalar code of ``VIPRSMix`` / ``VIPRS`` itself (``m_step``, ``elbo``, ``mse``,
  the stopping rules of ``VIPRS._after_e_step``) run with that chromosome's hyper-parameters, sample size, SNP count, history
  and optimisation record swapped in (``_as_model``) and its row of device sums in place of the model's own reduction -- the
  dtypes and roundings are the serial fit's by construction (~40 us per model and iteration);
* ``host="vector"`` (default): the same as array operations over the active models (``_lockstep_mix.LockstepMixEM``).

A group's prep and sums are bit-identical to those of a plan of that chromosome alone, so the batch reproduces
``{c: VIPRSMix(loader_of_c, K).fit() for c in chromosomes}`` on this device bit for bit.

Results are keyed by chromosome as in ``VIPRSPerChromosome``: ``pi[c]`` / ``tau_beta[c]`` are K-vectors, ``history[c]`` and
``optim_results[c]`` are the chromosome's own.
"""
import contextlib
import logging

import numpy as np

from ..utils.optim import OptimizeResult
from .VIPRSMix import VIPRSMix
from .VIPRSPerChromosome import PerChromosomeGroups

logger = logging.getLogger(__name__)

# what makes one chromosome's model: swapped onto the instance while the scalar code of VIPRSMix runs for it
_MODEL_ATTRS = ("pi", "sigma_epsilon", "tau_beta", "_sigma_g", "fix_params", "history", "optim_result", "_last_prep",
                "_max_eta_diff")


class VIPRSMixPerChromosome(PerChromosomeGroups, VIPRSMix):

    def __init__(self, gdl, K=1, prior_multipliers=None, lambda_min=None, host="vector", **kwargs):
        """Arguments of ``VIPRSMix``.  ``host``: how the host side of an iteration runs -- "vector" (default): all active
        models' M-steps / ELBOs / stopping rules as array operations (`_lockstep_mix.LockstepMixEM`); "scalar": the serial
        code of ``VIPRSMix`` itself per model (22 x ~40 us per round; the cross-check of the former)."""
        assert host in ("vector", "scalar")
        self._host = host
        self._cur = None                      # index of the chromosome whose model is swapped in (None: none)
        super().__init__(gdl, lambda_min=lambda_min, K=K, prior_multipliers=prior_multipliers, **kwargs)
        self._models = []

    # ---- sizes of the model that is swapped in ---------------------------------------------------------------------------
    @property
    def m(self):
        return int(self._m_group[self._cur]) if self._cur is not None else int(self.gdl.m)

    n_snps = m

    @property
    def chromosomes(self):
        if self._cur is None:
            return sorted(self.shapes.keys())
        c = self.groups[self._cur]
        return [c] if c in self.shapes else []

    @contextlib.contextmanager
    def _as_model(self, g):
        rec = self._models[g]
        extra = ("_sample_size", "lambda_min", "_n_chroms_total", "_cur", "_sums", "_sums_valid")
        saved = {k: getattr(self, k) for k in _MODEL_ATTRS + extra}
        for k in _MODEL_ATTRS:
            setattr(self, k, rec[k])
        self._sample_size, self.lambda_min = float(self._n_group[g]), self._T.type(self._lambda_group[g])
        self._n_chroms_total, self._cur, self._sums, self._sums_valid = 1, g, None, False
        try:
            yield rec
        finally:
            for k in _MODEL_ATTRS:
                rec[k] = getattr(self, k)
            for k, v in saved.items():
                setattr(self, k, v)

    def _theta0_of(self, c, theta_0):
        t0 = theta_0
        if isinstance(theta_0, dict) and theta_0 and all(k in self._gindex for k in theta_0):
            t0 = theta_0.get(c)                                    # {chromosome: theta_0}
        return dict(t0) if t0 else None                            # (`_merge_theta` writes into its argument)

    # ---- variational state of one chromosome (VIPRS.py:330-359 with (m, K) arrays) -----------------------------------------
    def _init_chromosome_state(self, c, rec):
        T, shp = self._T, self._shape(c)
        self.var_tau[c] = (self.n_per_snp[c] / rec["sigma_epsilon"]) + rec["tau_beta"]
        self.var_mu[c] = np.zeros(shp, T, order=self.order)
        self.var_gamma[c] = (rec["pi"] * np.ones(shp, dtype=T, order=self.order)).astype(T, order=self.order)
        self.eta[c] = (self.var_gamma[c] * self.var_mu[c]).sum(axis=1)
        self.zeta[c] = (self.var_gamma[c] * (self.var_mu[c] ** 2 + (1.0 / self.var_tau[c]))).sum(axis=1)
        self.eta_diff[c] = np.zeros_like(self.eta[c], dtype=T)
        self.q[c] = np.zeros_like(self.eta[c], dtype=T)
        self._log_var_tau[c] = np.log(self.var_tau[c])

    def _upload_log_var_tau(self):
        if self._e_step_fn is None:
            chroms = sorted(self.shapes.keys())
            self._dstate["*"].set_log_var_tau(np.concatenate(
                [np.asarray(self._log_var_tau[c], dtype=np.float64) * np.ones(self._shape(c)) for c in chroms]))

    def _restart_state(self, theta_0, param_0):
        """`VIPRS._restart_state` for the chromosome whose model is swapped in (negative MSE with a free sigma_epsilon,
        VIPRS.py:1025-1037): its hyper-parameters are drawn again, its arrays start over; the other chromosomes keep theirs."""
        c = self.groups[self._cur]
        logger.info("Chromosome %s: restarting with sigma_epsilon fixed.", c)
        self.initialize_theta(self._theta0_of(c, theta_0))
        if self._e_step_fn is None:
            self._pull_state()
        if c in self.shapes:
            self._init_chromosome_state(c, dict(pi=self.pi, sigma_epsilon=self.sigma_epsilon, tau_beta=self.tau_beta))
        if self._e_step_fn is None:
            self._push_state()
            self._upload_log_var_tau()

    # ---- one lock-step iteration ---------------------------------------------------------------------------------------------
    def _sweep_models(self, a):
        if self._e_step_fn is None:
            rows = []
            for g in a:
                with self._as_model(int(g)):
                    pi, tau_beta = np.asarray(self.pi), np.asarray(self.tau_beta)
                    logit_pi = np.log(pi) - np.log(1.0 - pi)                         # dtype semantics of VIPRSMix.py:211
                    log_null_pi = np.log(1.0 - self.pi.sum())
                    rows.append(np.concatenate([[float(g), float(log_null_pi), float(self.sigma_epsilon),
                                                 float(1.0 + self.lambda_min)], np.asarray(logit_pi, dtype=np.float64),
                                                np.asarray(np.log(tau_beta), dtype=np.float64),
                                                np.asarray(tau_beta, dtype=np.float64)]))
                    self._last_prep = (self.sigma_epsilon, tau_beta, self.lambda_min)
            ds = self._dstate["*"]
            ds.prep_mixture_groups(np.array(rows))
            ds.e_step(self.dequantize_scale, sync=False)
            self._host_stale = True
            return
        for g in a:                                               # CPU test hook: the oracle's kernel per chromosome
            c = self.groups[g]
            if c not in self.shapes:
                continue
            with self._as_model(int(g)):
                log_null_pi, u_logs, shvt, mu_mult = self._prep(c)
                self._e_step_fn(self.ld_left_bound[c], self.ld_indptr[c], self.ld_data[c], self.std_beta[c],
                                self.var_gamma[c], self.var_mu[c], self.eta[c], self.q[c], self.eta_diff[c],
                                log_null_pi, u_logs, shvt, mu_mult, self.dequantize_scale, self.threads, self.low_memory)
            self.zeta[c] = (self.var_gamma[c] * (self.var_mu[c] ** 2 + (1.0 / self.var_tau[c]))).sum(axis=1)

    def _host_model_sums(self, a):
        s = np.zeros((len(a), 7 + 6 * self.K))
        for k, g in enumerate(a):
            c = self.groups[g]
            if c in self.shapes:
                with self._as_model(int(g)):
                    s[k, :-1] = VIPRSMix._partial_sums(self)        # (over `self.chromosomes` = this chromosome)
                s[k, -1] = float(np.max(np.abs(self.eta_diff[c]))) if self.eta_diff[c].size else 0.0
        return s

    def _model_sums(self, a, on_host=False):
        """(len(a), 7 + 6 K) rows in the layout of `viprs_state_sums_mixture_end`, over all ranks."""
        if self._e_step_fn is not None or on_host:
            s, reduced = self._host_model_sums(a), False
        else:
            ds = self._dstate["*"]
            ds.sums_mixture_groups_begin(a, [float(1.0 + self._T.type(self._lambda_group[g])) for g in a])
            s, reduced = ds.sums_mixture_groups_end(), self._device_reduce
        if self.comm.world_size > 1 and not reduced:
            tot = self.comm.allreduce_sum(np.ascontiguousarray(s[:, :-1]).ravel()).reshape(len(a), -1)
            mx = self.comm.allreduce_max(np.ascontiguousarray(s[:, -1]))
            s = np.column_stack([tot, mx])
        return s

    # ---- the fit ----------------------------------------------------------------------------------------------------------------
    def fit(self, max_iter=1000, theta_0=None, param_0=None, continued=False, disable_pbar=True, min_iter=3,
            f_abs_tol=1e-6, x_abs_tol=1e-6, patience=10, on_iteration=None, **kwargs):
        """All chromosomes' EM iterations together; arguments of ``VIPRSMix.fit``.  ``theta_0`` is one dict for every
        chromosome or ``{chromosome: dict}``."""
        if continued or param_0 is not None:
            raise NotImplementedError("VIPRSMixPerChromosome.fit: `continued` / `param_0` are not supported")
        if any(callable(t) for t in self.tracked_params):
            raise NotImplementedError("callable tracked_params are not supported by the per-chromosome fit")
        G = len(self.groups)
        if self._models and isinstance(self.pi, dict):           # a second fit() on the same object: back to scalars
            self.fix_params = self._base_fixed
        base_fixed = self._base_fixed = dict(self.fix_params)
        self._models = [dict(pi=None, sigma_epsilon=None, tau_beta=None, _sigma_g=None, fix_params=dict(base_fixed),
                             history={}, optim_result=OptimizeResult(), _last_prep=None, _max_eta_diff=0.0)
                        for _ in range(G)]
        # ---- hyper-parameters of every model as VIPRSMix.initialize_theta leaves them (several ranks: rank 0's draws) ----
        for g, c in enumerate(self.groups):
            with self._as_model(g):
                self.initialize_theta(self._theta0_of(c, theta_0))
                self.init_optim_meta()
        # ---- standard start of every chromosome + its initial ELBO ----
        self.var_mu, self.var_tau, self.var_gamma, self._log_var_tau = {}, {}, {}, {}
        self.eta, self.zeta, self.eta_diff, self.q = {}, {}, {}, {}
        for c in sorted(self.shapes.keys()):
            self._init_chromosome_state(c, self._models[self._gindex[c]])
        self._sums_valid, self._host_stale = False, False
        self._push_state()
        self._upload_log_var_tau()
        self._set_active(np.arange(G))
        all_groups = np.arange(G)
        s0 = self._model_sums(all_groups, on_host=True)
        progress = []
        for g in all_groups:
            with self._as_model(int(g)):
                self._sums, self._sums_valid, self._max_eta_diff = s0[g, :-1], True, float(s0[g, -1])
                self.update_theta_history()
                progress.append(self._new_fit_progress())

        if self._host == "vector":
            self._iterate_vector(max_iter, theta_0, min_iter, f_abs_tol, x_abs_tol, patience, on_iteration)
            return self._publish_models()
        active = all_groups
        for i in range(1, max_iter + 1):
            if active.size == 0:
                break
            self._sweep_models(active)
            s = self._model_sums(active)
            keep = np.ones(active.size, dtype=bool)
            for k, g in enumerate(active):
                with self._as_model(int(g)):
                    self._sums, self._sums_valid, self._max_eta_diff = s[k, :-1], True, float(s[k, -1])
                    self._after_e_step(i, progress[g], theta_0, None, min_iter, f_abs_tol, x_abs_tol, patience)
                    keep[k] = not self.optim_result.stop_iteration
            if not keep.all():
                self._set_active(active[keep])
            active = active[keep]
            if on_iteration is not None:
                on_iteration(i)
        for g in all_groups:
            rec = self._models[g]
            if not rec["optim_result"].stop_iteration:
                rec["optim_result"].update(rec["history"]["ELBO"][-1], stop_iteration=True, success=False, increment=False,
                                           message="Maximum iterations reached without convergence.\n"
                                                   "You may need to run the model for more iterations.")
        self._set_active(all_groups)
        return self._publish_models()

    # ---- the same loop with the host side as array operations over the active models ----------------------------------------
    def _iterate_vector(self, max_iter, theta_0, min_iter, f_abs_tol, x_abs_tol, patience, on_iteration):
        from ._lockstep import RESTART
        from ._lockstep_mix import LockstepMixEM
        M, G, T = self._models, len(self.groups), self._T
        th = [dict(pi=M[g]["pi"], sigma_epsilon=M[g]["sigma_epsilon"], tau_beta=M[g]["tau_beta"],
                   lam=T.type(self._lambda_group[g]), fixed=M[g]["fix_params"]) for g in range(G)]
        em = self._em = LockstepMixEM(T, th, self.d, self._m_group, self._n_group, min_iter=min_iter, f_abs_tol=f_abs_tol,
                                      x_abs_tol=x_abs_tol, patience=patience)
        em.sigma_g[:] = [float(M[g]["_sigma_g"]) for g in range(G)]
        em.prev_sigma_g[:] = em.sigma_g

        def sync_model(g):                   # the model's record as the scalar code expects it (CPU hook, restart, publishing)
            rec = M[g]
            rec["pi"], rec["sigma_epsilon"], rec["tau_beta"] = em.theta(g)
            rec["_sigma_g"], rec["_max_eta_diff"] = em.sigma_g[g], float(em.max_eta_diff[g])

        active = np.arange(G)
        for i in range(1, max_iter + 1):
            if active.size == 0:
                break
            a = active
            em.mark_e_step(a)
            if self._e_step_fn is None:
                ds = self._dstate["*"]
                ds.prep_mixture_groups(em.prep_rows(a))
                ds.e_step(self.dequantize_scale, sync=False)
                self._host_stale = True
            else:
                for g in a:
                    sync_model(int(g))
                self._sweep_models(a)
            code = em.update(a, self._model_sums(a), i)
            for g in a:
                h = M[g]["history"]
                h["ELBO"].append(float(em.elbos[g]))
                if self.tracked_params:
                    pi, sig, tau = em.theta(g)
                    for t in self.tracked_params:
                        h[t].append({"pi": lambda: np.sum(pi), "heritability": lambda: em.sigma_g[g] / (em.sigma_g[g] + sig),
                                     "sigma_epsilon": lambda: sig, "tau_beta": lambda: tau, "sigma_g": lambda: em.sigma_g[g],
                                     "max_eta_diff": lambda: float(em.max_eta_diff[g])}[t]())
            for g in a[code == RESTART]:
                g = int(g)
                sync_model(g)
                with self._as_model(g):
                    self._restart_state(theta_0, None)
                    self.fix_params["sigma_epsilon"] = self.sigma_epsilon = 0.95
                em.restart(g, M[g]["pi"], 0.95, M[g]["tau_beta"])
            keep = (code == 0) | (code == RESTART)
            if not keep.all():
                self._set_active(a[keep])
            active = a[keep]
            if on_iteration is not None:
                on_iteration(i)
        em.finish()
        for g in range(G):
            sync_model(g)
            M[g]["optim_result"], M[g]["_last_prep"] = em.results[g], em.last_prep(g)
        self._set_active(np.arange(G))

    def _publish_models(self):
        groups, M = self.groups, self._models
        if self._e_step_fn is None and self._host_stale:
            self._pull_state()
        for c in sorted(self.shapes.keys()):
            lp = M[self._gindex[c]]["_last_prep"]
            if lp is not None:                                     # var_tau of the chromosome's LAST E-step (sync_host)
                sigma_epsilon, tau_beta, lam = lp
                self.var_tau[c] = (self.n_per_snp[c] * (1.0 + lam) / sigma_epsilon) + tau_beta
        self._host_stale = False
        self.zeta = self.compute_zeta()
        self.pi = {c: M[g]["pi"] for g, c in enumerate(groups)}
        self.sigma_epsilon = {c: M[g]["sigma_epsilon"] for g, c in enumerate(groups)}
        self.tau_beta = {c: M[g]["tau_beta"] for g, c in enumerate(groups)}
        self._sigma_g = {c: M[g]["_sigma_g"] for g, c in enumerate(groups)}
        self.history = {c: M[g]["history"] for g, c in enumerate(groups)}
        self.optim_results = {c: M[g]["optim_result"] for g, c in enumerate(groups)}
        res = self.optim_result = OptimizeResult()
        res.nit = max(r.nit for r in self.optim_results.values())
        res.success = all(r.success for r in self.optim_results.values())
        res.stop_iteration = True
        res.fun = float(np.sum([h["ELBO"][-1] for h in self.history.values()]))
        failed = [c for c in groups if not self.optim_results[c].success]
        res.message = "All chromosomes converged." if not failed else \
            "Not converged: chromosome(s) " + ", ".join(str(c) for c in failed)
        self.update_posterior_moments()
        self._gather_posterior()
        for c in failed:
            logger.warning("\tchromosome %s: %s", c, self.optim_results[c].message)
        return self

    # ---- summaries: one value per chromosome --------------------------------------------------------------------------------
    def get_null_pi(self, chrom=None):
        if isinstance(self.pi, dict):
            return {c: 1.0 - np.sum(p) for c, p in self.pi.items()} if chrom is None else 1.0 - np.sum(self.pi[chrom])
        return VIPRSMix.get_null_pi(self, chrom)

    def get_proportion_causal(self):
        if isinstance(self.pi, dict):
            return {c: np.sum(p) for c, p in self.pi.items()}
        return VIPRSMix.get_proportion_causal(self)

    def get_heritability(self):
        if isinstance(self.pi, dict):
            return PerChromosomeGroups.get_heritability(self)
        return VIPRSMix.get_heritability(self)

    def get_average_effect_size_variance(self):
        if isinstance(self.pi, dict):
            return {c: float(np.sum(np.asarray(self.pi[c], dtype=np.float64) / np.asarray(self.tau_beta[c], dtype=np.float64)))
                    for c in self.groups}
        return VIPRSMix.get_average_effect_size_variance(self)

    def elbo(self, sum_axis=None):
        if isinstance(self.pi, dict):
            return {c: h["ELBO"][-1] for c, h in self.history.items()}
        return VIPRSMix.elbo(self, sum_axis)

    objective = elbo

    def to_theta_table(self):
        """Hyper-parameter rows of every chromosome's model (`VIPRSMix.to_theta_table`) with a `Chromosome` column."""
        import pandas as pd
        h2, rows = self.get_heritability(), []
        for c in self.groups:
            pi, tau = np.asarray(self.pi[c], dtype=np.float64), np.asarray(self.tau_beta[c], dtype=np.float64)
            rows += [("ELBO", self.history[c]["ELBO"][-1], c), ("Residual_variance", self.sigma_epsilon[c], c),
                     ("Heritability", h2[c], c), ("Proportion_causal", float(pi.sum()), c),
                     ("Average_effect_variance", float(np.sum(pi / tau)), c),
                     ("Lambda_min", self._lambda_group[self._gindex[c]], c)]
            rows += [("tau_beta" if tau.size == 1 else f"tau_beta_{i + 1}", float(t), c) for i, t in enumerate(tau)]
            rows += [(f"pi_{i + 1}", float(p), c) for i, p in enumerate(pi)]
        return pd.DataFrame([{"Parameter": k, "Value": v, "Chromosome": c} for k, v, c in rows])
