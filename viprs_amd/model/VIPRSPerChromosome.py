"""``VIPRSPerChromosome`` -- one independent spike-and-slab model PER CHROMOSOME, all of them fitted in lock step on one
device plan.

This is the reference's DEFAULT operating mode: ``viprs_fit`` splits the data loader by chromosome unless
``--genomewide`` is given (bin/viprs_fit:232-238) and fits one ``VIPRS`` model per chromosome, fanned out over processes
with joblib (:1079-1086); every model has its own ``(pi, tau_beta, sigma_epsilon)``, its own ELBO history and stops at its
own iteration (VIPRS.py:909-1124).  On the GPU a chromosome-sized fit cannot fill the device -- its sweep is bound by the
serial chain of its largest LD block (chr22: ~0.17 ms of chain against 0.007 ms of LD streaming) -- and 22 of them one
after the other cost ten genome-wide sweeps per EM round.  Here the LD blocks of all chromosomes sit in ONE plan (they are
independent units of the E-step whatever model they belong to):

* the chromosomes are SNP GROUPS of one device state (``viprs_state_set_groups``): ``viprs_state_prep_groups`` writes each
  group's E-step inputs from its own hyper-parameters, ONE sweep updates every group that is still iterating, and
  ``viprs_state_sums_groups_*`` returns the M-step / ELBO sums per group from one launch;
* the host side of an iteration (M-step, ELBO, stopping rules per model) is ``LockstepEM`` -- the vectorised form of
  ``VIPRS.m_step / elbo / fit`` in the serial fit's dtypes;
* a chromosome that has converged leaves the sweep (``viprs_plan_set_active_blocks``): its state stays as its last
  E-step left it, exactly as if its own ``fit()`` had returned.

A group's prep and sums are bit-identical to those of a plan that holds only that chromosome, so the batched fit
reproduces, bit for bit, what ``{c: VIPRS(loader_of_c).fit() for c in chromosomes}`` computes on this device -- and the
reference's per-chromosome trajectories at the common tolerances (tests/golden/fitchr_*.npz).

Results are keyed by chromosome: ``pi / tau_beta / sigma_epsilon / _sigma_g`` are dicts, ``history[c]["ELBO"]``,
``optim_results[c]``; the posterior dicts (``pip / post_mean_beta / post_var_beta / q``) have the usual layout.
"""
import contextlib
import logging

import numpy as np

from ..parallel import broadcast_from_root
from ..utils.optim import OptimizeResult
from ._lockstep import RESTART, LockstepEM
from .VIPRS import VIPRS, _is_numeric

logger = logging.getLogger(__name__)


class PerChromosomeGroups:
    """What the per-chromosome batches of every model family share (mixed in FRONT of the model class): the chromosomes as
    SNP groups of the one device state, their sample sizes / SNP counts / lambda_min, the convergence mask over LD blocks, the
    per-chromosome result tables."""

    _always_merge = True

    def __init__(self, gdl, lambda_min=None, **kwargs):
        """Arguments of the model class; ``lambda_min='infer'`` gives every chromosome the value of ITS LD matrix (each of the
        reference's per-chromosome models infers its own, VIPRS.py:186-191)."""
        infer = lambda_min is not None and not _is_numeric(lambda_min)
        super().__init__(gdl, lambda_min=None if infer else lambda_min, **kwargs)
        self.groups = sorted(self._all_shapes)                     # one model per chromosome of the data loader
        self._gindex = {c: g for g, c in enumerate(self.groups)}
        ss = gdl.sumstats_table
        self._n_group = np.array([float(np.max(ss[c].n_per_snp)) for c in self.groups])      # BayesPRSModel.py:75
        self._m_group = np.array([int(self._all_shapes[c]) for c in self.groups], dtype=np.int64)
        if infer:
            ld = gdl.get_ld_matrices()
            self._lambda_group = [ld[c].get_lambda_min(min_max_ratio=1e-3) for c in self.groups]
        else:
            self._lambda_group = [self.lambda_min] * len(self.groups)
        self.optim_results = {}
        self._em = None
        if self._e_step_fn is None:
            if not self._merged:
                raise NotImplementedError(type(self).__name__ + " runs on the device-resident one-plan layout "
                                          "(device_resident=True, merge_chromosomes=True)")
            ds, plan = self._dstate["*"], self._plans["*"]
            sizes = [int(self.shapes.get(c, 0)) for c in self.groups]
            self._group_start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
            ds.set_groups(self._group_start)
            starts, _ = plan.blocks()
            # (`right`: a chromosome without local SNPs shares its start with the next one, which owns the block)
            self._block_group = np.searchsorted(self._group_start, starts[:-1], side="right") - 1
            self._active_mask = None

    def _set_active(self, active_groups):
        """The chromosomes whose models still iterate: the LD blocks of the others leave the sweep."""
        if self._e_step_fn is not None:
            return
        flags = np.zeros(len(self.groups), dtype=bool)
        flags[active_groups] = True
        mask = flags[self._block_group]
        self._plans["*"].set_active_blocks(None if mask.all() else mask)

    def get_heritability(self):
        return {c: self._sigma_g[c] / (self._sigma_g[c] + self.sigma_epsilon[c]) for c in self.groups}

    def to_history_table(self):
        import pandas as pd
        return pd.concat([pd.DataFrame(h).assign(Chromosome=c) for c, h in self.history.items()])


class VIPRSPerChromosome(PerChromosomeGroups, VIPRS):

    # ---- per-group scalar context: lets the base class's scalar code run for one chromosome ------------------------
    @contextlib.contextmanager
    def _as_group(self, g, pi, sigma_epsilon, tau_beta, sigma_g=None):
        saved = (self.pi, self.sigma_epsilon, self.tau_beta, self._sigma_g, self._sample_size, self.lambda_min,
                 self._n_chroms_total)
        T = self._T.type
        self.pi, self.sigma_epsilon, self.tau_beta = pi, sigma_epsilon, tau_beta
        self._sigma_g = T(0.0) if sigma_g is None else sigma_g
        self._sample_size, self.lambda_min, self._n_chroms_total = float(self._n_group[g]), T(self._lambda_group[g]), 1
        try:
            yield
        finally:
            (self.pi, self.sigma_epsilon, self.tau_beta, self._sigma_g, self._sample_size, self.lambda_min,
             self._n_chroms_total) = saved

    def _theta_for(self, c, theta_0):
        """(pi, sigma_epsilon, tau_beta) of chromosome c's model as `VIPRS.initialize_theta` would leave them."""
        t0 = theta_0
        if isinstance(theta_0, dict) and theta_0 and all(k in self._gindex for k in theta_0):
            t0 = theta_0.get(c)                                    # {chromosome: theta_0}
        th = self._merge_theta(dict(t0) if t0 else None)
        return self._theta_values(th, int(self._m_group[self._gindex[c]]))

    def _cast_group_theta(self, raw):
        """The casts of `_cast_theta` for every group; several ranks take rank 0's values (random draws differ)."""
        if self.comm.world_size > 1:
            v = broadcast_from_root(self.comm, np.array([float(x) for t in raw for x in t], dtype=np.float64)).reshape(-1, 3)
            raw = [(float(v[g, 0]), float(v[g, 1]), float(v[g, 2])) for g in range(len(raw))]
        T = self._T.type
        return [(T(pi), T(sig), tau) for pi, sig, tau in raw]

    def _init_chromosome_state(self, c, pi, sigma_epsilon, tau_beta):
        """`VIPRS.initialize_variational_parameters` (VIPRS.py:330-359) for one local chromosome."""
        T, shp = self._T, self._shape(c)
        self.var_tau[c] = (self.n_per_snp[c] / sigma_epsilon) + tau_beta
        self.var_mu[c] = np.zeros(shp, T, order=self.order)
        self.var_gamma[c] = (pi * np.ones(shp, dtype=T, order=self.order)).astype(T, order=self.order)
        self.eta[c] = self.var_gamma[c] * self.var_mu[c]
        self.zeta[c] = np.multiply(self.var_gamma[c], self.var_mu[c].astype(np.float64) ** 2 + 1.0 / self.var_tau[c].astype(np.float64))
        self.eta_diff[c] = np.zeros_like(self.eta[c], dtype=T)
        self.q[c] = np.zeros_like(self.eta[c], dtype=T)
        self._log_var_tau[c] = np.log(self.var_tau[c])

    # ---- one lock-step iteration: E-step of the active groups, their sums ------------------------------------------------
    def _sweep(self, a, em):
        if self._e_step_fn is None:
            ds = self._dstate["*"]
            ds.prep_groups(em.prep_rows(a))
            ds.e_step(self.dequantize_scale, sync=False)
            self._host_stale = True
            return
        for g in a:                                               # CPU test hook: the oracle's kernel per chromosome
            c = self.groups[g]
            if c not in self.shapes:
                continue
            with self._as_group(g, *em.theta(g)):
                u_logs, shvt, mu_mult = self._prep(c)
                self._e_step_fn(self.ld_left_bound[c], self.ld_indptr[c], self.ld_data[c], self.std_beta[c],
                                self.var_gamma[c], self.var_mu[c], self.eta[c], self.q[c], self.eta_diff[c],
                                u_logs, shvt, mu_mult, self.dequantize_scale, self.threads, self.low_memory)
            self.zeta[c] = np.multiply(self.var_gamma[c], self.var_mu[c].astype(np.float64) ** 2
                                       + 1.0 / self.var_tau[c].astype(np.float64))

    def _host_group_sums(self, a):
        s = np.zeros((len(a), 11))
        for k, g in enumerate(a):
            c = self.groups[g]
            if c in self.shapes:
                with self._as_group(g, None, None, None):
                    s[k, :10] = self._host_partial_sums([c])       # [0] is already the mean over the chromosome
                s[k, 10] = float(np.max(np.abs(self.eta_diff[c]))) if self.eta_diff[c].size else 0.0
        return s

    def _group_sums(self, a, em, on_host=False):
        """(len(a), 11) rows in the layout of `viprs_state_sums`, [0] = mean of gamma over the chromosome, all ranks."""
        if self._e_step_fn is not None or on_host:
            s = self._host_group_sums(a)
            reduced = False
        else:
            ds = self._dstate["*"]
            ds.sums_groups_begin(a, em.lam1[a])
            s = ds.sums_groups_end()
            s[:, 0] /= self._m_group[a]
            reduced = self._device_reduce
        if self.comm.world_size > 1 and not reduced:
            tot = self.comm.allreduce_sum(np.ascontiguousarray(s[:, :10]).ravel()).reshape(-1, 10)
            mx = self.comm.allreduce_max(np.ascontiguousarray(s[:, 10]))
            s = np.column_stack([tot, mx])
        return s

    # ---- the fit -----------------------------------------------------------------------------------------------------
    def fit(self, max_iter=1000, theta_0=None, param_0=None, continued=False, disable_pbar=True, min_iter=3,
            f_abs_tol=1e-6, x_abs_tol=1e-6, patience=10, on_iteration=None, **kwargs):
        """All chromosomes' EM iterations together; arguments of ``VIPRS.fit``.  ``theta_0`` is one dict for every
        chromosome or ``{chromosome: dict}``."""
        if continued or param_0 is not None:
            raise NotImplementedError("VIPRSPerChromosome.fit: `continued` / `param_0` are not supported")
        T, G = self._T, len(self.groups)
        base_fixed = dict(self.fix_params)
        theta = self._cast_group_theta([self._theta_for(c, theta_0) for c in self.groups])
        th = [dict(pi=theta[g][0], sigma_epsilon=theta[g][1], tau_beta=theta[g][2], lam=T.type(self._lambda_group[g]),
                   fixed=set(base_fixed)) for g in range(G)]
        em = self._em = LockstepEM(T, th, self._m_group, self._n_group, n_chroms_total=1, min_iter=min_iter,
                                   f_abs_tol=f_abs_tol, x_abs_tol=x_abs_tol, patience=patience, restart_free_sigma=True)
        # ---- standard start of every model + its initial ELBO (VIPRS.py:330-359, update_theta_history) ----
        self.var_mu, self.var_tau, self.var_gamma, self._log_var_tau = {}, {}, {}, {}
        self.eta, self.zeta, self.eta_diff, self.q = {}, {}, {}, {}
        for c in self.chromosomes:
            self._init_chromosome_state(c, *theta[self._gindex[c]])
        self._host_stale = False
        self._push_state()
        self._set_active(np.arange(G))
        names = [t if isinstance(t, str) else t.__name__ for t in self.tracked_params]
        self.history = {c: dict({"ELBO": []}, **{n: [] for n in names}) for c in self.groups}
        all_groups = np.arange(G)
        s0 = self._group_sums(all_groups, em, on_host=True)
        for g, c in enumerate(self.groups):
            with self._as_group(g, *theta[g]):
                self._sums, self._sums_valid = s0[g, :10], True
                self.history[c]["ELBO"].append(VIPRS.elbo(self))
        self._sums, self._sums_valid = None, False
        self._track(all_groups, em)

        active = all_groups
        for i in range(1, max_iter + 1):
            if active.size == 0:
                break
            a = active
            em.mark_e_step(a)
            self._sweep(a, em)
            code = em.update(a, self._group_sums(a, em), i)
            for g in a:
                self.history[self.groups[g]]["ELBO"].append(float(em.elbos[g]))
            self._track(a, em)
            for g in a[code == RESTART]:
                self._restart_group(int(g), em, theta_0, i)
            keep = (code == 0) | (code == RESTART)
            if not keep.all():
                self._set_active(a[keep])
            active = a[keep]
            if on_iteration is not None:
                on_iteration(i)
        em.finish()
        self._set_active(all_groups)
        self.fix_params = base_fixed
        return self._publish(em)

    def _track(self, a, em):
        for t in self.tracked_params:
            for g in a:
                h, (pi, sig, tau) = self.history[self.groups[g]], em.theta(g)
                if t == "pi":
                    h["pi"].append(pi)
                elif t == "heritability":
                    h["heritability"].append(em.sigma_g[g] / (em.sigma_g[g] + sig))
                elif t == "sigma_epsilon":
                    h["sigma_epsilon"].append(sig)
                elif t == "tau_beta":
                    h["tau_beta"].append(tau)
                elif t == "sigma_g":
                    h["sigma_g"].append(em.sigma_g[g])
                elif t == "max_eta_diff":
                    h["max_eta_diff"].append(em.max_eta_diff[g])
                elif callable(t):
                    raise NotImplementedError("callable tracked_params are not supported by the per-chromosome fit")

    def _restart_group(self, g, em, theta_0, i):
        """Negative MSE with a free sigma_epsilon: that chromosome's model starts again with sigma_epsilon = 0.95
        fixed (VIPRS.py:1025-1037).  Rare; the state of the chromosome is re-initialised through the host."""
        c = self.groups[g]
        logger.info("Chromosome %s | iteration %d | MSE is negative; restarting with sigma_epsilon fixed.", c, i)
        (pi, sig, tau), = self._cast_group_theta([self._theta_for(c, theta_0)])
        if self._e_step_fn is None:
            self._pull_state()
        if c in self.shapes:
            self._init_chromosome_state(c, pi, sig, tau)
        if self._e_step_fn is None:
            self._push_state()
        em.restart(g, pi, 0.95, tau)

    def _publish(self, em):
        groups = self.groups
        # NumPy state back from the device; var_tau is what the LAST E-step of each chromosome was built from
        if self._e_step_fn is None and self._host_stale:
            self._pull_state()
        for c in self.chromosomes:
            g = self._gindex[c]
            lam = self._T.type(self._lambda_group[g])
            self.var_tau[c] = (self.n_per_snp[c] * (1.0 + lam) / em.sig_e[g]) + em.tau_e[g]
            self._log_var_tau[c] = np.log(self.var_tau[c])
        self._host_stale = False
        self.zeta = self.compute_zeta()
        self.pi = {c: em.theta(g)[0] for g, c in enumerate(groups)}
        self.sigma_epsilon = {c: em.theta(g)[1] for g, c in enumerate(groups)}
        self.tau_beta = {c: em.theta(g)[2] for g, c in enumerate(groups)}
        self._sigma_g = {c: em.sigma_g[g] for g, c in enumerate(groups)}
        self.optim_results = {c: em.results[g] for g, c in enumerate(groups)}
        res = self.optim_result = OptimizeResult()
        res.nit = max(r.nit for r in em.results)
        res.success = all(r.success for r in em.results)
        res.stop_iteration = True
        res.fun = float(np.sum(em.elbos))
        failed = [str(c) for c in groups if not self.optim_results[c].success]
        res.message = "All chromosomes converged." if not failed else "Not converged: chromosome(s) " + ", ".join(failed)
        self.update_posterior_moments()
        self._gather_posterior()
        for c in groups:
            if not self.optim_results[c].success:
                logger.warning("\tchromosome %s: %s", c, self.optim_results[c].message)
        return self

    # ---- summaries: one value per chromosome ------------------------------------------------------------------------------
    def elbo(self, sum_axis=None):
        """Final ELBO of every chromosome's model."""
        return {c: h["ELBO"][-1] for c, h in self.history.items()}

    objective = elbo

    def get_proportion_causal(self):
        return dict(self.pi)

    def get_average_effect_size_variance(self):
        return {c: float(np.float64(self.pi[c]) / np.float64(self.tau_beta[c])) for c in self.groups}

    def to_theta_table(self):
        """Hyper-parameter rows of every chromosome's model with a `Chromosome` column (the table `viprs_fit` writes,
        bin/viprs_fit:1100-1106)."""
        import pandas as pd
        h2, rows = self.get_heritability(), []
        for c in self.groups:
            for k, v in (("ELBO", self.history[c]["ELBO"][-1]), ("Residual_variance", self.sigma_epsilon[c]),
                         ("Heritability", h2[c]), ("Proportion_causal", self.pi[c]),
                         ("Average_effect_variance", float(np.float64(self.pi[c]) / np.float64(self.tau_beta[c]))),
                         ("Lambda_min", self._lambda_group[self._gindex[c]]), ("tau_beta", float(self.tau_beta[c]))):
                rows.append({"Parameter": k, "Value": v, "Chromosome": c})
        return pd.DataFrame(rows)
