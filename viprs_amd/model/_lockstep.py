"""Host side of an EM iteration for SEVERAL independent spike-and-slab models that advance in lock step on one device
state: the grid points of ``VIPRSGrid.fit(batched=True)`` (columns of a grid state) and the per-chromosome models of
``VIPRSPerChromosome`` (SNP groups of a spike-and-slab state).

Per model it is ``VIPRS.m_step`` (VIPRS.py:426-484), ``VIPRS.elbo`` (:497-581) and the stopping rules of ``VIPRS.fit``
(:1003-1094) over the model's row of device sums, written as a handful of NumPy calls over the ACTIVE models instead of
~40 Python statements per model (for 32 models that loop cost a third of the batched sweep).  The arithmetic follows
the serial fit's DTYPES, which decide roundings: a hyper-parameter that is fixed stays a scalar of the state precision
(``VIPRS.set_fixed_params`` / ``_cast_theta``), one that the M-step updates becomes a float64 (``VIPRS.m_step``) --
``*_is32`` track which is which, and ``in_dtype`` evaluates an expression in float32 for the former.  (Array ufuncs give
the same bits as the scalar calls they replace.)
"""
import numpy as np

from ..utils.optim import OptimizeResult

MESSAGES = (None, "The MSE is negative ({:.6f}).", "Objective (ELBO) is undefined.", "Residual variance estimate is negative.",
            "Estimated heritability is out of bounds.", "Objective (ELBO) converged successfully.",
            "Variational parameters converged successfully.", "LD-weighted variational parameters converged successfully.",
            "The objective (ELBO) is decreasing.")
SUCCESS = (False, False, False, False, False, True, True, True, False)
RESTART = 9        # negative MSE while sigma_epsilon is free: VIPRS.fit restarts that model with sigma_epsilon fixed (:1025-1037)
MAX_ITER_MESSAGE = "Maximum iterations reached without convergence.\n" "You may need to run the model for more iterations."

f32, f64 = np.float32, np.float64


def _is32(v):
    return isinstance(v, np.floating) and v.dtype == np.float32


def _close(a, b, atol, rtol=0.0):
    """np.isclose(a, b, atol=atol, rtol=rtol) for the finite / one-sided-infinite values that occur here, without its
    argument checking (it costs 25 us a call; three calls per iteration)."""
    with np.errstate(invalid="ignore"):
        return np.abs(a - b) <= atol + rtol * np.abs(b)


def in_dtype(v, m32, fn):
    """fn evaluated in float32 where the serial fit holds a float32 scalar (mask m32), in float64 elsewhere."""
    out = fn(v)
    if m32.any():
        out = np.where(m32, fn(v.astype(f32)).astype(f64), out)
    return out


class LockstepEM:
    """Hyper-parameters, ELBO history and stopping state of G models as arrays.

    :param T: the state precision (np.dtype).
    :param th: per model ``dict(pi=, sigma_epsilon=, tau_beta=, lam=, fixed=set)`` with the values as
        ``VIPRS.initialize_theta`` leaves them (their NumPy scalar types matter).
    :param n_snps: variants of a model (``VIPRS.n_snps``): one int, or one per model.
    :param n: sample size of a model (``VIPRS.n``): one float, or one per model.
    :param m_mean: what sum [0] is divided by to give pi (``update_pi``): None when the sums already carry the mean of
        gamma (per-SNP weights on the device), otherwise one count per model.
    :param restart_free_sigma: a model whose MSE turns negative while its sigma_epsilon is free is reported with code
        `RESTART` instead of being stopped (the caller re-initialises it and calls `restart`).
    """

    def __init__(self, T, th, n_snps, n, n_chroms_total=1, m_mean=None, min_iter=3, f_abs_tol=1e-6, x_abs_tol=1e-6,
                 patience=10, restart_free_sigma=False):
        G = len(th)
        self.T, self.G = np.dtype(T), G
        self.th = th
        self.pi = np.array([p["pi"] for p in th], dtype=self.T)              # always of the state precision (m_step casts)
        self.sig = np.array([p["sigma_epsilon"] for p in th], dtype=f64)
        self.tau = np.array([p["tau_beta"] for p in th], dtype=f64)
        self.sig_is32 = np.array([_is32(p["sigma_epsilon"]) for p in th])
        self.tau_is32 = np.array([_is32(p["tau_beta"]) for p in th])
        self.fx_pi = np.array(["pi" in p["fixed"] for p in th])
        self.fx_tau = np.array(["tau_beta" in p["fixed"] for p in th])
        self.fx_sig = np.array(["sigma_epsilon" in p["fixed"] for p in th])
        self.lam1 = np.array([float(1.0 + p["lam"]) for p in th], dtype=f64)
        self.sig_e, self.tau_e = self.sig.copy(), self.tau.copy()           # what var_tau of the last E-step was built from
        # pi * n_snps is a product in the state precision in the serial fit (a T scalar times a Python int)
        self._n_snps_scalar = np.ndim(n_snps) == 0
        self.n_snps = n_snps if self._n_snps_scalar else np.asarray(n_snps)
        self._n_snps_T = None if self._n_snps_scalar else np.asarray(n_snps).astype(self.T)
        self.n = float(n) if np.ndim(n) == 0 else np.asarray(n, dtype=f64)
        self.n_chroms_total = n_chroms_total
        self.m_mean = None if m_mean is None else np.asarray(m_mean, dtype=f64)
        self.min_iter, self.f_abs_tol, self.x_abs_tol, self.patience = min_iter, f_abs_tol, x_abs_tol, patience
        self.restart_free_sigma = restart_free_sigma
        self.results = [OptimizeResult() for _ in range(G)]
        self.sigma_g = np.zeros(G)
        self.prev_elbo = np.full(G, -np.inf)
        self.prev_sigma_g = np.zeros(G)
        self.plateau_n, self.dropping_n = np.zeros(G, dtype=np.int64), np.zeros(G, dtype=np.int64)     # ConditionStreak counters
        self.elbos = np.zeros(G)
        self.max_eta_diff = np.zeros(G)
        self.last_mse = np.zeros(G)

    def _per(self, v, a):
        return v if np.ndim(v) == 0 else v[a]

    def prep_rows(self, a):
        """Rows (model, logit_pi, log_tau_beta, sigma_epsilon, tau_beta, one_plus_lambda) of the device prep for models `a`."""
        pa = self.pi[a]
        logit = (np.log(pa) - np.log(1.0 - pa)).astype(f64)
        return np.column_stack([a.astype(f64), logit, in_dtype(self.tau[a], self.tau_is32[a], np.log), self.sig[a], self.tau[a],
                                self.lam1[a]])

    def mark_e_step(self, a):
        self.sig_e[a], self.tau_e[a] = self.sig[a], self.tau[a]

    def update(self, a, s, i):
        """M-step, ELBO and stopping rules of models `a` on iteration `i` from their sums `s` ((len(a), 11), the layout of
        `viprs_state_sums`).  Returns the stop code per model (0 = keeps going)."""
        T = self.T
        fx_pi, fx_tau, fx_sig = self.fx_pi[a], self.fx_tau[a], self.fx_sig[a]
        # ---- VIPRS.m_step, per model (VIPRS.py:426-484) ----
        mean_g = s[:, 0] if self.m_mean is None else s[:, 0] / self.m_mean[a]
        self.pi[a] = np.where(fx_pi, self.pi[a], (mean_g / self.n_chroms_total).astype(T))
        upd = ~fx_tau
        nsn = self.n_snps if self._n_snps_scalar else self._n_snps_T[a]
        self.tau[a] = np.where(upd, self.pi[a] * nsn / s[:, 1], self.tau[a])
        self.tau_is32[a] &= ~upd
        self.sigma_g[a] = s[:, 2]
        upd = ~fx_sig
        self.sig[a] = np.where(upd, (1.0 + (-2.0 * s[:, 3]).astype(T)) + self.sigma_g[a], self.sig[a])
        self.sig_is32[a] &= ~upd
        # ---- ELBO (VIPRS.py:497-581) in the serial fit's dtypes ----
        sg, sa, ta, pa = self.sigma_g[a], self.sig[a], self.tau[a], self.pi[a]
        sig32, tau32 = self.sig_is32[a], self.tau_is32[a]
        e = in_dtype(sa, sig32, lambda v: -np.log(2.0 * np.pi * v))
        e = np.where(fx_sig, e - in_dtype(sa, sig32, lambda v: 1.0 / v) * (1.0 - 2.0 * s[:, 3] + sg), e - 1.0)
        e = e * (0.5 * self._per(self.n, a))
        e = e - (s[:, 5] - np.log(pa) * s[:, 7])
        e = e - (s[:, 6] - np.log(1.0 - pa) * s[:, 8])
        e = e + 0.5 * (in_dtype(ta, tau32, lambda v: 1.0 + np.log(v)) * s[:, 7] - s[:, 9])
        e = e - 0.5 * ta * s[:, 1]
        self.elbos[a] = e
        self.max_eta_diff[a] = s[:, 10]
        mse = 1.0 - 2.0 * s[:, 3] + (sg - s[:, 1] + s[:, 4])
        self.last_mse[a] = mse
        h2 = sg / (sg + sa)
        # ---- VIPRS.fit stopping rules (VIPRS.py:1003-1080), first match wins ----
        min_iter, x_abs_tol, f_abs_tol = self.min_iter, self.x_abs_tol, self.f_abs_tol
        prev_e, late = self.prev_elbo[a], i > min_iter
        pl = late & _close(sg, self.prev_sigma_g[a], x_abs_tol) & (s[:, 10] < x_abs_tol * 10)
        dr = (e < prev_e) & ~_close(e, prev_e, 1e3 * f_abs_tol, 1e-4)
        pn = self.plateau_n[a] = np.where(pl, self.plateau_n[a] + 1, 0)
        dn = self.dropping_n[a] = np.where(dr, self.dropping_n[a] + 1, 0)
        code = np.zeros(len(a), dtype=np.int64)
        for c, cond in ((8, dn > self.patience), (7, pn > self.patience), (6, late & (s[:, 10] < x_abs_tol)),
                        (5, late & _close(prev_e, e, f_abs_tol)), (4, (h2 > 1.0) | (h2 < 0.0)), (3, sa < 0.0),
                        (2, ~np.isfinite(e)), (1, mse < 0.0)):          # (applied last = matched first)
            code = np.where(cond, c, code)
        if self.restart_free_sigma:
            code = np.where((code == 1) & ~fx_sig, RESTART, code)
        for k, g in enumerate(a):
            c = int(code[k])
            if c == 0:
                self.results[g].update(float(e[k]))
            elif c != RESTART:
                msg = MESSAGES[c].format(float(mse[k])) if c == 1 else MESSAGES[c]
                self.results[g].update(float(e[k]), stop_iteration=True, success=SUCCESS[c], message=msg)
        keep = code != RESTART              # (VIPRS.fit `continue`s past the bookkeeping on a restart)
        self.prev_elbo[a[keep]], self.prev_sigma_g[a[keep]] = e[keep], sg[keep]
        return code

    def restart(self, g, pi, sigma_epsilon, tau_beta):
        """Model g starts again from (pi, tau_beta) with sigma_epsilon FIXED at the given value (VIPRS.py:1030-1036)."""
        self.pi[g] = pi
        self.tau[g], self.tau_is32[g] = tau_beta, _is32(tau_beta)
        self.sig[g], self.sig_is32[g] = sigma_epsilon, _is32(sigma_epsilon)
        self.fx_sig[g] = True

    def finish(self):
        """Models that never stopped: the maximum-iterations record (VIPRS.py:1107-1114)."""
        for g in range(self.G):
            if not self.results[g].stop_iteration:
                self.results[g].update(self.elbos[g], stop_iteration=True, success=False, increment=False,
                                       message=MAX_ITER_MESSAGE)

    def theta(self, g):
        """(pi, sigma_epsilon, tau_beta) of model g in the serial fit's dtypes."""
        sig = f32(self.sig[g]) if self.sig_is32[g] else self.sig[g]
        tau = f32(self.tau[g]) if self.tau_is32[g] else self.tau[g]
        return self.pi[g], sig, tau
