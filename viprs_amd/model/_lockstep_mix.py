"""Host side of an EM iteration for SEVERAL independent ``VIPRSMix`` models that advance in lock step on one device state
(the per-chromosome models of ``VIPRSMixPerChromosome``): ``VIPRSMix.m_step`` (VIPRSMix.py:227-260), ``VIPRSMix.elbo`` /
``VIPRS.elbo`` (VIPRS.py:497-581) and the stopping rules of ``VIPRS.fit`` (:1003-1094) over the ACTIVE models' rows of device
sums as a handful of NumPy calls on (models, K) arrays instead of ~60 NumPy calls on K-vectors per model.

As in ``_lockstep.LockstepEM`` the arithmetic follows the serial fit's DTYPES, which decide roundings: ``pi`` is always of the
state precision (``_cast_theta`` / ``m_step`` cast it); ``tau_beta`` is a float64 vector once the M-step has written it, before
that whatever ``initialize_theta`` left (float64, or the state precision when it came from a fixed scalar); ``sigma_epsilon`` a
scalar of the state precision until the M-step makes it a float64.  Row sums over the K components reduce in NumPy's order
for a K-vector (checked: ``x.sum(axis=1)[i] == x[i].sum()`` bit for bit); the one BLAS call of the M-step (``np.dot(d, kv)``)
is made per model.  `VIPRSMixPerChromosome(host="scalar")` runs the serial code itself per model: the two must agree `==`
(tests/test_per_chromosome_mix.py).
"""
import numpy as np

from ..utils.optim import OptimizeResult
from ._lockstep import MAX_ITER_MESSAGE, MESSAGES, RESTART, SUCCESS, _close, _is32, in_dtype

f32, f64 = np.float32, np.float64


def _vec_is32(v):
    return isinstance(v, np.ndarray) and v.dtype == np.float32


class LockstepMixEM:
    """Hyper-parameters, ELBO history and stopping state of G K-component models as arrays.

    :param T: the state precision (np.dtype).
    :param th: per model ``dict(pi=(K,), sigma_epsilon=, tau_beta=(K,), lam=, fixed=dict)`` with the values as
        ``VIPRSMix.initialize_theta`` leaves them (their NumPy types matter); ``fixed`` is the model's ``fix_params``.
    :param d: the prior multipliers (``VIPRSMix.d``, state precision).
    :param n_snps: variants per model; :param n: sample size per model.
    """

    def __init__(self, T, th, d, n_snps, n, min_iter=3, f_abs_tol=1e-6, x_abs_tol=1e-6, patience=10):
        G = len(th)
        self.T, self.G, self.K = np.dtype(T), G, len(d)
        self.d = np.asarray(d)
        self.d64 = self.d.astype(f64)
        self.pi = np.array([np.asarray(p["pi"]) for p in th], dtype=self.T).reshape(G, self.K)
        self.tau = np.array([np.asarray(p["tau_beta"], dtype=f64) for p in th]).reshape(G, self.K)
        self.tau_is32 = np.array([_vec_is32(np.asarray(p["tau_beta"])) for p in th])
        self.sig = np.array([p["sigma_epsilon"] for p in th], dtype=f64)
        self.sig_is32 = np.array([_is32(p["sigma_epsilon"]) for p in th])
        # (a restart leaves a PYTHON float 0.95: a weak scalar in NumPy's promotion, unlike np.float64 -- it matters where the
        #  host forms var_tau from the (m, 1) array of the state precision)
        self.sig_is_py = np.array([type(p["sigma_epsilon"]) is float for p in th])
        self.fx_pis = np.array(["pis" in p["fixed"] for p in th])
        self.fx_pi = np.array(["pi" in p["fixed"] for p in th])                 # the OVERALL proportion is fixed (VIPRSMix.py:238)
        self.fix_pi_value = np.array([float(p["fixed"].get("pi", 0.0)) for p in th], dtype=f64)
        self.fx_tau = np.array(["tau_betas" in p["fixed"] for p in th])
        self.fx_sig = np.array(["sigma_epsilon" in p["fixed"] for p in th])
        self.lam1 = np.array([float(1.0 + p["lam"]) for p in th], dtype=f64)
        self.lam = [p["lam"] for p in th]
        # what var_tau of each model's LAST E-step was built from (values and the dtypes they had)
        self.sig_e, self.tau_e = self.sig.copy(), self.tau.copy()
        self.sig_e_is32, self.tau_e_is32, self.sig_e_is_py = self.sig_is32.copy(), self.tau_is32.copy(), self.sig_is_py.copy()
        self.e_done = np.zeros(G, dtype=bool)
        self.n_snps = np.asarray(n_snps, dtype=np.int64)
        self._n_snps_T = self.n_snps.astype(self.T)                             # (a T scalar times a Python int is a T product)
        self.n = np.asarray(n, dtype=f64)
        self.min_iter, self.f_abs_tol, self.x_abs_tol, self.patience = min_iter, f_abs_tol, x_abs_tol, patience
        self.results = [OptimizeResult() for _ in range(G)]
        self.sigma_g = np.zeros(G)
        self.prev_elbo = np.full(G, -np.inf)
        self.prev_sigma_g = np.zeros(G)
        self.plateau_n, self.dropping_n = np.zeros(G, dtype=np.int64), np.zeros(G, dtype=np.int64)
        self.elbos = np.zeros(G)
        self.max_eta_diff = np.zeros(G)

    # ---- E-step inputs ------------------------------------------------------------------------------------------------------
    def prep_rows(self, a):
        """Rows (model, log_null_pi, sigma_epsilon, 1 + lambda, logit_pi[K], log_tau_beta[K], tau_beta[K]) of
        `viprs_state_prep_mixture_groups` for models `a` (the expressions and dtypes of `VIPRSMix.e_step`)."""
        pa = self.pi[a]
        logit = (np.log(pa) - np.log(1.0 - pa)).astype(f64)
        log_null = np.log(1.0 - pa.sum(axis=1)).astype(f64)                     # (in the state precision, VIPRSMix.py:196)
        m32 = self.tau_is32[a][:, None] & np.ones((1, self.K), dtype=bool)
        log_tau = in_dtype(self.tau[a], m32, np.log)
        return np.column_stack([a.astype(f64), log_null, self.sig[a], self.lam1[a], logit, log_tau, self.tau[a]])

    def mark_e_step(self, a):
        self.sig_e[a], self.tau_e[a], self.e_done[a] = self.sig[a], self.tau[a], True
        self.sig_e_is32[a], self.tau_e_is32[a], self.sig_e_is_py[a] = self.sig_is32[a], self.tau_is32[a], self.sig_is_py[a]

    def last_prep(self, g):
        """(sigma_epsilon, tau_beta, lambda_min) of model g's last E-step in the dtypes they had (None: never swept)."""
        if not self.e_done[g]:
            return None
        sig = f32(self.sig_e[g]) if self.sig_e_is32[g] else float(self.sig_e[g]) if self.sig_e_is_py[g] else self.sig_e[g]
        tau = self.tau_e[g].astype(f32) if self.tau_e_is32[g] else self.tau_e[g].copy()
        return sig, tau, self.lam[g]

    # ---- M-step, ELBO, stopping rules ---------------------------------------------------------------------------------------
    def update(self, a, s, i):
        """Models `a` on iteration `i` from their sums `s` ((len(a), 7 + 6 K), the layout of `viprs_state_sums_mixture_end`).
        Returns the stop code per model (0 = keeps going, `RESTART` = negative MSE with a free sigma_epsilon)."""
        T, K = self.T, self.K
        kv = [s[:, 6 + j * K: 6 + (j + 1) * K] for j in range(6)]
        fx_sig = self.fx_sig[a]
        # ---- VIPRSMix.m_step (VIPRSMix.py:227-260) ----
        est = kv[0]
        fxp = self.fx_pi[a]
        if fxp.any():
            with np.errstate(all="ignore"):
                est = np.where(fxp[:, None], self.fix_pi_value[a][:, None] * est / est.sum(axis=1)[:, None],
                               est / self.n_snps[a][:, None])
        else:
            est = est / self.n_snps[a][:, None]
        self.pi[a] = np.where(self.fx_pis[a][:, None], self.pi[a], est.astype(T))
        upd = ~self.fx_tau[a]
        if upd.any():
            pa = self.pi[a]
            num = pa.sum(axis=1) * self._n_snps_T[a]                            # np.sum(pi) * m: a product in the state precision
            dots = np.array([np.dot(self.d, kv[1][k]) for k in range(len(a))])  # (BLAS, per model: its order is its own)
            tau_s = num.astype(f64) / dots
            new_tau = np.clip(self.d64[None, :] * tau_s[:, None], 1.0, None)    # d * (a float64 scalar): float64
            self.tau[a] = np.where(upd[:, None], new_tau, self.tau[a])
            self.tau_is32[a] &= ~upd
        self.sigma_g[a] = s[:, 1]
        upd = ~fx_sig
        self.sig[a] = np.where(upd, (1.0 + (-2.0 * s[:, 2]).astype(T)) + self.sigma_g[a], self.sig[a])
        self.sig_is32[a] &= ~upd
        self.sig_is_py[a] &= ~upd
        # ---- VIPRSMix.elbo in the serial fit's dtypes ----
        sg, sa, ta = self.sigma_g[a], self.sig[a], self.tau[a]
        pa = self.pi[a]
        sig32 = self.sig_is32[a]
        tau32 = self.tau_is32[a][:, None] & np.ones((1, K), dtype=bool)
        pi64 = pa.astype(f64)
        null_pi = 1.0 - pa.sum(axis=1)                                          # (state precision, VIPRSMix.get_null_pi)
        with np.errstate(all="ignore"):
            e = in_dtype(sa, sig32, lambda v: -np.log(2.0 * np.pi * v))
            e = np.where(fx_sig, e - in_dtype(sa, sig32, lambda v: 1.0 / v) * (1.0 - 2.0 * s[:, 2] + sg), e - 1.0)
            e = e * (0.5 * self.n[a])
            e = e - (kv[2] - np.log(pi64) * kv[3]).sum(axis=1)
            e = e - (s[:, 4] - np.log(null_pi).astype(f64) * s[:, 5])
            e = e + 0.5 * ((in_dtype(ta, tau32, lambda v: 1.0 + np.log(v))) * kv[3] - kv[4]).sum(axis=1)
            e = e - 0.5 * (ta * kv[5]).sum(axis=1)
        self.elbos[a] = e
        self.max_eta_diff[a] = s[:, -1]
        mse = 1.0 - 2.0 * s[:, 2] + (sg - s[:, 0] + s[:, 3])                    # VIPRSMix.mse
        with np.errstate(all="ignore"):
            h2 = sg / (sg + sa)
        # ---- VIPRS.fit stopping rules (VIPRS.py:1003-1080), first match wins ----
        min_iter, x_abs_tol, f_abs_tol = self.min_iter, self.x_abs_tol, self.f_abs_tol
        prev_e, late = self.prev_elbo[a], i > min_iter
        pl = late & _close(sg, self.prev_sigma_g[a], x_abs_tol) & (s[:, -1] < x_abs_tol * 10)
        dr = (e < prev_e) & ~_close(e, prev_e, 1e3 * f_abs_tol, 1e-4)
        pn = self.plateau_n[a] = np.where(pl, self.plateau_n[a] + 1, 0)
        dn = self.dropping_n[a] = np.where(dr, self.dropping_n[a] + 1, 0)
        code = np.zeros(len(a), dtype=np.int64)
        for c, cond in ((8, dn > self.patience), (7, pn > self.patience), (6, late & (s[:, -1] < x_abs_tol)),
                        (5, late & _close(prev_e, e, f_abs_tol)), (4, (h2 > 1.0) | (h2 < 0.0)), (3, sa < 0.0),
                        (2, ~np.isfinite(e)), (1, mse < 0.0)):                  # (applied last = matched first)
            code = np.where(cond, c, code)
        code = np.where((code == 1) & ~fx_sig, RESTART, code)
        for k, g in enumerate(a):
            c = int(code[k])
            if c == 0:
                self.results[g].update(float(e[k]))
            elif c != RESTART:
                msg = MESSAGES[c].format(float(mse[k])) if c == 1 else MESSAGES[c]
                self.results[g].update(float(e[k]), stop_iteration=True, success=SUCCESS[c], message=msg)
        keep = code != RESTART              # (VIPRS.fit returns from the iteration before the bookkeeping on a restart)
        self.prev_elbo[a[keep]], self.prev_sigma_g[a[keep]] = e[keep], sg[keep]
        return code

    def restart(self, g, pi, sigma_epsilon, tau_beta):
        """Model g starts again from (pi, tau_beta) with sigma_epsilon FIXED at the given value (VIPRS.py:1030-1036)."""
        self.pi[g] = np.asarray(pi, dtype=self.T)
        self.tau[g], self.tau_is32[g] = np.asarray(tau_beta, dtype=f64), _vec_is32(np.asarray(tau_beta))
        self.sig[g], self.sig_is32[g], self.sig_is_py[g] = sigma_epsilon, _is32(sigma_epsilon), type(sigma_epsilon) is float
        self.fx_sig[g] = True

    def finish(self):
        for g in range(self.G):
            if not self.results[g].stop_iteration:
                self.results[g].update(self.elbos[g], stop_iteration=True, success=False, increment=False,
                                       message=MAX_ITER_MESSAGE)

    def theta(self, g):
        """(pi, sigma_epsilon, tau_beta) of model g in the serial fit's dtypes."""
        sig = f32(self.sig[g]) if self.sig_is32[g] else float(self.sig[g]) if self.sig_is_py[g] else self.sig[g]
        tau = self.tau[g].astype(f32) if self.tau_is32[g] else self.tau[g].copy()
        return self.pi[g].copy(), sig, tau
