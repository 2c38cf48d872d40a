"""``VIPRSMix`` -- sparse scale-mixture prior (K causal components), E-step on MI355X.

Mirrors the surface and update rules of viprs/model/VIPRSMix.py; the native call it replaces is
``cpp_e_step_mixture`` (e_step_cpp.pyx:125-159 -> e_step.hpp:447-551).  State arrays have shape
(m, K), C-ordered, as the reference forces (VIPRSMix.py:40).
"""
import numpy as np

from .VIPRS import VIPRS, _DOUBLE_RES


class VIPRSMix(VIPRS):

    def __init__(self, gdl, K=1, prior_multipliers=None, **kwargs):
        kwargs["order"] = "C"
        assert K > 0
        self.K = K
        super().__init__(gdl, **kwargs)
        if prior_multipliers is not None:
            assert len(prior_multipliers) == K
            self.d = np.array(prior_multipliers).astype(self._T)
        else:
            self.d = 2 ** np.linspace(-min(K - 1, 7), 0, K).astype(self._T)            # VIPRSMix.py:52
        self.n_per_snp = {c: n[:, None].astype(self._T, order=self.order) for c, n in self.n_per_snp.items()}

    def _shape(self, c):
        return (self.shapes[c], self.K)

    def _make_device_state(self, plan):
        from ..plan import DeviceState
        return DeviceState(plan, self.float_precision, "mixture", self.K)

    def _supports_resident(self):
        return self.K <= 8    # device prep / sums kernels cover the panel kernels' K range

    def _supports_merged(self):
        return True

    # ---- hyper-parameter initialisation (VIPRSMix.py:60-167) --------------------------------------
    def initialize_theta(self, theta_0=None):
        th = self._merge_theta(theta_0)
        if "pis" in th:
            self.pi = np.asarray(th["pis"])
        else:
            overall = th["pi"] if "pi" in th else np.random.uniform(low=max(0.005, 1.0 / self.n_snps), high=0.1)
            self.pi = overall * np.random.dirichlet(np.ones(self.K))
        if "sigma_epsilon" in th:
            self.sigma_epsilon = th["sigma_epsilon"]
            if "tau_betas" in th:
                self.tau_beta = th["tau_betas"]
            elif "tau_beta" in th:
                self.tau_beta = np.repeat(th["tau_beta"], self.K)
            else:
                self.tau_beta = self.d * (self.n_snps * np.dot(1.0 / self.d, self.pi) / (1.0 - self.sigma_epsilon))
        elif "tau_betas" in th:
            self.tau_beta = th["tau_betas"]
            self.sigma_epsilon = np.clip(1.0 - np.dot(1.0 / self.tau_beta, self.pi), 1e-4, 1.0 - 1e-4)
        elif "tau_beta" in th:
            self.tau_beta = th["tau_beta"] * self.d
            self.sigma_epsilon = np.clip(1.0 - (self.n_snps * self.pi / self.tau_beta).sum(), 1e-4, 1.0 - 1e-4)
        else:
            h2 = np.random.uniform(low=0.001, high=0.999)
            self.sigma_epsilon = 1.0 - h2
            self.tau_beta = self.d * (self.n_snps * np.dot(1.0 / self.d, self.pi) / h2)
        self._cast_theta()

    def initialize_variational_parameters(self, param_0=None):
        super().initialize_variational_parameters(param_0)
        if self._resident:
            # the reference's ELBO keeps the log var_tau of the initial state: device copy
            for key, ds in self._dstate.items():
                chroms = self.chromosomes if key == "*" else [key]
                ds.set_log_var_tau(np.concatenate([np.asarray(self._log_var_tau[c], dtype=np.float64)
                                                   * np.ones(self._shape(c)) for c in chroms]))

    def sync_host(self):
        """Refresh the NumPy state from the GPU; `_log_var_tau` stays the initial one (as on the host path)."""
        if self._resident and self._host_stale:
            self._pull_state()
            if self._last_prep is not None:
                sigma_epsilon, tau_beta, lam = self._last_prep
                for c in self.chromosomes:
                    self.var_tau[c] = (self.n_per_snp[c] * (1.0 + lam) / sigma_epsilon) + tau_beta
            self.zeta = self.compute_zeta()
            self._host_stale = False

    def get_null_pi(self, chrom=None):
        return 1.0 - np.sum(self.get_pi(chrom))

    def get_proportion_causal(self):
        return np.sum(self.pi)

    # ---- E-step (VIPRSMix.py:169-225) ----------------------------------------------------------------
    def _prep(self, c):
        tau_beta, pi = self.get_tau_beta(c), self.get_pi(c)
        self.var_tau[c] = (self.n_per_snp[c] * (1.0 + self.lambda_min) / self.sigma_epsilon) + tau_beta
        T = self._T
        log_null_pi = (np.ones_like(self.eta[c]) * np.log(1.0 - self.pi.sum())).astype(T)
        mu_mult = np.ascontiguousarray((self.n_per_snp[c] / (self.var_tau[c] * self.sigma_epsilon)).astype(T))
        u_logs = np.ascontiguousarray((np.log(pi) - np.log(1.0 - pi)
                                       + 0.5 * (np.log(tau_beta) - np.log(self.var_tau[c]))).astype(T))
        shvt = np.ascontiguousarray(np.sqrt(0.5 * self.var_tau[c]).astype(T))
        return log_null_pi, u_logs, shvt, mu_mult

    def e_step(self):
        if self._resident:
            # whole iteration on the device: prep kernel + sweep per plan, nothing crosses PCIe
            pi, tau_beta = np.asarray(self.pi), np.asarray(self.tau_beta)
            logit_pi = np.log(pi) - np.log(1.0 - pi)                     # dtype semantics of VIPRSMix.py:211
            log_null_pi = np.log(1.0 - self.pi.sum())
            for ds in self._dstate.values():
                ds.prep_mixture(logit_pi, np.log(tau_beta), tau_beta, log_null_pi, self.sigma_epsilon,
                                1.0 + self.lambda_min)
                ds.e_step(self.dequantize_scale, sync=False)
            self._last_prep = (self.sigma_epsilon, tau_beta, self.lambda_min)
            self._host_stale = True
            self._sums_valid = False
            return
        if self._e_step_fn is not None:
            for c in self.chromosomes:
                log_null_pi, u_logs, shvt, mu_mult = self._prep(c)
                self._e_step_fn(self.ld_left_bound[c], self.ld_indptr[c], self.ld_data[c], self.std_beta[c],
                                self.var_gamma[c], self.var_mu[c], self.eta[c], self.q[c], self.eta_diff[c],
                                log_null_pi, u_logs, shvt, mu_mult, self.dequantize_scale, self.threads,
                                self.low_memory)
        else:
            for c in self.chromosomes:
                log_null_pi, u_logs, shvt, mu_mult = self._prep(c)
                ds = self._dstate[c]
                ds.upload("log_null_pi", log_null_pi)
                ds.upload("u_logs", u_logs)
                ds.upload("sqrt_half_var_tau", shvt)
                ds.upload("mu_mult", mu_mult)
                ds.e_step(self.dequantize_scale, sync=False)
            self._pull_state()
        self.zeta = self.compute_zeta()
        self._sums_valid = False

    # ---- posterior summaries (VIPRSMix.py:291-316) ---------------------------------------------------
    def compute_pip(self):
        return {c: g.sum(axis=1) for c, g in self.var_gamma.items()}

    def compute_eta(self):
        return {c: (g * self.var_mu[c]).sum(axis=1) for c, g in self.var_gamma.items()}

    def compute_zeta(self, sum_axis=1):
        return {c: (g * (self.var_mu[c] ** 2 + (1.0 / self.var_tau[c]))).sum(axis=sum_axis)
                for c, g in self.var_gamma.items()}

    # ---- partial sums -----------------------------------------------------------------------------------
    def _partial_sums(self):
        """[0] sum zeta  [1] sum((1+lambda) zeta + q eta)  [2] sum std_beta.eta  [3] sum eta^2
        [4] sum null_gamma log null_gamma  [5] sum null_gamma
        then K-vectors: sum_j gamma_jk | sum_j zeta_jk | sum gamma log gamma | sum gamma (clipped)
        | sum gamma log_var_tau (stale, VIPRSMix never refreshes it: SURVEY Appendix A) | sum gamma (mu^2 + 1/tau)"""
        K, lam = self.K, self.lambda_min
        if self._resident and self._host_stale:
            tot = np.zeros(6 + 6 * K, dtype=np.float64)
            self._dev_max_eta_diff = 0.0
            for ds in self._dstate.values():
                ds.sums_mixture_begin(1.0 + lam)
            for ds in self._dstate.values():
                v = ds.sums_mixture_end()
                tot += v[:-1]
                self._dev_max_eta_diff = max(self._dev_max_eta_diff, float(v[-1]))
            return tot
        s = np.zeros(6, dtype=np.float64)
        kv = np.zeros((6, K), dtype=np.float64)
        for c in self.chromosomes:
            g, z = self.var_gamma[c], self.zeta[c]
            s[0] += z.sum()
            s[1] += np.sum((1.0 + lam) * z + np.multiply(self.q[c], self.eta[c]), axis=0)
            s[2] += self.std_beta[c].dot(self.eta[c])
            s[3] += (self.eta[c].astype(np.float64) ** 2).sum()
            ng = np.clip(1.0 - g.sum(axis=1).astype(np.float64), _DOUBLE_RES, 1.0 - _DOUBLE_RES)
            s[4] += (ng * np.log(ng)).sum()
            s[5] += ng.sum()
            gc = np.clip(g.astype(np.float64), _DOUBLE_RES, 1.0 - _DOUBLE_RES)
            kv[0] += g.sum(axis=0)
            kv[1] += (g * (self.var_mu[c] ** 2 + (1.0 / self.var_tau[c]))).sum(axis=0)
            kv[2] += (gc * np.log(gc)).sum(axis=0)
            kv[3] += gc.sum(axis=0)
            kv[4] += (gc * self._log_var_tau[c]).sum(axis=0)
            kv[5] += (gc * (self.var_mu[c].astype(np.float64) ** 2 + 1.0 / self.var_tau[c])).sum(axis=0)
        return np.concatenate([s, kv.ravel()])

    def _kv(self, i):
        return self._sums[6 + i * self.K: 6 + (i + 1) * self.K]

    # ---- M-step (VIPRSMix.py:227-260 + VIPRS.py:446-471) ---------------------------------------------
    def m_step(self):
        s = self._reduce()
        T = self._T.type
        if "pis" not in self.fix_params:
            est = self._kv(0).copy()
            if "pi" in self.fix_params:
                est = self.fix_params["pi"] * est / est.sum()
            else:
                est /= self.n_snps
            self.pi = est.astype(self._T)
        if "tau_betas" not in self.fix_params:
            tau = np.sum(self.pi) * self.m / np.dot(self.d, self._kv(1))
            self.tau_beta = np.clip(self.d * tau, a_min=1.0, a_max=None)
        self._sigma_g = s[1]
        if "sigma_epsilon" not in self.fix_params:
            self.sigma_epsilon = 1.0 + T(-2.0 * s[2]) + self._sigma_g

    update_pi = m_step

    def elbo(self, sum_axis=None):
        s = self._sums if (self._sums is not None and self._sums_valid) else self._reduce()
        pi, null_pi, tau_beta = np.asarray(self.pi, dtype=np.float64), self.get_null_pi(), np.asarray(self.tau_beta)
        e = -np.log(2.0 * np.pi * self.sigma_epsilon)
        if "sigma_epsilon" not in self.fix_params:
            e -= 1.0
        else:
            e -= (1.0 / self.sigma_epsilon) * (1.0 - 2.0 * s[2] + self._sigma_g)
        e *= 0.5 * self.n
        e -= (self._kv(2) - np.log(pi) * self._kv(3)).sum()
        e -= s[4] - np.log(null_pi) * s[5]
        e += 0.5 * ((1.0 + np.log(tau_beta)) * self._kv(3) - self._kv(4)).sum()
        e -= 0.5 * (tau_beta * self._kv(5)).sum()                                       # VIPRS.py:570-573
        return float(e)

    objective = elbo

    # ---- the ELBO's parts for K components (VIPRS.py:583-687 with (m, K) arrays) ------------------------
    def entropy(self, sum_axis=None):
        s = self._current_sums()
        return float(0.5 * self.n_snps * (np.log(2.0 * np.pi) + 1.0) - self._kv(2).sum() - s[4] - 0.5 * self._kv(4).sum())

    def loglikelihood(self):
        s = self._current_sums()
        return float(-0.5 * self.n * (np.log(2.0 * np.pi * self.sigma_epsilon)
                                      + (1.0 / self.sigma_epsilon) * (1.0 - 2.0 * s[2] + self._sigma_g)))

    def log_prior(self, sum_axis=None):
        s = self._current_sums()
        pi, tau_beta = np.asarray(self.pi, dtype=np.float64), np.asarray(self.tau_beta, dtype=np.float64)
        lp = (0.5 * np.log(tau_beta) * self._kv(3)).sum() + (np.log(pi) * self._kv(3)).sum()
        lp += np.log(self.get_null_pi()) * s[5]
        lp -= 0.5 * (tau_beta * self._kv(5)).sum()
        return float(lp - 0.5 * self.n_snps * np.log(2.0 * np.pi))

    def get_average_effect_size_variance(self):
        return float(np.sum(np.asarray(self.pi, dtype=np.float64) / np.asarray(self.tau_beta, dtype=np.float64)))

    def to_theta_table(self):                                                          # VIPRSMix.py:318-335
        import pandas as pd
        extra = pd.DataFrame([{"Parameter": f"pi_{i + 1}", "Value": float(p)} for i, p in enumerate(np.asarray(self.pi))])
        return pd.concat([super().to_theta_table(), extra])

    def mse(self, sum_axis=None):
        s = self._sums
        return 1.0 - 2.0 * s[2] + (self._sigma_g - s[0] + s[3])
