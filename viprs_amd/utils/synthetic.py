"""Synthetic LD blocks + summary statistics in the reference's array layout.

The reference gets these arrays from magenpy (`LDMatrix.load()` -> ld_data / ld_indptr /
leftmost_idx, VIPRS.py:167-172; `std_beta`, `n_per_snp`, BayesPRSModel.py:133-136); magenpy is
not part of the reference tree, so benchmarks and tests use analytic AR(1) blocks instead:
R[i, j] = rho^|i-j| inside a block (positive definite), nothing across blocks.
"""
from dataclasses import dataclass, field

import numpy as np

SEED = 7209   # the reference's default seed (benchmarks/benchmark_e_step.py:249, bin/viprs_fit:996)


def block_sizes(config, seed=SEED):
    """Block-size lists of BASELINE.json's configs."""
    rng = np.random.default_rng(seed)
    if config in ("cfg1", "single_block"):
        return np.array([500], dtype=np.int64)
    if config in ("cfg2", "chr22"):
        n, mu, sigma, lo, hi, total = 40, np.log(420.0), 0.5, 80, 2500, 19_000
    elif config in ("cfg3", "genome"):
        n, mu, sigma, lo, hi, total = 1700, np.log(560.0), 0.55, 50, 6000, 1_100_000
    else:
        raise ValueError(f"unknown config {config!r}")
    s = np.clip(rng.lognormal(mu, sigma, n), lo, hi)
    s = np.clip(np.round(s * (total / s.sum())), lo, hi).astype(np.int64)
    return s


@dataclass
class SyntheticLD:
    """LD arrays exactly as `cpp_e_step` takes them (e_step_cpp.pyx:91-93)."""
    ld_left_bound: np.ndarray      # (m,)  int32
    ld_indptr: np.ndarray          # (m+1,) int64 (or int32)
    ld_data: np.ndarray            # (nnz,) float32 / int8 / ...
    block_start: np.ndarray        # (n_blocks+1,) int64
    rho: np.ndarray                # (n_blocks,) AR(1) coefficient per block
    low_memory: bool
    dq_scale: float = 1.0
    meta: dict = field(default_factory=dict)

    @property
    def m(self):
        return int(self.ld_left_bound.shape[0])


def _ar1_row(b, rho, dtype, quant_max):
    pw = np.power(np.float64(rho), np.arange(b))
    if quant_max is None:
        return pw.astype(dtype)
    # magenpy-style symmetric int quantisation (the scale is capped at 2^52 so that the int32/int64
    # storage types of the Cython boundary stay exactly representable in the float64 intermediate)
    return np.round(pw * min(float(quant_max), 2.0 ** 30 if dtype == np.int32 else 2.0 ** 52)).astype(dtype)


def _fill_block(data, off, b, row, low_memory):
    """Write one AR(1) block (R[i, j] = row[|i - j|]) into `data` starting at `off`, row by row with
    contiguous copies (numpy releases the GIL for them, so blocks are filled by a thread pool)."""
    if low_memory:
        o = off
        for r in range(b - 1):                              # row r holds R[r, r+1:] = row[1 : b-r]
            data[o:o + b - 1 - r] = row[1:b - r]
            o += b - 1 - r
    else:
        vals = np.concatenate([row[::-1], row[1:]])         # R[r, :] = vals[b-1-r : 2b-1-r]
        dst = data[off:off + b * b].reshape(b, b)
        for r in range(b):
            dst[r] = vals[b - 1 - r:2 * b - 1 - r]


def make_ld(sizes, low_memory=False, ld_dtype=np.float32, indptr_dtype=np.int64, seed=SEED, rho_range=(0.3, 0.8),
            rho=None):
    """Block-diagonal AR(1) LD in symmetric (`low_memory=False`: every row of a block stores the whole
    block, diagonal included) or upper-triangular form (`low_memory=True`: row j stores columns
    j+1 .. block_end-1, left bound j+1) -- the two layouts e_step.hpp:389-392,423-440 consumes."""
    sizes = np.asarray(sizes, dtype=np.int64)
    rng = np.random.default_rng(seed + 1)
    rho = rng.uniform(rho_range[0], rho_range[1], len(sizes)) if rho is None else np.asarray(rho, dtype=np.float64)
    ld_dtype = np.dtype(ld_dtype)
    quant_max = None
    dq_scale = 1.0
    if np.issubdtype(ld_dtype, np.integer):
        quant_max = np.iinfo(ld_dtype).max
        dq_scale = 1.0 / quant_max                         # VIPRS.py:203-207
    m = int(sizes.sum())
    starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    if low_memory:
        nnz = int((sizes * (sizes - 1) // 2).sum())
    else:
        nnz = int((sizes * sizes).sum())
    data = np.empty(nnz, dtype=ld_dtype)
    lb = np.empty(m, dtype=np.int32)
    rowlen = np.empty(m, dtype=np.int64)
    off = 0
    jobs = []
    for bi, b in enumerate(sizes):
        b = int(b)
        s = int(starts[bi])
        jobs.append((off, b, bi))
        if low_memory:
            lb[s:s + b] = np.arange(s + 1, s + b + 1, dtype=np.int32)
            rowlen[s:s + b] = np.arange(b - 1, -1, -1)
            off += b * (b - 1) // 2
        else:
            lb[s:s + b] = s
            rowlen[s:s + b] = b
            off += b * b

    def _job(j):
        o, b, bi = j
        _fill_block(data, o, b, _ar1_row(b, rho[bi], ld_dtype, quant_max), low_memory)

    if nnz > (1 << 24):
        from concurrent.futures import ThreadPoolExecutor
        import os
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
            list(ex.map(_job, sorted(jobs, key=lambda j: -j[1])))
    else:
        for j in jobs:
            _job(j)
    indptr = np.concatenate([[0], np.cumsum(rowlen)]).astype(indptr_dtype)
    return SyntheticLD(lb, indptr, data, starts, rho, bool(low_memory), dq_scale)


@dataclass
class SyntheticSumstats:
    std_beta: np.ndarray           # (m,) marginal standardized effects
    n_per_snp: np.ndarray          # (m,) float64
    beta_true: np.ndarray
    n: float


def make_sumstats(ld, n=1e5, h2=0.2, pi=0.01, seed=SEED, float_precision=np.float32):
    """std_beta = R beta + e, e ~ N(0, R / N), beta spike-and-slab."""
    rng = np.random.default_rng(seed + 2)
    m = ld.m
    causal = rng.random(m) < pi
    n_causal = max(int(causal.sum()), 1)
    beta = np.zeros(m)
    beta[causal] = rng.normal(0.0, np.sqrt(h2 / n_causal), int(causal.sum()))
    z = rng.standard_normal(m)
    std_beta = np.empty(m)
    for bi in range(len(ld.rho)):
        s, e = int(ld.block_start[bi]), int(ld.block_start[bi + 1])
        b = e - s
        rho = ld.rho[bi]
        k = np.arange(b)
        # R beta for sparse beta: sum of shifted geometric profiles
        rb = np.zeros(b)
        for c in np.nonzero(causal[s:e])[0]:
            rb += beta[s + c] * np.power(rho, np.abs(k - c))
        # AR(1) noise with covariance R
        eps = np.empty(b)
        zz = z[s:e]
        eps[0] = zz[0]
        sd = np.sqrt(1.0 - rho * rho)
        for t in range(1, b):
            eps[t] = rho * eps[t - 1] + sd * zz[t]
        std_beta[s:e] = rb + eps / np.sqrt(n)
    return SyntheticSumstats(std_beta.astype(float_precision), np.full(m, float(n)), beta, float(n))


@dataclass
class EStepInputs:
    """Everything one `cpp_e_step` call takes besides the LD arrays."""
    std_beta: np.ndarray
    var_gamma: np.ndarray
    var_mu: np.ndarray
    eta: np.ndarray
    q: np.ndarray
    eta_diff: np.ndarray
    u_logs: np.ndarray
    sqrt_half_var_tau: np.ndarray
    mu_mult: np.ndarray
    pi: float
    sigma_epsilon: float
    tau_beta: float

    def state_copy(self):
        return {k: getattr(self, k).copy() for k in ("var_gamma", "var_mu", "eta", "q", "eta_diff")}


def host_prep(n_per_snp, pi, sigma_epsilon, tau_beta, lambda_min=0.0, float_precision=np.float32):
    """The per-iteration host prep of VIPRS.e_step (VIPRS.py:400-418), float64 -> float_precision."""
    var_tau = n_per_snp * (1.0 + lambda_min) / sigma_epsilon + tau_beta
    log_var_tau = np.log(var_tau)
    mu_mult = (n_per_snp / (var_tau * sigma_epsilon)).astype(float_precision)
    u_logs = (np.log(pi) - np.log(1.0 - pi) + 0.5 * (np.log(tau_beta) - log_var_tau)).astype(float_precision)
    shvt = np.sqrt(0.5 * var_tau).astype(float_precision)
    return var_tau, mu_mult, u_logs, shvt


def make_inputs(ss, pi=0.01, sigma_epsilon=0.8, h2=0.2, float_precision=np.float32):
    """Hyper-parameters pi=0.01, sigma_eps=0.8, tau_beta = M pi / h2 and the standard initial state
    var_gamma = pi, var_mu = eta = q = eta_diff = 0 (VIPRS.py:344-358)."""
    m = ss.std_beta.shape[0]
    tau_beta = m * pi / h2
    _, mu_mult, u_logs, shvt = host_prep(ss.n_per_snp, pi, sigma_epsilon, tau_beta, 0.0, float_precision)
    T = np.dtype(float_precision)
    return EStepInputs(
        std_beta=ss.std_beta.astype(T), var_gamma=np.full(m, pi, dtype=T), var_mu=np.zeros(m, dtype=T),
        eta=np.zeros(m, dtype=T), q=np.zeros(m, dtype=T), eta_diff=np.zeros(m, dtype=T),
        u_logs=u_logs, sqrt_half_var_tau=shvt, mu_mult=mu_mult, pi=pi, sigma_epsilon=sigma_epsilon,
        tau_beta=tau_beta)


def make_problem(config="cfg1", low_memory=False, ld_dtype=np.float32, seed=SEED, sizes=None,
                 indptr_dtype=np.int64, float_precision=np.float32):
    sizes = block_sizes(config, seed) if sizes is None else np.asarray(sizes)
    ld = make_ld(sizes, low_memory=low_memory, ld_dtype=ld_dtype, seed=seed, indptr_dtype=indptr_dtype)
    ss = make_sumstats(ld, seed=seed, float_precision=float_precision)
    inp = make_inputs(ss, float_precision=float_precision)
    return ld, ss, inp


def make_mixture_inputs(ss, K=4, float_precision=np.float32, pi=0.01, sigma_eps=0.8, h2=0.2):
    """Per-SNP inputs of `e_step_mixture` for a K-component sparse mixture prior, C-order (m, K):
    prior multipliers d = 2^linspace(-min(K-1, 7), 0, K) (VIPRSMix.py:52; K = 4: 2^-3 .. 1), mixing proportions
    pi_k = pi * [.4, .3, .2, .1] for K = 4 (in general decreasing weights 2 (K - k) / (K (K + 1))), component
    precisions tau_k = d_k M sum_k(pi_k / d_k) / h2 (VIPRSMix.py:126-128), inputs as VIPRSMix.py:181-223."""
    T = np.dtype(float_precision)
    m = ss.n_per_snp.shape[0]
    d = 2.0 ** np.linspace(-min(K - 1, 7), 0, K)
    w = 2.0 * (K - np.arange(K)) / (K * (K + 1.0))           # K = 4: [.4, .3, .2, .1]
    pis = pi * w
    tau = d * (m * np.dot(1.0 / d, pis) / h2)
    n = np.asarray(ss.n_per_snp, dtype=np.float64)[:, None]
    var_tau = n / sigma_eps + tau[None, :]
    return dict(
        log_null_pi=np.full(m, np.log(1.0 - pis.sum()), dtype=T),
        u_logs=np.ascontiguousarray((np.log(pis) - np.log(1 - pis) + 0.5 * (np.log(tau) - np.log(var_tau))).astype(T)),
        sqrt_half_var_tau=np.ascontiguousarray(np.sqrt(0.5 * var_tau).astype(T)),
        mu_mult=np.ascontiguousarray((n / (var_tau * sigma_eps)).astype(T)),
        pi=float(pis[0]))


def grid_points(G=32, n_snps=1_100_000):
    """The (sigma_epsilon, pi) grid of BASELINE configs[4]: the reference's own `HyperparameterGrid`
    with h2 = 0.1 +- 0.1 -- G / 8 sigma_epsilon values (the slow axis of `itertools.product`,
    HyperparameterGrid.py:110-166, :238-245) x 8 pi values (log-spaced, the fast axis, :184-208); G = 32: 4 x 8.
    Returns (sigma_epsilon (G,), pi (G,))."""
    from ..model.gridsearch.HyperparameterGrid import HyperparameterGrid
    n_pi = 8 if G % 8 == 0 and G >= 8 else G
    grid = HyperparameterGrid(sigma_epsilon_steps=G // n_pi, pi_steps=n_pi, h2_est=0.1, h2_se=0.1, n_snps=n_snps)
    pts = grid.combine_grids()
    assert len(pts) == G
    return (np.array([p["sigma_epsilon"] for p in pts], dtype=np.float64),
            np.array([p["pi"] for p in pts], dtype=np.float64))


def make_grid_inputs(ss, G=32, float_precision=np.float32):
    """Per-SNP inputs of `e_step_grid` for the G grid points of `grid_points`, column-major (m, G) (VIPRSGrid.py:
    one column per model); per model tau_beta = pi M / (1 - sigma_epsilon) (VIPRS.py:310), inputs as VIPRS.py:400-418
    with half_var_tau in place of its square root (e_step.hpp:616)."""
    T = np.dtype(float_precision)
    m = ss.n_per_snp.shape[0]
    sig, pis = grid_points(G, m)
    tau = pis * m / (1 - sig)
    n = np.asarray(ss.n_per_snp, dtype=np.float64)[:, None]
    var_tau = n / sig[None, :] + tau[None, :]
    mk = lambda a: np.asfortranarray(a.astype(T))
    return dict(u_logs=mk(np.log(pis) - np.log(1 - pis) + 0.5 * (np.log(tau) - np.log(var_tau))),
                half_var_tau=mk(0.5 * var_tau), mu_mult=mk(n / (var_tau * sig[None, :])), pi=float(pis[0]))
