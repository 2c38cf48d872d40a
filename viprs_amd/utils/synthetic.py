"""Synthetic LD blocks + summary statistics in the reference's array layout.

The reference gets these arrays from magenpy (`LDMatrix.load()` -> ld_data / ld_indptr /
leftmost_idx, VIPRS.py:167-172; `std_beta`, `n_per_snp`, BayesPRSModel.py:133-136); magenpy is
not part of the reference tree, so benchmarks and tests use synthetic blocks instead (nothing
across blocks).  Three kinds of block (`make_ld(kind=...)`):

* ``"ar1"``  -- analytic R[i, j] = rho^|i-j| (Toeplitz, positive definite, cheapest).  Entries more than
  ~2 panels off the diagonal are below half an ulp of `q`, so an E-step result does NOT depend on them:
  good for timing, blind as a parity input for everything far from the diagonal.
* ``"longrange"`` -- S (alpha A + (1 - alpha)(U U^T + D)) S: A the AR(1) matrix, U a b x 2 factor loading
  with entries of magnitude 0.3-0.7, D = diag(1 - |u_i|^2), S random signs.  Unit diagonal, positive
  definite, NOT Toeplitz, every entry of a block is O(0.05-0.3) wherever it sits: each LD entry the
  kernel streams changes the result.  O(b) parameters per block, so summary statistics stay cheap at
  genome scale.
* ``"sample"`` -- SURVEY 8d's "realistic" variant: the sample correlation of n = 4 b simulated AR(1)
  genotypes (noise of size 1/sqrt(n) everywhere in the block on top of the AR(1) decay).  O(b^3): for
  blocks up to a few thousand SNPs.
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
    kind: str = "ar1"
    params: list = None            # "longrange": per block (alpha, U (b, r) with the signs folded in, signs (b,))
    ld_dtype: np.dtype = None      # element type when `ld_data` is None (a skeleton: the entries exist on the device only)

    @property
    def m(self):
        return int(self.ld_left_bound.shape[0])

    @property
    def itemsize(self):
        return int(np.dtype(self.ld_dtype).itemsize if self.ld_data is None else self.ld_data.dtype.itemsize)

    @property
    def nnz(self):
        return int(self.ld_indptr[-1])


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
        dst[...] = np.lib.stride_tricks.as_strided(vals[b - 1:], shape=(b, b), strides=(-vals.itemsize, vals.itemsize))


def _store_dense_block(data, off, R, low_memory, ld_dtype, quant_max):
    """Write the dense b x b correlation matrix `R` (float) of one block into `data` at `off` in the symmetric
    (whole rows) or the upper-triangular layout (row j = R[j, j+1:]), quantised magenpy-style for integer LD."""
    b = R.shape[0]
    if quant_max is not None:
        R = np.round(R * min(float(quant_max), 2.0 ** 30 if ld_dtype == np.int32 else 2.0 ** 52))
    if low_memory:
        o = off
        for r in range(b - 1):
            data[o:o + b - 1 - r] = R[r, r + 1:]
            o += b - 1 - r
    else:
        dst = data[off:off + b * b].reshape(b, b)
        if dst is not R:
            dst[...] = R


def _longrange_params(rng, b, n_factors=2):
    """(alpha, U with the signs folded in, signs): R = S (alpha A + (1 - alpha)(U0 U0^T + D)) S."""
    alpha = float(rng.uniform(0.4, 0.7))
    signs = np.where(rng.random(b) < 0.5, -1.0, 1.0)
    U0 = rng.uniform(0.3, 0.7, (b, n_factors)) * np.where(rng.random((b, n_factors)) < 0.5, -1.0, 1.0)
    # |u_i|^2 <= 0.98 < 1 for two factors: D = diag(1 - |u_i|^2) > 0
    return alpha, U0 * signs[:, None], signs


def _longrange_block(b, rho, alpha, U, signs, out=None):
    """Dense float32 long-range block: alpha s_i s_j rho^|i-j| + (1 - alpha) u_i . u_j off the diagonal, 1 on it.
    Built from outer products (exactly symmetric), a handful of passes over the b x b array."""
    f32 = np.float32
    pw = np.power(np.float64(rho), np.arange(b)).astype(f32)
    vals = np.concatenate([pw[::-1], pw[1:]])
    T = np.lib.stride_tricks.as_strided(vals[b - 1:], shape=(b, b), strides=(-vals.itemsize, vals.itemsize))
    M = np.empty((b, b), dtype=f32) if out is None else out
    Uf = (U * np.sqrt(1.0 - alpha)).astype(f32)
    np.multiply.outer(Uf[:, 0], Uf[:, 0], out=M)
    for k in range(1, Uf.shape[1]):
        M += np.multiply.outer(Uf[:, k], Uf[:, k])
    sf = signs.astype(f32)
    tmp = np.multiply(T, (sf * f32(alpha))[:, None])
    tmp *= sf[None, :]
    M += tmp
    np.fill_diagonal(M, 1.0)
    return M


def _sample_block(rng, b, rho, n_mult=4):
    """Sample correlation (float64) of n = n_mult * b AR(1) genotype vectors (the "realistic" variant)."""
    n = n_mult * b
    Z = rng.standard_normal((n, b))
    X = np.empty((n, b))
    X[:, 0] = Z[:, 0]
    sd = np.sqrt(1.0 - rho * rho)
    for t in range(1, b):
        X[:, t] = rho * X[:, t - 1] + sd * Z[:, t]
    X -= X.mean(axis=0)
    X /= np.sqrt((X * X).sum(axis=0))
    R = X.T @ X
    R = 0.5 * (R + R.T)
    np.fill_diagonal(R, 1.0)
    return R


LD_KINDS = ("ar1", "longrange", "sample")


def longrange_params(sizes, seed=SEED):
    """The per-block parameters of "longrange" LD for a whole workload (one sequential draw: a block keeps its
    parameters whichever subset of the workload a rank builds)."""
    prng = np.random.default_rng(seed + 3)
    return [_longrange_params(prng, int(b)) for b in sizes]


def longrange_device_params(sizes, rho, params):
    """The per-SNP float32 vectors `viprs_plan_create_synthetic` / `viprs_synthetic_ld_host` take (include/viprs_hip.h):
    exactly the float32 intermediates of `_longrange_block`, concatenated over the blocks --
    pw[k] = float32(rho^k), uf0 / uf1 = float32(U sqrt(1 - alpha)), sa = float32(signs) * float32(alpha), sf = float32(signs)."""
    f32 = np.float32
    m = int(np.sum(sizes))
    pw, uf0, uf1, sa, sf = (np.empty(m, dtype=f32) for _ in range(5))
    o = 0
    for b, r, (alpha, U, signs) in zip(sizes, rho, params):
        b = int(b)
        assert U.shape == (b, 2), "the device generator implements the two-factor blocks of `_longrange_params`"
        pw[o:o + b] = np.power(np.float64(r), np.arange(b)).astype(f32)
        Uf = (U * np.sqrt(1.0 - alpha)).astype(f32)
        uf0[o:o + b], uf1[o:o + b] = Uf[:, 0], Uf[:, 1]
        s = signs.astype(f32)
        sa[o:o + b] = s * f32(alpha)
        sf[o:o + b] = s
        o += b
    return pw, uf0, uf1, sa, sf


def make_ld(sizes, low_memory=False, ld_dtype=np.float32, indptr_dtype=np.int64, seed=SEED, rho_range=(0.3, 0.8),
            rho=None, kind="ar1", params=None, data=True):
    """Block-diagonal LD in symmetric (`low_memory=False`: every row of a block stores the whole
    block, diagonal included) or upper-triangular form (`low_memory=True`: row j stores columns
    j+1 .. block_end-1, left bound j+1) -- the two layouts e_step.hpp:389-392,423-440 consumes.
    `kind`: "ar1" | "longrange" | "sample" (module docstring).  `data=False`: the skeleton only (index arrays, block
    parameters, `ld_data = None`) -- for plans whose entries are generated on the device (`LDPlan.synthetic`)."""
    if kind not in LD_KINDS:
        raise ValueError(f"unknown LD kind {kind!r}")
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
    data = np.empty(nnz, dtype=ld_dtype) if data else None
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

    if kind == "longrange":
        if params is None:
            params = longrange_params(sizes, seed)
        assert len(params) == len(sizes) and all(p[2].shape[0] == int(b) for p, b in zip(params, sizes))
    elif kind == "sample":
        # one generator per block so that the thread pool below cannot change the draws
        params = [np.random.default_rng([seed + 4, bi]) for bi in range(len(sizes))]

    def _job(j):
        o, b, bi = j
        if kind == "ar1":
            _fill_block(data, o, b, _ar1_row(b, rho[bi], ld_dtype, quant_max), low_memory)
        elif kind == "longrange":
            direct = (not low_memory) and ld_dtype == np.float32
            out = data[o:o + b * b].reshape(b, b) if direct else None
            R = _longrange_block(b, rho[bi], *params[bi], out=out)
            if not direct:
                _store_dense_block(data, o, R, low_memory, ld_dtype, quant_max)
        else:
            _store_dense_block(data, o, _sample_block(params[bi], b, rho[bi]), low_memory, ld_dtype, quant_max)

    if data is None:
        pass
    elif nnz > (1 << 24):
        from concurrent.futures import ThreadPoolExecutor
        import os
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
            list(ex.map(_job, sorted(jobs, key=lambda j: -j[1])))
    else:
        for j in jobs:
            _job(j)
    indptr = np.concatenate([[0], np.cumsum(rowlen)]).astype(indptr_dtype)
    return SyntheticLD(lb, indptr, data, starts, rho, bool(low_memory), dq_scale, kind=kind,
                       params=params if kind == "longrange" else None, ld_dtype=ld_dtype)


def dense_block(ld, bi):
    """The dense float64 correlation matrix of block `bi`, rebuilt from the stored arrays (dequantised)."""
    s, e = int(ld.block_start[bi]), int(ld.block_start[bi + 1])
    b = e - s
    o = int(ld.ld_indptr[s])
    if ld.low_memory:
        R = np.eye(b)
        for r in range(b - 1):
            R[r, r + 1:] = ld.ld_data[o:o + b - 1 - r] * ld.dq_scale
            o += b - 1 - r
        return R + np.triu(R, 1).T
    return ld.ld_data[o:o + b * b].reshape(b, b).astype(np.float64) * ld.dq_scale


@dataclass
class SyntheticSumstats:
    std_beta: np.ndarray           # (m,) marginal standardized effects
    n_per_snp: np.ndarray          # (m,) float64
    beta_true: np.ndarray
    n: float


def _ar1_profile_and_noise(b, rho, idx, coef, zz):
    """(A beta, e) for the AR(1) matrix A = rho^|i-j|: `A beta` for the sparse vector with entries `coef` at `idx`
    as a sum of shifted geometric profiles, `e ~ N(0, A)` by the AR(1) recursion driven by `zz`."""
    k = np.arange(b)
    rb = np.zeros(b)
    for c, v in zip(idx, coef):
        rb += v * np.power(rho, np.abs(k - c))
    eps = np.empty(b)
    eps[0] = zz[0]
    sd = np.sqrt(1.0 - rho * rho)
    for t in range(1, b):
        eps[t] = rho * eps[t - 1] + sd * zz[t]
    return rb, eps


def make_sumstats(ld, n=1e5, h2=0.2, pi=0.01, seed=SEED, float_precision=np.float32, noise_seed=None):
    """std_beta = R beta + e, e ~ N(0, R / N), beta spike-and-slab.  `R` is the block's model
    matrix (before quantisation); AR(1) and long-range blocks need O(b) work per block, sample-correlation
    blocks a Cholesky factor of the stored matrix.  `noise_seed`: same effects `beta`, an independent draw of
    the noise -- the marginal effects of a second cohort (validation set) on the same LD."""
    rng = np.random.default_rng(seed + 2)
    m = ld.m
    causal = rng.random(m) < pi
    n_causal = max(int(causal.sum()), 1)
    beta = np.zeros(m)
    beta[causal] = rng.normal(0.0, np.sqrt(h2 / n_causal), int(causal.sum()))
    z = rng.standard_normal(m)
    kind = getattr(ld, "kind", "ar1")
    zrng = np.random.default_rng(seed + 5) if kind == "longrange" else None
    if noise_seed is not None:
        z = np.random.default_rng([noise_seed, 0]).standard_normal(m)
        zrng = np.random.default_rng([noise_seed, 1]) if kind == "longrange" else None
    std_beta = np.empty(m)
    for bi in range(len(ld.rho)):
        s, e = int(ld.block_start[bi]), int(ld.block_start[bi + 1])
        b = e - s
        rho = ld.rho[bi]
        idx = np.nonzero(causal[s:e])[0]
        bb = beta[s:e]
        if kind == "ar1":
            rb, eps = _ar1_profile_and_noise(b, rho, idx, bb[idx], z[s:e])
        elif kind == "longrange":
            alpha, U, signs = ld.params[bi]
            d = 1.0 - (U * U).sum(axis=1)
            ab, ea = _ar1_profile_and_noise(b, rho, idx, signs[idx] * bb[idx], z[s:e])
            rb = alpha * signs * ab + (1.0 - alpha) * (U @ (U[idx].T @ bb[idx]) + d * bb)
            zf, zd = zrng.standard_normal(U.shape[1]), zrng.standard_normal(b)
            eps = np.sqrt(alpha) * signs * ea + np.sqrt(1.0 - alpha) * (U @ zf + np.sqrt(d) * zd)
        else:
            R = dense_block(ld, bi)
            rb = R[:, idx] @ bb[idx]
            eps = np.linalg.cholesky(R + 1e-6 * np.eye(b)) @ z[s:e]
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
                 indptr_dtype=np.int64, float_precision=np.float32, kind="ar1"):
    sizes = block_sizes(config, seed) if sizes is None else np.asarray(sizes)
    ld = make_ld(sizes, low_memory=low_memory, ld_dtype=ld_dtype, seed=seed, indptr_dtype=indptr_dtype, kind=kind)
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
