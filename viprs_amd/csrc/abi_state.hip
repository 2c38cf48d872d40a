// C ABI, part 2 (include/viprs_hip.h): the device-resident variational state and the device-side pieces of
// the EM iteration around the E-step (host prep VIPRS.py:400-418, compute_zeta :888-897, the M-step / ELBO
// partial sums :426-581; VIPRSMix.py:169-260).
#include "internal.h"

using namespace viprs;

namespace {

// device-side re-initialisation to the standard start (VIPRS.py:344-358) in ONE launch
template <typename T>
__global__ void reset_state_kernel(T* var_gamma, T* var_mu, int64_t n_wide, T* eta, T* q, T* eta_diff, int64_t n_vec,
                                   T pi) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_wide) { var_gamma[i] = pi; var_mu[i] = (T)0; }
    if (i < n_vec) { eta[i] = (T)0; q[i] = (T)0; eta_diff[i] = (T)0; }
}

// VIPRS.py:400-418 on the device (float64, cast to T at the end)
template <typename T>
__global__ void prep_kernel(const double* __restrict__ n, int64_t m, double logit_pi, double log_tau_beta,
                            double sigma_eps, double tau_beta, double one_plus_lambda, T* __restrict__ mu_mult,
                            T* __restrict__ u_logs, T* __restrict__ shvt, double* __restrict__ var_tau_out,
                            int half_not_sqrt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double vt = n[i] * one_plus_lambda / sigma_eps + tau_beta;
    if (var_tau_out) var_tau_out[i] = vt;              // (grid columns: not stored, the sums form it again)
    mu_mult[i] = (T)(n[i] / (vt * sigma_eps));
    u_logs[i] = (T)(logit_pi + 0.5 * (log_tau_beta - log(vt)));
    shvt[i] = half_not_sqrt ? (T)(0.5 * vt) : (T)sqrt(0.5 * vt);     // e_step_grid takes var_tau / 2 (e_step.hpp:616)
}

// the same for several columns of a grid state in one launch: blockIdx.y picks a row of `params`
// (column, logit_pi, log_tau_beta, sigma_eps, tau_beta, one_plus_lambda)
// (var_tau itself is not stored for a grid state: m x G doubles to write here and to read back in the sums -- the sums
//  kernel forms it again from n_j and the column's scalars, the same expression, the same bits)
template <typename T>
__global__ void prep_columns_kernel(const double* __restrict__ n, int64_t m, const double* __restrict__ params,
                                    T* __restrict__ mu_mult, T* __restrict__ u_logs, T* __restrict__ shvt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double* __restrict__ p = params + 6 * (int64_t)blockIdx.y;
    const int64_t off = (int64_t)p[0] * m;
    const double logit_pi = p[1], log_tau_beta = p[2], sigma_eps = p[3], tau_beta = p[4], one_plus_lambda = p[5];
    const double vt = n[i] * one_plus_lambda / sigma_eps + tau_beta;
    mu_mult[off + i] = (T)(n[i] / (vt * sigma_eps));
    u_logs[off + i] = (T)(logit_pi + 0.5 * (log_tau_beta - log(vt)));
    shvt[off + i] = (T)(0.5 * vt);                                    // e_step_grid takes var_tau / 2 (e_step.hpp:616)
}

// the same for SNP GROUPS of a spike-and-slab state (one model per chromosome in one plan): blockIdx.y picks a row of
// `params` (group, logit_pi, log_tau_beta, sigma_eps, tau_beta, one_plus_lambda); only the group's SNPs are written
template <typename T>
__global__ void prep_groups_kernel(const double* __restrict__ n, const int64_t* __restrict__ gstart,
                                   const double* __restrict__ params, T* __restrict__ mu_mult, T* __restrict__ u_logs,
                                   T* __restrict__ shvt, double* __restrict__ var_tau_out) {
    const double* __restrict__ p = params + 6 * (int64_t)blockIdx.y;
    const int g = (int)p[0];
    const double logit_pi = p[1], log_tau_beta = p[2], sigma_eps = p[3], tau_beta = p[4], one_plus_lambda = p[5];
    const int64_t end = gstart[g + 1];
    for (int64_t i = gstart[g] + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += (int64_t)gridDim.x * blockDim.x) {
        const double vt = n[i] * one_plus_lambda / sigma_eps + tau_beta;
        var_tau_out[i] = vt;
        mu_mult[i] = (T)(n[i] / (vt * sigma_eps));
        u_logs[i] = (T)(logit_pi + 0.5 * (log_tau_beta - log(vt)));
        shvt[i] = (T)sqrt(0.5 * vt);
    }
}

constexpr int kSumsBlock = 256;
constexpr int kNSums = VIPRS_N_SUMS;
constexpr int kSumsMaxBlocks = 1024;
// workgroups of the reduction over `count` SNPs (the same for a whole plan and for one group of it: a group's sums are
// bit-identical to those of a plan that holds only that group)
__host__ __device__ inline int sums_blocks(int64_t count) {
    const int64_t nb = (count + kSumsBlock - 1) / kSumsBlock;
    return (int)(nb < kSumsMaxBlocks ? nb : kSumsMaxBlocks);
}

// stage 1: per-workgroup partial sums (fixed assignment of elements to threads, tree reduction in
// LDS: deterministic); stage 2 adds the partials in index order.  `sums_body`: workgroup `bx` of `nb` over SNPs [i0, i1).
template <typename T>
__device__ __forceinline__ void sums_body(int64_t i0, int64_t i1, int nb, int bx, const T* __restrict__ gam,
                                          const T* __restrict__ mu, const T* __restrict__ eta, const T* __restrict__ q,
                                          const T* __restrict__ ed, const T* __restrict__ beta,
                                          const double* __restrict__ var_tau, double one_plus_lambda,
                                          const double* __restrict__ weight, double* __restrict__ out,
                                          const double* __restrict__ n_snp = nullptr, double p_lam1 = 0.0, double p_sig = 1.0,
                                          double p_tau = 0.0) {
    // n_snp != nullptr: var_tau is not stored (grid states) -- formed as the prep kernels form it, n (1 + lambda) / sigma_eps + tau_beta
    __shared__ double red[kNSums][kSumsBlock];
    double acc[kNSums];
#pragma unroll
    for (int k = 0; k < kNSums; ++k) acc[k] = 0.0;
    const double lo = 1e-15, hi = 1.0 - 1e-15;       // np.finfo(float64).resolution (VIPRS.py:509)
    for (int64_t i = i0 + (int64_t)bx * kSumsBlock + threadIdx.x; i < i1; i += (int64_t)nb * kSumsBlock) {
        const double g = (double)gam[i], mud = (double)mu[i];
        const double vt = n_snp ? n_snp[i] * p_lam1 / p_sig + p_tau : var_tau[i];
        const double zeta = g * (mud * mud + 1.0 / vt);                       // VIPRS.py:896
        acc[0] += weight ? g * weight[i] : g;                                  // sum_c mean(gamma_c) over merged chromosomes
        acc[1] += zeta;
        acc[2] += one_plus_lambda * zeta + (double)(q[i] * eta[i]);        // :455 (q*eta in T, as np.multiply)
        acc[3] += (double)beta[i] * (double)eta[i];
        acc[4] += (double)eta[i] * (double)eta[i];
        const double gc = fmin(fmax(g, lo), hi), ng = fmin(fmax(1.0 - g, lo), hi);
        acc[5] += gc * log(gc);
        acc[6] += ng * log(ng);
        acc[7] += gc;
        acc[8] += ng;
        acc[9] += gc * log(vt);
        acc[10] = fmax(acc[10], fabs((double)ed[i]));
    }
#pragma unroll
    for (int k = 0; k < kNSums; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int s = kSumsBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
#pragma unroll
            for (int k = 0; k < kNSums - 1; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
            red[kNSums - 1][threadIdx.x] = fmax(red[kNSums - 1][threadIdx.x], red[kNSums - 1][threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x < kNSums) out[threadIdx.x] = red[threadIdx.x][0];
}

template <typename T>
__global__ __launch_bounds__(kSumsBlock) void sums_kernel(int64_t m, const T* __restrict__ gam, const T* __restrict__ mu,
                                                          const T* __restrict__ eta, const T* __restrict__ q,
                                                          const T* __restrict__ ed, const T* __restrict__ beta,
                                                          const double* __restrict__ var_tau, double one_plus_lambda,
                                                          const double* __restrict__ weight, double* __restrict__ partials,
                                                          const double* __restrict__ cols = nullptr) {
    if (cols) {
        // columns of a grid state, one per blockIdx.y: a row of `cols` = (column, one_plus_lambda, and the scalars the
        // column's last prep built var_tau from: one_plus_lambda, sigma_eps, tau_beta); `var_tau` is the per-SNP n here
        const double* __restrict__ c = cols + 5 * (int64_t)blockIdx.y;
        const int64_t off = (int64_t)c[0] * m;
        gam += off; mu += off; eta += off; q += off; ed += off;
        partials += (int64_t)blockIdx.y * gridDim.x * kNSums;
        sums_body<T>(0, m, (int)gridDim.x, (int)blockIdx.x, gam, mu, eta, q, ed, beta, nullptr, c[1], weight,
                     partials + (int64_t)blockIdx.x * kNSums, var_tau, c[2], c[3], c[4]);
        return;
    }
    sums_body<T>(0, m, (int)gridDim.x, (int)blockIdx.x, gam, mu, eta, q, ed, beta, var_tau, one_plus_lambda, weight,
                 partials + (int64_t)blockIdx.x * kNSums);
}

// the sums of SNP groups of a spike-and-slab state: blockIdx.y picks a row (group, one_plus_lambda) of `rows`; group g is
// reduced by sums_blocks(its SNPs) workgroups exactly as a plan of its own would be (the rest of the row's grid leaves)
template <typename T>
__global__ __launch_bounds__(kSumsBlock) void sums_groups_kernel(const int64_t* __restrict__ gstart, const double* __restrict__ rows,
                                                                 const T* __restrict__ gam, const T* __restrict__ mu,
                                                                 const T* __restrict__ eta, const T* __restrict__ q,
                                                                 const T* __restrict__ ed, const T* __restrict__ beta,
                                                                 const double* __restrict__ var_tau, double* __restrict__ partials) {
    const int g = (int)rows[2 * blockIdx.y];
    const int64_t i0 = gstart[g], i1 = gstart[g + 1];
    const int nb = sums_blocks(i1 - i0);
    if ((int)blockIdx.x >= nb) return;
    sums_body<T>(i0, i1, nb, (int)blockIdx.x, gam, mu, eta, q, ed, beta, var_tau, rows[2 * blockIdx.y + 1], nullptr,
                 partials + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * kNSums);
}

// one wave per sum: lane l adds the partials of blocks l, l + 64, ... in order, then a fixed xor-shuffle
// tree combines the 64 lanes -- a deterministic order whatever the timing
__global__ void sums_final_kernel(const double* __restrict__ partials, int n_blocks, double* __restrict__ out) {
    partials += (int64_t)blockIdx.x * n_blocks * kNSums;            // one workgroup per column (grid states)
    out += (int64_t)blockIdx.x * kNSums;
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (k >= kNSums) return;
    const bool is_max = (k == kNSums - 1);
    double a = 0.0;
    for (int b = lane; b < n_blocks; b += 64) {
        const double v = partials[(int64_t)b * kNSums + k];
        a = is_max ? fmax(a, v) : a + v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(a, off, 64);
        a = is_max ? fmax(a, o) : a + o;
    }
    if (lane == 0) out[k] = a;
}

// the same per group: workgroup y adds the sums_blocks(group's SNPs) partials of row y (`stride` slots per row)
__global__ void sums_final_groups_kernel(const double* __restrict__ partials, int stride, const int64_t* __restrict__ gstart,
                                         const double* __restrict__ rows, double* __restrict__ out) {
    const int g = (int)rows[2 * blockIdx.x];
    const int n_blocks = sums_blocks(gstart[g + 1] - gstart[g]);
    partials += (int64_t)blockIdx.x * stride * kNSums;
    out += (int64_t)blockIdx.x * kNSums;
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (k >= kNSums) return;
    const bool is_max = (k == kNSums - 1);
    double a = 0.0;
    for (int b = lane; b < n_blocks; b += 64) {
        const double v = partials[(int64_t)b * kNSums + k];
        a = is_max ? fmax(a, v) : a + v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(a, off, 64);
        a = is_max ? fmax(a, o) : a + o;
    }
    if (lane == 0) out[k] = a;
}

// ---- device-resident EM iteration of the mixture model (VIPRSMix.py:169-225 prep, :227-260 M-step, elbo) ----
constexpr int kMixResidentK = 8;                                  // = kPanelMaxK: the lane-parallel panel chain
constexpr int kMixSums(int K) { return 7 + 6 * K; }               // s[0..5] | kv[6][K] | max |eta_diff|
struct MixPrepArgs { double logit_pi[kMixResidentK], log_tau[kMixResidentK], tau[kMixResidentK]; };

// per SNP and component (C-order (m, K)): var_tau = n (1 + lambda) / sigma_eps + tau_k and the three E-step inputs
template <typename T>
__global__ void prep_mixture_kernel(const double* __restrict__ n, int64_t m, int K, MixPrepArgs a, double sigma_eps,
                                    double one_plus_lambda, double log_null_pi, T* __restrict__ mu_mult,
                                    T* __restrict__ u_logs, T* __restrict__ shvt, T* __restrict__ lnp,
                                    double* __restrict__ var_tau_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    lnp[i] = (T)log_null_pi;
    for (int k = 0; k < K; ++k) {
        const double vt = n[i] * one_plus_lambda / sigma_eps + a.tau[k];
        var_tau_out[i * K + k] = vt;
        mu_mult[i * K + k] = (T)(n[i] / (vt * sigma_eps));
        u_logs[i * K + k] = (T)(a.logit_pi[k] + 0.5 * (a.log_tau[k] - log(vt)));
        shvt[i * K + k] = (T)sqrt(0.5 * vt);
    }
}

// VIPRSMix._partial_sums on the device (float64): per-workgroup partials, fixed order.  `sums_mixture_body`: workgroup `bx`
// of `nb` over SNPs [i0, i1) -- a plan's SNPs, or one SNP group of it reduced exactly as a plan of its own would be.
template <typename T>
__device__ __forceinline__ void sums_mixture_body(int64_t i0, int64_t i1, int nb, int bx, int K, const T* __restrict__ gam,
                                                  const T* __restrict__ mu, const T* __restrict__ eta,
                                                  const T* __restrict__ q, const T* __restrict__ ed,
                                                  const T* __restrict__ beta, const double* __restrict__ var_tau,
                                                  const double* __restrict__ log_var_tau0, double one_plus_lambda,
                                                  double* __restrict__ partials) {
    constexpr int NMAX = kMixSums(kMixResidentK);
    const int N = kMixSums(K);
    double acc[NMAX];
#pragma unroll
    for (int k = 0; k < NMAX; ++k) acc[k] = 0.0;
    const double lo = 1e-15, hi = 1.0 - 1e-15;
    for (int64_t i = i0 + (int64_t)bx * kSumsBlock + threadIdx.x; i < i1; i += (int64_t)nb * kSumsBlock) {
        double zeta = 0.0, gsum = 0.0;
#pragma unroll
        for (int k = 0; k < kMixResidentK; ++k) {
            if (k < K) {
                const double g = (double)gam[i * K + k], mud = (double)mu[i * K + k], vt = var_tau[i * K + k];
                const double z = g * (mud * mud + 1.0 / vt);
                zeta += z;
                gsum += g;
                const double gc = fmin(fmax(g, lo), hi);
                acc[6 + 0 * kMixResidentK + k] += g;
                acc[6 + 1 * kMixResidentK + k] += z;
                acc[6 + 2 * kMixResidentK + k] += gc * log(gc);
                acc[6 + 3 * kMixResidentK + k] += gc;
                acc[6 + 4 * kMixResidentK + k] += gc * log_var_tau0[i * K + k];
                acc[6 + 5 * kMixResidentK + k] += gc * (mud * mud + 1.0 / vt);
            }
        }
        acc[0] += zeta;
        acc[1] += one_plus_lambda * zeta + (double)(q[i] * eta[i]);
        acc[2] += (double)beta[i] * (double)eta[i];
        acc[3] += (double)eta[i] * (double)eta[i];
        const double ng = fmin(fmax(1.0 - gsum, lo), hi);
        acc[4] += ng * log(ng);
        acc[5] += ng;
        acc[NMAX - 1] = fmax(acc[NMAX - 1], fabs((double)ed[i]));
    }
    // wave shuffle tree, then the 4 waves in order: fixed summation order
    __shared__ double red[NMAX][kSumsBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
        double a = acc[k];
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(a, off, 64);
            a = (k == NMAX - 1) ? fmax(a, o) : a + o;
        }
        if (lane == 0) red[k][wave] = a;
    }
    __syncthreads();
    // compact to the K actually used: out index n -> internal index
    if ((int)threadIdx.x < N) {
        const int nidx = threadIdx.x;
        int src;
        if (nidx < 6) src = nidx;
        else if (nidx == N - 1) src = NMAX - 1;
        else src = 6 + ((nidx - 6) / K) * kMixResidentK + (nidx - 6) % K;
        double a = red[src][0];
        for (int w = 1; w < kSumsBlock / 64; ++w) a = (src == NMAX - 1) ? fmax(a, red[src][w]) : a + red[src][w];
        partials[nidx] = a;
    }
}

template <typename T>
__global__ __launch_bounds__(kSumsBlock) void sums_mixture_kernel(int64_t m, int K, const T* __restrict__ gam,
                                                                  const T* __restrict__ mu, const T* __restrict__ eta,
                                                                  const T* __restrict__ q, const T* __restrict__ ed,
                                                                  const T* __restrict__ beta, const double* __restrict__ var_tau,
                                                                  const double* __restrict__ log_var_tau0, double one_plus_lambda,
                                                                  double* __restrict__ partials) {
    sums_mixture_body<T>(0, m, (int)gridDim.x, (int)blockIdx.x, K, gam, mu, eta, q, ed, beta, var_tau, log_var_tau0, one_plus_lambda,
                         partials + (int64_t)blockIdx.x * kMixSums(K));
}

// SNP groups of a mixture state (one model per chromosome in one plan): blockIdx.y picks a row (group, one_plus_lambda)
template <typename T>
__global__ __launch_bounds__(kSumsBlock) void sums_mixture_groups_kernel(const int64_t* __restrict__ gstart, const double* __restrict__ rows,
                                                                         int K, const T* __restrict__ gam, const T* __restrict__ mu,
                                                                         const T* __restrict__ eta, const T* __restrict__ q,
                                                                         const T* __restrict__ ed, const T* __restrict__ beta,
                                                                         const double* __restrict__ var_tau,
                                                                         const double* __restrict__ log_var_tau0,
                                                                         double* __restrict__ partials) {
    const int g = (int)rows[2 * blockIdx.y];
    const int64_t i0 = gstart[g], i1 = gstart[g + 1];
    const int nb = sums_blocks(i1 - i0);
    if ((int)blockIdx.x >= nb) return;
    sums_mixture_body<T>(i0, i1, nb, (int)blockIdx.x, K, gam, mu, eta, q, ed, beta, var_tau, log_var_tau0, rows[2 * blockIdx.y + 1],
                         partials + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * kMixSums(K));
}

// per-SNP inputs of the groups listed in `params`: rows of 4 + 3 K doubles (group, log_null_pi, sigma_eps, one_plus_lambda,
// logit_pi[K], log_tau_beta[K], tau_beta[K]); blockIdx.y picks the row, only the group's SNPs are written
template <typename T>
__global__ void prep_mixture_groups_kernel(const double* __restrict__ n, const int64_t* __restrict__ gstart, int K,
                                           const double* __restrict__ params, T* __restrict__ mu_mult, T* __restrict__ u_logs,
                                           T* __restrict__ shvt, T* __restrict__ lnp, double* __restrict__ var_tau_out) {
    const double* __restrict__ p = params + (int64_t)(4 + 3 * K) * blockIdx.y;
    const int g = (int)p[0];
    const double log_null_pi = p[1], sigma_eps = p[2], one_plus_lambda = p[3];
    const double* __restrict__ logit_pi = p + 4;
    const double* __restrict__ log_tau = p + 4 + K;
    const double* __restrict__ tau = p + 4 + 2 * K;
    const int64_t end = gstart[g + 1];
    for (int64_t i = gstart[g] + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += (int64_t)gridDim.x * blockDim.x) {
        lnp[i] = (T)log_null_pi;
        for (int k = 0; k < K; ++k) {
            const double vt = n[i] * one_plus_lambda / sigma_eps + tau[k];
            var_tau_out[i * K + k] = vt;
            mu_mult[i * K + k] = (T)(n[i] / (vt * sigma_eps));
            u_logs[i * K + k] = (T)(logit_pi[k] + 0.5 * (log_tau[k] - log(vt)));
            shvt[i * K + k] = (T)sqrt(0.5 * vt);
        }
    }
}

// one workgroup (one wave) per sum over the per-block partials; the last sum is a maximum
__global__ void sums_final_generic_kernel(const double* __restrict__ partials, int n_blocks, int n_sums,
                                          double* __restrict__ out) {
    const int k = blockIdx.x, lane = threadIdx.x;
    const bool is_max = (k == n_sums - 1);
    double a = 0.0;
    for (int b = lane; b < n_blocks; b += 64) {
        const double v = partials[(int64_t)b * n_sums + k];
        a = is_max ? fmax(a, v) : a + v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(a, off, 64);
        a = is_max ? fmax(a, o) : a + o;
    }
    if (lane == 0) out[k] = a;
}

// the same per group: workgroup (k, y) adds the sums_blocks(group's SNPs) partials of sum k of row y (`stride` slots per row)
__global__ void sums_final_generic_groups_kernel(const double* __restrict__ partials, int stride, int n_sums,
                                                 const int64_t* __restrict__ gstart, const double* __restrict__ rows,
                                                 double* __restrict__ out) {
    const int g = (int)rows[2 * blockIdx.y];
    const int n_blocks = sums_blocks(gstart[g + 1] - gstart[g]);
    partials += (int64_t)blockIdx.y * stride * n_sums;
    const int k = blockIdx.x, lane = threadIdx.x;
    const bool is_max = (k == n_sums - 1);
    double a = 0.0;
    for (int b = lane; b < n_blocks; b += 64) {
        const double v = partials[(int64_t)b * n_sums + k];
        a = is_max ? fmax(a, v) : a + v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(a, off, 64);
        a = is_max ? fmax(a, o) : a + o;
    }
    if (lane == 0) out[(int64_t)blockIdx.y * n_sums + k] = a;
}


}  // namespace

extern "C" {

// ---- state -------------------------------------------------------------------------------------
int viprs_state_create(viprs_state** out, viprs_plan* plan, int float_dtype, int model_kind, int width) {
    if (!out || !plan) return fail(VIPRS_EINVAL, "null argument");
    *out = nullptr;
    if (float_size(float_dtype) == 0) return fail(VIPRS_EINVAL, "bad float dtype code");
    if (model_kind < VIPRS_MODEL_SPIKE_SLAB || model_kind > VIPRS_MODEL_GRID) return fail(VIPRS_EINVAL, "bad model kind");
    if (model_kind == VIPRS_MODEL_SPIKE_SLAB) width = 1;
    if (width < 1) return fail(VIPRS_EINVAL, "width must be >= 1");
    HIP_TRY(hipSetDevice(plan->device));
    (void)hipGetLastError();                 // (an error left behind by an earlier, unchecked call of this thread is not ours)
    std::unique_ptr<viprs_state> S(new viprs_state());
    S->plan = plan;
    S->device = plan->device;
    S->float_dtype = float_dtype;
    S->model_kind = model_kind;
    S->width = width;
    if (plan->n_granule_rows > 0) {
        HIP_TRY(S->eta_out.alloc(S->field_elems(VIPRS_FIELD_ETA) * float_size(float_dtype)));
        HIP_TRY(S->q_out.alloc(S->field_elems(VIPRS_FIELD_ETA) * float_size(float_dtype)));
    }
    for (int k = 0; k < VIPRS_FIELD_COUNT; ++k) {
        const size_t bytes = S->field_elems(k) * float_size(float_dtype);
        HIP_TRY(S->f[k].alloc(bytes));
        // on the plan's stream (non-blocking: the null stream is NOT ordered with it -- a late null-stream
        // memset would wipe data uploaded in the meantime)
        if (bytes) HIP_TRY(hipMemsetAsync(S->f[k].p, 0, bytes, plan->stream));
    }
    HIP_TRY(hipStreamSynchronize(plan->stream));
    *out = S.release();
    return VIPRS_OK;
}

int viprs_state_destroy(viprs_state* S) {
    if (!S) return VIPRS_OK;
    // (S->device, not S->plan->device: finalizers of a garbage collector run in any order, and a state destroyed after its
    //  plan would read freed memory here -- hipSetDevice(garbage) then leaves "invalid device ordinal" as the thread's last
    //  error, which the next hipGetLastError() behind a kernel launch reports as that launch's failure)
    (void)hipSetDevice(S->device);
    delete S;
    return VIPRS_OK;
}

int viprs_state_upload(viprs_state* S, int field, const void* host) {
    if (!S || field < 0 || field >= VIPRS_FIELD_COUNT) return fail(VIPRS_EINVAL, "bad state/field");
    const size_t bytes = S->field_elems(field) * float_size(S->float_dtype);
    if (bytes == 0) return VIPRS_OK;
    if (!host) return fail(VIPRS_EINVAL, "host buffer is null");
    HIP_TRY(hipSetDevice(S->plan->device));
    HIP_TRY(hipMemcpyAsync(S->f[field].p, host, bytes, hipMemcpyHostToDevice, S->plan->stream));
    HIP_TRY(hipStreamSynchronize(S->plan->stream));
    return VIPRS_OK;
}

int viprs_state_download(viprs_state* S, int field, void* host) {
    if (!S || field < 0 || field >= VIPRS_FIELD_COUNT) return fail(VIPRS_EINVAL, "bad state/field");
    const size_t bytes = S->field_elems(field) * float_size(S->float_dtype);
    if (bytes == 0) return VIPRS_OK;
    if (!host) return fail(VIPRS_EINVAL, "host buffer is null");
    HIP_TRY(hipSetDevice(S->plan->device));
    HIP_TRY(hipMemcpyAsync(host, S->f[field].p, bytes, hipMemcpyDeviceToHost, S->plan->stream));
    HIP_TRY(hipStreamSynchronize(S->plan->stream));
    return check_device_error(S->plan);
}

int viprs_state_reset(viprs_state* S, double pi) {
    if (!S) return fail(VIPRS_EINVAL, "null state");
    viprs_plan* P = S->plan;
    HIP_TRY(hipSetDevice(P->device));
    const int64_t n_wide = (int64_t)S->field_elems(VIPRS_FIELD_VAR_GAMMA);
    const int64_t n_vec = (int64_t)S->field_elems(VIPRS_FIELD_ETA);
    const int64_t n = std::max(n_wide, n_vec);
    if (n == 0) return VIPRS_OK;
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (S->float_dtype == VIPRS_F32)
        reset_state_kernel<float><<<grid, 256, 0, P->stream>>>(
            (float*)S->f[VIPRS_FIELD_VAR_GAMMA].p, (float*)S->f[VIPRS_FIELD_VAR_MU].p, n_wide,
            (float*)S->f[VIPRS_FIELD_ETA].p, (float*)S->f[VIPRS_FIELD_Q].p, (float*)S->f[VIPRS_FIELD_ETA_DIFF].p, n_vec,
            (float)pi);
    else
        reset_state_kernel<double><<<grid, 256, 0, P->stream>>>(
            (double*)S->f[VIPRS_FIELD_VAR_GAMMA].p, (double*)S->f[VIPRS_FIELD_VAR_MU].p, n_wide,
            (double*)S->f[VIPRS_FIELD_ETA].p, (double*)S->f[VIPRS_FIELD_Q].p, (double*)S->f[VIPRS_FIELD_ETA_DIFF].p, n_vec,
            pi);
    HIP_TRY(hipGetLastError());
    return VIPRS_OK;
}

}  // extern "C"

namespace viprs {
// after a synchronisation point: did a team hand-off give up (bounded spin)?
int check_device_error(viprs_plan* P) {
    int32_t e = 0;
    HIP_TRY(hipMemcpy(&e, P->d_error.p, sizeof(e), hipMemcpyDeviceToHost));
    if (e != 0) {
        HIP_TRY(hipMemsetAsync(P->d_error.p, 0, sizeof(e), P->stream));
        HIP_TRY(hipStreamSynchronize(P->stream));
        return fail(VIPRS_EDEVICE, "E-step kernel: a team hand-off timed out (results of this sweep are invalid)");
    }
    return VIPRS_OK;
}
}  // namespace viprs

// A rank whose plan holds no SNP still takes part in the collective of viprs_state_set_comm: it contributes zeros.
static int sums_enqueue_empty(viprs_state* S, int n, int group) {
    viprs_plan* P = S->plan;
    HIP_TRY(hipSetDevice(P->device));
    if (S->d_sums.n < (size_t)n) HIP_TRY(S->d_sums.alloc((size_t)n));
    if (S->h_sums_cap < (size_t)n + 1) {
        if (S->h_sums) HIP_TRY(hipHostFree(S->h_sums));
        S->h_sums = nullptr;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S->h_sums), ((size_t)n + 1) * sizeof(double), hipHostMallocDefault));
        S->h_sums_cap = (size_t)n + 1;
    }
    HIP_TRY(hipMemsetAsync(S->d_sums.p, 0, (size_t)n * sizeof(double), P->stream));
    const int rc = comm_reduce_on_stream(S->comm, S->d_sums.p, n, group, P->stream);
    if (rc != VIPRS_OK) return rc;
    HIP_TRY(hipMemcpyAsync(S->h_sums, S->d_sums.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, P->stream));
    HIP_TRY(hipMemcpyAsync(S->h_sums + n, P->d_error.p, sizeof(int32_t), hipMemcpyDeviceToHost, P->stream));
    S->sums_pending = true;
    S->sums_empty = false;
    return VIPRS_OK;
}

template <typename T>
static int sums_enqueue(viprs_state* S, int64_t off, int64_t vt_off, double one_plus_lambda) {
    viprs_plan* P = S->plan;
    const int nb = sums_blocks(P->m);
    if (S->d_partials.n < (size_t)nb * kNSums) HIP_TRY(S->d_partials.alloc((size_t)nb * kNSums));
    if (!S->d_sums.p) HIP_TRY(S->d_sums.alloc(kNSums));
    // pinned landing buffer: kNSums doubles + the plan's device error word (no second synchronisation)
    if (!S->h_sums) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S->h_sums), (kNSums + 1) * sizeof(double), hipHostMallocDefault));
        S->h_sums_cap = kNSums + 1;
    }
    sums_kernel<T><<<nb, kSumsBlock, 0, P->stream>>>(
        P->m, (const T*)S->f[VIPRS_FIELD_VAR_GAMMA].p + off, (const T*)S->f[VIPRS_FIELD_VAR_MU].p + off,
        (const T*)S->f[VIPRS_FIELD_ETA].p + off, (const T*)S->f[VIPRS_FIELD_Q].p + off,
        (const T*)S->f[VIPRS_FIELD_ETA_DIFF].p + off, (const T*)S->f[VIPRS_FIELD_STD_BETA].p, S->d_var_tau.p + vt_off,
        one_plus_lambda, S->d_weight.p, S->d_partials.p);
    HIP_TRY(hipGetLastError());
    sums_final_kernel<<<1, 64 * kNSums, 0, P->stream>>>(S->d_partials.p, nb, S->d_sums.p);
    HIP_TRY(hipGetLastError());
    if (S->comm) {                      // all ranks: ONE all-gather + rank-ordered reduction, still on the plan's stream
        const int rc = comm_reduce_on_stream(S->comm, S->d_sums.p, kNSums, kNSums, P->stream);
        if (rc != VIPRS_OK) return rc;
    }
    // pinned host buffer: the copy is truly asynchronous, several plans' sums overlap
    HIP_TRY(hipMemcpyAsync(S->h_sums, S->d_sums.p, kNSums * sizeof(double), hipMemcpyDeviceToHost, P->stream));
    HIP_TRY(hipMemcpyAsync(S->h_sums + kNSums, P->d_error.p, sizeof(int32_t), hipMemcpyDeviceToHost, P->stream));
    S->sums_pending = true;
    return VIPRS_OK;
}

static int sums_finish(viprs_state* S, double* out) {
    viprs_plan* P = S->plan;
    if (!S->sums_pending) return fail(VIPRS_EINVAL, "no device sums in flight (viprs_state_sums_begin)");
    HIP_TRY(hipStreamSynchronize(P->stream));
    S->sums_pending = false;
    for (int k = 0; k < kNSums; ++k) out[k] = S->h_sums[k];
    int32_t e = 0;
    memcpy(&e, S->h_sums + kNSums, sizeof(e));
    return e != 0 ? check_device_error(P) : VIPRS_OK;       // slow path only when a hand-off timed out
}

template <typename T>
static int sums_launch(viprs_state* S, int64_t off, int64_t vt_off, double one_plus_lambda, double* out) {
    const int rc = sums_enqueue<T>(S, off, vt_off, one_plus_lambda);
    return rc != VIPRS_OK ? rc : sums_finish(S, out);
}

template <typename T>
static int sums_columns_enqueue(viprs_state* S, int n) {
    viprs_plan* P = S->plan;
    const int nb = (int)std::min<int64_t>((P->m + kSumsBlock - 1) / kSumsBlock, 256);
    const size_t need = (size_t)nb * kNSums * n;
    if (S->d_partials.n < need) HIP_TRY(S->d_partials.alloc(need));
    if (S->d_sums.n < (size_t)kNSums * S->width) HIP_TRY(S->d_sums.alloc((size_t)kNSums * S->width));
    const size_t hcap = (size_t)kNSums * S->width + 1;
    if (S->h_sums_cap < hcap) {
        if (S->h_sums) HIP_TRY(hipHostFree(S->h_sums));
        S->h_sums = nullptr;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S->h_sums), hcap * sizeof(double), hipHostMallocDefault));
        S->h_sums_cap = hcap;
    }
    sums_kernel<T><<<dim3(nb, n), kSumsBlock, 0, P->stream>>>(
        P->m, (const T*)S->f[VIPRS_FIELD_VAR_GAMMA].p, (const T*)S->f[VIPRS_FIELD_VAR_MU].p, (const T*)S->f[VIPRS_FIELD_ETA].p,
        (const T*)S->f[VIPRS_FIELD_Q].p, (const T*)S->f[VIPRS_FIELD_ETA_DIFF].p, (const T*)S->f[VIPRS_FIELD_STD_BETA].p,
        S->d_n.p, 0.0, S->d_weight.p, S->d_partials.p, S->d_sumcols.p);
    HIP_TRY(hipGetLastError());
    sums_final_kernel<<<n, 64 * kNSums, 0, P->stream>>>(S->d_partials.p, nb, S->d_sums.p);
    HIP_TRY(hipGetLastError());
    if (S->comm) {
        const int rc = comm_reduce_on_stream(S->comm, S->d_sums.p, kNSums * n, kNSums, P->stream);
        if (rc != VIPRS_OK) return rc;
    }
    HIP_TRY(hipMemcpyAsync(S->h_sums, S->d_sums.p, (size_t)kNSums * n * sizeof(double), hipMemcpyDeviceToHost, P->stream));
    HIP_TRY(hipMemcpyAsync(S->h_sums + (size_t)kNSums * n, P->d_error.p, sizeof(int32_t), hipMemcpyDeviceToHost, P->stream));
    S->sums_cols = n;
    S->sums_pending = true;
    return VIPRS_OK;
}

extern "C" {

int viprs_state_synchronize(viprs_state* S) {
    if (!S) return fail(VIPRS_EINVAL, "null state");
    HIP_TRY(hipSetDevice(S->plan->device));
    HIP_TRY(hipStreamSynchronize(S->plan->stream));
    return check_device_error(S->plan);
}

}  // extern "C"

extern "C" {

int viprs_state_set_n_per_snp(viprs_state* S, const double* n) {
    if (!S || !n) return fail(VIPRS_EINVAL, "null argument");
    viprs_plan* P = S->plan;
    HIP_TRY(hipSetDevice(P->device));
    const size_t m = (size_t)P->m;
    if (m == 0) return VIPRS_OK;
    HIP_TRY(S->d_n.alloc(m));
    HIP_TRY(S->d_var_tau.alloc(m));
    HIP_TRY(hipMemcpyAsync(S->d_n.p, n, m * sizeof(double), hipMemcpyHostToDevice, P->stream));
    HIP_TRY(hipMemsetAsync(S->d_var_tau.p, 0, m * sizeof(double), P->stream));
    HIP_TRY(hipStreamSynchronize(P->stream));
    return VIPRS_OK;
}

int viprs_state_set_snp_weights(viprs_state* S, const double* w) {
    if (!S) return fail(VIPRS_EINVAL, "null argument");
    viprs_plan* P = S->plan;
    HIP_TRY(hipSetDevice(P->device));
    const size_t m = (size_t)P->m;
    if (!w || m == 0) { HIP_TRY(S->d_weight.alloc(0)); return VIPRS_OK; }
    HIP_TRY(S->d_weight.alloc(m));
    HIP_TRY(hipMemcpyAsync(S->d_weight.p, w, m * sizeof(double), hipMemcpyHostToDevice, P->stream));
    HIP_TRY(hipStreamSynchronize(P->stream));
    return VIPRS_OK;
}

int viprs_state_prep(viprs_state* S, double logit_pi, double log_tau_beta, double sigma_epsilon, double tau_beta,
                     double one_plus_lambda) {
    if (!S) return fail(VIPRS_EINVAL, "null state");
    if (S->model_kind != VIPRS_MODEL_SPIKE_SLAB) return fail(VIPRS_EUNSUPPORTED, "device prep: spike-and-slab only");
    viprs_plan* P = S->plan;
    if (P->m == 0) return VIPRS_OK;
    if (!S->d_n.p) return fail(VIPRS_EINVAL, "viprs_state_set_n_per_snp has not been called");
    HIP_TRY(hipSetDevice(P->device));
    const unsigned grid = (unsigned)((P->m + 255) / 256);
    if (S->float_dtype == VIPRS_F32)
        prep_kernel<float><<<grid, 256, 0, P->stream>>>(S->d_n.p, P->m, logit_pi, log_tau_beta, sigma_epsilon, tau_beta,
                                                        one_plus_lambda, (float*)S->f[VIPRS_FIELD_MU_MULT].p,
                                                        (float*)S->f[VIPRS_FIELD_U_LOGS].p,
                                                        (float*)S->f[VIPRS_FIELD_SQRT_HALF_VAR_TAU].p, S->d_var_tau.p, 0);
    else
        prep_kernel<double><<<grid, 256, 0, P->stream>>>(S->d_n.p, P->m, logit_pi, log_tau_beta, sigma_epsilon, tau_beta,
                                                         one_plus_lambda, (double*)S->f[VIPRS_FIELD_MU_MULT].p,
                                                         (double*)S->f[VIPRS_FIELD_U_LOGS].p,
                                                         (double*)S->f[VIPRS_FIELD_SQRT_HALF_VAR_TAU].p, S->d_var_tau.p, 0);
    HIP_TRY(hipGetLastError());
    return VIPRS_OK;
}

int viprs_state_sums(viprs_state* S, double one_plus_lambda, double* out) {
    if (!S || !out) return fail(VIPRS_EINVAL, "null argument");
    if (S->model_kind != VIPRS_MODEL_SPIKE_SLAB) return fail(VIPRS_EUNSUPPORTED, "device sums: spike-and-slab only");
    viprs_plan* P = S->plan;
    for (int k = 0; k < kNSums; ++k) out[k] = 0.0;
    if (P->m == 0 && S->comm) {          // an empty rank still takes part in the collective (it contributes zeros)
        const int rc = sums_enqueue_empty(S, kNSums, kNSums);
        return rc != VIPRS_OK ? rc : sums_finish(S, out);
    }
    if (P->m == 0) return VIPRS_OK;
    if (!S->d_var_tau.p) return fail(VIPRS_EINVAL, "viprs_state_set_n_per_snp / viprs_state_prep have not been called");
    HIP_TRY(hipSetDevice(P->device));
    return S->float_dtype == VIPRS_F32 ? sums_launch<float>(S, 0, 0, one_plus_lambda, out)
                                       : sums_launch<double>(S, 0, 0, one_plus_lambda, out);
}

int viprs_state_sums_begin(viprs_state* S, double one_plus_lambda) {
    if (!S) return fail(VIPRS_EINVAL, "null argument");
    if (S->model_kind != VIPRS_MODEL_SPIKE_SLAB) return fail(VIPRS_EUNSUPPORTED, "device sums: spike-and-slab only");
    viprs_plan* P = S->plan;
    if (P->m == 0 && S->comm) return sums_enqueue_empty(S, kNSums, kNSums);
    if (P->m == 0) { S->sums_pending = false; S->sums_empty = true; return VIPRS_OK; }
    S->sums_empty = false;
    if (!S->d_var_tau.p) return fail(VIPRS_EINVAL, "viprs_state_set_n_per_snp / viprs_state_prep have not been called");
    HIP_TRY(hipSetDevice(P->device));
    return S->float_dtype == VIPRS_F32 ? sums_enqueue<float>(S, 0, 0, one_plus_lambda)
                                       : sums_enqueue<double>(S, 0, 0, one_plus_lambda);
}

int viprs_state_sums_end(viprs_state* S, double* out) {
    if (!S || !out) return fail(VIPRS_EINVAL, "null argument");
    if (S->sums_empty) {
        for (int k = 0; k < kNSums; ++k) out[k] = 0.0;
        return VIPRS_OK;
    }
    HIP_TRY(hipSetDevice(S->plan->device));
    return sums_finish(S, out);
}

static void record_col_prep(viprs_state* S, int g, double one_plus_lambda, double sigma_eps, double tau_beta) {
    if (S->col_prep.size() != (size_t)3 * S->width) S->col_prep.assign((size_t)3 * S->width, NAN);
    S->col_prep[3 * (size_t)g] = one_plus_lambda;
    S->col_prep[3 * (size_t)g + 1] = sigma_eps;
    S->col_prep[3 * (size_t)g + 2] = tau_beta;
}
static bool col_prepped(const viprs_state* S, int g) {
    return S->col_prep.size() == (size_t)3 * S->width && !std::isnan(S->col_prep[3 * (size_t)g]);
}

static int grid_column_check(viprs_state* S, int g) {
    if (!S) return fail(VIPRS_EINVAL, "null state");
    if (S->model_kind != VIPRS_MODEL_GRID) return fail(VIPRS_EINVAL, "not a grid state");
    if (g < 0 || g >= S->width) return fail(VIPRS_EINVAL, "model index out of range");
    // (an empty plan has no per-SNP sample sizes to set: a rank without LD blocks passes)
    if (!S->d_n.p && S->plan->m > 0) return fail(VIPRS_EINVAL, "viprs_state_set_n_per_snp has not been called");
    return VIPRS_OK;
}

int viprs_state_prep_column(viprs_state* S, int g, double logit_pi, double log_tau_beta, double sigma_epsilon,
                            double tau_beta, double one_plus_lambda) {
    int rc = grid_column_check(S, g);
    if (rc != VIPRS_OK) return rc;
    viprs_plan* P = S->plan;
    if (P->m == 0) return VIPRS_OK;
    HIP_TRY(hipSetDevice(P->device));
    record_col_prep(S, g, one_plus_lambda, sigma_epsilon, tau_beta);
    const int64_t off = (int64_t)g * P->m;
    const unsigned grid = (unsigned)((P->m + 255) / 256);
    if (S->float_dtype == VIPRS_F32)
        prep_kernel<float><<<grid, 256, 0, P->stream>>>(S->d_n.p, P->m, logit_pi, log_tau_beta, sigma_epsilon, tau_beta,
                                                        one_plus_lambda, (float*)S->f[VIPRS_FIELD_MU_MULT].p + off,
                                                        (float*)S->f[VIPRS_FIELD_U_LOGS].p + off,
                                                        (float*)S->f[VIPRS_FIELD_SQRT_HALF_VAR_TAU].p + off,
                                                        nullptr, 1);
    else
        prep_kernel<double><<<grid, 256, 0, P->stream>>>(S->d_n.p, P->m, logit_pi, log_tau_beta, sigma_epsilon, tau_beta,
                                                         one_plus_lambda, (double*)S->f[VIPRS_FIELD_MU_MULT].p + off,
                                                         (double*)S->f[VIPRS_FIELD_U_LOGS].p + off,
                                                         (double*)S->f[VIPRS_FIELD_SQRT_HALF_VAR_TAU].p + off,
                                                         nullptr, 1);
    HIP_TRY(hipGetLastError());
    return VIPRS_OK;
}

int viprs_state_set_log_var_tau(viprs_state* S, const double* log_var_tau) {
    if (!S || !log_var_tau) return fail(VIPRS_EINVAL, "null argument");
    if (S->model_kind != VIPRS_MODEL_MIXTURE) return fail(VIPRS_EINVAL, "not a mixture state");
    viprs_plan* P = S->plan;
    const size_t n = (size_t)P->m * S->width;
    if (n == 0) return VIPRS_OK;
    HIP_TRY(hipSetDevice(P->device));
    HIP_TRY(S->d_log_var_tau0.alloc(n));
    HIP_TRY(hipMemcpyAsync(S->d_log_var_tau0.p, log_var_tau, n * sizeof(double), hipMemcpyHostToDevice, P->stream));
    HIP_TRY(hipStreamSynchronize(P->stream));
    return VIPRS_OK;
}

int viprs_state_prep_mixture(viprs_state* S, const double* logit_pi, const double* log_tau_beta, const double* tau_beta,
                             double log_null_pi, double sigma_epsilon, double one_plus_lambda) {
    if (!S || !logit_pi || !log_tau_beta || !tau_beta) return fail(VIPRS_EINVAL, "null argument");
    if (S->model_kind != VIPRS_MODEL_MIXTURE) return fail(VIPRS_EINVAL, "not a mixture state");
    if (S->width > kMixResidentK) return fail(VIPRS_EUNSUPPORTED, "device-resident mixture iteration: K <= 8");
    if (!S->d_n.p) return fail(VIPRS_EINVAL, "viprs_state_set_n_per_snp has not been called");
    viprs_plan* P = S->plan;
    if (P->m == 0) return VIPRS_OK;
    HIP_TRY(hipSetDevice(P->device));
    const int K = S->width;
    if (S->d_var_tau.n < (size_t)P->m * K) {
        HIP_TRY(hipStreamSynchronize(P->stream));
        HIP_TRY(S->d_var_tau.alloc((size_t)P->m * K));
    }
    MixPrepArgs a{};
    for (int k = 0; k < K; ++k) { a.logit_pi[k] = logit_pi[k]; a.log_tau[k] = log_tau_beta[k]; a.tau[k] = tau_beta[k]; }
    const unsigned grid = (unsigned)((P->m + 255) / 256);
    if (S->float_dtype == VIPRS_F32)
        prep_mixture_kernel<float><<<grid, 256, 0, P->stream>>>(
            S->d_n.p, P->m, K, a, sigma_epsilon, one_plus_lambda, log_null_pi, (float*)S->f[VIPRS_FIELD_MU_MULT].p,
            (float*)S->f[VIPRS_FIELD_U_LOGS].p, (float*)S->f[VIPRS_FIELD_SQRT_HALF_VAR_TAU].p,
            (float*)S->f[VIPRS_FIELD_LOG_NULL_PI].p, S->d_var_tau.p);
    else
        prep_mixture_kernel<double><<<grid, 256, 0, P->stream>>>(
            S->d_n.p, P->m, K, a, sigma_epsilon, one_plus_lambda, log_null_pi, (double*)S->f[VIPRS_FIELD_MU_MULT].p,
            (double*)S->f[VIPRS_FIELD_U_LOGS].p, (double*)S->f[VIPRS_FIELD_SQRT_HALF_VAR_TAU].p,
            (double*)S->f[VIPRS_FIELD_LOG_NULL_PI].p, S->d_var_tau.p);
    HIP_TRY(hipGetLastError());
    return VIPRS_OK;
}

int viprs_state_sums_mixture_begin(viprs_state* S, double one_plus_lambda) {
    if (!S) return fail(VIPRS_EINVAL, "null argument");
    if (S->model_kind != VIPRS_MODEL_MIXTURE) return fail(VIPRS_EINVAL, "not a mixture state");
    if (S->width > kMixResidentK) return fail(VIPRS_EUNSUPPORTED, "device-resident mixture iteration: K <= 8");
    viprs_plan* P = S->plan;
    const int K = S->width, N = kMixSums(K);
    S->sums_cols = N;
    if (P->m == 0 && S->comm) return sums_enqueue_empty(S, N, N);
    if (P->m == 0) { S->sums_pending = false; S->sums_empty = true; return VIPRS_OK; }
    S->sums_empty = false;
    if (S->d_var_tau.n < (size_t)P->m * K || !S->d_log_var_tau0.p)
        return fail(VIPRS_EINVAL, "viprs_state_prep_mixture / viprs_state_set_log_var_tau have not been called");
    HIP_TRY(hipSetDevice(P->device));
    const int nb = (int)std::min<int64_t>((P->m + kSumsBlock - 1) / kSumsBlock, 1024);
    if (S->d_partials.n < (size_t)nb * N) HIP_TRY(S->d_partials.alloc((size_t)nb * N));
    if (S->d_sums.n < (size_t)N) HIP_TRY(S->d_sums.alloc((size_t)N));
    if (S->h_sums_cap < (size_t)N + 1) {
        if (S->h_sums) HIP_TRY(hipHostFree(S->h_sums));
        S->h_sums = nullptr;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S->h_sums), ((size_t)N + 1) * sizeof(double), hipHostMallocDefault));
        S->h_sums_cap = (size_t)N + 1;
    }
    if (S->float_dtype == VIPRS_F32)
        sums_mixture_kernel<float><<<nb, kSumsBlock, 0, P->stream>>>(
            P->m, K, (const float*)S->f[VIPRS_FIELD_VAR_GAMMA].p, (const float*)S->f[VIPRS_FIELD_VAR_MU].p,
            (const float*)S->f[VIPRS_FIELD_ETA].p, (const float*)S->f[VIPRS_FIELD_Q].p, (const float*)S->f[VIPRS_FIELD_ETA_DIFF].p,
            (const float*)S->f[VIPRS_FIELD_STD_BETA].p, S->d_var_tau.p, S->d_log_var_tau0.p, one_plus_lambda, S->d_partials.p);
    else
        sums_mixture_kernel<double><<<nb, kSumsBlock, 0, P->stream>>>(
            P->m, K, (const double*)S->f[VIPRS_FIELD_VAR_GAMMA].p, (const double*)S->f[VIPRS_FIELD_VAR_MU].p,
            (const double*)S->f[VIPRS_FIELD_ETA].p, (const double*)S->f[VIPRS_FIELD_Q].p, (const double*)S->f[VIPRS_FIELD_ETA_DIFF].p,
            (const double*)S->f[VIPRS_FIELD_STD_BETA].p, S->d_var_tau.p, S->d_log_var_tau0.p, one_plus_lambda, S->d_partials.p);
    HIP_TRY(hipGetLastError());
    sums_final_generic_kernel<<<N, 64, 0, P->stream>>>(S->d_partials.p, nb, N, S->d_sums.p);
    HIP_TRY(hipGetLastError());
    if (S->comm) {
        const int rc = comm_reduce_on_stream(S->comm, S->d_sums.p, N, N, P->stream);
        if (rc != VIPRS_OK) return rc;
    }
    HIP_TRY(hipMemcpyAsync(S->h_sums, S->d_sums.p, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, P->stream));
    HIP_TRY(hipMemcpyAsync(S->h_sums + N, P->d_error.p, sizeof(int32_t), hipMemcpyDeviceToHost, P->stream));
    S->sums_pending = true;
    return VIPRS_OK;
}

int viprs_state_sums_mixture_end(viprs_state* S, double* out) {
    if (!S || !out) return fail(VIPRS_EINVAL, "null argument");
    if (S->model_kind != VIPRS_MODEL_MIXTURE) return fail(VIPRS_EINVAL, "not a mixture state");
    const int N = kMixSums(S->width);
    if (S->sums_empty) {
        for (int k = 0; k < N; ++k) out[k] = 0.0;
        return VIPRS_OK;
    }
    if (!S->sums_pending) return fail(VIPRS_EINVAL, "no device sums in flight (viprs_state_sums_mixture_begin)");
    viprs_plan* P = S->plan;
    HIP_TRY(hipSetDevice(P->device));
    HIP_TRY(hipStreamSynchronize(P->stream));
    S->sums_pending = false;
    for (int k = 0; k < N; ++k) out[k] = S->h_sums[k];
    int32_t e = 0;
    memcpy(&e, S->h_sums + N, sizeof(e));
    return e != 0 ? check_device_error(P) : VIPRS_OK;
}

int viprs_state_prep_columns(viprs_state* S, int n, const double* params) {
    if (!S || !params) return fail(VIPRS_EINVAL, "null argument");
    if (S->model_kind != VIPRS_MODEL_GRID) return fail(VIPRS_EINVAL, "not a grid state");
    if (n < 0 || n > S->width) return fail(VIPRS_EINVAL, "bad column count");
    for (int i = 0; i < n; ++i)
        if (params[6 * i] < 0 || params[6 * i] >= S->width || params[6 * i] != floor(params[6 * i]))
            return fail(VIPRS_EINVAL, "model index out of range");
    if (!S->d_n.p) return fail(VIPRS_EINVAL, "viprs_state_set_n_per_snp has not been called");
    viprs_plan* P = S->plan;
    if (P->m == 0 || n == 0) return VIPRS_OK;
    HIP_TRY(hipSetDevice(P->device));
    for (int i = 0; i < n; ++i) record_col_prep(S, (int)params[6 * i], params[6 * i + 5], params[6 * i + 3], params[6 * i + 4]);
    if (S->d_colparams.n < (size_t)6 * S->width) HIP_TRY(S->d_colparams.alloc((size_t)6 * S->width));
    if (!S->h_params) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S->h_params), (size_t)11 * S->width * sizeof(double), hipHostMallocDefault));
    if (!S->ev_prep) HIP_TRY(hipEventCreateWithFlags(&S->ev_prep, hipEventDisableTiming));
    else HIP_TRY(hipEventSynchronize(S->ev_prep));            // the previous launch has read its parameters
    memcpy(S->h_params, params, (size_t)6 * n * sizeof(double));
    // pinned staging + copy ON the plan's stream: ordered with the kernel below (the null stream is not)
    HIP_TRY(hipMemcpyAsync(S->d_colparams.p, S->h_params, (size_t)6 * n * sizeof(double), hipMemcpyHostToDevice, P->stream));
    const dim3 grid((unsigned)((P->m + 255) / 256), (unsigned)n);
    if (S->float_dtype == VIPRS_F32)
        prep_columns_kernel<float><<<grid, 256, 0, P->stream>>>(S->d_n.p, P->m, S->d_colparams.p,
                                                                (float*)S->f[VIPRS_FIELD_MU_MULT].p, (float*)S->f[VIPRS_FIELD_U_LOGS].p,
                                                                (float*)S->f[VIPRS_FIELD_SQRT_HALF_VAR_TAU].p);
    else
        prep_columns_kernel<double><<<grid, 256, 0, P->stream>>>(S->d_n.p, P->m, S->d_colparams.p,
                                                                 (double*)S->f[VIPRS_FIELD_MU_MULT].p, (double*)S->f[VIPRS_FIELD_U_LOGS].p,
                                                                 (double*)S->f[VIPRS_FIELD_SQRT_HALF_VAR_TAU].p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(S->ev_prep, P->stream));
    return VIPRS_OK;
}

int viprs_state_sums_columns_begin(viprs_state* S, int n, const double* cols) {
    if (!S || !cols) return fail(VIPRS_EINVAL, "null argument");
    if (S->model_kind != VIPRS_MODEL_GRID) return fail(VIPRS_EINVAL, "not a grid state");
    if (n < 0 || n > S->width) return fail(VIPRS_EINVAL, "bad column count");
    for (int i = 0; i < n; ++i)
        if (cols[2 * i] < 0 || cols[2 * i] >= S->width || cols[2 * i] != floor(cols[2 * i]))
            return fail(VIPRS_EINVAL, "model index out of range");
    viprs_plan* P = S->plan;
    S->sums_cols = n;
    if (P->m == 0 && n > 0 && S->comm) return sums_enqueue_empty(S, kNSums * n, kNSums);
    if (P->m == 0 || n == 0) { S->sums_pending = false; S->sums_empty = true; return VIPRS_OK; }
    S->sums_empty = false;
    for (int i = 0; i < n; ++i)
        if (!col_prepped(S, (int)cols[2 * i])) return fail(VIPRS_EINVAL, "viprs_state_prep_column(s) has not been called");
    HIP_TRY(hipSetDevice(P->device));
    // (own buffer: the previous reduction that read it has been collected, nothing else does)
    if (S->d_sumcols.n < (size_t)5 * S->width) HIP_TRY(S->d_sumcols.alloc((size_t)5 * S->width));
    if (!S->h_params) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S->h_params), (size_t)11 * S->width * sizeof(double), hipHostMallocDefault));
    // device rows: (column, one_plus_lambda of these sums | what the column's last prep built var_tau from)
    double* h = S->h_params + (size_t)6 * S->width;
    for (int i = 0; i < n; ++i) {
        const int g = (int)cols[2 * i];
        h[5 * i] = cols[2 * i];
        h[5 * i + 1] = cols[2 * i + 1];
        h[5 * i + 2] = S->col_prep[3 * (size_t)g];
        h[5 * i + 3] = S->col_prep[3 * (size_t)g + 1];
        h[5 * i + 4] = S->col_prep[3 * (size_t)g + 2];
    }
    HIP_TRY(hipMemcpyAsync(S->d_sumcols.p, h, (size_t)5 * n * sizeof(double), hipMemcpyHostToDevice, P->stream));
    return S->float_dtype == VIPRS_F32 ? sums_columns_enqueue<float>(S, n) : sums_columns_enqueue<double>(S, n);
}

int viprs_state_sums_columns_end(viprs_state* S, double* out) {
    if (!S || !out) return fail(VIPRS_EINVAL, "null argument");
    const int n = S->sums_cols;
    if (S->sums_empty) {
        for (int k = 0; k < kNSums * n; ++k) out[k] = 0.0;
        return VIPRS_OK;
    }
    if (!S->sums_pending) return fail(VIPRS_EINVAL, "no device sums in flight (viprs_state_sums_columns_begin)");
    viprs_plan* P = S->plan;
    HIP_TRY(hipSetDevice(P->device));
    HIP_TRY(hipStreamSynchronize(P->stream));
    S->sums_pending = false;
    for (int k = 0; k < kNSums * n; ++k) out[k] = S->h_sums[k];
    int32_t e = 0;
    memcpy(&e, S->h_sums + (size_t)kNSums * n, sizeof(e));
    return e != 0 ? check_device_error(P) : VIPRS_OK;
}

int viprs_state_sums_column(viprs_state* S, int g, double one_plus_lambda, double* out) {
    if (!out) return fail(VIPRS_EINVAL, "null argument");
    int rc = grid_column_check(S, g);
    if (rc != VIPRS_OK) return rc;
    viprs_plan* P = S->plan;
    for (int k = 0; k < kNSums; ++k) out[k] = 0.0;
    if (P->m == 0 && S->comm) {          // an empty rank still takes part in the collective (it contributes zeros)
        rc = sums_enqueue_empty(S, kNSums, kNSums);
        return rc != VIPRS_OK ? rc : sums_finish(S, out);
    }
    if (P->m == 0) return VIPRS_OK;
    const double row[2] = {(double)g, one_plus_lambda};
    rc = viprs_state_sums_columns_begin(S, 1, row);
    return rc != VIPRS_OK ? rc : viprs_state_sums_columns_end(S, out);
}

int viprs_state_reset_column(viprs_state* S, int g, double pi) {
    if (!S) return fail(VIPRS_EINVAL, "null state");
    if (S->model_kind != VIPRS_MODEL_GRID) return fail(VIPRS_EINVAL, "not a grid state");
    if (g < 0 || g >= S->width) return fail(VIPRS_EINVAL, "model index out of range");
    viprs_plan* P = S->plan;
    if (P->m == 0) return VIPRS_OK;
    HIP_TRY(hipSetDevice(P->device));
    const int64_t off = (int64_t)g * P->m, n = P->m;
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (S->float_dtype == VIPRS_F32)
        reset_state_kernel<float><<<grid, 256, 0, P->stream>>>(
            (float*)S->f[VIPRS_FIELD_VAR_GAMMA].p + off, (float*)S->f[VIPRS_FIELD_VAR_MU].p + off, n,
            (float*)S->f[VIPRS_FIELD_ETA].p + off, (float*)S->f[VIPRS_FIELD_Q].p + off,
            (float*)S->f[VIPRS_FIELD_ETA_DIFF].p + off, n, (float)pi);
    else
        reset_state_kernel<double><<<grid, 256, 0, P->stream>>>(
            (double*)S->f[VIPRS_FIELD_VAR_GAMMA].p + off, (double*)S->f[VIPRS_FIELD_VAR_MU].p + off, n,
            (double*)S->f[VIPRS_FIELD_ETA].p + off, (double*)S->f[VIPRS_FIELD_Q].p + off,
            (double*)S->f[VIPRS_FIELD_ETA_DIFF].p + off, n, pi);
    HIP_TRY(hipGetLastError());
    return VIPRS_OK;
}


// doubles per row of the groups' prep parameters (viprs_state_prep_groups / viprs_state_prep_mixture_groups)
static size_t group_prep_width(const viprs_state* S) {
    return S->model_kind == VIPRS_MODEL_MIXTURE ? (size_t)4 + 3 * (size_t)S->width : 6;
}

// ---- SNP groups: one spike-and-slab (or mixture) model per chromosome, all of them in ONE plan / state (bin/viprs_fit:232-238 fits one
// model per chromosome unless --genomewide; the chromosomes' LD blocks are independent, so their E-steps share one sweep)
int viprs_state_set_groups(viprs_state* S, int n_groups, const int64_t* group_start) {
    if (!S) return fail(VIPRS_EINVAL, "null state");
    if (S->model_kind != VIPRS_MODEL_SPIKE_SLAB && S->model_kind != VIPRS_MODEL_MIXTURE)
        return fail(VIPRS_EUNSUPPORTED, "SNP groups: spike-and-slab and mixture states only");
    if (S->model_kind == VIPRS_MODEL_MIXTURE && S->width > kMixResidentK)
        return fail(VIPRS_EUNSUPPORTED, "device-resident mixture iteration: K <= 8");
    viprs_plan* P = S->plan;
    if (n_groups == 0) {                         // back to one set of hyper-parameters
        S->n_groups = 0;
        S->group_start.clear();
        return VIPRS_OK;
    }
    if (n_groups < 0 || !group_start) return fail(VIPRS_EINVAL, "bad group list");
    if (group_start[0] != 0 || group_start[n_groups] != P->m) return fail(VIPRS_EINVAL, "the groups must cover SNPs 0 .. m");
    for (int g = 0; g < n_groups; ++g)
        if (group_start[g + 1] < group_start[g]) return fail(VIPRS_EINVAL, "group_start must not decrease");
    // a group is a set of whole LD blocks (its hyper-parameters are fixed within a block's sweep)
    for (const Block& b : P->blocks) {
        const int64_t* it = std::upper_bound(group_start, group_start + n_groups + 1, b.start);      // first boundary > b.start
        if (it != group_start + n_groups + 1 && *it < b.end) return fail(VIPRS_EINVAL, "a group boundary cuts through an LD block");
    }
    HIP_TRY(hipSetDevice(P->device));
    HIP_TRY(hipStreamSynchronize(P->stream));
    S->n_groups = n_groups;
    S->group_start.assign(group_start, group_start + n_groups + 1);
    S->group_max_nb = 1;
    for (int g = 0; g < n_groups; ++g) S->group_max_nb = std::max(S->group_max_nb, sums_blocks(group_start[g + 1] - group_start[g]));
    HIP_TRY(S->d_group_start.alloc((size_t)n_groups + 1));
    HIP_TRY(hipMemcpy(S->d_group_start.p, group_start, sizeof(int64_t) * ((size_t)n_groups + 1), hipMemcpyHostToDevice));
    const size_t pw = group_prep_width(S);
    HIP_TRY(S->d_group_prep.alloc(pw * n_groups));
    HIP_TRY(S->d_group_sumrows.alloc((size_t)2 * n_groups));
    if (S->h_gparams) { HIP_TRY(hipHostFree(S->h_gparams)); S->h_gparams = nullptr; }
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S->h_gparams), (pw + 2) * n_groups * sizeof(double), hipHostMallocDefault));
    return VIPRS_OK;
}

static int group_rows_check(const viprs_state* S, int n, const double* rows, int width) {
    if (!S || !rows) return fail(VIPRS_EINVAL, "null argument");
    if (S->n_groups == 0) return fail(VIPRS_EINVAL, "viprs_state_set_groups has not been called");
    if (n < 0 || n > S->n_groups) return fail(VIPRS_EINVAL, "bad group count");
    for (int i = 0; i < n; ++i) {
        const double g = rows[(size_t)width * i];
        if (g < 0 || g >= S->n_groups || g != floor(g)) return fail(VIPRS_EINVAL, "group index out of range");
    }
    return VIPRS_OK;
}

int viprs_state_prep_groups(viprs_state* S, int n, const double* params) {
    int rc = group_rows_check(S, n, params, 6);
    if (rc != VIPRS_OK) return rc;
    if (S->model_kind != VIPRS_MODEL_SPIKE_SLAB) return fail(VIPRS_EINVAL, "not a spike-and-slab state (viprs_state_prep_mixture_groups)");
    viprs_plan* P = S->plan;
    if (P->m == 0 || n == 0) return VIPRS_OK;
    if (!S->d_n.p) return fail(VIPRS_EINVAL, "viprs_state_set_n_per_snp has not been called");
    HIP_TRY(hipSetDevice(P->device));
    if (!S->ev_prep) HIP_TRY(hipEventCreateWithFlags(&S->ev_prep, hipEventDisableTiming));
    else HIP_TRY(hipEventSynchronize(S->ev_prep));            // the previous launch has read its parameters
    memcpy(S->h_gparams, params, (size_t)6 * n * sizeof(double));
    HIP_TRY(hipMemcpyAsync(S->d_group_prep.p, S->h_gparams, (size_t)6 * n * sizeof(double), hipMemcpyHostToDevice, P->stream));
    int64_t longest = 0;
    for (int i = 0; i < n; ++i) {
        const int g = (int)params[6 * i];
        longest = std::max(longest, S->group_start[(size_t)g + 1] - S->group_start[(size_t)g]);
    }
    const dim3 grid((unsigned)std::max<int64_t>(1, (longest + 255) / 256), (unsigned)n);
    if (S->float_dtype == VIPRS_F32)
        prep_groups_kernel<float><<<grid, 256, 0, P->stream>>>(S->d_n.p, S->d_group_start.p, S->d_group_prep.p,
                                                               (float*)S->f[VIPRS_FIELD_MU_MULT].p, (float*)S->f[VIPRS_FIELD_U_LOGS].p,
                                                               (float*)S->f[VIPRS_FIELD_SQRT_HALF_VAR_TAU].p, S->d_var_tau.p);
    else
        prep_groups_kernel<double><<<grid, 256, 0, P->stream>>>(S->d_n.p, S->d_group_start.p, S->d_group_prep.p,
                                                                (double*)S->f[VIPRS_FIELD_MU_MULT].p, (double*)S->f[VIPRS_FIELD_U_LOGS].p,
                                                                (double*)S->f[VIPRS_FIELD_SQRT_HALF_VAR_TAU].p, S->d_var_tau.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(S->ev_prep, P->stream));
    return VIPRS_OK;
}

}  // extern "C"

template <typename T>
static int sums_groups_enqueue(viprs_state* S, int n) {
    viprs_plan* P = S->plan;
    const int nb = S->group_max_nb;
    const size_t need = (size_t)nb * kNSums * n;
    if (S->d_partials.n < need) HIP_TRY(S->d_partials.alloc(need));
    if (S->d_sums.n < (size_t)kNSums * S->n_groups) HIP_TRY(S->d_sums.alloc((size_t)kNSums * S->n_groups));
    const size_t hcap = (size_t)kNSums * S->n_groups + 1;
    if (S->h_sums_cap < hcap) {
        if (S->h_sums) HIP_TRY(hipHostFree(S->h_sums));
        S->h_sums = nullptr;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S->h_sums), hcap * sizeof(double), hipHostMallocDefault));
        S->h_sums_cap = hcap;
    }
    sums_groups_kernel<T><<<dim3(nb, n), kSumsBlock, 0, P->stream>>>(
        S->d_group_start.p, S->d_group_sumrows.p, (const T*)S->f[VIPRS_FIELD_VAR_GAMMA].p, (const T*)S->f[VIPRS_FIELD_VAR_MU].p,
        (const T*)S->f[VIPRS_FIELD_ETA].p, (const T*)S->f[VIPRS_FIELD_Q].p, (const T*)S->f[VIPRS_FIELD_ETA_DIFF].p,
        (const T*)S->f[VIPRS_FIELD_STD_BETA].p, S->d_var_tau.p, S->d_partials.p);
    HIP_TRY(hipGetLastError());
    sums_final_groups_kernel<<<n, 64 * kNSums, 0, P->stream>>>(S->d_partials.p, nb, S->d_group_start.p, S->d_group_sumrows.p, S->d_sums.p);
    HIP_TRY(hipGetLastError());
    if (S->comm) {
        const int rc = comm_reduce_on_stream(S->comm, S->d_sums.p, kNSums * n, kNSums, P->stream);
        if (rc != VIPRS_OK) return rc;
    }
    HIP_TRY(hipMemcpyAsync(S->h_sums, S->d_sums.p, (size_t)kNSums * n * sizeof(double), hipMemcpyDeviceToHost, P->stream));
    HIP_TRY(hipMemcpyAsync(S->h_sums + (size_t)kNSums * n, P->d_error.p, sizeof(int32_t), hipMemcpyDeviceToHost, P->stream));
    S->sums_cols = n;
    S->sums_pending = true;
    return VIPRS_OK;
}

extern "C" {

int viprs_state_sums_groups_begin(viprs_state* S, int n, const double* rows) {
    int rc = group_rows_check(S, n, rows, 2);
    if (rc != VIPRS_OK) return rc;
    if (S->model_kind != VIPRS_MODEL_SPIKE_SLAB) return fail(VIPRS_EINVAL, "not a spike-and-slab state (viprs_state_sums_mixture_groups_begin)");
    viprs_plan* P = S->plan;
    S->sums_cols = n;
    if (P->m == 0 && n > 0 && S->comm) return sums_enqueue_empty(S, kNSums * n, kNSums);
    if (P->m == 0 || n == 0) { S->sums_pending = false; S->sums_empty = true; return VIPRS_OK; }
    S->sums_empty = false;
    if (!S->d_var_tau.p) return fail(VIPRS_EINVAL, "viprs_state_set_n_per_snp / viprs_state_prep_groups have not been called");
    HIP_TRY(hipSetDevice(P->device));
    // (own half of the pinned staging: the previous reduction that read it has been collected)
    double* h = S->h_gparams + group_prep_width(S) * S->n_groups;
    memcpy(h, rows, (size_t)2 * n * sizeof(double));
    HIP_TRY(hipMemcpyAsync(S->d_group_sumrows.p, h, (size_t)2 * n * sizeof(double), hipMemcpyHostToDevice, P->stream));
    return S->float_dtype == VIPRS_F32 ? sums_groups_enqueue<float>(S, n) : sums_groups_enqueue<double>(S, n);
}

int viprs_state_sums_groups_end(viprs_state* S, double* out) {
    if (S && S->n_groups == 0) return fail(VIPRS_EINVAL, "viprs_state_set_groups has not been called");
    return viprs_state_sums_columns_end(S, out);          // same landing buffer and bookkeeping: sums_cols rows of VIPRS_N_SUMS
}

// ---- the same for a mixture state (VIPRSMix per chromosome) ----
int viprs_state_prep_mixture_groups(viprs_state* S, int n, const double* params) {
    if (!S) return fail(VIPRS_EINVAL, "null argument");
    if (S->model_kind != VIPRS_MODEL_MIXTURE) return fail(VIPRS_EINVAL, "not a mixture state");
    const int K = S->width, W = 4 + 3 * K;
    int rc = group_rows_check(S, n, params, W);
    if (rc != VIPRS_OK) return rc;
    viprs_plan* P = S->plan;
    if (P->m == 0 || n == 0) return VIPRS_OK;
    if (!S->d_n.p) return fail(VIPRS_EINVAL, "viprs_state_set_n_per_snp has not been called");
    HIP_TRY(hipSetDevice(P->device));
    if (S->d_var_tau.n < (size_t)P->m * K) {
        HIP_TRY(hipStreamSynchronize(P->stream));
        HIP_TRY(S->d_var_tau.alloc((size_t)P->m * K));
    }
    if (!S->ev_prep) HIP_TRY(hipEventCreateWithFlags(&S->ev_prep, hipEventDisableTiming));
    else HIP_TRY(hipEventSynchronize(S->ev_prep));            // the previous launch has read its parameters
    memcpy(S->h_gparams, params, (size_t)W * n * sizeof(double));
    HIP_TRY(hipMemcpyAsync(S->d_group_prep.p, S->h_gparams, (size_t)W * n * sizeof(double), hipMemcpyHostToDevice, P->stream));
    int64_t longest = 0;
    for (int i = 0; i < n; ++i) {
        const int g = (int)params[(size_t)W * i];
        longest = std::max(longest, S->group_start[(size_t)g + 1] - S->group_start[(size_t)g]);
    }
    const dim3 grid((unsigned)std::max<int64_t>(1, (longest + 255) / 256), (unsigned)n);
    if (S->float_dtype == VIPRS_F32)
        prep_mixture_groups_kernel<float><<<grid, 256, 0, P->stream>>>(
            S->d_n.p, S->d_group_start.p, K, S->d_group_prep.p, (float*)S->f[VIPRS_FIELD_MU_MULT].p,
            (float*)S->f[VIPRS_FIELD_U_LOGS].p, (float*)S->f[VIPRS_FIELD_SQRT_HALF_VAR_TAU].p,
            (float*)S->f[VIPRS_FIELD_LOG_NULL_PI].p, S->d_var_tau.p);
    else
        prep_mixture_groups_kernel<double><<<grid, 256, 0, P->stream>>>(
            S->d_n.p, S->d_group_start.p, K, S->d_group_prep.p, (double*)S->f[VIPRS_FIELD_MU_MULT].p,
            (double*)S->f[VIPRS_FIELD_U_LOGS].p, (double*)S->f[VIPRS_FIELD_SQRT_HALF_VAR_TAU].p,
            (double*)S->f[VIPRS_FIELD_LOG_NULL_PI].p, S->d_var_tau.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(S->ev_prep, P->stream));
    return VIPRS_OK;
}

}  // extern "C"

template <typename T>
static int sums_mixture_groups_enqueue(viprs_state* S, int n) {
    viprs_plan* P = S->plan;
    const int K = S->width, N = kMixSums(K), nb = S->group_max_nb;
    const size_t need = (size_t)nb * N * n;
    if (S->d_partials.n < need) HIP_TRY(S->d_partials.alloc(need));
    if (S->d_sums.n < (size_t)N * S->n_groups) HIP_TRY(S->d_sums.alloc((size_t)N * S->n_groups));
    const size_t hcap = (size_t)N * S->n_groups + 1;
    if (S->h_sums_cap < hcap) {
        if (S->h_sums) HIP_TRY(hipHostFree(S->h_sums));
        S->h_sums = nullptr;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S->h_sums), hcap * sizeof(double), hipHostMallocDefault));
        S->h_sums_cap = hcap;
    }
    sums_mixture_groups_kernel<T><<<dim3(nb, n), kSumsBlock, 0, P->stream>>>(
        S->d_group_start.p, S->d_group_sumrows.p, K, (const T*)S->f[VIPRS_FIELD_VAR_GAMMA].p, (const T*)S->f[VIPRS_FIELD_VAR_MU].p,
        (const T*)S->f[VIPRS_FIELD_ETA].p, (const T*)S->f[VIPRS_FIELD_Q].p, (const T*)S->f[VIPRS_FIELD_ETA_DIFF].p,
        (const T*)S->f[VIPRS_FIELD_STD_BETA].p, S->d_var_tau.p, S->d_log_var_tau0.p, S->d_partials.p);
    HIP_TRY(hipGetLastError());
    sums_final_generic_groups_kernel<<<dim3(N, n), 64, 0, P->stream>>>(S->d_partials.p, nb, N, S->d_group_start.p,
                                                                      S->d_group_sumrows.p, S->d_sums.p);
    HIP_TRY(hipGetLastError());
    if (S->comm) {
        const int rc = comm_reduce_on_stream(S->comm, S->d_sums.p, N * n, N, P->stream);
        if (rc != VIPRS_OK) return rc;
    }
    HIP_TRY(hipMemcpyAsync(S->h_sums, S->d_sums.p, (size_t)N * n * sizeof(double), hipMemcpyDeviceToHost, P->stream));
    HIP_TRY(hipMemcpyAsync(S->h_sums + (size_t)N * n, P->d_error.p, sizeof(int32_t), hipMemcpyDeviceToHost, P->stream));
    S->sums_cols = n;
    S->sums_pending = true;
    return VIPRS_OK;
}

extern "C" {

int viprs_state_sums_mixture_groups_begin(viprs_state* S, int n, const double* rows) {
    if (!S) return fail(VIPRS_EINVAL, "null argument");
    if (S->model_kind != VIPRS_MODEL_MIXTURE) return fail(VIPRS_EINVAL, "not a mixture state");
    int rc = group_rows_check(S, n, rows, 2);
    if (rc != VIPRS_OK) return rc;
    viprs_plan* P = S->plan;
    const int N = kMixSums(S->width);
    S->sums_cols = n;
    if (P->m == 0 && n > 0 && S->comm) return sums_enqueue_empty(S, N * n, N);
    if (P->m == 0 || n == 0) { S->sums_pending = false; S->sums_empty = true; return VIPRS_OK; }
    S->sums_empty = false;
    if (S->d_var_tau.n < (size_t)P->m * S->width || !S->d_log_var_tau0.p)
        return fail(VIPRS_EINVAL, "viprs_state_prep_mixture_groups / viprs_state_set_log_var_tau have not been called");
    HIP_TRY(hipSetDevice(P->device));
    double* h = S->h_gparams + group_prep_width(S) * S->n_groups;
    memcpy(h, rows, (size_t)2 * n * sizeof(double));
    HIP_TRY(hipMemcpyAsync(S->d_group_sumrows.p, h, (size_t)2 * n * sizeof(double), hipMemcpyHostToDevice, P->stream));
    return S->float_dtype == VIPRS_F32 ? sums_mixture_groups_enqueue<float>(S, n) : sums_mixture_groups_enqueue<double>(S, n);
}

int viprs_state_sums_mixture_groups_end(viprs_state* S, double* out) {
    if (!S || !out) return fail(VIPRS_EINVAL, "null argument");
    if (S->model_kind != VIPRS_MODEL_MIXTURE) return fail(VIPRS_EINVAL, "not a mixture state");
    if (S->n_groups == 0) return fail(VIPRS_EINVAL, "viprs_state_set_groups has not been called");
    const size_t total = (size_t)kMixSums(S->width) * S->sums_cols;
    if (S->sums_empty) {
        for (size_t k = 0; k < total; ++k) out[k] = 0.0;
        return VIPRS_OK;
    }
    if (!S->sums_pending) return fail(VIPRS_EINVAL, "no device sums in flight (viprs_state_sums_mixture_groups_begin)");
    viprs_plan* P = S->plan;
    HIP_TRY(hipSetDevice(P->device));
    HIP_TRY(hipStreamSynchronize(P->stream));
    S->sums_pending = false;
    for (size_t k = 0; k < total; ++k) out[k] = S->h_sums[k];
    int32_t e = 0;
    memcpy(&e, S->h_sums + total, sizeof(e));
    return e != 0 ? check_device_error(P) : VIPRS_OK;
}

}  // extern "C"
