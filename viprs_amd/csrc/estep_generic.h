// Generic (any window structure) spike-and-slab E-step kernel: one workgroup walks one LD
// component SNP by SNP, exactly as e_step<T,U,I> does (e_step.hpp:387-433), with the axpy over
// the row window (e_step.hpp:421 -> :172-174) spread over the workgroup's lanes.  It is the
// fallback for ragged / banded components and for (T, U) combinations the panel kernels do not
// specialise; dense LDetect-style blocks take the panel kernels (estep_panel.h).
//
// q and eta_diff of the component live in LDS when they fit, in global memory otherwise.
#pragma once
#include "device_math.h"
#include "kernels_common.h"

namespace viprs {

constexpr int kGenericThreads = 256;

template <typename T, typename U, bool EXACT, bool IN_LDS>
__global__ __launch_bounds__(kGenericThreads) void estep_generic_kernel(EStepArgs<T> A) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ int s_blk;
    ExpTab tab;
    tab.init();
    const int tid = threadIdx.x;
    const U* __restrict__ ld = static_cast<const U*>(A.ld_rows);
    const T eps = Eps<T>::value;
    unsigned long long my_skipped = 0;

    for (;;) {
        if (tid == 0) s_blk = atomicAdd(A.counter, 1);
        __syncthreads();
        const int blk = s_blk;
        __syncthreads();
        if (blk >= A.n_blocks) break;
        const BlockDesc bd = A.blocks[blk];
        const int64_t s0 = bd.start;
        const int n = bd.size;

        T* qv;
        T* edv;
        if (IN_LDS) {
            qv = reinterpret_cast<T*>(smem_raw);
            edv = qv + n;
            for (int i = tid; i < n; i += kGenericThreads) qv[i] = A.q[s0 + i];
        } else {
            qv = A.q + s0;
            edv = A.eta_diff + s0;
        }
        __syncthreads();

        for (int jj = 0; jj < n; ++jj) {
            const int64_t j = s0 + jj;
            const int64_t ls = A.rowstart[j];
            const int len = A.rowlen[j];
            const int wstart = A.lb[j] - (int)s0;          // window start, component-local
            const T qj = qv[jj];
            T mu, gamma, d;
            snp_update<EXACT, kLookupPerLane>(A.mu_mult[j], A.std_beta[j], A.shvt[j], A.u_logs[j], A.eta[j], qj,
                                    tab, mu, gamma, d);
            const bool skip = abs_t<T>(d) < eps;           // e_step.hpp:410
            if (!skip) {
                const T a = A.dq * d;
                const bool has_j = (!A.low_memory) && jj >= wstart && jj < wstart + len;
                for (int i = tid; i < len; i += kGenericThreads) {
                    T v = fma_t<T>(static_cast<T>(ld[ls + i]), a, qv[wstart + i]);   // :421
                    if (has_j && wstart + i == jj) v -= d;                            // :427
                    qv[wstart + i] = v;
                }
                if (tid == 0) {
                    if (!A.low_memory && !has_j) qv[jj] = qv[jj] - d;                 // :427 (j outside its window)
                    A.var_mu[j] = mu;                                                 // :416-418
                    A.var_gamma[j] = gamma;
                    A.eta[j] = A.eta[j] + d;                                          // :431
                    edv[jj] = d;
                }
            } else if (tid == 0) {
                edv[jj] = (T)0;                                                       // :412
                ++my_skipped;
            }
            __syncthreads();
        }

        if (A.low_memory) {
            // update_q_factor (e_step.hpp:331-337): q[j] += dq * dot(eta_diff[win(j)], row(j)),
            // the dot being a serial fma chain from 0 in index order (:100-102).
            for (int jj = tid; jj < n; jj += kGenericThreads) {
                const int64_t j = s0 + jj;
                const int64_t ls = A.rowstart[j];
                const int len = A.rowlen[j];
                const int wstart = A.lb[j] - (int)s0;
                T s = (T)0;
                for (int i = 0; i < len; ++i) s = fma_t<T>(static_cast<T>(ld[ls + i]), edv[wstart + i], s);
                qv[jj] += A.dq * s;
            }
            __syncthreads();
        }
        if (IN_LDS) {
            for (int i = tid; i < n; i += kGenericThreads) {
                A.q[s0 + i] = qv[i];
                A.eta_diff[s0 + i] = edv[i];
            }
        }
        __syncthreads();
    }
    if (tid == 0 && my_skipped) atomicAdd(A.skipped, my_skipped);
}

// ---------------------------------------------------------------------------------------------
// Sparse-mixture E-step (e_step_mixture, e_step.hpp:447-551): same walk, K posterior means, a
// (K+1)-way softmax (e_step.hpp:222-241) per SNP, no skip branch.  (m, K) arrays are C-ordered.
// Every thread evaluates the K-loop redundantly (K is small); the axpy is spread over the lanes.
// ---------------------------------------------------------------------------------------------

template <typename T> __device__ __forceinline__ T exp_nonpos(T x, const ExpTab& tab);
template <> __device__ __forceinline__ float exp_nonpos<float>(float x, const ExpTab& tab) {
    return expf_glibc_nonpos<kLookupPerLane>(x, tab);
}
template <> __device__ __forceinline__ double exp_nonpos<double>(double x, const ExpTab&) { return exp_glibc_f64_nonpos(x); }

template <typename T, typename U, bool IN_LDS>
__global__ __launch_bounds__(kGenericThreads) void estep_mixture_generic_kernel(EStepArgs<T> A) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ int s_blk;
    ExpTab tab;
    tab.init();
    const int tid = threadIdx.x;
    const int K = A.width;
    const U* __restrict__ ld = static_cast<const U*>(A.ld_rows);

    for (;;) {
        if (tid == 0) s_blk = atomicAdd(A.counter, 1);
        __syncthreads();
        const int blk = s_blk;
        __syncthreads();
        if (blk >= A.n_blocks) break;
        const BlockDesc bd = A.blocks[blk];
        const int64_t s0 = bd.start;
        const int n = bd.size;
        T* qv;
        T* edv;
        if (IN_LDS) {
            qv = reinterpret_cast<T*>(smem_raw);
            edv = qv + n;
            for (int i = tid; i < n; i += kGenericThreads) qv[i] = A.q[s0 + i];
        } else {
            qv = A.q + s0;
            edv = A.eta_diff + s0;
        }
        __syncthreads();

        for (int jj = 0; jj < n; ++jj) {
            const int64_t j = s0 + jj;
            const int64_t ls = A.rowstart[j];
            const int len = A.rowlen[j];
            const int wstart = A.lb[j] - (int)s0;
            const T r = A.std_beta[j] - qv[jj];                                   // :505
            T u[kMaxMixtureK + 1], mu[kMaxMixtureK];
            T mx;
#pragma unroll 1
            for (int k = 0; k < K; ++k) {                                         // :507-512
                const int64_t idx = j * K + k;
                mu[k] = A.mu_mult[idx] * r;
                const T t = A.shvt[idx] * mu[k];
                u[k] = fma_t<T>(t, t, A.u_logs[idx]);
            }
            u[K] = A.log_null_pi[j];                                              // :515
            mx = u[0];
            for (int i = 1; i <= K; ++i) if (mx < u[i]) mx = u[i];                // c_max :58-71
            T ssum = (T)0;
            for (int i = 0; i <= K; ++i) { u[i] = exp_nonpos<T>(u[i] - mx, tab); ssum += u[i]; }   // :233-236
            T d = -A.eta[j];                                                      // :519
            for (int k = 0; k < K; ++k) {
                const T g = u[k] / ssum;                                          // :239
                d = fma_t<T>(g, mu[k], d);                                        // :523
                if (tid == 0) { A.var_gamma[j * K + k] = g; A.var_mu[j * K + k] = mu[k]; }
            }
            const T a = A.dq * d;
            const bool has_j = (!A.low_memory) && jj >= wstart && jj < wstart + len;
            for (int i = tid; i < len; i += kGenericThreads) {
                T v = fma_t<T>(static_cast<T>(ld[ls + i]), a, qv[wstart + i]);    // :527
                if (has_j && wstart + i == jj) v -= d;                            // :533
                qv[wstart + i] = v;
            }
            if (tid == 0) {
                if (!A.low_memory && !has_j) qv[jj] = qv[jj] - d;
                A.eta[j] = A.eta[j] + d;                                          // :536
                edv[jj] = d;
            }
            __syncthreads();
        }
        if (A.low_memory) {                                                        // :543-549
            for (int jj = tid; jj < n; jj += kGenericThreads) {
                const int64_t j = s0 + jj;
                const int64_t ls = A.rowstart[j];
                const int len = A.rowlen[j];
                const int wstart = A.lb[j] - (int)s0;
                T s = (T)0;
                for (int i = 0; i < len; ++i) s = fma_t<T>(static_cast<T>(ld[ls + i]), edv[wstart + i], s);
                qv[jj] += A.dq * s;
            }
            __syncthreads();
        }
        if (IN_LDS) {
            for (int i = tid; i < n; i += kGenericThreads) {
                A.q[s0 + i] = qv[i];
                A.eta_diff[s0 + i] = edv[i];
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Grid of spike-and-slab models (e_step_grid, e_step.hpp:555-647): per SNP the active models are
// independent, so thread t carries model active[t] through the scalar update and the workgroup
// then spreads the n_active axpys (one per model column of q) over its lanes.  (m, G) arrays are
// column-major; no skip branch; `half_var_tau` (not its square root) and no fma in the logit
// (e_step.hpp:616).
// ---------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T sigmoid_t(T x, const ExpTab& tab);
template <> __device__ __forceinline__ float sigmoid_t<float>(float x, const ExpTab& tab) {
    return sigmoid_exact<kLookupPerLane>(x, tab);
}
template <> __device__ __forceinline__ double sigmoid_t<double>(double x, const ExpTab&) { return sigmoid_f64(x); }

template <typename T, typename U>
__global__ __launch_bounds__(kGenericThreads) void estep_grid_generic_kernel(EStepArgs<T> A) {
    __shared__ int s_blk;
    __shared__ T s_a[kGenericThreads];
    __shared__ T s_d[kGenericThreads];
    ExpTab tab;
    tab.init();
    const int tid = threadIdx.x;
    const int64_t m = A.m;
    const int na = A.n_active;
    const U* __restrict__ ld = static_cast<const U*>(A.ld_rows);

    for (;;) {
        if (tid == 0) s_blk = atomicAdd(A.counter, 1);
        __syncthreads();
        const int blk = s_blk;
        __syncthreads();
        if (blk >= A.n_blocks) break;
        const BlockDesc bd = A.blocks[blk];
        const int64_t s0 = bd.start;
        const int n = bd.size;

        for (int jj = 0; jj < n; ++jj) {
            const int64_t j = s0 + jj;
            const int64_t ls = A.rowstart[j];
            const int len = A.rowlen[j];
            const int64_t wstart = A.lb[j];
            const bool has_j = (!A.low_memory) && j >= wstart && j < wstart + len;
            // models are processed in chunks of the workgroup size (n_active may exceed it)
            for (int a0 = 0; a0 < na; a0 += kGenericThreads) {
                const int t = a0 + tid;
                T av = (T)0, dv = (T)0;
                int64_t g = 0;
                const bool has_model = t < na;
                {
                    g = has_model ? A.active[t] : A.active[0];
                    const int64_t idx = g * m + j;
                    const T mu = A.mu_mult[idx] * (A.std_beta[j] - A.q[idx]);              // :613
                    const T uj = A.u_logs[idx] + A.shvt[idx] * mu * mu;                      // :616
                    const T gam = sigmoid_t<T>(uj, tab);                                     // :617
                    const T d = gam * mu - A.eta[idx];                                       // :620
                    if (has_model) {
                        A.var_mu[idx] = mu;
                        A.var_gamma[idx] = gam;
                        A.eta_diff[idx] = d;
                        A.eta[idx] = A.eta[idx] + d;                                         // :633
                        if (!A.low_memory && !has_j) A.q[idx] = A.q[idx] - d;                // :629 (j outside its window)
                        av = A.dq * d;
                        dv = d;
                    }
                }
                s_a[tid] = av;
                s_d[tid] = dv;
                __syncthreads();
                const int nm = min(kGenericThreads, na - a0);
                const int64_t total = (int64_t)nm * len;
                for (int64_t w = tid; w < total; w += kGenericThreads) {
                    const int tm = (int)(w / len);
                    const int i = (int)(w - (int64_t)tm * len);
                    const int64_t gg = A.active[a0 + tm];
                    const int64_t qi = gg * m + wstart + i;
                    T v = fma_t<T>(static_cast<T>(ld[ls + i]), s_a[tm], A.q[qi]);             // :623
                    if (has_j && wstart + i == j) v -= s_d[tm];                              // :629
                    A.q[qi] = v;
                }
                __syncthreads();
            }
        }
        if (A.low_memory) {                                                                   // :637-645
            const int64_t total = (int64_t)n * na;
            for (int64_t w = tid; w < total; w += kGenericThreads) {
                const int jj = (int)(w % n);
                const int64_t gg = A.active[(int)(w / n)];
                const int64_t j = s0 + jj;
                const int64_t ls = A.rowstart[j];
                const int len = A.rowlen[j];
                T s = (T)0;
                for (int i = 0; i < len; ++i)
                    s = fma_t<T>(static_cast<T>(ld[ls + i]), A.eta_diff[gg * m + A.lb[j] + i], s);
                A.q[gg * m + j] += A.dq * s;
            }
        }
        __syncthreads();
    }
}

}  // namespace viprs
