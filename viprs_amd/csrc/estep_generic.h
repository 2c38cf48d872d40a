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
    const U* __restrict__ ld = static_cast<const U*>(A.ld_raw);
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
            const int64_t ls = A.ip[j];
            const int len = (int)(A.ip[j + 1] - ls);
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
                const int64_t ls = A.ip[j];
                const int len = (int)(A.ip[j + 1] - ls);
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

}  // namespace viprs
