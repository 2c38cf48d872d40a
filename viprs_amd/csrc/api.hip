// C-ABI implementation (include/viprs_hip.h): plan (device-resident LD + block schedule),
// device-resident variational state, kernel dispatch, measurement hooks.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <array>
#include <mutex>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/viprs_hip.h"
#include "estep_generic.h"
#include "estep_grid_mfma.h"
#include "estep_band.h"
#include "estep_panel.h"
#include "planner.h"

using namespace viprs;

// ------------------------------------------------------------------------------------------------
namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return fail(VIPRS_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)

size_t ld_elem_size(int ld_dtype) {
    switch (ld_dtype) {
        case VIPRS_LD_I8: return 1;
        case VIPRS_LD_I16: return 2;
        case VIPRS_LD_I32: return 4;
        case VIPRS_LD_I64: return 8;
        case VIPRS_LD_F32: return 4;
        case VIPRS_LD_F64: return 8;
        default: return 0;
    }
}
size_t float_size(int t) { return t == VIPRS_F32 ? 4 : (t == VIPRS_F64 ? 8 : 0); }

// workgroup-size classes of the panel kernel (waves per workgroup; wave 0 is the chain)
// Every class uses 4-wave workgroups (1 chain + 3 updaters) so that a team member needs exactly
// the same CU resources as a small-block workgroup; larger blocks get more CUs, not bigger groups.
static int kLargeBlock = 2304, kMediumBlock = 1280;
constexpr int kClassWaves[3] = {4, 4, 4};
static int kClassTeam[3] = {8, 2, 1};       // workgroups (CUs) sharing one block of the class (0/1: teams)
// mixture: the chain step is ~4x the spike-and-slab one, every team member replicates it -- smaller teams
static int kClassTeamMix[3] = {4, 1, 1};
static bool g_team_env = false;             // VIPRS_TEAM0/1 given: they apply to every model
constexpr int kEpiWaves = 4;

template <typename V> struct DevBuf {
    V* p = nullptr;
    size_t n = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) {
        if (p) { (void)hipFree(p); p = nullptr; }
        n = count;
        if (count == 0) return hipSuccess;
        return hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(V));
    }
};

// repack one dense block from the caller's row-concatenated layout into the padded row-major
// device layout (pure data movement; values are not touched)
template <typename U>
__global__ void repack_dense_kernel(const U* __restrict__ src, const int64_t* __restrict__ ip, U* __restrict__ dst,
                                    const BlockDesc* __restrict__ blocks, int upper) {
    const BlockDesc bd = blocks[blockIdx.y];
    const int b = bd.size;
    for (int r = blockIdx.x; r < b; r += gridDim.x) {
        const int64_t rs = ip[bd.start + r];
        U* __restrict__ drow = dst + bd.ld_off + (int64_t)r * bd.stride;
        if (upper) {
            for (int c = r + 1 + threadIdx.x; c < b; c += blockDim.x) drow[c] = src[rs + (c - r - 1)];
        } else {
            for (int c = threadIdx.x; c < b; c += blockDim.x) drow[c] = src[rs + c];
        }
    }
}

// symmetric expansion on the device: row j of the symmetric store = [mirror of the upper rows that
// reach j | diagonal | row j of the upper store].  Replaces the host-side symmetric load of
// VIPRS.py:167-172 (`ld_mat.load(return_symmetric=True)`): the compact store crosses PCIe once and
// the symmetric copy never exists in host memory.  Pure data movement, values are not touched.
template <typename U>
__global__ void expand_symmetric_kernel(const U* __restrict__ up, const int64_t* __restrict__ ipu,
                                        const int32_t* __restrict__ lb, const int64_t* __restrict__ ip,
                                        U* __restrict__ out, int64_t m, U diag) {
    for (int64_t j = blockIdx.x; j < m; j += gridDim.x) {
        const int64_t o = ip[j];
        const int len = (int)(ip[j + 1] - o);
        const int64_t c0 = lb[j];
        const int64_t uj = ipu[j];
        for (int p = threadIdx.x; p < len; p += blockDim.x) {
            const int64_t c = c0 + p;
            U v = diag;
            if (c > j) v = up[uj + (c - j - 1)];
            else if (c < j) v = up[ipu[c] + (j - c - 1)];
            out[o + p] = v;
        }
    }
}

template <typename U>
static hipError_t launch_expand(const void* up, const int64_t* ipu, const int32_t* lb, const int64_t* ip, void* out,
                                int64_t m, double diag) {
    const unsigned grid = (unsigned)std::min<int64_t>(m, 1 << 16);
    expand_symmetric_kernel<U><<<grid, 256>>>((const U*)up, ipu, lb, ip, (U*)out, m, (U)diag);
    return hipGetLastError();
}

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
    var_tau_out[i] = vt;
    mu_mult[i] = (T)(n[i] / (vt * sigma_eps));
    u_logs[i] = (T)(logit_pi + 0.5 * (log_tau_beta - log(vt)));
    shvt[i] = half_not_sqrt ? (T)(0.5 * vt) : (T)sqrt(0.5 * vt);     // e_step_grid takes var_tau / 2 (e_step.hpp:616)
}

// the same for several columns of a grid state in one launch: blockIdx.y picks a row of `params`
// (column, logit_pi, log_tau_beta, sigma_eps, tau_beta, one_plus_lambda)
template <typename T>
__global__ void prep_columns_kernel(const double* __restrict__ n, int64_t m, const double* __restrict__ params,
                                    T* __restrict__ mu_mult, T* __restrict__ u_logs, T* __restrict__ shvt,
                                    double* __restrict__ var_tau_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double* __restrict__ p = params + 6 * (int64_t)blockIdx.y;
    const int64_t off = (int64_t)p[0] * m;
    const double logit_pi = p[1], log_tau_beta = p[2], sigma_eps = p[3], tau_beta = p[4], one_plus_lambda = p[5];
    const double vt = n[i] * one_plus_lambda / sigma_eps + tau_beta;
    var_tau_out[off + i] = vt;
    mu_mult[off + i] = (T)(n[i] / (vt * sigma_eps));
    u_logs[off + i] = (T)(logit_pi + 0.5 * (log_tau_beta - log(vt)));
    shvt[off + i] = (T)(0.5 * vt);                                    // e_step_grid takes var_tau / 2 (e_step.hpp:616)
}

constexpr int kSumsBlock = 256;
constexpr int kNSums = VIPRS_N_SUMS;

// stage 1: per-workgroup partial sums (fixed assignment of elements to threads, tree reduction in
// LDS: deterministic); stage 2 adds the partials in index order
template <typename T>
__global__ __launch_bounds__(kSumsBlock) void sums_kernel(int64_t m, const T* __restrict__ gam, const T* __restrict__ mu,
                                                          const T* __restrict__ eta, const T* __restrict__ q,
                                                          const T* __restrict__ ed, const T* __restrict__ beta,
                                                          const double* __restrict__ var_tau, double one_plus_lambda,
                                                          const double* __restrict__ weight, double* __restrict__ partials,
                                                          const double* __restrict__ cols = nullptr) {
    if (cols) {
        // several columns of a grid state in one launch: blockIdx.y picks (column, one_plus_lambda)
        const int64_t off = (int64_t)cols[2 * blockIdx.y] * m;
        one_plus_lambda = cols[2 * blockIdx.y + 1];
        gam += off; mu += off; eta += off; q += off; ed += off; var_tau += off;
        partials += (int64_t)blockIdx.y * gridDim.x * kNSums;
    }
    __shared__ double red[kNSums][kSumsBlock];
    double acc[kNSums];
#pragma unroll
    for (int k = 0; k < kNSums; ++k) acc[k] = 0.0;
    const double lo = 1e-15, hi = 1.0 - 1e-15;       // np.finfo(float64).resolution (VIPRS.py:509)
    for (int64_t i = (int64_t)blockIdx.x * kSumsBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kSumsBlock) {
        const double g = (double)gam[i], mud = (double)mu[i], vt = var_tau[i];
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
    if (threadIdx.x < kNSums) partials[(int64_t)blockIdx.x * kNSums + threadIdx.x] = red[threadIdx.x][0];
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

// VIPRSMix._partial_sums on the device (float64): per-workgroup partials, fixed order
template <typename T>
__global__ __launch_bounds__(kSumsBlock) void sums_mixture_kernel(int64_t m, int K, const T* __restrict__ gam,
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
    for (int64_t i = (int64_t)blockIdx.x * kSumsBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kSumsBlock) {
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
        partials[(int64_t)blockIdx.x * N + nidx] = a;
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

// zeroes the work-queue heads, the skip counter and the team hand-off granules in ONE launch
__global__ void sweep_prologue_kernel(int32_t* counters, int n_counters, unsigned long long* skipped,
                                      unsigned long long* granules, int64_t n_granules) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_counters) counters[i] = 0;
    if (i == 0) *skipped = 0ull;
    for (int64_t k = i; k < n_granules; k += (int64_t)gridDim.x * blockDim.x) granules[k] = 0ull;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
struct viprs_plan {
    int64_t m = 0;
    int64_t nnz = 0;
    int low_memory = 0;
    int ld_dtype = 0;
    int device = 0;
    int n_cu = 0;
    int math_mode = VIPRS_MATH_EXACT;
    hipStream_t stream = nullptr;
    std::vector<Block> blocks;              // SNP order
    std::vector<BlockDesc> dense_h, ragged_h;  // schedule order (descending cost)
    // dense blocks are served by panel kernels of three workgroup sizes (more updater waves =
    // more row loads in flight = a larger share of HBM bandwidth for the larger blocks); class c
    // covers dense_h[class_begin[c] .. class_begin[c+1])
    int class_begin[4] = {0, 0, 0, 0};
    DevBuf<BlockDesc> d_dense, d_ragged;
    hipStream_t class_stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    DevBuf<EpiItem> d_epi;
    DevBuf<unsigned long long> d_granules;  // team hand-off granules (one row of 64 per panel of a team block)
    int64_t n_granule_rows = 0;
    DevBuf<int32_t> d_error;
    DevBuf<int32_t> d_admit;                // admission thresholds of the small-block class
    int admit_grid = 0;
    double admit_factor = 1.5;
    int grid_mfma = -1;                     // batched grid E-step on the matrix cores: 1 always, 0 never (per-(block,
                                            // model) items), -1 when the plan has enough blocks to fill the CUs (VIPRS_GRID_MFMA)
    int64_t n_epi = 0;
    int epi_begin[4] = {0, 0, 0, 0};        // per size class ranges of d_epi
    DevBuf<EpiItem> d_epi_all;              // all items, plan-wide block indices, longest first
    DevBuf<EpiItem> d_low_items;            // symmetric form: (block, 128-column tile) items of the batched grid lower pass
    int64_t n_low_items = 0;
    DevBuf<int32_t> d_lb;
    DevBuf<int64_t> d_ip;
    DevBuf<int32_t> d_rowlen;               // indptr[j+1] - indptr[j]
    DevBuf<int64_t> d_rowstart_dense;       // row starts inside the repacked dense buffer (generic kernels)
    DevBuf<char> d_ld_raw;                  // kept only when ragged blocks exist
    DevBuf<char> d_ld_dense;
    int64_t dense_elems = 0;
    int max_dense = 0, max_ragged = 0;
    int max_band_panels = 0;                // ragged blocks: widest (band_left + band_right + 2), sizes the band kernel's q ring
    DevBuf<int32_t> d_counters;             // [0] dense queue head, [1] ragged queue head
    DevBuf<unsigned long long> d_skipped;
    // HIP-event ring: per sweep {sweep start, sweep end, panel start, panel end}, recorded on
    // the stream the kernels are launched on
    static constexpr int kRing = 256;
    std::vector<hipEvent_t> ev;             // 4 * kRing
    int64_t sweeps = 0;                     // sweeps recorded since the last timing reset
    viprs_state* scratch = nullptr;         // state used by the one-shot host-buffer calls

    ~viprs_plan();
};

struct viprs_state {
    viprs_plan* plan = nullptr;
    int float_dtype = VIPRS_F32;
    int model_kind = VIPRS_MODEL_SPIKE_SLAB;
    int width = 1;
    DevBuf<char> f[VIPRS_FIELD_COUNT];
    DevBuf<int32_t> d_active;               // grid: active model indices of the current call
    DevBuf<char> eta_out, q_out;            // team kernels' in-out staging (see kernels_common.h)
    DevBuf<double> d_n, d_var_tau, d_partials, d_sums;   // device-resident EM iteration
    DevBuf<double> d_weight;                // optional per-SNP weight of sum [0] (several chromosomes in one plan)
    DevBuf<double> d_log_var_tau0;          // mixture: the log var_tau of the initial state (the reference's ELBO never refreshes it)
    DevBuf<double> d_colparams, d_sumcols;  // grid: per-column parameters of the batched prep / of the batched sums
    int sums_cols = 0;                      // columns of the reduction in flight (grid: sums_columns_begin)
    size_t h_sums_cap = 0;
    double* h_sums = nullptr;               // pinned landing buffer of the device sums
    bool sums_pending = false, sums_empty = false;
    hipEvent_t ev_prep = nullptr;           // the last batched prep launch (it reads d_colparams)
    double* h_params = nullptr;             // pinned staging of the batched prep (6 x width) / sums (2 x width) parameters
    ~viprs_state() {
        if (h_sums) (void)hipHostFree(h_sums);
        if (h_params) (void)hipHostFree(h_params);
        if (ev_prep) (void)hipEventDestroy(ev_prep);
    }
    size_t field_elems(int field) const {
        const size_t m = (size_t)plan->m;
        switch (field) {
            case VIPRS_FIELD_STD_BETA: return m;
            case VIPRS_FIELD_LOG_NULL_PI: return model_kind == VIPRS_MODEL_MIXTURE ? m : 0;
            case VIPRS_FIELD_ETA: case VIPRS_FIELD_Q: case VIPRS_FIELD_ETA_DIFF:
                return model_kind == VIPRS_MODEL_GRID ? m * width : m;
            default: return m * width;
        }
    }
};

viprs_plan::~viprs_plan() {
    delete scratch;
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    for (auto& e : ev_join) if (e) (void)hipEventDestroy(e);
    if (class_stream[2]) (void)hipStreamDestroy(class_stream[2]);       // [0], [1]: shared per device, never destroyed
    if (stream) (void)hipStreamDestroy(stream);
}

// ------------------------------------------------------------------------------------------------
static int check_device_error(viprs_plan* P);

// what viprs_plan_create_expanded hands to the common path: the compact upper-triangular store the
// symmetric rows are built from on the device
struct ExpandSource {
    std::vector<int64_t> ip_upper;
    const void* data = nullptr;
    double diag = 1.0;
};

static int widen_indptr(int64_t m, const void* indptr, int indptr_dtype, std::vector<int64_t>& ip64) {
    ip64.assign((size_t)m + 1, 0);
    if (indptr_dtype == VIPRS_IP_I64) {
        if (m > 0) std::memcpy(ip64.data(), indptr, sizeof(int64_t) * ((size_t)m + 1));
    } else if (indptr_dtype == VIPRS_IP_I32) {
        const int32_t* p = static_cast<const int32_t*>(indptr);
        for (int64_t i = 0; i <= m && m > 0; ++i) ip64[(size_t)i] = p[i];
    } else {
        return fail(VIPRS_EINVAL, "bad indptr dtype code");
    }
    return VIPRS_OK;
}

static int plan_create_impl(viprs_plan** out, int64_t m, const int32_t* lb, const std::vector<int64_t>& ip64,
                            const void* ld_data, int ld_dtype, int low_memory, int device, const ExpandSource* ex) {
    const size_t es = ld_elem_size(ld_dtype);
    std::unique_ptr<viprs_plan> P(new viprs_plan());
    P->m = m;
    if (const char* f = getenv("VIPRS_ADMIT_FACTOR")) P->admit_factor = atof(f);
    if (const char* f = getenv("VIPRS_LARGE_BLOCK")) kLargeBlock = atoi(f);
    if (const char* f = getenv("VIPRS_MEDIUM_BLOCK")) kMediumBlock = atoi(f);
    if (const char* f = getenv("VIPRS_GRID_MFMA")) P->grid_mfma = atoi(f);
    if (const char* f = getenv("VIPRS_TEAM0")) { kClassTeam[0] = std::max(1, atoi(f)); g_team_env = true; }
    if (const char* f = getenv("VIPRS_TEAM1")) { kClassTeam[1] = std::max(1, atoi(f)); g_team_env = true; }
    P->low_memory = low_memory != 0;
    P->ld_dtype = ld_dtype;
    P->device = device;
    std::string err;
    int rc = plan_blocks(m, lb, ip64.data(), low_memory != 0, P->blocks, err);
    if (rc != VIPRS_OK) return fail(rc, err);
    P->nnz = m > 0 ? ip64[(size_t)m] : 0;
    if (P->nnz > 0 && !ld_data && !ex) return fail(VIPRS_EINVAL, "ld_data is null");

    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    P->n_cu = prop.multiProcessorCount;
    HIP_TRY(hipStreamCreateWithFlags(&P->stream, hipStreamNonBlocking));
    {   // team classes get the highest stream priority: their workgroups must become co-resident quickly
        int prio_lo = 0, prio_hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        // The two team classes run on streams SHARED by all plans of a device: a team kernel needs all its
        // workgroups resident at once (members spin on each other's hand-offs), so two team kernels of the
        // same class from different plans must never be half-resident together.  The shared streams
        // serialise them; the streams live as long as the process.
        static std::mutex team_mu;
        static std::map<int, std::array<hipStream_t, 2>> team_streams;
        {
            std::lock_guard<std::mutex> lock(team_mu);
            auto it = team_streams.find(device);
            if (it == team_streams.end()) {
                std::array<hipStream_t, 2> st{};
                for (int c = 0; c < 2; ++c) HIP_TRY(hipStreamCreateWithPriority(&st[c], hipStreamNonBlocking, prio_hi));
                it = team_streams.emplace(device, st).first;
            }
            P->class_stream[0] = it->second[0];
            P->class_stream[1] = it->second[1];
        }
        HIP_TRY(hipStreamCreateWithPriority(&P->class_stream[2], hipStreamNonBlocking, prio_lo));
    }
    HIP_TRY(hipEventCreateWithFlags(&P->ev_fork, hipEventDisableTiming));
    for (auto& e : P->ev_join) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    P->ev.assign(4 * viprs_plan::kRing, nullptr);
    for (auto& e : P->ev) HIP_TRY(hipEventCreate(&e));

    // ---- schedule: dense blocks -> panel kernels, everything else -> generic kernel ----------
    // The panel kernels specialise T = float and U in {f32, i8, i16}; other LD dtypes run generic.
    bool panel_ld = (ld_dtype == VIPRS_LD_F32 || ld_dtype == VIPRS_LD_I8 || ld_dtype == VIPRS_LD_I16);
    if (const char* f = getenv("VIPRS_NO_DENSE")) panel_ld = panel_ld && !atoi(f);      // experiments: every block as a windowed component
    int64_t dense_off = 0;
    for (const Block& b : P->blocks) {
        BlockDesc d;
        d.start = (int32_t)b.start;
        d.size = (int32_t)(b.end - b.start);
        d.kind = b.kind;
        d.stride = 0;
        d.ld_off = 0;
        d.gr_off = 0;
        d.band_left = d.band_right = 0;
        // the panel kernels keep q of a whole block in LDS: a dense block beyond that (~29 000 SNPs) is
        // scheduled like a windowed component (band kernel if its ring fits, generic kernel otherwise)
        constexpr int kMaxDenseBlock = (160 * 1024 / 4 - panel_lds_floats(kStrip) - kMixLdsFloats) / kPanel * kPanel;
        const bool dense = panel_ld && (b.kind == VIPRS_BLOCK_DENSE_SYM || b.kind == VIPRS_BLOCK_DENSE_UPPER) &&
                           d.size <= kMaxDenseBlock;
        if (dense) {
            d.stride = (d.size + kPanel - 1) / kPanel * kPanel;
            d.ld_off = dense_off;
            dense_off += (int64_t)d.size * d.stride;
            dense_off = (dense_off + 63) / 64 * 64;
            P->dense_h.push_back(d);
            P->max_dense = std::max(P->max_dense, d.size);
        } else {
            // reach of the row windows around the diagonal, in panels (band kernel, estep_band.h)
            int64_t wl = 0, wr = 0;
            for (int64_t j = b.start; j < b.end; ++j) {
                const int64_t len = ip64[(size_t)j + 1] - ip64[(size_t)j];
                if (len <= 0) continue;
                wl = std::max<int64_t>(wl, j - lb[j]);
                wr = std::max<int64_t>(wr, lb[j] + len - 1 - j);
            }
            d.band_left = (int32_t)(wl / kPanel + 1);
            d.band_right = (int32_t)(wr / kPanel + 1);
            P->max_band_panels = std::max(P->max_band_panels, d.band_left + d.band_right + 2);
            P->ragged_h.push_back(d);
            P->max_ragged = std::max(P->max_ragged, d.size);
        }
    }
    P->dense_elems = dense_off;
    auto by_cost = [](const BlockDesc& a, const BlockDesc& b) {
        return a.size != b.size ? a.size > b.size : a.start < b.start;
    };
    std::sort(P->dense_h.begin(), P->dense_h.end(), by_cost);
    std::sort(P->ragged_h.begin(), P->ragged_h.end(), by_cost);
    {   // descending order: [large | medium | small]
        int i = 0, n = (int)P->dense_h.size();
        P->class_begin[0] = 0;
        while (i < n && P->dense_h[i].size >= kLargeBlock) ++i;
        P->class_begin[1] = i;
        while (i < n && P->dense_h[i].size >= kMediumBlock) ++i;
        P->class_begin[2] = i;
        P->class_begin[3] = n;
        // hand-off granules for the blocks served by teams (classes 0 and 1)
        int64_t rows = 0;
        for (int k = 0; k < P->class_begin[2]; ++k) {
            P->dense_h[(size_t)k].gr_off = rows;
            rows += (P->dense_h[(size_t)k].size + kPanel - 1) / kPanel;
        }
        P->n_granule_rows = rows;
    }

    // ---- upload -------------------------------------------------------------------------------
    HIP_TRY(P->d_counters.alloc(16));
    HIP_TRY(P->d_error.alloc(1));
    HIP_TRY(hipMemset(P->d_error.p, 0, sizeof(int32_t)));
    HIP_TRY(P->d_skipped.alloc(1));
    HIP_TRY(hipMemset(P->d_skipped.p, 0, sizeof(unsigned long long)));
    if (m > 0) {
        HIP_TRY(P->d_lb.alloc((size_t)m));
        HIP_TRY(P->d_ip.alloc((size_t)m + 1));
        HIP_TRY(hipMemcpy(P->d_lb.p, lb, sizeof(int32_t) * (size_t)m, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(P->d_ip.p, ip64.data(), sizeof(int64_t) * ((size_t)m + 1), hipMemcpyHostToDevice));
        std::vector<int32_t> rowlen((size_t)m);
        for (int64_t j = 0; j < m; ++j) rowlen[(size_t)j] = (int32_t)(ip64[(size_t)j + 1] - ip64[(size_t)j]);
        HIP_TRY(P->d_rowlen.alloc((size_t)m));
        HIP_TRY(hipMemcpy(P->d_rowlen.p, rowlen.data(), sizeof(int32_t) * (size_t)m, hipMemcpyHostToDevice));
        if (!P->dense_h.empty()) {
            // element (row r, column c) of a repacked block sits at ld_off + r*stride + c; the row's
            // window starts at column 0 (symmetric) or r + 1 (upper-triangular)
            std::vector<int64_t> rs((size_t)m, 0);
            for (const BlockDesc& d : P->dense_h)
                for (int r = 0; r < d.size; ++r)
                    rs[(size_t)d.start + r] = d.ld_off + (int64_t)r * d.stride + (P->low_memory ? r + 1 : 0);
            HIP_TRY(P->d_rowstart_dense.alloc((size_t)m));
            HIP_TRY(hipMemcpy(P->d_rowstart_dense.p, rs.data(), sizeof(int64_t) * (size_t)m, hipMemcpyHostToDevice));
        }
    }
    if (P->nnz > 0) {
        HIP_TRY(P->d_ld_raw.alloc((size_t)P->nnz * es + 64));      // + slack: the band kernel clamps empty rows to their start
        if (!ex) {
            HIP_TRY(hipMemcpy(P->d_ld_raw.p, ld_data, (size_t)P->nnz * es, hipMemcpyHostToDevice));
        } else {
            // upload the compact store, mirror it into the symmetric rows on the device, drop it
            const size_t nnz_u = (size_t)ex->ip_upper[(size_t)m];
            DevBuf<char> d_up;
            DevBuf<int64_t> d_ipu;
            HIP_TRY(d_up.alloc(std::max<size_t>(nnz_u, 1) * es));
            HIP_TRY(d_ipu.alloc((size_t)m + 1));
            if (nnz_u) HIP_TRY(hipMemcpy(d_up.p, ex->data, nnz_u * es, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(d_ipu.p, ex->ip_upper.data(), sizeof(int64_t) * ((size_t)m + 1), hipMemcpyHostToDevice));
            hipError_t e = hipSuccess;
            switch (ld_dtype) {
                case VIPRS_LD_I8:  e = launch_expand<int8_t>(d_up.p, d_ipu.p, P->d_lb.p, P->d_ip.p, P->d_ld_raw.p, m, ex->diag); break;
                case VIPRS_LD_I16: e = launch_expand<int16_t>(d_up.p, d_ipu.p, P->d_lb.p, P->d_ip.p, P->d_ld_raw.p, m, ex->diag); break;
                case VIPRS_LD_I32: e = launch_expand<int32_t>(d_up.p, d_ipu.p, P->d_lb.p, P->d_ip.p, P->d_ld_raw.p, m, ex->diag); break;
                case VIPRS_LD_I64: e = launch_expand<int64_t>(d_up.p, d_ipu.p, P->d_lb.p, P->d_ip.p, P->d_ld_raw.p, m, ex->diag); break;
                case VIPRS_LD_F32: e = launch_expand<float>(d_up.p, d_ipu.p, P->d_lb.p, P->d_ip.p, P->d_ld_raw.p, m, ex->diag); break;
                default:           e = launch_expand<double>(d_up.p, d_ipu.p, P->d_lb.p, P->d_ip.p, P->d_ld_raw.p, m, ex->diag); break;
            }
            HIP_TRY(e);
            HIP_TRY(hipDeviceSynchronize());
        }
    }
    if (P->n_granule_rows > 0) HIP_TRY(P->d_granules.alloc((size_t)P->n_granule_rows * kPanel));
    if (!P->dense_h.empty()) {
        HIP_TRY(P->d_dense.alloc(P->dense_h.size()));
        HIP_TRY(hipMemcpy(P->d_dense.p, P->dense_h.data(), sizeof(BlockDesc) * P->dense_h.size(), hipMemcpyHostToDevice));
        // + slack so that partial-panel tile loads stay inside the allocation: one strip for the panel
        // kernels, one panel of rows of the widest block for the batched grid kernel (estep_grid_mfma.h)
        int max_stride = 0;
        for (const BlockDesc& d : P->dense_h) max_stride = std::max(max_stride, d.stride);
        const size_t bytes = ((size_t)P->dense_elems + 4 * kStrip + (size_t)kPanel * max_stride) * es;
        HIP_TRY(P->d_ld_dense.alloc(bytes));
        HIP_TRY(hipMemset(P->d_ld_dense.p, 0, bytes));
        dim3 grid(64, (unsigned)P->dense_h.size());
        const int upper = P->low_memory;
        switch (ld_dtype) {
            case VIPRS_LD_F32:
                repack_dense_kernel<float><<<grid, 256>>>((const float*)P->d_ld_raw.p, P->d_ip.p, (float*)P->d_ld_dense.p, P->d_dense.p, upper);
                break;
            case VIPRS_LD_I8:
                repack_dense_kernel<int8_t><<<grid, 256>>>((const int8_t*)P->d_ld_raw.p, P->d_ip.p, (int8_t*)P->d_ld_dense.p, P->d_dense.p, upper);
                break;
            case VIPRS_LD_I16:
                repack_dense_kernel<int16_t><<<grid, 256>>>((const int16_t*)P->d_ld_raw.p, P->d_ip.p, (int16_t*)P->d_ld_dense.p, P->d_dense.p, upper);
                break;
            default: break;
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        if (!P->low_memory) {
            // batched grid E-step, symmetric form: the columns left of the chain are finished by
            // estep_grid_lower_pass_kernel, one item per 128-column tile that has rows below it, longest first
            std::vector<EpiItem> low;
            for (size_t i = 0; i < P->dense_h.size(); ++i) {
                const int np = (P->dense_h[i].size + kPanel - 1) / kPanel;
                for (int T = 0; 2 * T + 1 < np; ++T) low.push_back({(int32_t)i, T});
            }
            std::stable_sort(low.begin(), low.end(), [&](const EpiItem& x, const EpiItem& y) {
                const int npx = (P->dense_h[(size_t)x.blk].size + kPanel - 1) / kPanel, npy = (P->dense_h[(size_t)y.blk].size + kPanel - 1) / kPanel;
                return npx - 2 * x.row0 > npy - 2 * y.row0;
            });
            P->n_low_items = (int64_t)low.size();
            if (!low.empty()) {
                HIP_TRY(P->d_low_items.alloc(low.size()));
                HIP_TRY(hipMemcpy(P->d_low_items.p, low.data(), sizeof(EpiItem) * low.size(), hipMemcpyHostToDevice));
            }
        }
        if (P->low_memory) {
            std::vector<EpiItem> items;
            for (int c = 0; c < 3; ++c) {
                P->epi_begin[c] = (int)items.size();
                std::vector<EpiItem> cls;
                for (int i = P->class_begin[c]; i < P->class_begin[c + 1]; ++i)
                    for (int r0 = 0; r0 < P->dense_h[(size_t)i].size; r0 += kPanel)
                        cls.push_back({(int32_t)(i - P->class_begin[c]), r0});        // block index inside its class
                // longest rows first (the item cost is the number of columns right of its rows)
                const int cb = P->class_begin[c];
                std::stable_sort(cls.begin(), cls.end(), [&](const EpiItem& x, const EpiItem& y) {
                    return P->dense_h[(size_t)(cb + x.blk)].size - x.row0 > P->dense_h[(size_t)(cb + y.blk)].size - y.row0;
                });
                items.insert(items.end(), cls.begin(), cls.end());
            }
            P->epi_begin[3] = (int)items.size();
            P->n_epi = (int64_t)items.size();
            // the same items once more with plan-wide block indices, longest first across all classes
            // (one launch of the batched grid second pass)
            std::vector<EpiItem> all;
            for (int c = 0; c < 3; ++c)
                for (int k = P->epi_begin[c]; k < P->epi_begin[c + 1]; ++k)
                    all.push_back({(int32_t)(items[(size_t)k].blk + P->class_begin[c]), items[(size_t)k].row0});
            std::stable_sort(all.begin(), all.end(), [&](const EpiItem& x, const EpiItem& y) {
                return P->dense_h[(size_t)x.blk].size - x.row0 > P->dense_h[(size_t)y.blk].size - y.row0;
            });
            HIP_TRY(P->d_epi_all.alloc(all.size()));
            HIP_TRY(hipMemcpy(P->d_epi_all.p, all.data(), sizeof(EpiItem) * all.size(), hipMemcpyHostToDevice));
            HIP_TRY(P->d_epi.alloc(items.size()));
            HIP_TRY(hipMemcpy(P->d_epi.p, items.data(), sizeof(EpiItem) * items.size(), hipMemcpyHostToDevice));
        }
    }
    if (!P->ragged_h.empty()) {
        HIP_TRY(P->d_ragged.alloc(P->ragged_h.size()));
        HIP_TRY(hipMemcpy(P->d_ragged.p, P->ragged_h.data(), sizeof(BlockDesc) * P->ragged_h.size(), hipMemcpyHostToDevice));
    } else {
        // raw copy no longer needed: every block was repacked
        HIP_TRY(P->d_ld_raw.alloc(0));
    }
    // every copy / memset above went through the null stream; the plan's own streams are non-blocking
    // (not ordered with it), so nothing may still be in flight when the first sweep is launched
    HIP_TRY(hipDeviceSynchronize());
    *out = P.release();
    return VIPRS_OK;
}


extern "C" {

const char* viprs_last_error(void) { return g_err.c_str(); }
const char* viprs_version(void) { return "viprs_amd 0.1.0 (gfx950)"; }

int viprs_device_count(int* count) {
    if (!count) return fail(VIPRS_EINVAL, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(VIPRS_EDEVICE, hipGetErrorString(e)); }
    *count = n;
    return VIPRS_OK;
}

int viprs_check_blas_support(void) { return 0; }
int viprs_check_omp_support(void) { return 0; }

int viprs_plan_blocks(int64_t m, const int32_t* lb, const void* indptr, int indptr_dtype, int low_memory,
                      int64_t* n_blocks, int64_t* block_start, int32_t* block_kind) {
    if (!n_blocks || !block_start) return fail(VIPRS_EINVAL, "null output");
    if (m > 0 && (!lb || !indptr)) return fail(VIPRS_EINVAL, "null LD index array");
    std::vector<int64_t> ip64;
    const int64_t* ip = nullptr;
    if (indptr_dtype == VIPRS_IP_I64) {
        ip = static_cast<const int64_t*>(indptr);
    } else if (indptr_dtype == VIPRS_IP_I32) {
        const int32_t* p = static_cast<const int32_t*>(indptr);
        ip64.assign(p, p + m + 1);
        ip = ip64.data();
    } else {
        return fail(VIPRS_EINVAL, "bad indptr dtype code");
    }
    std::vector<Block> blocks;
    std::string err;
    int rc = plan_blocks(m, lb, ip, low_memory != 0, blocks, err);
    if (rc != VIPRS_OK) return fail(rc, err);
    *n_blocks = (int64_t)blocks.size();
    for (size_t i = 0; i < blocks.size(); ++i) {
        block_start[i] = blocks[i].start;
        if (block_kind) block_kind[i] = blocks[i].kind;
    }
    block_start[blocks.size()] = m;
    return VIPRS_OK;
}

int viprs_plan_create(viprs_plan** out, int64_t m, const int32_t* lb, const void* indptr, int indptr_dtype,
                      const void* ld_data, int ld_dtype, int low_memory, int device) {
    if (!out) return fail(VIPRS_EINVAL, "plan output is null");
    *out = nullptr;
    if (ld_elem_size(ld_dtype) == 0) return fail(VIPRS_EINVAL, "bad LD dtype code");
    if (m < 0 || m > INT32_MAX) return fail(VIPRS_EINVAL, "m out of range");
    if (m > 0 && (!lb || !indptr)) return fail(VIPRS_EINVAL, "null LD index array");
    std::vector<int64_t> ip64;
    int rc = widen_indptr(m, indptr, indptr_dtype, ip64);
    if (rc != VIPRS_OK) return rc;
    return plan_create_impl(out, m, lb, ip64, ld_data, ld_dtype, low_memory, device, nullptr);
}

int viprs_plan_create_expanded(viprs_plan** out, int64_t m, const void* upper_indptr, int indptr_dtype,
                               const void* upper_data, int ld_dtype, double diag_value, int device) {
    if (!out) return fail(VIPRS_EINVAL, "plan output is null");
    *out = nullptr;
    if (ld_elem_size(ld_dtype) == 0) return fail(VIPRS_EINVAL, "bad LD dtype code");
    if (m < 0 || m > INT32_MAX) return fail(VIPRS_EINVAL, "m out of range");
    if (m > 0 && !upper_indptr) return fail(VIPRS_EINVAL, "null LD index array");
    ExpandSource ex;
    int rc = widen_indptr(m, upper_indptr, indptr_dtype, ex.ip_upper);
    if (rc != VIPRS_OK) return rc;
    ex.data = upper_data;
    ex.diag = diag_value;
    if (m > 0 && ex.ip_upper[(size_t)m] > 0 && !upper_data) return fail(VIPRS_EINVAL, "ld_data is null");
    // symmetric windows: row j = [first row that reaches j .. j + len_j].  They are contiguous (the
    // layout e_step.hpp:389-392 needs) iff the right ends j + len_j never decrease.
    std::vector<int32_t> lb((size_t)m, 0);
    std::vector<int64_t> ip((size_t)m + 1, 0);
    int64_t first = 0, prev_reach = -1;
    for (int64_t j = 0; j < m; ++j) {
        const int64_t len = ex.ip_upper[(size_t)j + 1] - ex.ip_upper[(size_t)j];
        if (len < 0) return fail(VIPRS_EINVAL, "ld_indptr is not non-decreasing");
        const int64_t reach = j + len;
        if (reach >= m) return fail(VIPRS_EINVAL, "an upper-triangular LD row runs past the last SNP");
        if (reach < prev_reach) return fail(VIPRS_EINVAL, "the upper-triangular windows do not mirror into contiguous symmetric windows");
        prev_reach = reach;
        while (first < j && first + (ex.ip_upper[(size_t)first + 1] - ex.ip_upper[(size_t)first]) < j) ++first;
        lb[(size_t)j] = (int32_t)first;
        ip[(size_t)j + 1] = ip[(size_t)j] + (j - first) + 1 + len;
    }
    return plan_create_impl(out, m, lb.data(), ip, nullptr, ld_dtype, 0, device, &ex);
}

int viprs_plan_get_windows(const viprs_plan* P, int32_t* left_bound, int64_t* indptr) {
    if (!P || !left_bound || !indptr) return fail(VIPRS_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(P->device));
    indptr[0] = 0;
    if (P->m > 0) {
        HIP_TRY(hipMemcpy(left_bound, P->d_lb.p, sizeof(int32_t) * (size_t)P->m, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(indptr, P->d_ip.p, sizeof(int64_t) * ((size_t)P->m + 1), hipMemcpyDeviceToHost));
    }
    return VIPRS_OK;
}
int viprs_plan_destroy(viprs_plan* plan) {
    if (!plan) return VIPRS_OK;
    (void)hipSetDevice(plan->device);
    delete plan;
    return VIPRS_OK;
}

int viprs_plan_info(const viprs_plan* P, int key, int64_t* value) {
    if (!P || !value) return fail(VIPRS_EINVAL, "null argument");
    switch (key) {
        case VIPRS_INFO_M: *value = P->m; break;
        case VIPRS_INFO_NNZ: *value = P->nnz; break;
        case VIPRS_INFO_N_BLOCKS: *value = (int64_t)P->blocks.size(); break;
        case VIPRS_INFO_N_DENSE: *value = (int64_t)P->dense_h.size(); break;
        case VIPRS_INFO_N_RAGGED: *value = (int64_t)P->ragged_h.size(); break;
        case VIPRS_INFO_MAX_BLOCK: *value = std::max(P->max_dense, P->max_ragged); break;
        case VIPRS_INFO_LD_BYTES_DEVICE: *value = (int64_t)(P->d_ld_raw.n + P->d_ld_dense.n); break;
        case VIPRS_INFO_LD_ELEM_SIZE: *value = (int64_t)ld_elem_size(P->ld_dtype); break;
        case VIPRS_INFO_DEVICE: *value = P->device; break;
        case VIPRS_INFO_LOW_MEMORY: *value = P->low_memory; break;
        case VIPRS_INFO_N_CU: *value = P->n_cu; break;
        default: return fail(VIPRS_EINVAL, "unknown info key");
    }
    return VIPRS_OK;
}

int viprs_plan_get_blocks(const viprs_plan* P, int64_t* block_start, int32_t* block_kind) {
    if (!P || !block_start) return fail(VIPRS_EINVAL, "null argument");
    for (size_t i = 0; i < P->blocks.size(); ++i) {
        block_start[i] = P->blocks[i].start;
        if (block_kind) block_kind[i] = P->blocks[i].kind;
    }
    block_start[P->blocks.size()] = P->m;
    return VIPRS_OK;
}

int viprs_plan_set_math_mode(viprs_plan* P, int mode) {
    if (!P) return fail(VIPRS_EINVAL, "null plan");
    if (mode != VIPRS_MATH_EXACT && mode != VIPRS_MATH_FAST) return fail(VIPRS_EINVAL, "bad math mode");
    P->math_mode = mode;
    return VIPRS_OK;
}

// ---- state -------------------------------------------------------------------------------------
int viprs_state_create(viprs_state** out, viprs_plan* plan, int float_dtype, int model_kind, int width) {
    if (!out || !plan) return fail(VIPRS_EINVAL, "null argument");
    *out = nullptr;
    if (float_size(float_dtype) == 0) return fail(VIPRS_EINVAL, "bad float dtype code");
    if (model_kind < VIPRS_MODEL_SPIKE_SLAB || model_kind > VIPRS_MODEL_GRID) return fail(VIPRS_EINVAL, "bad model kind");
    if (model_kind == VIPRS_MODEL_SPIKE_SLAB) width = 1;
    if (width < 1) return fail(VIPRS_EINVAL, "width must be >= 1");
    HIP_TRY(hipSetDevice(plan->device));
    std::unique_ptr<viprs_state> S(new viprs_state());
    S->plan = plan;
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
    (void)hipSetDevice(S->plan->device);
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

// after a synchronisation point: did a team hand-off give up (bounded spin)?
static int check_device_error(viprs_plan* P) {
    int32_t e = 0;
    HIP_TRY(hipMemcpy(&e, P->d_error.p, sizeof(e), hipMemcpyDeviceToHost));
    if (e != 0) {
        HIP_TRY(hipMemsetAsync(P->d_error.p, 0, sizeof(e), P->stream));
        HIP_TRY(hipStreamSynchronize(P->stream));
        return fail(VIPRS_EDEVICE, "E-step kernel: a team hand-off timed out (results of this sweep are invalid)");
    }
    return VIPRS_OK;
}

template <typename T>
static int sums_enqueue(viprs_state* S, int64_t off, int64_t vt_off, double one_plus_lambda) {
    viprs_plan* P = S->plan;
    const int nb = (int)std::min<int64_t>((P->m + kSumsBlock - 1) / kSumsBlock, 1024);
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
        S->d_var_tau.p, 0.0, S->d_weight.p, S->d_partials.p, S->d_sumcols.p);
    HIP_TRY(hipGetLastError());
    sums_final_kernel<<<n, 64 * kNSums, 0, P->stream>>>(S->d_partials.p, nb, S->d_sums.p);
    HIP_TRY(hipGetLastError());
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

// ---- kernel dispatch ----------------------------------------------------------------------------
namespace {


template <typename T>
EStepArgs<T> make_args(viprs_state* S, double dq) {
    viprs_plan* P = S->plan;
    EStepArgs<T> A{};
    A.skipped = P->d_skipped.p;
    A.lb = P->d_lb.p;
    A.rowstart = P->d_ip.p;
    A.rowlen = P->d_rowlen.p;
    A.ld_rows = P->d_ld_raw.p;
    A.ld_dense = P->d_ld_dense.p;
    A.std_beta = (const T*)S->f[VIPRS_FIELD_STD_BETA].p;
    A.u_logs = (const T*)S->f[VIPRS_FIELD_U_LOGS].p;
    A.shvt = (const T*)S->f[VIPRS_FIELD_SQRT_HALF_VAR_TAU].p;
    A.mu_mult = (const T*)S->f[VIPRS_FIELD_MU_MULT].p;
    A.log_null_pi = (const T*)S->f[VIPRS_FIELD_LOG_NULL_PI].p;
    A.var_gamma = (T*)S->f[VIPRS_FIELD_VAR_GAMMA].p;
    A.var_mu = (T*)S->f[VIPRS_FIELD_VAR_MU].p;
    A.eta = (T*)S->f[VIPRS_FIELD_ETA].p;
    A.q = (T*)S->f[VIPRS_FIELD_Q].p;
    A.eta_diff = (T*)S->f[VIPRS_FIELD_ETA_DIFF].p;
    A.eta_out = (T*)S->eta_out.p;
    A.q_out = (T*)S->q_out.p;
    A.granules = P->d_granules.p;
    A.granule_rows = P->n_granule_rows;
    A.error = P->d_error.p;
    A.dq = (T)dq;
    A.low_memory = P->low_memory;
    A.width = S->width;
    A.m = P->m;
    return A;
}

enum { kGenSpikeSlab = 0, kGenMixture = 1, kGenGrid = 2 };

// Generic kernels over one block list: `dense` = the repacked dense blocks (addressed through
// d_rowstart_dense), otherwise the ragged blocks in the caller's own layout.
template <typename T, typename U>
int launch_generic(viprs_plan* P, EStepArgs<T> A, int model, bool dense) {
    const std::vector<BlockDesc>& list = dense ? P->dense_h : P->ragged_h;
    if (list.empty()) return VIPRS_OK;
    A.blocks = dense ? P->d_dense.p : P->d_ragged.p;
    A.n_blocks = (int)list.size();
    A.counter = P->d_counters.p + (dense ? 2 : 1);
    if (dense) {
        A.rowstart = P->d_rowstart_dense.p;
        A.ld_rows = P->d_ld_dense.p;
    }
    const int max_b = dense ? P->max_dense : P->max_ragged;
    const size_t need = 2 * (size_t)max_b * sizeof(T);
    const bool in_lds = need <= 128 * 1024;
    const size_t shmem = in_lds ? need : 0;
    const int grid = std::min<int>(A.n_blocks, P->n_cu * 2);
    const bool exact = P->math_mode == VIPRS_MATH_EXACT;
#define GEN_LAUNCH(KFN, SH)                                                                                   \
    do {                                                                                                      \
        auto kfn = KFN;                                                                                       \
        if ((SH) > 48 * 1024)                                                                                 \
            HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SH))); \
        kfn<<<grid, kGenericThreads, (SH), P->stream>>>(A);                                                   \
    } while (0)
    if (model == kGenSpikeSlab) {
        if (exact && in_lds) GEN_LAUNCH((estep_generic_kernel<T, U, true, true>), shmem);
        else if (exact) GEN_LAUNCH((estep_generic_kernel<T, U, true, false>), shmem);
        else if (in_lds) GEN_LAUNCH((estep_generic_kernel<T, U, false, true>), shmem);
        else GEN_LAUNCH((estep_generic_kernel<T, U, false, false>), shmem);
    } else if (model == kGenMixture) {
        if (A.width > kMaxMixtureK) return fail(VIPRS_EUNSUPPORTED, "mixture: K > 64 is not supported");
        if (in_lds) GEN_LAUNCH((estep_mixture_generic_kernel<T, U, true>), shmem);
        else GEN_LAUNCH((estep_mixture_generic_kernel<T, U, false>), shmem);
    } else {
        GEN_LAUNCH((estep_grid_generic_kernel<T, U>), (size_t)0);
    }
#undef GEN_LAUNCH
    HIP_TRY(hipGetLastError());
    return VIPRS_OK;
}

template <typename T>
int launch_generic_u(viprs_plan* P, const EStepArgs<T>& A, int model, bool dense) {
    switch (P->ld_dtype) {
        case VIPRS_LD_I8: return launch_generic<T, int8_t>(P, A, model, dense);
        case VIPRS_LD_I16: return launch_generic<T, int16_t>(P, A, model, dense);
        case VIPRS_LD_I32: return launch_generic<T, int32_t>(P, A, model, dense);
        case VIPRS_LD_I64: return launch_generic<T, int64_t>(P, A, model, dense);
        case VIPRS_LD_F32: return launch_generic<T, float>(P, A, model, dense);
        case VIPRS_LD_F64: return launch_generic<T, double>(P, A, model, dense);
        default: return fail(VIPRS_EINVAL, "bad LD dtype");
    }
}

enum { kPanelSpikeSlab = 0, kPanelGridColumn = 1, kPanelMixture = 2 };

template <typename U, int NW, bool TEAM, int CPL>
int launch_panel_class(viprs_plan* P, EStepArgs<float> A, int cls, hipStream_t stream, int model) {
    const int begin = P->class_begin[cls], end = P->class_begin[cls + 1];
    if (end <= begin) return VIPRS_OK;
    A.blocks = P->d_dense.p + begin;
    A.n_blocks = end - begin;
    A.counter = P->d_counters.p + 4 + cls;
    const int max_b = P->dense_h[begin].size;      // descending order
    const int qcap = (max_b + kPanel - 1) / kPanel * kPanel + kStrip;
    const size_t shmem = (size_t)(panel_lds_floats(qcap) + (model == kPanelMixture ? kMixLdsFloats : 0)) * sizeof(float);
    const bool exact = P->math_mode == VIPRS_MATH_EXACT;
    const bool upper = P->low_memory != 0;
    const void* kfn = nullptr;
#define PK(MODEL) (upper ? (const void*)estep_panel_kernel<U, MODEL, false, NW, TEAM, CPL> \
                          : (const void*)estep_panel_kernel<U, MODEL, true, NW, TEAM, CPL>)
    if (model == kPanelGridColumn) kfn = PK(GridColumnModel);
    else if (model == kPanelMixture) kfn = PK(MixtureModel);
    else kfn = exact ? PK(SpikeSlabModel<true>) : PK(SpikeSlabModel<false>);
#undef PK
    if (shmem > 160 * 1024) return fail(VIPRS_EUNSUPPORTED, "LD block too large for the LDS-resident panel kernel");
    if (shmem > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, NW * 64, shmem));
    per_cu = std::max(1, per_cu);
    if (const char* f = getenv("VIPRS_MAX_WG_PER_CU")) per_cu = std::max(1, std::min(per_cu, atoi(f)));   // experiments
    const int n_models = std::max(1, A.n_active);
    const int64_t n_items = (int64_t)A.n_blocks * n_models;
    int grid = (int)std::min<int64_t>(n_items, (int64_t)P->n_cu * per_cu);
    const int* team_of = (model == kPanelMixture && !g_team_env) ? kClassTeamMix : kClassTeam;
    const int TS = TEAM ? team_of[cls] : 1;
    A.team_size = TS;
    if (!TEAM) {
        // leave room for the team workgroups of the larger classes: they must all become resident
        // while this class's persistent workgroups hold their slots
        int reserved = 0;
        for (int c = 0; c < 3; ++c) {
            const int nb = P->class_begin[c + 1] - P->class_begin[c];
            if (c < 2 && nb > 0)
                reserved += (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)nb * n_models, P->n_cu / team_of[c])) * team_of[c];
        }
        grid = (int)std::min<int64_t>(n_items, std::max(P->n_cu / 2, P->n_cu * per_cu - reserved));
    }
    if (TEAM) {
        // teams of TS workgroups, statically assigned blocks; the whole grid must be able to be
        // resident at once (at most one workgroup per CU is assumed)
        A.n_teams = (int)std::max<int64_t>(1, std::min<int64_t>(n_items, P->n_cu / TS));
        grid = A.n_teams * TS;
    }
    A.admit = nullptr;
    if (cls == 2 && P->admit_factor > 0.0 && n_models == 1) {
        if (P->admit_grid != grid) {
            // Concurrency needed to keep HBM busy while the head is at a block of b SNPs:
            // a workgroup streams 64 rows x b columns per chain phase (~10 us), i.e. about
            // kRatePerSnp * b bytes/s; need ~kHbmRate in total.  The larger classes (their own
            // kernels, never gated) are credited with their demand.
            const double kRatePerSnp = 64.0 * 4.0 / 10.5e-6, kHbmRate = 5.5e12;
            double big_demand = 0.0;
            for (int i = 0; i < begin; ++i) big_demand += std::min(9.0e10, kRatePerSnp * P->dense_h[i].size);
            std::vector<int32_t> th((size_t)grid, 0);
            int r = 0;
            for (int n = 0; n < A.n_blocks && r < grid; ++n) {
                const double b = P->dense_h[begin + n].size;
                // the big-class kernels drain while the small class advances; credit them linearly
                const double credit = big_demand * std::max(0.0, 1.0 - 2.0 * n / (double)A.n_blocks);
                double need = P->admit_factor * std::max(0.0, kHbmRate - credit) / (kRatePerSnp * b);
                int allowed = (int)std::min<double>(grid, std::max(64.0, need));
                while (r < allowed) th[(size_t)r++] = n;
            }
            while (r < grid) th[(size_t)r++] = A.n_blocks;
            HIP_TRY(P->d_admit.alloc((size_t)grid));
            HIP_TRY(hipMemcpyAsync(P->d_admit.p, th.data(), sizeof(int32_t) * (size_t)grid, hipMemcpyHostToDevice, stream));
            HIP_TRY(hipStreamSynchronize(stream));         // one-time set-up; `th` is a local buffer
            P->admit_grid = grid;
        }
        A.admit = P->d_admit.p;
    }
    void* params[] = {(void*)&A, (void*)&qcap};
    HIP_TRY(hipLaunchKernel(kfn, dim3(grid), dim3(NW * 64), params, shmem, stream));
    if (TEAM) {
        commit_team_kernel<<<dim3(A.n_blocks, n_models), 256, 0, stream>>>(A);
        HIP_TRY(hipGetLastError());
    }
    if (upper) {
        // the reference's second pass for this class, right behind its forward kernel on the same
        // stream (so it overlaps the forward kernels of the other classes)
        const int eb = P->epi_begin[cls], en = P->epi_begin[cls + 1] - eb;
        if (en > 0) {
            const int eg = (int)std::min<int64_t>(((int64_t)en * n_models + kEpiWaves - 1) / kEpiWaves, (int64_t)P->n_cu * 2);
            estep_upper_epilogue_kernel<U, kEpiWaves><<<eg, kEpiWaves * 64, 0, stream>>>(A, P->d_epi.p + eb, en,
                                                                                      P->d_counters.p + 8 + cls);
            HIP_TRY(hipGetLastError());
        }
    }
    return VIPRS_OK;
}

// one wave that holds its stream for `ticks` of the 100 MHz wall clock (see launch_panel)
__global__ void stream_delay_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

// Panel kernels of the three size classes run concurrently on their own streams (forked from /
// joined back into the plan's stream with events); the upper-triangular second pass follows.
template <typename U>
int launch_panel(viprs_plan* P, EStepArgs<float> A, int model = kPanelSpikeSlab) {
    HIP_TRY(hipEventRecord(P->ev_fork, P->stream));
    int rc;
    for (int c = 0; c < 3; ++c) HIP_TRY(hipStreamWaitEvent(P->class_stream[c], P->ev_fork, 0));
    if ((rc = launch_panel_class<U, kClassWaves[0], true, panel_cols<U>()>(P, A, 0, P->class_stream[0], model)) != VIPRS_OK) return rc;
    if ((rc = launch_panel_class<U, kClassWaves[1], true, panel_cols<U>()>(P, A, 1, P->class_stream[1], model)) != VIPRS_OK) return rc;
    {
        // The team kernels are the critical path and need all their workgroups resident; when the persistent
        // workgroups of the small-block queue reach the CUs first, a sweep takes 1.15-1.2 ms instead of
        // 0.87-0.9 ms (cfg3; it happened in about every third sweep).  One wave holds the small-block stream
        // for a few microseconds so that the team kernels are placed first.  Only when the small-block queue
        // is long enough to fill the chip, and only for fp32 LD: with int8 / int16 LD the small-block queue is
        // itself the critical path (1.12 -> 1.17 ms with the delay).  VIPRS_SMALL_DELAY_US overrides, 0 = off.
        static const int env_us = [] { const char* f = getenv("VIPRS_SMALL_DELAY_US"); return f ? atoi(f) : -1; }();
        const int delay_us = env_us >= 0 ? env_us : (P->ld_dtype == VIPRS_LD_F32 ? 15 : 0);
        if (delay_us > 0 && P->class_begin[2] > 0 && P->class_begin[3] - P->class_begin[2] >= P->n_cu) {
            stream_delay_kernel<<<1, 64, 0, P->class_stream[2]>>>((long long)delay_us * 100);
            HIP_TRY(hipGetLastError());
        }
    }
    if ((rc = launch_panel_class<U, kClassWaves[2], false, panel_cols<U>()>(P, A, 2, P->class_stream[2], model)) != VIPRS_OK) return rc;
    for (int c = 0; c < 3; ++c) {
        HIP_TRY(hipEventRecord(P->ev_join[c], P->class_stream[c]));
        HIP_TRY(hipStreamWaitEvent(P->stream, P->ev_join[c], 0));
    }
    return VIPRS_OK;
}

// Windowed (ragged) components, fp32 state, f32 / int8 / int16 LD, lane-per-SNP model policies: the band
// kernel (estep_band.h).  VIPRS_BAND=0 sends them back to the generic kernel.
enum { kBandSpikeSlab = 0, kBandGridColumn = 1, kBandMixture = 2 };

static int band_ring_panels(const viprs_plan* P) {
    int rp = 4;
    while (rp < P->max_band_panels) rp *= 2;
    return rp;
}

static bool use_band(const viprs_plan* P) {
    const char* f = getenv("VIPRS_BAND");
    if ((f && !atoi(f)) || P->ragged_h.empty()) return false;
    if (P->ld_dtype != VIPRS_LD_F32 && P->ld_dtype != VIPRS_LD_I8 && P->ld_dtype != VIPRS_LD_I16) return false;
    return band_ring_panels(P) <= kBandMaxRingPanels;
}

template <typename U>
int launch_band(viprs_plan* P, EStepArgs<float> A, int model) {
    A.blocks = P->d_ragged.p;
    A.n_blocks = (int)P->ragged_h.size();
    A.counter = P->d_counters.p + 1;
    const int ring = band_ring_panels(P);
    const size_t shmem = (size_t)band_lds_floats(ring) * sizeof(float);
    const bool exact = P->math_mode == VIPRS_MATH_EXACT;
    const bool upper = P->low_memory != 0;
    void (*kfn)(EStepArgs<float>, int) = nullptr;
#define BK(MODEL) (upper ? estep_band_kernel<U, MODEL, false> : estep_band_kernel<U, MODEL, true>)
    if (model == kBandGridColumn) kfn = BK(GridColumnModel);
    else if (model == kBandMixture) kfn = BK(MixtureSerialModel);
    else kfn = exact ? BK(SpikeSlabModel<true>) : BK(SpikeSlabModel<false>);
#undef BK
    HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const int items = A.n_blocks * std::max(A.n_active, 1);
    const int grid = std::min(items, 2 * P->n_cu);
    kfn<<<grid, 64 * kBandWaves, shmem, P->stream>>>(A, ring);
    HIP_TRY(hipGetLastError());
    if (upper) {
        const dim3 egrid((unsigned)std::min(256, (P->max_ragged + 255) / 256), (unsigned)items);
        band_upper_epilogue_kernel<U><<<egrid, 256, 0, P->stream>>>(A);
        HIP_TRY(hipGetLastError());
    }
    return VIPRS_OK;
}

static int launch_band_u(viprs_plan* P, const EStepArgs<float>& A, int model) {
    switch (P->ld_dtype) {
        case VIPRS_LD_F32: return launch_band<float>(P, A, model);
        case VIPRS_LD_I8: return launch_band<int8_t>(P, A, model);
        case VIPRS_LD_I16: return launch_band<int16_t>(P, A, model);
        default: return fail(VIPRS_EINVAL, "band schedule with unsupported LD dtype");
    }
}

int run_spike_slab(viprs_state* S, double dq) {
    viprs_plan* P = S->plan;
    HIP_TRY(hipSetDevice(P->device));
    {
        const int64_t ng = P->n_granule_rows * kPanel;
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((ng + 255) / 256, 1024));
        sweep_prologue_kernel<<<grid, 256, 0, P->stream>>>(P->d_counters.p, 16, P->d_skipped.p, P->d_granules.p, ng);
        HIP_TRY(hipGetLastError());
    }
    hipEvent_t* ev = P->ev.data() + 4 * (P->sweeps % viprs_plan::kRing);
    HIP_TRY(hipEventRecord(ev[0], P->stream));
    int rc = VIPRS_OK;
    if (S->float_dtype == VIPRS_F32) {
        EStepArgs<float> A = make_args<float>(S, dq);
        if (!P->dense_h.empty()) {
            HIP_TRY(hipEventRecord(ev[2], P->stream));
            switch (P->ld_dtype) {
                case VIPRS_LD_F32: rc = launch_panel<float>(P, A); break;
                case VIPRS_LD_I8: rc = launch_panel<int8_t>(P, A); break;
                case VIPRS_LD_I16: rc = launch_panel<int16_t>(P, A); break;
                default: rc = fail(VIPRS_EINVAL, "dense schedule with unsupported LD dtype"); break;
            }
            if (rc != VIPRS_OK) return rc;
            HIP_TRY(hipEventRecord(ev[3], P->stream));
        }
        if (use_band(P)) rc = launch_band_u(P, A, kBandSpikeSlab);
        else if (!P->ragged_h.empty()) rc = launch_generic_u<float>(P, A, kGenSpikeSlab, false);
    } else {
        // float64 state: the panel kernels specialise float; every block takes the generic kernel
        EStepArgs<double> A = make_args<double>(S, dq);
        rc = launch_generic_u<double>(P, A, kGenSpikeSlab, true);
        if (rc == VIPRS_OK) rc = launch_generic_u<double>(P, A, kGenSpikeSlab, false);
    }
    if (rc != VIPRS_OK) return rc;
    HIP_TRY(hipEventRecord(ev[1], P->stream));
    P->sweeps++;
    return VIPRS_OK;
}

static int sweep_prologue(viprs_plan* P, int n_models = 1) {
    const int64_t ng = P->n_granule_rows * kPanel * std::max(1, n_models);
    if (P->d_granules.n < (size_t)ng) {                      // grid launches: one granule set per active model
        HIP_TRY(hipStreamSynchronize(P->stream));
        HIP_TRY(P->d_granules.alloc((size_t)ng));
    }
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((ng + 255) / 256, 1024));
    sweep_prologue_kernel<<<grid, 256, 0, P->stream>>>(P->d_counters.p, 16, P->d_skipped.p, P->d_granules.p, ng);
    HIP_TRY(hipGetLastError());
    return VIPRS_OK;
}

static int launch_panel_u(viprs_plan* P, const EStepArgs<float>& A, int model) {
    switch (P->ld_dtype) {
        case VIPRS_LD_F32: return launch_panel<float>(P, A, model);
        case VIPRS_LD_I8: return launch_panel<int8_t>(P, A, model);
        case VIPRS_LD_I16: return launch_panel<int16_t>(P, A, model);
        default: return fail(VIPRS_EINVAL, "dense schedule with unsupported LD dtype");
    }
}

// Batched grid E-step on the matrix cores: every LD row is read once for up to 32 models.
template <typename U>
static int launch_grid_mfma(viprs_plan* P, EStepArgs<float> A) {
    A.blocks = P->d_dense.p;
    A.n_blocks = (int)P->dense_h.size();
    A.counter = P->d_counters.p + 12;
    const size_t shmem = (size_t)kGridLdsFloats * sizeof(float);
    const void* kfn = P->low_memory ? (const void*)estep_grid_mfma_kernel<U, false> : (const void*)estep_grid_mfma_kernel<U, true>;
    HIP_TRY(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 64 * kGridWaves, shmem));
    const int grid = std::min<int>(A.n_blocks, P->n_cu * std::max(1, per_cu));
    void* params[] = {(void*)&A};
    HIP_TRY(hipLaunchKernel(kfn, dim3(grid), dim3(64 * kGridWaves), params, shmem, P->stream));
    if (!P->low_memory && P->n_low_items > 0) {
        // symmetric form: q of the SNPs already visited receives the later rows of its block here
        const size_t lshmem = (size_t)kGridLowWaves * kGridLowWaveFloats * sizeof(float);
        const void* lfn = (const void*)estep_grid_lower_pass_kernel<U>;
        HIP_TRY(hipFuncSetAttribute(lfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lshmem));
        const int ln = (int)P->n_low_items;
        const int lg = (int)std::min<int64_t>(((int64_t)ln + kGridLowWaves - 1) / kGridLowWaves, (int64_t)P->n_cu * 2);
        estep_grid_lower_pass_kernel<U><<<lg, 64 * kGridLowWaves, lshmem, P->stream>>>(A, P->d_low_items.p, ln, P->d_counters.p + 9);
        HIP_TRY(hipGetLastError());
    }
    if (P->low_memory) {
        // update_q_factor_matrix (e_step.hpp:266-303) for all models of the chunk: (block, 64-row group) items
        const size_t eshmem = (size_t)kGridEpiWaves * kGridEpiWaveFloats * sizeof(float);
        const void* efn = (const void*)estep_grid_upper_epilogue_kernel<U>;
        HIP_TRY(hipFuncSetAttribute(efn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)eshmem));
        const int en = (int)P->n_epi;
        if (en > 0) {
            const int eg = (int)std::min<int64_t>(((int64_t)en + kGridEpiWaves - 1) / kGridEpiWaves, (int64_t)P->n_cu);
            estep_grid_upper_epilogue_kernel<U><<<eg, 64 * kGridEpiWaves, eshmem, P->stream>>>(A, P->d_epi_all.p, en,
                                                                                             P->d_counters.p + 8);
            HIP_TRY(hipGetLastError());
        }
    }
    return VIPRS_OK;
}

// One workgroup per LD block: worth it once the blocks can occupy a good part of the chip; with a few
// blocks the (block, model) work items of the panel kernel spread better.  32-bit element offsets.
static bool use_grid_mfma(const viprs_plan* P, int width) {
    if ((int64_t)P->m * std::max(1, width) >= (1LL << 31)) return false;
    if (P->grid_mfma >= 0) return P->grid_mfma != 0;
    return (int64_t)P->dense_h.size() * 8 >= (int64_t)P->n_cu * 3;
}

static int launch_grid_mfma_u(viprs_plan* P, const EStepArgs<float>& A) {
    switch (P->ld_dtype) {
        case VIPRS_LD_F32: return launch_grid_mfma<float>(P, A);
        case VIPRS_LD_I8: return launch_grid_mfma<int8_t>(P, A);
        case VIPRS_LD_I16: return launch_grid_mfma<int16_t>(P, A);
        default: return fail(VIPRS_EINVAL, "dense schedule with unsupported LD dtype");
    }
}

// mixture / grid.  fp32 state on dense blocks: the panel kernels with the model's policy (the grid
// runs its independent models one after the other, each on its own column of the (m, G) arrays);
// everything else (ragged blocks, fp64 state, K > kPanelMaxK): the generic kernels.
int run_generic_model(viprs_state* S, double dq, int model, const int32_t* d_active, int n_active,
                      const int32_t* h_active) {
    viprs_plan* P = S->plan;
    HIP_TRY(hipSetDevice(P->device));
    int rc = sweep_prologue(P, model == kGenGrid ? n_active : 1);
    if (rc != VIPRS_OK) return rc;
    hipEvent_t* ev = P->ev.data() + 4 * (P->sweeps % viprs_plan::kRing);
    HIP_TRY(hipEventRecord(ev[0], P->stream));
    HIP_TRY(hipEventRecord(ev[2], P->stream));
    if (S->float_dtype == VIPRS_F32) {
        EStepArgs<float> A = make_args<float>(S, dq);
        A.active = d_active;
        A.n_active = n_active;
        const bool panel_ok = !P->dense_h.empty() && (model == kGenGrid || S->width <= kPanelMaxK);
        if (panel_ok && model == kGenMixture) {
            rc = launch_panel_u(P, A, kPanelMixture);
        } else if (panel_ok && model == kGenGrid && use_grid_mfma(P, A.width)) {
            // matrix-core path: chunks of 32 models, each LD row read once per chunk
            for (int off = 0; off < n_active && rc == VIPRS_OK; off += kGridModels) {
                EStepArgs<float> Ac = A;
                Ac.active = d_active + off;
                Ac.n_active = std::min(kGridModels, n_active - off);
                if (off > 0) rc = sweep_prologue(P, 1);
                if (rc == VIPRS_OK) rc = launch_grid_mfma_u(P, Ac);
            }
        } else if (panel_ok && model == kGenGrid) {
            // ONE launch over (block, model) work items (select_model offsets the columns in-kernel)
            A.granules = P->d_granules.p;
            rc = launch_panel_u(P, A, kPanelGridColumn);
        } else {
            rc = launch_generic_u<float>(P, A, model, true);
        }
        if (rc == VIPRS_OK) {
            if (model == kGenGrid && use_band(P)) rc = launch_band_u(P, A, kBandGridColumn);
            else if (model == kGenMixture && S->width <= kPanelMaxK && use_band(P)) rc = launch_band_u(P, A, kBandMixture);
            else rc = launch_generic_u<float>(P, A, model, false);
        }
    } else {
        EStepArgs<double> A = make_args<double>(S, dq);
        A.active = d_active;
        A.n_active = n_active;
        rc = launch_generic_u<double>(P, A, model, true);
        if (rc == VIPRS_OK) rc = launch_generic_u<double>(P, A, model, false);
    }
    if (rc != VIPRS_OK) return rc;
    HIP_TRY(hipEventRecord(ev[3], P->stream));
    HIP_TRY(hipEventRecord(ev[1], P->stream));
    P->sweeps++;
    return VIPRS_OK;
}

}  // namespace

extern "C" {

int viprs_state_e_step(viprs_state* S, double dq_scale, const int32_t* active, int n_active, int sync) {
    if (!S) return fail(VIPRS_EINVAL, "null state");
    int rc;
    switch (S->model_kind) {
        case VIPRS_MODEL_SPIKE_SLAB: rc = run_spike_slab(S, dq_scale); break;
        case VIPRS_MODEL_MIXTURE: rc = run_generic_model(S, dq_scale, kGenMixture, nullptr, 0, nullptr); break;
        case VIPRS_MODEL_GRID: {
            std::vector<int32_t> all;
            if (!active) {                                    // default: every model is active
                all.resize((size_t)S->width);
                for (int g = 0; g < S->width; ++g) all[(size_t)g] = g;
                active = all.data();
                n_active = S->width;
            }
            for (int i = 0; i < n_active; ++i)
                if (active[i] < 0 || active[i] >= S->width) return fail(VIPRS_EINVAL, "active_model_idx out of range");
            if (n_active == 0) return VIPRS_OK;
            HIP_TRY(hipSetDevice(S->plan->device));
            if (S->d_active.n < (size_t)n_active) HIP_TRY(S->d_active.alloc((size_t)std::max(n_active, S->width)));
            HIP_TRY(hipMemcpyAsync(S->d_active.p, active, sizeof(int32_t) * (size_t)n_active, hipMemcpyHostToDevice,
                                   S->plan->stream));
            HIP_TRY(hipStreamSynchronize(S->plan->stream));   // `active` may be a temporary
            rc = run_generic_model(S, dq_scale, kGenGrid, S->d_active.p, n_active, active);
            break;
        }
        default: return fail(VIPRS_EINVAL, "bad model kind");
    }
    if (rc != VIPRS_OK) return rc;
    if (sync) HIP_TRY(hipStreamSynchronize(S->plan->stream));
    return VIPRS_OK;
}

int viprs_e_step(viprs_plan* P, int float_dtype, const void* std_beta, void* var_gamma, void* var_mu, void* eta,
                 void* q, void* eta_diff, const void* u_logs, const void* shvt, const void* mu_mult,
                 double dq_scale, int threads, int low_memory) {
    (void)threads;
    if (!P) return fail(VIPRS_EINVAL, "null plan");
    if ((low_memory != 0) != (P->low_memory != 0))
        return fail(VIPRS_EINVAL, "low_memory differs from the value the plan was created with");
    if (P->m == 0) return VIPRS_OK;
    if (!P->scratch || P->scratch->float_dtype != float_dtype || P->scratch->model_kind != VIPRS_MODEL_SPIKE_SLAB) {
        delete P->scratch;
        P->scratch = nullptr;
        int rc = viprs_state_create(&P->scratch, P, float_dtype, VIPRS_MODEL_SPIKE_SLAB, 1);
        if (rc != VIPRS_OK) return rc;
    }
    viprs_state* S = P->scratch;
    const size_t bytes = (size_t)P->m * float_size(float_dtype);
    const void* ins[] = {std_beta, u_logs, shvt, mu_mult, var_gamma, var_mu, eta, q, eta_diff};
    const int in_fields[] = {VIPRS_FIELD_STD_BETA, VIPRS_FIELD_U_LOGS, VIPRS_FIELD_SQRT_HALF_VAR_TAU, VIPRS_FIELD_MU_MULT,
                             VIPRS_FIELD_VAR_GAMMA, VIPRS_FIELD_VAR_MU, VIPRS_FIELD_ETA, VIPRS_FIELD_Q, VIPRS_FIELD_ETA_DIFF};
    HIP_TRY(hipSetDevice(P->device));
    for (int i = 0; i < 9; ++i) {
        if (!ins[i]) return fail(VIPRS_EINVAL, "null buffer");
        HIP_TRY(hipMemcpyAsync(S->f[in_fields[i]].p, ins[i], bytes, hipMemcpyHostToDevice, P->stream));
    }
    int rc = run_spike_slab(S, dq_scale);
    if (rc != VIPRS_OK) return rc;
    void* outs[] = {var_gamma, var_mu, eta, q, eta_diff};
    const int out_fields[] = {VIPRS_FIELD_VAR_GAMMA, VIPRS_FIELD_VAR_MU, VIPRS_FIELD_ETA, VIPRS_FIELD_Q, VIPRS_FIELD_ETA_DIFF};
    for (int i = 0; i < 5; ++i)
        HIP_TRY(hipMemcpyAsync(outs[i], S->f[out_fields[i]].p, bytes, hipMemcpyDeviceToHost, P->stream));
    HIP_TRY(hipStreamSynchronize(P->stream));
    return check_device_error(P);
}

static int scratch_state(viprs_plan* P, int float_dtype, int model_kind, int width, viprs_state** out) {
    viprs_state* S = P->scratch;
    if (!S || S->float_dtype != float_dtype || S->model_kind != model_kind || S->width != width) {
        delete P->scratch;
        P->scratch = nullptr;
        int rc = viprs_state_create(&P->scratch, P, float_dtype, model_kind, width);
        if (rc != VIPRS_OK) return rc;
    }
    *out = P->scratch;
    return VIPRS_OK;
}

static int one_shot(viprs_state* S, const void* const* ins, const int* in_fields, int n_in, void* const* outs,
                    const int* out_fields, int n_out, double dq, const int32_t* active, int n_active) {
    viprs_plan* P = S->plan;
    const size_t fs = float_size(S->float_dtype);
    HIP_TRY(hipSetDevice(P->device));
    for (int i = 0; i < n_in; ++i) {
        const size_t bytes = S->field_elems(in_fields[i]) * fs;
        if (bytes == 0) continue;
        if (!ins[i]) return fail(VIPRS_EINVAL, "null buffer");
        HIP_TRY(hipMemcpyAsync(S->f[in_fields[i]].p, ins[i], bytes, hipMemcpyHostToDevice, P->stream));
    }
    int rc = viprs_state_e_step(S, dq, active, n_active, 0);
    if (rc != VIPRS_OK) return rc;
    for (int i = 0; i < n_out; ++i) {
        const size_t bytes = S->field_elems(out_fields[i]) * fs;
        HIP_TRY(hipMemcpyAsync(outs[i], S->f[out_fields[i]].p, bytes, hipMemcpyDeviceToHost, P->stream));
    }
    HIP_TRY(hipStreamSynchronize(P->stream));
    return VIPRS_OK;
}

int viprs_e_step_mixture(viprs_plan* P, int float_dtype, int K, const void* std_beta, void* var_gamma, void* var_mu,
                         void* eta, void* q, void* eta_diff, const void* log_null_pi, const void* u_logs,
                         const void* shvt, const void* mu_mult, double dq_scale, int threads, int low_memory) {
    (void)threads;
    if (!P) return fail(VIPRS_EINVAL, "null plan");
    if ((low_memory != 0) != (P->low_memory != 0))
        return fail(VIPRS_EINVAL, "low_memory differs from the value the plan was created with");
    if (K < 1) return fail(VIPRS_EINVAL, "K must be >= 1");
    if (P->m == 0) return VIPRS_OK;
    viprs_state* S = nullptr;
    int rc = scratch_state(P, float_dtype, VIPRS_MODEL_MIXTURE, K, &S);
    if (rc != VIPRS_OK) return rc;
    const void* ins[] = {std_beta, log_null_pi, u_logs, shvt, mu_mult, var_gamma, var_mu, eta, q, eta_diff};
    const int in_fields[] = {VIPRS_FIELD_STD_BETA, VIPRS_FIELD_LOG_NULL_PI, VIPRS_FIELD_U_LOGS,
                             VIPRS_FIELD_SQRT_HALF_VAR_TAU, VIPRS_FIELD_MU_MULT, VIPRS_FIELD_VAR_GAMMA,
                             VIPRS_FIELD_VAR_MU, VIPRS_FIELD_ETA, VIPRS_FIELD_Q, VIPRS_FIELD_ETA_DIFF};
    void* outs[] = {var_gamma, var_mu, eta, q, eta_diff};
    const int out_fields[] = {VIPRS_FIELD_VAR_GAMMA, VIPRS_FIELD_VAR_MU, VIPRS_FIELD_ETA, VIPRS_FIELD_Q, VIPRS_FIELD_ETA_DIFF};
    return one_shot(S, ins, in_fields, 10, outs, out_fields, 5, dq_scale, nullptr, 0);
}

int viprs_e_step_grid(viprs_plan* P, int float_dtype, int G, const void* std_beta, void* var_gamma, void* var_mu,
                      void* eta, void* q, void* eta_diff, const void* u_logs, const void* half_var_tau,
                      const void* mu_mult, double dq_scale, const int32_t* active_model_idx, int n_active, int threads,
                      int low_memory) {
    (void)threads;
    if (!P) return fail(VIPRS_EINVAL, "null plan");
    if ((low_memory != 0) != (P->low_memory != 0))
        return fail(VIPRS_EINVAL, "low_memory differs from the value the plan was created with");
    if (G < 1) return fail(VIPRS_EINVAL, "G must be >= 1");
    if (n_active < 0 || (n_active > 0 && !active_model_idx)) return fail(VIPRS_EINVAL, "bad active_model_idx");
    if (P->m == 0 || n_active == 0) return VIPRS_OK;
    viprs_state* S = nullptr;
    int rc = scratch_state(P, float_dtype, VIPRS_MODEL_GRID, G, &S);
    if (rc != VIPRS_OK) return rc;
    const void* ins[] = {std_beta, u_logs, half_var_tau, mu_mult, var_gamma, var_mu, eta, q, eta_diff};
    const int in_fields[] = {VIPRS_FIELD_STD_BETA, VIPRS_FIELD_U_LOGS, VIPRS_FIELD_SQRT_HALF_VAR_TAU, VIPRS_FIELD_MU_MULT,
                             VIPRS_FIELD_VAR_GAMMA, VIPRS_FIELD_VAR_MU, VIPRS_FIELD_ETA, VIPRS_FIELD_Q, VIPRS_FIELD_ETA_DIFF};
    void* outs[] = {var_gamma, var_mu, eta, q, eta_diff};
    const int out_fields[] = {VIPRS_FIELD_VAR_GAMMA, VIPRS_FIELD_VAR_MU, VIPRS_FIELD_ETA, VIPRS_FIELD_Q, VIPRS_FIELD_ETA_DIFF};
    return one_shot(S, ins, in_fields, 9, outs, out_fields, 5, dq_scale, active_model_idx, n_active);
}

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

static int grid_column_check(viprs_state* S, int g) {
    if (!S) return fail(VIPRS_EINVAL, "null state");
    if (S->model_kind != VIPRS_MODEL_GRID) return fail(VIPRS_EINVAL, "not a grid state");
    if (g < 0 || g >= S->width) return fail(VIPRS_EINVAL, "model index out of range");
    if (!S->d_n.p) return fail(VIPRS_EINVAL, "viprs_state_set_n_per_snp has not been called");
    return VIPRS_OK;
}

int viprs_state_prep_column(viprs_state* S, int g, double logit_pi, double log_tau_beta, double sigma_epsilon,
                            double tau_beta, double one_plus_lambda) {
    int rc = grid_column_check(S, g);
    if (rc != VIPRS_OK) return rc;
    viprs_plan* P = S->plan;
    if (P->m == 0) return VIPRS_OK;
    HIP_TRY(hipSetDevice(P->device));
    if (S->d_var_tau.n < (size_t)P->m * S->width) {      // one var_tau column per model
        HIP_TRY(hipStreamSynchronize(P->stream));
        HIP_TRY(S->d_var_tau.alloc((size_t)P->m * S->width));
    }
    const int64_t off = (int64_t)g * P->m;
    const unsigned grid = (unsigned)((P->m + 255) / 256);
    if (S->float_dtype == VIPRS_F32)
        prep_kernel<float><<<grid, 256, 0, P->stream>>>(S->d_n.p, P->m, logit_pi, log_tau_beta, sigma_epsilon, tau_beta,
                                                        one_plus_lambda, (float*)S->f[VIPRS_FIELD_MU_MULT].p + off,
                                                        (float*)S->f[VIPRS_FIELD_U_LOGS].p + off,
                                                        (float*)S->f[VIPRS_FIELD_SQRT_HALF_VAR_TAU].p + off,
                                                        S->d_var_tau.p + off, 1);
    else
        prep_kernel<double><<<grid, 256, 0, P->stream>>>(S->d_n.p, P->m, logit_pi, log_tau_beta, sigma_epsilon, tau_beta,
                                                         one_plus_lambda, (double*)S->f[VIPRS_FIELD_MU_MULT].p + off,
                                                         (double*)S->f[VIPRS_FIELD_U_LOGS].p + off,
                                                         (double*)S->f[VIPRS_FIELD_SQRT_HALF_VAR_TAU].p + off,
                                                         S->d_var_tau.p + off, 1);
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
    if (S->d_var_tau.n < (size_t)P->m * S->width) HIP_TRY(S->d_var_tau.alloc((size_t)P->m * S->width));
    if (S->d_colparams.n < (size_t)6 * S->width) HIP_TRY(S->d_colparams.alloc((size_t)6 * S->width));
    if (!S->h_params) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S->h_params), (size_t)8 * S->width * sizeof(double), hipHostMallocDefault));
    if (!S->ev_prep) HIP_TRY(hipEventCreateWithFlags(&S->ev_prep, hipEventDisableTiming));
    else HIP_TRY(hipEventSynchronize(S->ev_prep));            // the previous launch has read its parameters
    memcpy(S->h_params, params, (size_t)6 * n * sizeof(double));
    // pinned staging + copy ON the plan's stream: ordered with the kernel below (the null stream is not)
    HIP_TRY(hipMemcpyAsync(S->d_colparams.p, S->h_params, (size_t)6 * n * sizeof(double), hipMemcpyHostToDevice, P->stream));
    const dim3 grid((unsigned)((P->m + 255) / 256), (unsigned)n);
    if (S->float_dtype == VIPRS_F32)
        prep_columns_kernel<float><<<grid, 256, 0, P->stream>>>(S->d_n.p, P->m, S->d_colparams.p,
                                                                (float*)S->f[VIPRS_FIELD_MU_MULT].p, (float*)S->f[VIPRS_FIELD_U_LOGS].p,
                                                                (float*)S->f[VIPRS_FIELD_SQRT_HALF_VAR_TAU].p, S->d_var_tau.p);
    else
        prep_columns_kernel<double><<<grid, 256, 0, P->stream>>>(S->d_n.p, P->m, S->d_colparams.p,
                                                                 (double*)S->f[VIPRS_FIELD_MU_MULT].p, (double*)S->f[VIPRS_FIELD_U_LOGS].p,
                                                                 (double*)S->f[VIPRS_FIELD_SQRT_HALF_VAR_TAU].p, S->d_var_tau.p);
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
    if (P->m == 0 || n == 0) { S->sums_pending = false; S->sums_empty = true; return VIPRS_OK; }
    S->sums_empty = false;
    if (S->d_var_tau.n < (size_t)P->m * S->width) return fail(VIPRS_EINVAL, "viprs_state_prep_column(s) has not been called");
    HIP_TRY(hipSetDevice(P->device));
    // (own buffer: the previous reduction that read it has been collected, nothing else does)
    if (S->d_sumcols.n < (size_t)2 * S->width) HIP_TRY(S->d_sumcols.alloc((size_t)2 * S->width));
    if (!S->h_params) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S->h_params), (size_t)8 * S->width * sizeof(double), hipHostMallocDefault));
    memcpy(S->h_params + (size_t)6 * S->width, cols, (size_t)2 * n * sizeof(double));
    HIP_TRY(hipMemcpyAsync(S->d_sumcols.p, S->h_params + (size_t)6 * S->width, (size_t)2 * n * sizeof(double), hipMemcpyHostToDevice, P->stream));
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
    if (P->m == 0) return VIPRS_OK;
    if (S->d_var_tau.n < (size_t)P->m * S->width) return fail(VIPRS_EINVAL, "viprs_state_prep_column has not been called");
    HIP_TRY(hipSetDevice(P->device));
    const int64_t off = (int64_t)g * P->m;
    return S->float_dtype == VIPRS_F32 ? sums_launch<float>(S, off, off, one_plus_lambda, out)
                                       : sums_launch<double>(S, off, off, one_plus_lambda, out);
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

static int sweep_ms(viprs_plan* P, int64_t sweep, int which, double* ms) {
    hipEvent_t* ev = P->ev.data() + 4 * (sweep % viprs_plan::kRing);
    float t = 0.f;
    if (which == 1) {
        if (P->dense_h.empty()) { *ms = 0.0; return VIPRS_OK; }
        HIP_TRY(hipEventSynchronize(ev[3]));
        HIP_TRY(hipEventElapsedTime(&t, ev[2], ev[3]));
    } else {
        HIP_TRY(hipEventSynchronize(ev[1]));
        HIP_TRY(hipEventElapsedTime(&t, ev[0], ev[1]));
    }
    *ms = t;
    return VIPRS_OK;
}

int viprs_plan_last_kernel_ms(viprs_plan* P, int which, double* ms) {
    if (!P || !ms) return fail(VIPRS_EINVAL, "null argument");
    if (P->sweeps == 0) return fail(VIPRS_EINVAL, "no timed sweep yet");
    HIP_TRY(hipSetDevice(P->device));
    return sweep_ms(P, P->sweeps - 1, which, ms);
}

int viprs_plan_timing_reset(viprs_plan* P) {
    if (!P) return fail(VIPRS_EINVAL, "null plan");
    HIP_TRY(hipSetDevice(P->device));
    HIP_TRY(hipStreamSynchronize(P->stream));
    P->sweeps = 0;
    return VIPRS_OK;
}

int viprs_plan_timing_history(viprs_plan* P, int which, double* ms, int capacity, int* n) {
    if (!P || !ms || !n) return fail(VIPRS_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(P->device));
    const int64_t have = std::min<int64_t>(P->sweeps, viprs_plan::kRing);
    const int count = (int)std::min<int64_t>(have, capacity);
    for (int i = 0; i < count; ++i) {
        int rc = sweep_ms(P, P->sweeps - count + i, which, &ms[i]);
        if (rc != VIPRS_OK) return rc;
    }
    *n = count;
    return VIPRS_OK;
}

int viprs_plan_last_skipped(viprs_plan* P, int64_t* n) {
    if (!P || !n) return fail(VIPRS_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(P->device));
    HIP_TRY(hipStreamSynchronize(P->stream));
    unsigned long long v = 0;
    HIP_TRY(hipMemcpy(&v, P->d_skipped.p, sizeof(v), hipMemcpyDeviceToHost));
    *n = (int64_t)v;
    return VIPRS_OK;
}

}  // extern "C"
