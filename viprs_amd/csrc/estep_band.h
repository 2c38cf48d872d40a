// Band kernel: the panel scheme of estep_panel.h for WINDOWED LD components (banded matrices from
// windowed / shrinkage LD estimators, ragged windows of any shape) -- what e_step<T,U,I> walks row by
// row through (ld_left_bound, ld_indptr) in e_step.hpp:387-433.
//
// A component of b SNPs is one Gauss-Seidel chain of b steps whatever its bandwidth, so one workgroup
// serves it: wave 0 runs the 64 serial updates of a panel against its diagonal tile (LDS), the other
// waves apply the previous panel's scaled eta_diff to the columns its rows reach -- at most
// `band_left` panels to the left (symmetric form) and `band_right` panels to the right -- and stage the
// next diagonal / next off-diagonal tile.  Rows are read in the caller's own concatenated layout
// (rowstart / lb / rowlen); an element outside its row's window is 0, and fma(0, a, q) == q is exactly
// "the reference never touches it".  q lives in an LDS RING of panels: a panel enters when the first row
// that reaches it is one phase away and is written back after the last row that reaches it.
//
// Order of the accumulation into any q[c] is row order, as in the reference: contributions of panel pp
// are applied in phase pp + 1 (other panels), by the T-tile at the start of phase pp + 1 (panel pp + 1)
// or inside the chain (panel pp itself), and phases are separated by workgroup barriers.
#pragma once
#include "estep_panel.h"

namespace viprs {

constexpr int kBandWaves = 8;                  // chain wave, 2 stagers, 4 strip waves, 1 spare (8 waves: 256 VGPRs for the chain)
constexpr int kBandStagers = 2;                // waves 1 .. 2: next diagonal + off-diagonal tile -> LDS
constexpr int kBandIdleWave = 4;               // shares its SIMD with the chain wave (waves go round-robin over the 4
                                               // SIMDs): it only keeps the barriers, so the chain issues alone
constexpr int kBandUpdaters = kBandWaves - 2 - kBandStagers;   // the rest: kBandTargets target panels each per round
constexpr int kBandTargets = 2;                // target panels per strip wave and round

// LDS carve (floats): q ring[ring_panels][64] | a[2][64] | T[2][64*64] | D[2][64*64]
__host__ __device__ constexpr int band_lds_floats(int ring_panels) {
    return ring_panels * kPanel + 2 * kPanel + 4 * kPanel * kPanel;
}

// Row windows of one panel: lane k describes row k (component-local window start; len 0 past the end of
// the component).  Loaded one phase ahead of their first use, handed from phase to phase in registers.
struct BandRows {
    int rs_lo, rs_hi, lb, len;
};
__device__ __forceinline__ BandRows band_rows(const EStepArgs<float>& A, int64_t s0, int b, int panel, int lane) {
    const int row = panel * kPanel + lane;
    const bool ok = row < b;
    const int64_t rs = ok ? A.rowstart[s0 + row] : 0;
    BandRows d;
    d.rs_lo = (int)(unsigned)(rs & 0xffffffffll);
    d.rs_hi = (int)(rs >> 32);
    d.lb = ok ? A.lb[s0 + row] - (int)s0 : 0;
    d.len = ok ? A.rowlen[s0 + row] : 0;
    return d;
}
__device__ __forceinline__ int64_t band_rowstart(const BandRows& d, int r) {
    return ((int64_t)__builtin_amdgcn_readlane(d.rs_hi, r) << 32) | (int64_t)(unsigned)__builtin_amdgcn_readlane(d.rs_lo, r);
}

// The two tiles the chain needs next phase -> LDS as float, 0 outside a row's window: T = rows `dT` (panel
// p) x columns of panel p + 1, D = rows `dD` (panel p + 1) x the same columns.  The 128 rows are dealt
// round-robin to the stager waves; every load is issued before the first is consumed (ONE memory round
// trip per phase) and none sits behind a guard (indices are clamped into the row, the select comes after).
template <typename U>
__device__ __forceinline__ void band_stage_tiles(const U* __restrict__ ld, const BandRows& dT, const BandRows& dD, int cp,
                                                 float* __restrict__ dstT, float* __restrict__ dstD, int sw, int lane) {
    constexpr int RPW = (2 * kPanel + kBandStagers - 1) / kBandStagers;
    const int c = cp * kPanel + lane;
    float v[RPW];
    unsigned long long okm = 0;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int t = min(sw + i * kBandStagers, 2 * kPanel - 1);             // wave-uniform
        const bool inT = t < kPanel;
        const int r = t & (kPanel - 1);
        const int len = inT ? __builtin_amdgcn_readlane(dT.len, r) : __builtin_amdgcn_readlane(dD.len, r);
        const int off = c - (inT ? __builtin_amdgcn_readlane(dT.lb, r) : __builtin_amdgcn_readlane(dD.lb, r));
        const int64_t rs = inT ? band_rowstart(dT, r) : band_rowstart(dD, r);
        v[i] = static_cast<float>(ld[rs + max(0, min(off, len - 1))]);
        okm |= (off >= 0 && off < len) ? (1ull << i) : 0ull;
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int t = sw + i * kBandStagers;
        if (t < 2 * kPanel) {
            float* __restrict__ dst = (t < kPanel) ? dstT : dstD;
            dst[(t & (kPanel - 1)) * kPanel + lane] = ((okm >> i) & 1ull) ? v[i] : 0.0f;
        }
    }
}

// q of one column (lane) of NT target panels after the 64 rows `d` of panel pp: q = fma(R[row][c], a_row, q)
// in row order, 64 row loads in flight per lane; the row windows are broadcast once per row for all NT targets.
template <typename U, int NT>
__device__ __forceinline__ void band_strip(const U* __restrict__ ld, const BandRows& d, const int (&c)[NT], float avec,
                                           float (&qv)[NT]) {
    constexpr int DEPTH = 64 / NT;          // 64 loads in flight per lane
#pragma unroll
    for (int g = 0; g < kPanel / DEPTH; ++g) {
        float x[NT][DEPTH];
        unsigned okm[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) okm[n] = 0;
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const int r = g * DEPTH + k;
            const int len = __builtin_amdgcn_readlane(d.len, r);
            const int lb = __builtin_amdgcn_readlane(d.lb, r);
            const U* __restrict__ row = ld + band_rowstart(d, r);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int off = c[n] - lb;
                x[n][k] = static_cast<float>(row[max(0, min(off, len - 1))]);
                okm[n] |= (off >= 0 && off < len) ? (1u << k) : 0u;
            }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const float a = rl(avec, g * DEPTH + k);
#pragma unroll
            for (int n = 0; n < NT; ++n) qv[n] = __builtin_fmaf(((okm[n] >> k) & 1u) ? x[n][k] : 0.0f, a, qv[n]);
        }
    }
}

// The mixture model with the K components of a SNP evaluated one after the other in the SNP's own lane
// (MixtureModel's lane-per-SNP update / finish: the same arithmetic as its lane-parallel chain in the
// panel kernels, K + 1 expf and K divides per step instead of one each).
template <bool EXACT = true>
struct MixtureSerialModel : MixtureModel<EXACT> {
    static constexpr bool kLaneParallel = false;
};

template <typename U, typename MODEL, bool SYM>
__global__ __launch_bounds__(64 * kBandWaves) void estep_band_kernel(EStepArgs<float> A0, int ring_panels) {
    static_assert(!MODEL::kLaneParallel, "band kernel: lane-per-SNP model policies only");
    extern __shared__ __attribute__((aligned(16))) float band_smem[];
    __shared__ int s_blk;
    ExpTab tab;
    tab.init();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int RM = ring_panels - 1;                        // power of two
    float* __restrict__ lq = band_smem;
    float* __restrict__ la = lq + ring_panels * kPanel;
    float* __restrict__ lT = la + 2 * kPanel;
    float* __restrict__ lD = lT + 2 * kPanel * kPanel;
    const U* __restrict__ ld = static_cast<const U*>(A0.ld_rows);
    const int n_models = max(A0.n_active, 1);
    const float dq = A0.dq;
    unsigned long long my_skipped = 0;

    for (;;) {
        if (tid == 0) s_blk = atomicAdd(A0.counter, 1);
        __syncthreads();
        const int item = s_blk;
        __syncthreads();
        if (item >= A0.n_blocks * n_models) break;
        const int blk = item / n_models, model_slot = item - blk * n_models;
        const EStepArgs<float> A = select_model(A0, model_slot);
        const BlockDesc bd = A.blocks[blk];
        const int64_t s0 = bd.start;
        const int b = bd.size;
        const int WL = bd.band_left, WR = bd.band_right;   // reach in panels (>= 1)
        const int np = (b + kPanel - 1) / kPanel;

        // panels 0 .. WR-1 of q, diagonal tile of panel 0
        for (int i = tid; i < min(WR, np) * kPanel; i += 64 * kBandWaves)
            lq[((i >> 6) & RM) * kPanel + (i & 63)] = (i < b) ? A.q[s0 + i] : 0.0f;
        // updater waves: row windows of panels p - 1 / p / p + 1 (rows applied / staged this phase)
        BandRows d_prev{}, d_cur{}, d_next{};
        if (wave > 0) {
            d_cur = band_rows(A, s0, b, 0, lane);
            d_next = band_rows(A, s0, b, 1, lane);
            // diagonal tile of panel 0 (the T half of the call lands in the unused T buffer 0)
            if (wave <= kBandStagers) band_stage_tiles<U>(ld, d_cur, d_cur, 0, lT, lD, wave - 1, lane);
        }
        __syncthreads();

        float a_prev = 0.0f;
        typename MODEL::In nxt_in{};
        if (wave == 0) nxt_in = MODEL::load(A, s0 + (lane < b ? lane : 0), lane < b);

        for (int p = 0; p < np + (SYM ? 1 : 0); ++p) {
            if (wave == 0) {
                // ================================ chain ======================================
                if (p < np) {
                    const int r0 = p * kPanel;
                    const int nrows = min(kPanel, b - r0);
                    const int64_t j = s0 + r0 + lane;
                    const bool live = lane < nrows;
                    const typename MODEL::In in = nxt_in;
                    {
                        const int rn = r0 + kPanel + lane;
                        const bool ln = rn < b;
                        nxt_in = MODEL::load(A, s0 + (ln ? rn : 0), ln);
                    }
                    float* __restrict__ slot = lq + (p & RM) * kPanel + lane;
                    float qc = *slot;
                    const float* __restrict__ Dt = lD + (p & 1) * kPanel * kPanel + lane;
                    float drow[kPanel];
#pragma unroll
                    for (int k = 0; k < kChainPrefetch; ++k) drow[k] = Dt[k * kPanel];
                    if (p > 0) {
                        // a_{p-1} through tile R[p-1, p]
                        const float* __restrict__ T = lT + (p & 1) * kPanel * kPanel;
#pragma unroll
                        for (int k = 0; k < kPanel; ++k) qc = __builtin_fmaf(T[k * kPanel + lane], rl(a_prev, k), qc);
                    }
                    float qcap_v = 0.0f;
#pragma unroll
                    for (int jj = 0; jj < kPanel; ++jj) {
                        if (jj + kChainPrefetch < kPanel) drow[jj + kChainPrefetch] = Dt[(jj + kChainPrefetch) * kPanel];
                        float d;
                        const bool upd = MODEL::template update<kLookupLane>(in, qc, tab, d, jj) && live;
                        const float a_lane = upd ? dq * d : 0.0f;
                        int l = lane;
                        asm volatile("" : "+v"(l));
                        const bool me = (l == jj);
                        qcap_v = me ? qc : qcap_v;
                        qc = __builtin_fmaf(drow[jj], rl(a_lane, jj), qc);
                        if (SYM) qc = (me && upd) ? qc - d : qc;          // e_step.hpp:427
                    }
                    bool skipped_lane;
                    a_prev = MODEL::template finish<false>(A, j, in, qcap_v, tab, live, true, skipped_lane);
                    my_skipped += __popcll(__ballot(skipped_lane));
                    la[(p & 1) * kPanel + lane] = a_prev;
                    *slot = qc;
                    if (!SYM && live) A.q[j] = qc;      // upper form: nothing to the left is updated in this pass
                }
            } else {
                // ================================ updaters ===================================
                const BandRows d_fetch = band_rows(A, s0, b, p + 2, lane);      // consumed next phase
                if (wave <= kBandStagers) {
                    if (p + 1 < np)
                        band_stage_tiles<U>(ld, d_cur, d_next, p + 1, lT + ((p + 1) & 1) * kPanel * kPanel,
                                            lD + ((p + 1) & 1) * kPanel * kPanel, wave - 1, lane);
                } else if (p > 0) {
                    // strip waves 3, 5, 6, 7 -> 0 .. 3; wave 4 (the chain's SIMD) joins as number 4 only when a
                    // phase has more targets than the other four take in one round (wide bands: strip-bound anyway)
                    const int uw = (wave == kBandIdleWave) ? kBandUpdaters : wave - 1 - kBandStagers - (wave > kBandIdleWave ? 1 : 0);
                    const int pp = p - 1;
                    const float avec = la[(pp & 1) * kPanel + lane];
                    const bool any_a = __ballot(avec != 0.0f) != 0;
                    // targets: every panel the rows of pp reach except pp itself (the chain did it) and p (the
                    // T tile does it) -- [pp - WL, pp) and (p, pp + WR]; kBandTargets of them per wave and round
                    const int first = SYM ? max(0, pp - WL) : p + 1;
                    const int last = min(np - 1, pp + WR);
                    const int n_targets = last - first + 1 - ((SYM && p < np) ? 2 : (SYM ? 1 : 0));
                    const int n_upd = (n_targets > kBandUpdaters * kBandTargets) ? kBandUpdaters + 1 : kBandUpdaters;
                    for (int t0 = uw * kBandTargets; t0 < n_targets && uw < n_upd; t0 += n_upd * kBandTargets) {
                        int cp[kBandTargets], c[kBandTargets];
                        float qv[kBandTargets];
#pragma unroll
                        for (int n = 0; n < kBandTargets; ++n) {
                            int x = first + t0 + n;
                            if (SYM && x >= pp) x += 2;
                            cp[n] = (t0 + n < n_targets) ? x : -1;
                            c[n] = (cp[n] >= 0) ? cp[n] * kPanel + lane : -(1 << 28);     // no row reaches it
                            qv[n] = (cp[n] >= 0) ? lq[(cp[n] & RM) * kPanel + lane] : 0.0f;
                        }
                        if (any_a) band_strip<U, kBandTargets>(ld, d_prev, c, avec, qv);
#pragma unroll
                        for (int n = 0; n < kBandTargets; ++n) {
                            if (cp[n] < 0) continue;
                            lq[(cp[n] & RM) * kPanel + lane] = qv[n];
                            // symmetric form: no later row reaches this panel -> it is final
                            if (SYM && p == min(cp[n] + WL + 1, np) && c[n] < b) A.q[s0 + c[n]] = qv[n];
                        }
                    }
                    // the last panel has no phase of its own in which it is a target
                    if (SYM && p == np && uw == 0 && pp * kPanel + lane < b)
                        A.q[s0 + pp * kPanel + lane] = lq[(pp & RM) * kPanel + lane];
                }
                d_prev = d_cur;
                d_cur = d_next;
                d_next = d_fetch;
                if (wave == 1) {
                    // the panel first touched in the next phase enters the ring
                    const int cn = p + WR;
                    if (cn < np) {
                        const int c = cn * kPanel + lane;
                        lq[(cn & RM) * kPanel + lane] = (c < b) ? A.q[s0 + c] : 0.0f;
                    }
                }
            }
            __syncthreads();
        }

        // (upper form: the second pass, update_q_factor, is band_upper_epilogue_kernel)
    }
    if (lane == 0 && my_skipped) atomicAdd(A0.skipped, my_skipped);
}

// Upper-triangular form, second pass (update_q_factor, e_step.hpp:331-337) of the windowed components:
// q[j] += dq * dot(row(j), eta_diff[win(j)]), the dot a serial fma chain from 0 in index order (:100-102),
// one thread per row.  blockIdx.y = (component, model) item.
template <typename U>
__global__ __launch_bounds__(256) void band_upper_epilogue_kernel(EStepArgs<float> A0) {
    const int n_models = max(A0.n_active, 1);
    const int blk = blockIdx.y / n_models, model_slot = blockIdx.y - blk * n_models;
    const EStepArgs<float> A = select_model(A0, model_slot);
    const BlockDesc bd = A.blocks[blk];
    const U* __restrict__ ld = static_cast<const U*>(A.ld_rows);
    for (int jj = blockIdx.x * 256 + threadIdx.x; jj < bd.size; jj += gridDim.x * 256) {
        const int64_t j = (int64_t)bd.start + jj;
        const U* __restrict__ row = ld + A.rowstart[j];
        const float* __restrict__ ed = A.eta_diff + A.lb[j];
        const int len = A.rowlen[j];
        float s = 0.0f;
        for (int i = 0; i < len; ++i) s = __builtin_fmaf(static_cast<float>(row[i]), ed[i], s);
        A.q[j] += A.dq * s;
    }
}

}  // namespace viprs
