// Panel kernel for dense LD blocks (spike-and-slab, T = float): the MI355X hot path.
//
// One workgroup owns one LD block; blocks are pulled from a work queue in descending cost order
// by persistent workgroups.  The serial Gauss-Seidel sweep of e_step<T,U,I>
// (e_step.hpp:387-433) is re-blocked into panels of 64 SNPs (= one wavefront) without changing
// a single floating-point operation or its order per q-entry:
//
//   wave 0 ("chain")     for panel p: q_p  <- LDS; apply a_{p-1} through the 64x64 tile
//                        R[p-1, p] (64 ordered fma per lane); then the 64 serial SNP updates
//                        against the diagonal tile R[p, p] -- every lane computes the scalar
//                        chain redundantly, lane i owns q[p*64 + i], the diagonal-tile rows are
//                        streamed from HBM 16 rows ahead of their use.
//   waves 1.. ("updaters") while the chain solves panel p they apply a_{p-1} -- the trailing
//                        rank-64 update -- to every column RIGHT of panel p: each lane owns 4
//                        columns, walks the 64 rows of panel p-1 in order (one coalesced 16-byte
//                        load per lane per row) and performs q[c] = fma(R[j][c], a_j, q[c]);
//                        they also stage tile R[p, p+1] into LDS for the chain's next phase.
//
// Symmetric form: columns LEFT of the current panel (q of SNPs already visited) keep receiving the
// later rows as well (they are what the next sweep starts from).  Upper-triangular form: the
// reference's second pass (update_q_factor, e_step.hpp:331-337) is folded into the sweep: the dense
// blocks hold the upper triangle MIRRORED into the lower one (kFormMirror) and the columns left of the
// chain take the later rows exactly as in the symmetric form -- into per-row sums s, with eta_diff as
// the multiplier: R[i, j] read as R[j, i], a row's sum in ascending column order, the reference's dot
// bit for bit.
//
// Large blocks (a single CU pulls only ~50 GB/s from HBM) are shared by a TEAM of TS workgroups on
// TS CUs: see the comment at `team` in the kernel.
#pragma once
#include <type_traits>

#include "device_math.h"
#include "kernels_common.h"

#ifndef PANEL_MIN_WAVES
#define PANEL_MIN_WAVES 2      // waves per SIMD the panel kernel is compiled for (register budget 512 / n): two workgroups per CU
#endif

namespace viprs {

#ifdef VIPRS_SWEEP_TRACE
// profiling builds: one record {workgroup, team flag, block size, start, end (100 MHz wall clock)} per (workgroup, block)
__device__ unsigned long long g_sweep_trace[1 << 15][4];
__device__ unsigned int g_sweep_trace_n;
#endif

#ifdef VIPRS_PANEL_PROFILE
#define PPROF(slot, cond) do { if (TEAM && item == 0 && member < 2 && lane == 0 && p < 64 && (cond)) s_pprof[p][slot] = (unsigned)wall_clock64(); } while (0)
#else
#define PPROF(slot, cond) do { } while (0)
#endif

// ---- 4-element row loads, converted with static_cast<float> as e_step.hpp:173 does ----------
template <typename U> __device__ __forceinline__ float4 load4(const U* p);
template <> __device__ __forceinline__ float4 load4<float>(const float* p) {
    return *reinterpret_cast<const float4*>(p);
}
template <> __device__ __forceinline__ float4 load4<int8_t>(const int8_t* p) {
    const int w = *reinterpret_cast<const int*>(p);
    return make_float4((float)(int8_t)(w), (float)(int8_t)(w >> 8), (float)(int8_t)(w >> 16),
                       (float)(int8_t)(w >> 24));
}
template <> __device__ __forceinline__ float4 load4<int16_t>(const int16_t* p) {
    const int2 w = *reinterpret_cast<const int2*>(p);
    return make_float4((float)(int16_t)(w.x), (float)(int16_t)(w.x >> 16), (float)(int16_t)(w.y),
                       (float)(int16_t)(w.y >> 16));
}

// CPL consecutive columns of one row as floats (CPL = 4: one 16-byte load for f32)
template <int CPL> struct RowVec { float v[CPL]; };

template <typename U, int CPL> __device__ __forceinline__ RowVec<CPL> load_cols(const U* p);
template <> __device__ __forceinline__ RowVec<4> load_cols<float, 4>(const float* p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    return RowVec<4>{{t.x, t.y, t.z, t.w}};
}
template <> __device__ __forceinline__ RowVec<8> load_cols<float, 8>(const float* p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    const float4 u = *reinterpret_cast<const float4*>(p + 4);
    return RowVec<8>{{t.x, t.y, t.z, t.w, u.x, u.y, u.z, u.w}};
}
template <> __device__ __forceinline__ RowVec<8> load_cols<int8_t, 8>(const int8_t* p) {
    const float4 t = load4<int8_t>(p), u = load4<int8_t>(p + 4);
    return RowVec<8>{{t.x, t.y, t.z, t.w, u.x, u.y, u.z, u.w}};
}
template <> __device__ __forceinline__ RowVec<8> load_cols<int16_t, 8>(const int16_t* p) {
    const float4 t = load4<int16_t>(p), u = load4<int16_t>(p + 4);
    return RowVec<8>{{t.x, t.y, t.z, t.w, u.x, u.y, u.z, u.w}};
}
template <> __device__ __forceinline__ RowVec<2> load_cols<float, 2>(const float* p) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    return RowVec<2>{{t.x, t.y}};
}
template <> __device__ __forceinline__ RowVec<1> load_cols<float, 1>(const float* p) { return RowVec<1>{{*p}}; }
template <> __device__ __forceinline__ RowVec<4> load_cols<int8_t, 4>(const int8_t* p) {
    const float4 t = load4<int8_t>(p);
    return RowVec<4>{{t.x, t.y, t.z, t.w}};
}
template <> __device__ __forceinline__ RowVec<2> load_cols<int8_t, 2>(const int8_t* p) {
    const short w = *reinterpret_cast<const short*>(p);
    return RowVec<2>{{(float)(int8_t)(w), (float)(int8_t)(w >> 8)}};
}
template <> __device__ __forceinline__ RowVec<1> load_cols<int8_t, 1>(const int8_t* p) { return RowVec<1>{{(float)*p}}; }
template <> __device__ __forceinline__ RowVec<4> load_cols<int16_t, 4>(const int16_t* p) {
    const float4 t = load4<int16_t>(p);
    return RowVec<4>{{t.x, t.y, t.z, t.w}};
}
template <> __device__ __forceinline__ RowVec<2> load_cols<int16_t, 2>(const int16_t* p) {
    const int w = *reinterpret_cast<const int*>(p);
    return RowVec<2>{{(float)(int16_t)(w), (float)(int16_t)(w >> 16)}};
}
template <> __device__ __forceinline__ RowVec<1> load_cols<int16_t, 1>(const int16_t* p) { return RowVec<1>{{(float)*p}}; }

// value of lane - N within a row of 16 lanes (v_mov_b32_dpp row_shr:N); lanes without a source get `fill`
template <int N> __device__ __forceinline__ float dpp_shr(float v, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x110 + N, 0xf, 0xf, false));
}

// max(x[lane - N], x[lane]) within a row of 16 lanes, lanes without a source keep x (no NaN canonicalisation:
// the operands are finite); the two wait states a DPP read of a fresh VALU result needs are in the asm
// Keeps N wave-uniform values (v_readlane results) in SGPRs at this point of the program: the reads are issued together and
// the scalar chain that consumes them follows without the two wait states a VALU read of a just-written SGPR costs per term.
template <int N> __device__ __forceinline__ void pin_sgprs(int (&v)[N]) {
    static_assert(N == 4 || N == 5 || N == 8 || N == 9, "chain lengths of the K <= 8 mixture step");
    if constexpr (N == 4) asm volatile("" : "+s"(v[0]), "+s"(v[1]), "+s"(v[2]), "+s"(v[3]));
    if constexpr (N == 5) asm volatile("" : "+s"(v[0]), "+s"(v[1]), "+s"(v[2]), "+s"(v[3]), "+s"(v[4]));
    if constexpr (N == 8) asm volatile("" : "+s"(v[0]), "+s"(v[1]), "+s"(v[2]), "+s"(v[3]), "+s"(v[4]), "+s"(v[5]), "+s"(v[6]), "+s"(v[7]));
    if constexpr (N == 9) asm volatile("" : "+s"(v[0]), "+s"(v[1]), "+s"(v[2]), "+s"(v[3]), "+s"(v[4]), "+s"(v[5]), "+s"(v[6]), "+s"(v[7]),
                                       "+s"(v[8]));
}
template <int N> __device__ __forceinline__ float dpp_max_shr(float x) {
    float r;
    if (N == 1) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "0"(x));
    if (N == 2) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "0"(x));
    if (N == 4) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "0"(x));
    if (N == 8) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "0"(x));
    return r;
}

// r = mask[lane] ? b : a with the lane mask in an SGPR pair (one v_cndmask, no per-step v_cmp)
__device__ __forceinline__ float sel_mask(float a, float b, unsigned long long m) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
    return r;
}

// staged outputs of a team block: another member -- possibly on another XCD, behind another L2 -- copies them into place
// when the block is done, so they are written past the caches (agent scope)
__device__ __forceinline__ void stage_store(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ float rl(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// CPL consecutive columns of one LD row exactly as stored (float / int8 / int16), converted with
// static_cast<float> (e_step.hpp:173) only when consumed: the prefetch buffer of an updater lane holds
// raw bytes, so a 16-byte load brings 4 fp32, 8 int16 or 16 int8 columns.
template <typename U, int CPL> struct RawRow {
    static constexpr int kWords = CPL * (int)sizeof(U) / 4;
    static_assert(CPL * sizeof(U) % 4 == 0 && (kWords == 1 || kWords == 2 || kWords == 4), "1, 2 or 4 dwords per lane and row");
    unsigned w[kWords];
    __device__ __forceinline__ float get(int i) const {
        if constexpr (sizeof(U) == 4) return __uint_as_float(w[i]);
        else if constexpr (sizeof(U) == 1) return static_cast<float>(static_cast<int8_t>(w[i >> 2] >> (8 * (i & 3))));
        else return static_cast<float>(static_cast<int16_t>(w[i >> 1] >> (16 * (i & 1))));
    }
};
template <typename U, int CPL> __device__ __forceinline__ RawRow<U, CPL> load_raw(const U* p) {
    RawRow<U, CPL> r;
    if constexpr (RawRow<U, CPL>::kWords == 1) {
        r.w[0] = *reinterpret_cast<const unsigned*>(p);
    } else if constexpr (RawRow<U, CPL>::kWords == 2) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        r.w[0] = t.x; r.w[1] = t.y;
    } else {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        r.w[0] = t.x; r.w[1] = t.y; r.w[2] = t.z; r.w[3] = t.w;
    }
    return r;
}
// columns per updater lane: 16-byte loads for fp32 and int16, 8-byte loads for int8 (wider strips leave
// too few updater waves per block busy)
template <typename U> __host__ __device__ constexpr int panel_cols() { return sizeof(U) == 4 ? 4 : 8; }
// ... and per updater lane of a LARGE TEAM's workgroup in the upper-triangular form (estep_sweep_kernel picks the role).
// A team block is the sweep's critical path, and part of what bounds its phase is the memory LATENCY of its updater
// waves: a strip of 64 x CPL columns is one wave's unit of work per phase (the fma chain of a column takes its rows in
// order), its 64 rows come in 64 / DEPTH round trips of ~4 us under load (25 MB of row loads in flight on the chip), and
// with 4 columns per lane the 3 619-SNP block of cfg3 has 15 strips for the 36 updater waves of its 12 members.  Narrower
// strips with proportionally more rows in flight (the same 64 VGPRs of row data per lane) give every wave of the team
// work and halve / quarter the round trips: 8-byte loads and 32 rows in flight for fp32 / int16 LD, 4-byte loads and all
// 64 rows in flight for int8.  Worth 1.5-3 % where the updaters also carry the second pass (the upper form); the
// symmetric form, whose sweep is bound by the stream, does better with the 16-byte loads.
template <typename U> __host__ __device__ constexpr int panel_team_cols() {
#ifdef PANEL_TEAM_CPL_F32
    return sizeof(U) == 4 ? PANEL_TEAM_CPL_F32 : 4;      // (experiments)
#else
    return sizeof(U) == 4 ? 2 : 4;
#endif
}

constexpr int kChainPrefetch = 16;   // diagonal-tile rows in flight ahead of the serial chain
#ifndef PANEL_STRIP_DEPTH
#define PANEL_STRIP_DEPTH 16
#endif
constexpr int kStripRowsInFlight = PANEL_STRIP_DEPTH;   // row loads in flight per updater lane (x 16 B for every LD type)


// Trailing update of one strip (64 * CPL columns) by one wave: q[c..c+CPL-1] = fma(R[row][c..], a_row, .)
// for the 64 rows of a panel, in row order, with DEPTH row loads in flight per lane (DEPTH * CPL = 64
// floats of row data per lane whatever the strip width).  The row loop is rolled in groups of DEPTH
// so that every load is consumed exactly one group later (a fully unrolled loop lets hipcc sink the
// loads next to their uses, leaving two in flight), and there is no runtime guard around any load
// (a guard makes hipcc wait vmcnt(0) per row).  FULL = false (partial last panel of a block): rows
// past its end are clamped to its last row; their a is 0, so fma(R, 0, q) == q leaves q untouched.
// MIXED (mirrored upper form, a strip with columns on both sides of the chain): the multiplier of row j is per lane,
// fvec * avec[j] -- avec = eta_diff of the panel, fvec = 1 for a column left of the chain (a term of its second-pass sum),
// dq right of it (dq * eta_diff[j] IS a_j, the same product the chain formed): one v_mul per row more.
template <typename U, int CPL, bool FULL, int DEPTH = kStripRowsInFlight, bool MIXED = false>
__device__ __forceinline__ void strip_update(const U* __restrict__ rowp, int stride, int last_row, float avec,
                                             float* __restrict__ lq_c, float fvec = 1.0f) {
    static_assert(kPanel % DEPTH == 0, "panel must be a whole number of prefetch groups");
    float qv[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) qv[i] = lq_c[i];
    RawRow<U, CPL> buf[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k)
        buf[k] = load_raw<U, CPL>(rowp + (int64_t)(FULL ? k : min(k, last_row)) * stride);
#pragma unroll 1
    for (int g = 0; g < kPanel / DEPTH - 1; ++g) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const RawRow<U, CPL> v = buf[k];
            const int rn = DEPTH * (g + 1) + k;
            buf[k] = load_raw<U, CPL>(rowp + (int64_t)(FULL ? rn : min(rn, last_row)) * stride);
            const float a = MIXED ? fvec * rl(avec, DEPTH * g + k) : rl(avec, DEPTH * g + k);
#pragma unroll
            for (int i = 0; i < CPL; ++i) qv[i] = __builtin_fmaf(v.get(i), a, qv[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
        const RawRow<U, CPL> v = buf[k];
        const float a = MIXED ? fvec * rl(avec, kPanel - DEPTH + k) : rl(avec, kPanel - DEPTH + k);
#pragma unroll
        for (int i = 0; i < CPL; ++i) qv[i] = __builtin_fmaf(v.get(i), a, qv[i]);
    }
#pragma unroll
    for (int i = 0; i < CPL; ++i) lq_c[i] = qv[i];
}

// Mirrored upper form: the terms of the second-pass sums that lie INSIDE a diagonal tile -- s[i] += R[i, j] ed[j] for the
// SNPs i < j of one panel, read as R[j, i] from row j (lane = column i), rows in ascending order.  The tile is the one the
// chain has just swept, still in LDS (fp32, staged rows past a partial last panel clamped; their eta_diff is 0).
__device__ __forceinline__ float diag_lower_update(const float* __restrict__ tile, float edvec, float sv, int lane) {
#pragma unroll 16
    for (int jr = 0; jr < kPanel; ++jr) {
        const float t = __builtin_fmaf(tile[jr * kPanel + lane], rl(edvec, jr), sv);
        sv = lane < jr ? t : sv;
    }
    return sv;
}

// ---------------------------------------------------------------------------------------------
// Model policies: what one SNP update computes (the serial chain evaluates `update` with the
// lane-select table lookup; after the 64 steps every lane replays its own SNP with the per-lane
// lookup -- same operations, same inputs, same bits -- and `finish` stores the outputs).
//   load    per-SNP inputs of SNP j (lane-resident for a whole panel)
//   update  d = new eta - old eta from the current q[j];  returns false on the skip branch
//   finish  replay + stores; returns the scaled eta_diff (0 for skipped SNPs)
// ---------------------------------------------------------------------------------------------
template <bool EXACT>
struct SpikeSlabModel {                      // e_step, e_step.hpp:387-433
    static constexpr bool kLaneParallel = false;
    struct In { float mm, beta, sv, ulog, eta_old; };
    static __device__ __forceinline__ In load(const EStepArgs<float>& A, int64_t j, bool live) {
        In in;
        in.mm = live ? A.mu_mult[j] : 0.0f;
        in.beta = live ? A.std_beta[j] : 0.0f;
        in.sv = live ? A.shvt[j] : 0.0f;
        in.ulog = live ? A.u_logs[j] : 0.0f;
        in.eta_old = live ? A.eta[j] : 0.0f;
        return in;
    }
    static constexpr bool kHasSkip = true;     // e_step.hpp:410-413
    // d = new eta - old eta of the lane's SNP from the current q (no skip handling)
    template <int LOOKUP>
    static __device__ __forceinline__ float delta(const In& in, float q, const ExpTab& tab, int sel) {
        float mu, gamma, d;
        snp_update<EXACT, LOOKUP>(in.mm, in.beta, in.sv, in.ulog, in.eta_old, q, tab, mu, gamma, d, sel);
        return d;
    }
    template <int LOOKUP>
    static __device__ __forceinline__ bool update(const In& in, float q, const ExpTab& tab, float& d, int sel) {
        d = delta<LOOKUP>(in, q, tab, sel);
        return !(fabsf(d) < Eps<float>::value);                       // :410
    }
    template <bool TEAM>
    static __device__ __forceinline__ float finish(const EStepArgs<float>& A, int64_t j, const In& in, float q,
                                                   const ExpTab& tab, bool live, bool writer, bool& skipped,
                                                   float* d_out = nullptr) {
        float mu, gamma, d;
        snp_update<EXACT, kLookupPerLane>(in.mm, in.beta, in.sv, in.ulog, in.eta_old, q, tab, mu, gamma, d);
        const bool skip = fabsf(d) < Eps<float>::value;
        if (live && writer) {
            if (!skip) {
                A.var_mu[j] = mu;                                         // :416-418
                A.var_gamma[j] = gamma;
                A.eta_diff[j] = d;
                if (!TEAM) A.eta[j] = in.eta_old + d;                     // :431
            } else {
                A.eta_diff[j] = 0.0f;                                     // :412
            }
            if (TEAM) stage_store(A.eta_out + j, skip ? in.eta_old : in.eta_old + d);
        }
        skipped = live && skip;
        if (d_out) *d_out = (live && !skip) ? d : 0.0f;
        return (live && !skip) ? A.dq * d : 0.0f;
    }
};

// One column of e_step_grid (e_step.hpp:599-635): models of a grid are independent, so the host runs
// this policy once per active model with the (m, G) column-major arrays offset to that column.
// Different arithmetic from e_step: no fma in mu / the logit / d, half_var_tau instead of its
// square root, no skip branch.  EXACT = false (math_mode = fast): the sigmoid on v_exp_f32 / v_rcp_f32.
template <bool EXACT = true>
struct GridColumnModel {
    static constexpr bool kLaneParallel = false;
    struct In { float mm, beta, hvt, ulog, eta_old; };
    static __device__ __forceinline__ In load(const EStepArgs<float>& A, int64_t j, bool live) {
        In in;
        in.mm = live ? A.mu_mult[j] : 0.0f;
        in.beta = live ? A.std_beta[j] : 0.0f;
        in.hvt = live ? A.shvt[j] : 0.0f;
        in.ulog = live ? A.u_logs[j] : 0.0f;
        in.eta_old = live ? A.eta[j] : 0.0f;
        return in;
    }
    template <int LOOKUP>
    static __device__ __forceinline__ void core(const In& in, float q, const ExpTab& tab, float& mu, float& gamma,
                                                float& d, int sel) {
        mu = in.mm * (in.beta - q);                                       // :613
        const float u = in.ulog + in.hvt * mu * mu;                       // :616
        gamma = EXACT ? sigmoid_exact<LOOKUP>(u, tab, sel) : sigmoid_fast(u);   // :617
        d = gamma * mu - in.eta_old;                                      // :620
    }
    static constexpr bool kHasSkip = false;
    template <int LOOKUP>
    static __device__ __forceinline__ float delta(const In& in, float q, const ExpTab& tab, int sel) {
        float mu, gamma, d;
        core<LOOKUP>(in, q, tab, mu, gamma, d, sel);
        return d;
    }
    template <int LOOKUP>
    static __device__ __forceinline__ bool update(const In& in, float q, const ExpTab& tab, float& d, int sel) {
        d = delta<LOOKUP>(in, q, tab, sel);
        return true;
    }
    template <bool TEAM>
    static __device__ __forceinline__ float finish(const EStepArgs<float>& A, int64_t j, const In& in, float q,
                                                   const ExpTab& tab, bool live, bool writer, bool& skipped,
                                                   float* d_out = nullptr) {
        float mu, gamma, d;
        core<kLookupPerLane>(in, q, tab, mu, gamma, d, 0);
        if (d_out) *d_out = live ? d : 0.0f;
        if (live && writer) {
            A.var_mu[j] = mu;
            A.var_gamma[j] = gamma;
            A.eta_diff[j] = d;
            if (TEAM) stage_store(A.eta_out + j, in.eta_old + d); else A.eta[j] = in.eta_old + d;   // :633
        }
        skipped = false;
        return live ? A.dq * d : 0.0f;
    }
};

// exp(x), x <= 0, of the softmax (e_step.hpp:231-240): glibc's expf bit for bit, or v_exp_f32 (math_mode = fast)
template <bool EXACT, int LOOKUP>
__device__ __forceinline__ float softmax_exp(float x, const ExpTab& tab, int sel = 0) {
    if constexpr (EXACT) return expf_glibc_nonpos<LOOKUP>(x, tab, sel);
    else return expf_fast_nonpos(x);
}
// e / ssum of the softmax (:239): the IEEE fp32 divide, or e * v_rcp_f32(ssum) (math_mode = fast; 1 ulp + 1 rounding)
template <bool EXACT>
__device__ __forceinline__ float softmax_div(float e, float ssum) {
    if constexpr (EXACT) return e / ssum;
    else return e * __builtin_amdgcn_rcpf(ssum);
}

// e_step_mixture (e_step.hpp:496-537) for K <= kPanelMaxK components ((m, K) arrays C-ordered).
template <bool EXACT = true>
struct MixtureModel {
    static constexpr bool kExact = EXACT;
    // the chain evaluates the K + 1 components of ONE SNP on K + 1 lanes (see the chain in panel_role)
    static constexpr bool kLaneParallel = true;
    struct In { float mm[kPanelMaxK], sv[kPanelMaxK], ulog[kPanelMaxK]; float lnp, beta, eta_old; int K; };
    static __device__ __forceinline__ In load(const EStepArgs<float>& A, int64_t j, bool live) {
        In in;
        in.K = A.width;
#pragma unroll
        for (int k = 0; k < kPanelMaxK; ++k) {
            const bool on = live && k < in.K;
            const int64_t idx = on ? j * in.K + k : 0;
            in.mm[k] = on ? A.mu_mult[idx] : 0.0f;
            in.sv[k] = on ? A.shvt[idx] : 0.0f;
            in.ulog[k] = on ? A.u_logs[idx] : 0.0f;
        }
        in.lnp = live ? A.log_null_pi[j] : 0.0f;
        in.beta = live ? A.std_beta[j] : 0.0f;
        in.eta_old = live ? A.eta[j] : 0.0f;
        return in;
    }
    template <int LOOKUP>
    static __device__ __forceinline__ void core(const In& in, float q, const ExpTab& tab, float (&mu)[kPanelMaxK],
                                                float (&gam)[kPanelMaxK], float& d, int sel) {
        const float r = in.beta - q;                                      // :505
        float u[kPanelMaxK];
        float mx = in.lnp;                                                // max over u_0..u_K (c_max, :58-71)
#pragma unroll
        for (int k = 0; k < kPanelMaxK; ++k) {
            mu[k] = in.mm[k] * r;                                         // :509
            const float t = in.sv[k] * mu[k];
            u[k] = __builtin_fmaf(t, t, in.ulog[k]);                      // :511
            if (k < in.K) mx = fmaxf(mx, u[k]);
        }
        float ssum = 0.0f;                                                // softmax, :231-240: k = 0..K in order
#pragma unroll
        for (int k = 0; k < kPanelMaxK; ++k) {
            if (k < in.K) {
                u[k] = softmax_exp<EXACT, LOOKUP>(u[k] - mx, tab, sel);
                ssum += u[k];
            }
        }
        ssum += softmax_exp<EXACT, LOOKUP>(in.lnp - mx, tab, sel);
        d = -in.eta_old;                                                  // :519
#pragma unroll
        for (int k = 0; k < kPanelMaxK; ++k) {
            if (k < in.K) {
                gam[k] = softmax_div<EXACT>(u[k], ssum);                  // :239
                d = __builtin_fmaf(gam[k], mu[k], d);                     // :523
            }
        }
    }
    template <int LOOKUP>
    static __device__ __forceinline__ bool update(const In& in, float q, const ExpTab& tab, float& d, int sel) {
        float mu[kPanelMaxK], gam[kPanelMaxK];
        core<LOOKUP>(in, q, tab, mu, gam, d, sel);
        return true;
    }
    template <bool TEAM>
    static __device__ __forceinline__ float finish(const EStepArgs<float>& A, int64_t j, const In& in, float q,
                                                   const ExpTab& tab, bool live, bool writer, bool& skipped) {
        float mu[kPanelMaxK], gam[kPanelMaxK], d;
        core<kLookupPerLane>(in, q, tab, mu, gam, d, 0);
        if (live && writer) {
#pragma unroll
            for (int k = 0; k < kPanelMaxK; ++k) {
                if (k < in.K) {
                    A.var_mu[j * in.K + k] = mu[k];
                    A.var_gamma[j * in.K + k] = gam[k];
                }
            }
            A.eta_diff[j] = d;
            if (TEAM) stage_store(A.eta_out + j, in.eta_old + d); else A.eta[j] = in.eta_old + d;   // :536
        }
        skipped = false;
        return live ? A.dq * d : 0.0f;
    }
};

// e_step_mixture for kPanelMaxK < K <= kPanelWideMaxK components: the K + 1 components of one SNP on K + 1 lanes as
// in MixtureModel, but (i) the component inputs of the coming SNPs are prefetched from global memory into a ring of
// registers (the (m, K) arrays keep a SNP's K values contiguous: one 128-byte line per SNP and array) and the
// per-component outputs are stored straight from the chain -- no LDS staging that would grow with K; (ii) the
// reference's ordered sums (softmax denominator e_step.hpp:231-240, eta :519-523) run as scalar chains over
// v_readlane values: KMAX terms whatever K is -- the terms of lanes > K are exactly neutral (e = +0, gamma = 0).
template <int KMAX>
struct MixtureWideModel {
    static constexpr bool kLaneParallel = true;
    static constexpr bool kWide = true;
    static constexpr int kMax = KMAX;
    struct In { float lnp, beta, eta_old; int K; };
    static __device__ __forceinline__ In load(const EStepArgs<float>& A, int64_t j, bool live) {
        In in;
        in.K = A.width;
        in.lnp = live ? A.log_null_pi[j] : 0.0f;
        in.beta = live ? A.std_beta[j] : 0.0f;
        in.eta_old = live ? A.eta[j] : 0.0f;
        return in;
    }
};
template <typename M, typename = void> struct is_wide_mixture { static constexpr bool value = false; };
template <typename M> struct is_wide_mixture<M, std::enable_if_t<M::kWide>> { static constexpr bool value = true; };

// One role of the sweep kernel below: a workgroup either works as member `wg % team_size` of team `wg / team_size`
// on the statically assigned blocks of a team class (TEAM), or pulls blocks from the small-block queue.
// FORM: what the LD buffer holds and which arithmetic runs over it --
//   kFormSym     the symmetric matrix, the reference's low_memory = False arithmetic
//   kFormMirror  the upper triangle MIRRORED into the lower one (zero diagonal), low_memory = True arithmetic: the second
//                pass q[i] += dq * sum_{j > i} R[i, j] ed[j] reads R[i, j] as R[j, i] -- row j of the buffer, lane = column i,
//                coalesced, the same strip update that carries the trailing updates, into the sums s instead of q (the
//                sum of a row still takes its terms in ascending j: the reference's dot, bit for bit).  No transposition,
//                no second kind of tile traffic: the instruction stream and the memory traffic of the symmetric form.
// (Rounds 1-5 also had a form over the PACKED upper triangle -- zeros on and left of the diagonal, the second pass as
//  per-row running sums behind an LDS transposition; removed in round 6 after a round on the mirrored form.)
constexpr int kFormSym = 1, kFormMirror = 2;

template <typename U, typename MODEL, int FORM, int NW, bool TEAM, int CPL>
__device__ __forceinline__ void panel_role(const EStepArgs<float>& A0, const int qcap, float* __restrict__ smem, const int wg) {
    // q[qcap] | a[2][64] | tiles[2][64 x 64] | mixture chain scratch.  The two tile buffers hold
    //   lane-per-SNP models: the DIAGONAL tiles R[p, p] / R[p+1, p+1] (staged by the updaters one phase ahead; the
    //     chain reads its row with one ds_read per step and touches no LD memory on its critical path -- the
    //     off-diagonal tile R[p, p+1] of its next phase it prefetches into registers, one row per step, a whole
    //     phase ahead of its use);
    //   mixtures (rolled chain loop): the same since round 4 (before: the off-diagonal tiles R[p-1, p] / R[p, p+1] here and the
    //     diagonal rows streamed from global memory by the chain wave; -DPANEL_MIX_UPPER_REGS keeps that for the upper form).
    // (the buffer holds both triangles in either form: strips left and right of the chain)
    constexpr bool MIR = FORM == kFormMirror;       // the arithmetic is the upper-triangular form's
    constexpr bool SUMS = FORM != kFormSym;         // second-pass sums s[] and eta_diff of the last two panels in LDS
    constexpr bool kDiagInLds = !MODEL::kLaneParallel;
    // The off-diagonal tile of the chain's next phase goes through LDS too (below), for every model and both forms: the
    // chain wave issues no vector-memory instruction inside a panel.  The K <= 8 mixture chain (components of one SNP across
    // the lanes) and the wide mixtures (K = 9 .. 31) follow the same scheme: diagonal tiles staged in LDS, the off-diagonal
    // tile of the next phase in the single gated buffer.
    constexpr bool kMixLds = MODEL::kLaneParallel;
    constexpr bool kStageDiag = kDiagInLds || kMixLds;          // what the updaters stage into lT: diagonal tiles
    float* lq = smem;
    float* la = smem + qcap;
    float* lT = la + 2 * kPanel;
    // lane-per-SNP models: the off-diagonal tile R[p, p+1] the chain's NEXT phase starts with (ONE buffer: staged by the
    // updaters during phase p once the chain has consumed its predecessor -- s_tdone below)
    float* lTo = lT + 2 * kPanel * kPanel;
    float* lmx = lTo + kPanel * kPanel;                      // mixture chain only (kMixLdsFloats)
    // upper-triangular form: eta_diff of the last two panels and the running second-pass sums s[j] of the block
    // (panel_mirror_lds_floats; behind the mixture scratch)
    float* led = lmx + ((MODEL::kLaneParallel && !is_wide_mixture<MODEL>::value) ? kMixLdsFloats : 0);
    float* ls = led + 2 * kPanel;
    // row loads in flight per updater lane: 64 VGPRs of row data whatever the strip width (team strips are narrower, see
    // panel_team_cols: more rows in flight, fewer memory round trips per phase on the critical path)
#ifdef PANEL_TEAM_STRIP_DEPTH
    constexpr int kDepth = TEAM ? PANEL_TEAM_STRIP_DEPTH : kStripRowsInFlight;
#else
    constexpr int kDepth = TEAM ? (kPanel / RawRow<U, CPL>::kWords < kPanel ? kPanel / RawRow<U, CPL>::kWords : kPanel) : kStripRowsInFlight;
#endif
    __shared__ int s_blk;
    __shared__ int s_tdone;        // phases whose off-diagonal tile the chain has consumed (gate of the single lTo buffer)
    __shared__ int s_ddone;        // mirrored upper form: phases whose diagonal-tile sums are done (gate of the tile buffer they read)

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const U* __restrict__ ldd = static_cast<const U*>(A0.ld_dense);
    ExpTab tab;
    tab.init();
    unsigned long long my_skipped = 0;
    const float dq = A0.dq;
    const int n_models = max(1, A0.n_active);          // grid: work items are (block, model) pairs

    // Teams (TS > 1): TS workgroups on TS different CUs share one large block.  Every member runs
    // the (deterministic) serial chain itself, so no a-vector ever has to be communicated; the
    // trailing updates are split by column strip (strip s belongs to member s % TS), and the owner of
    // a strip hands the q values of each of its panels to the other members once, just before the
    // chain reaches it, through 8-byte {tag, value} granules in global memory (one relaxed
    // agent-scope atomic store / load per lane; the data is the flag, so no fence is needed).
    // Blocks are assigned to teams statically (team t: blocks t, t + n_teams, ...).
    constexpr int kSW = kPanel * CPL;                   // strip width (columns per updater wave)
    const int TS = TEAM ? A0.team_size : 1;
    const int team = TEAM ? wg / TS : 0;
    const int member = TEAM ? wg % TS : 0;
    int team_iter = 0;

    for (;;) {
        int item;
        if (TEAM) {
            item = team + team_iter * A0.n_teams;
            ++team_iter;
        } else {
            if (tid == 0) {
                int it = atomicAdd(A0.counter, 1);
                if (A0.bottom_mod > 0 && it < A0.n_blocks * n_models) {
                    // claim `it` is valid: take the next block from this workgroup's end of the sorted list (the two ends
                    // cannot cross: at most n claims are valid)
                    const bool bottom = (wg % A0.bottom_mod) == A0.bottom_mod - 1;
                    it = bottom ? A0.n_blocks * n_models - 1 - atomicAdd(A0.counter + 2, 1) : atomicAdd(A0.counter + 1, 1);
                }
                s_blk = it;
            }
            __syncthreads();
            item = s_blk;
            __syncthreads();
        }
        if (item >= A0.n_blocks * n_models) break;
        // models of one block are adjacent in the queue, so the workgroups that stream the same LD
        // rows for different models run at about the same time and share them through L2 / MALL
        const int blk = item / n_models, model_slot = item - blk * n_models;
        const EStepArgs<float> A = select_model(A0, model_slot);

        const BlockDesc bd = A.blocks[blk];
        const int64_t s0 = bd.start;
        const int b = bd.size;
#ifdef VIPRS_SWEEP_TRACE
        const unsigned long long trace_t0 = wall_clock64();
#endif
        unsigned long long* __restrict__ gran =
            TEAM ? A.granules + ((int64_t)model_slot * A.granule_rows + bd.gr_off) * kPanel : nullptr;
        const int stride = bd.stride;
        const U* __restrict__ base = ldd + bd.ld_off;
        const int np = (b + kPanel - 1) / kPanel;
        const int bpad = np * kPanel;
        // A team member keeps in LDS only the q (and, upper-triangular form, the second-pass sums) of ITS OWN strips,
        // strip after strip (strip st = member + k TS is local strip k): nobody else's columns are ever read from or
        // written to its LDS -- the chain takes the other members' panels from their hand-off granules (panels 0 and 1,
        // which precede every hand-off, from the state itself).  LDS of the team classes: qcap / TS instead of qcap.
        const int nstrips_all = (bpad + kSW - 1) / kSW;
        const int n_own = TEAM ? (nstrips_all - member + TS - 1) / TS : nstrips_all;       // strips member, member + TS, ...
        auto own = [&](int c) { return !TEAM || ((c / kSW) % TS) == member; };
        auto loc = [&](int c) { return TEAM ? ((c / kSW) / TS) * kSW + (c % kSW) : c; };       // own columns only
        auto glob = [&](int li) { return TEAM ? ((li / kSW) * TS + member) * kSW + (li % kSW) : li; };

        for (int li = tid; li < (TEAM ? n_own * kSW : bpad) + kStrip; li += NW * 64) {
            const int c = glob(li);
            const bool in = (!TEAM || li < n_own * kSW) && c < b;
            lq[li] = in ? A.q[s0 + c] : 0.0f;
            if (SUMS) ls[li] = 0.0f;
        }
        if (tid == 0) { s_tdone = 0; s_ddone = 0; }
        if (kStageDiag) {
            // diagonal tile of panel 0 (rows past the end of a short block are clamped: finite values that
            // only ever meet a = 0)
            for (int i = tid; i < kPanel * kPanel / 4; i += NW * 64) {
                const int row = i >> 4, tcol = (i & 15) * 4;
                *reinterpret_cast<float4*>(lT + row * kPanel + tcol) = load4<U>(base + (int64_t)min(row, b - 1) * stride + tcol);
            }
        }
        __syncthreads();

        float a_prev = 0.0f;   // chain wave: lane j = dq * eta_diff of SNP j of the previous panel

        // ---- chain wave: inputs and first diagonal-tile rows of the NEXT panel, fetched under the
        //      current panel's serial updates so that no HBM latency sits between two panels
        typename MODEL::In nxt_in{};
        float dnext[kChainPrefetch];                    // mixture chain: first diagonal rows of the next panel
        if (wave == 0) {
            const bool live0 = lane < b;
            nxt_in = MODEL::load(A, s0 + (live0 ? lane : 0), live0);
            if (!kStageDiag) {
#pragma unroll
                for (int k = 0; k < kChainPrefetch; ++k)
                    dnext[k] = static_cast<float>(base[(int64_t)min(k, b - 1) * stride + lane]);
            }
        }

        // symmetric form: one extra phase applies the last panel's a-vector to the columns left of it
#ifdef VIPRS_PANEL_PROFILE
        __shared__ unsigned s_pprof[64][8];
#endif
        // The chain wave and the updater waves run their OWN loops over the phases (one workgroup barrier per phase
        // in each): what a role keeps in registers across phases -- the chain's 64 prefetched tile rows -- is then
        // live in its own branch only and does not add to the other role's register budget.
        if (wave == 0) {
#ifdef PANEL_CHAIN_PRIO
        __builtin_amdgcn_s_setprio(PANEL_CHAIN_PRIO);    // the chain wave wins the issue arbitration on its SIMD
#endif
        for (int p = 0; p < np + 1; ++p) {
            PPROF(0, true);
            {
                // ================================ chain ======================================
                if (p < np) {
                    const int r0 = p * kPanel;
                    const int nrows = min(kPanel, b - r0);
                    const int64_t j = s0 + r0 + lane;
                    const bool live = lane < nrows;
                    const typename MODEL::In in = nxt_in;
                    {   // next panel's inputs (consumed one phase later)
                        const int rn = r0 + kPanel + lane;
                        const bool ln = rn < b;
                        nxt_in = MODEL::load(A, s0 + (ln ? rn : 0), ln);
                    }

                    // diagonal tile rows, streamed kChainPrefetch rows ahead.  All 64 steps always
                    // run, straight-line (no runtime guards around loads, see strip_update): rows
                    // past a partial last panel are clamped and their steps forced onto the skip
                    // path (a = 0 leaves every q untouched).
                    const int last = nrows - 1;
                    const U* __restrict__ dptr = base + (int64_t)r0 * stride + r0 + lane;
                    float drow[kStageDiag ? 1 : kPanel];
                    if (!kStageDiag) {
#pragma unroll
                        for (int k = 0; k < kChainPrefetch; ++k) drow[k] = dnext[k];
                    }
                    // first rows of the next panel's diagonal tile (clamped to the block when there
                    // is no next panel: loaded, never used)
                    const int rn0 = min(r0 + kPanel, bpad - kPanel);
                    const U* __restrict__ nptr = base + (int64_t)rn0 * stride + rn0 + lane;

                    float qc;
                    // mirrored upper form: lane j keeps the q_j its own update consumed (what the upper form leaves in q[j] until
                    // the second pass is added)
                    float q_own = 0.0f;
                    if (own(r0)) qc = lq[loc(r0 + lane)];
                    else qc = (r0 + lane < b) ? A.q[s0 + r0 + lane] : 0.0f;       // (panels 0 / 1 of another member: no update has reached them yet)
                    if (TEAM && p >= 2 && (((p * kPanel) / kSW) % TS) != member) {
                        // panel p lives in another member's strip: take its q (all trailing updates
                        // a_0 .. a_{p-2} applied) from the owner's granules, tag = p + 1
                        unsigned long long g = 0;
#ifdef PANEL_TIMING_NO_HANDOFF_WAIT
                        // TIMING EXPERIMENT (wrong results): the chain never waits for a hand-off -- what a zero-latency hand-off
                        // would be worth
                        if (false)
#endif
                        for (unsigned spins = 0;; ++spins) {
                            g = __hip_atomic_load(gran + (int64_t)p * kPanel + lane, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
                            if (__all((unsigned)(g >> 32) == A.tag_base + (unsigned)(p + 1))) break;
                            if (spins > (1u << 22)) {          // ~seconds: give up loudly, never hang
                                if (lane == 0) atomicExch(A.error, 1);
                                break;
                            }
                            // somebody has given up already: the launch is lost, no further wait may cost another timeout
                            if ((spins & 1023u) == 1023u &&
                                __hip_atomic_load(A.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                            __builtin_amdgcn_s_sleep(8);
                        }
                        qc = __uint_as_float((unsigned)g);
                    }
                    PPROF(1, true);
                    if (p > 0) {
                        {
                            // a_{p-1} through tile R[p-1, p], staged in LDS by the updaters during the previous phase: the
                            // chain wave issues NO vector-memory instruction between two panels and none inside one -- a
                            // global load per step stalls at issue behind the updaters' loads when the chip is loaded
                            // (chain step 135 ns on an idle chip, 170-230 ns in mid-sweep; in-kernel timeline)
                            float trow[kPanel];
#pragma unroll
                            for (int k = 0; k < kPanel; ++k) trow[k] = lTo[k * kPanel + lane];
                            if constexpr (sizeof(U) < 4 || MODEL::kLaneParallel) {
                                // Integer LD and the K <= 8 mixture of the symmetric form (chain-bound sweeps; configs[3] 1.065 -> 1.054 ms): the 64 a-values as 16 broadcast LDS reads (la holds them for
                                // the updaters) instead of 64 v_readlane, each followed by two wait states before the fma that
                                // reads its SGPR: int8 upper 0.510 -> 0.500 ms, int8 symmetric 0.527 -> 0.518 (fp32 LD, a
                                // bandwidth-bound sweep, measured slower with it and keeps the v_readlane form).
                                const float4* __restrict__ av4 = reinterpret_cast<const float4*>(la + ((p - 1) & 1) * kPanel);
                                float4 av[kPanel / 4];
#pragma unroll
                                for (int i = 0; i < kPanel / 4; ++i) av[i] = av4[i];
#pragma unroll
                                for (int k = 0; k < kPanel; ++k) {
                                    const float4 t = av[k >> 2];
                                    const float ak = (k & 3) == 0 ? t.x : (k & 3) == 1 ? t.y : (k & 3) == 2 ? t.z : t.w;
                                    qc = __builtin_fmaf(trow[k], ak, qc);
                                }
                            } else {
#pragma unroll
                                for (int k = 0; k < kPanel; ++k) qc = __builtin_fmaf(trow[k], rl(a_prev, k), qc);
                            }
                        }
                    }

                    {
                        // the tile is consumed: the updaters may stage the next one (LDS executes a wave's operations in
                        // order, so the reads above precede this store)
                        if (lane == 0) *reinterpret_cast<volatile int*>(&s_tdone) = p + 1;
                    }
                    PPROF(2, true);
                    if constexpr (is_wide_mixture<MODEL>::value) {
                        constexpr int KMAX = MODEL::kMax;
                        const int K = in.K;
                        const int kc = min(lane, K - 1);                   // lanes >= K: harmless copies of component K - 1
                        const bool comp = lane < K;
                        float dvec = 0.0f, avec = 0.0f;                    // lane j: eta_diff / dq * eta_diff of SNP j
                        // component inputs of SNP (r0 + jj), lanes 0 .. K-1; SNPs past the block are clamped (never used)
                        const int64_t jlast = s0 + b - 1;
                        auto cptr = [&](const float* __restrict__ arr, int jj) {
                            return arr + min(s0 + r0 + jj, jlast) * K + kc;
                        };
                        static_assert(kChainPrefetch == 16 && kPanel == 64, "window indexing");
                        float win[kChainPrefetch], rmm[kChainPrefetch], rsv[kChainPrefetch], rul[kChainPrefetch];
                        float rvec = in.beta - qc;                         // lane j: beta_j - q_j, read by step j (:505)
                        const float* __restrict__ DtW = lT + (p & 1) * kPanel * kPanel + lane;    // kMixLds: the panel's diagonal tile in LDS
#pragma unroll
                        for (int k = 0; k < kChainPrefetch; ++k) {
                            win[k] = kMixLds ? 0.0f : drow[kMixLds ? 0 : k];
                            rmm[k] = *cptr(A.mu_mult, k);
                            rsv[k] = *cptr(A.shvt, k);
                            rul[k] = *cptr(A.u_logs, k);
                        }
#pragma unroll 1
                        for (int g = 0; g < kPanel / kChainPrefetch; ++g) {
#pragma unroll
                        for (int k = 0; k < kChainPrefetch; ++k) {
                            const int jj = kChainPrefetch * g + k;                             // wave-uniform
                            const float drow_jj = kMixLds ? DtW[jj * kPanel] : win[k];
                            const float cmm = rmm[k], csv = rsv[k], cul = rul[k];
                            {
                                if (!kMixLds) {
                                    const U* __restrict__ src = (g < kPanel / kChainPrefetch - 1)
                                        ? dptr + (int64_t)min(jj + kChainPrefetch, last) * stride
                                        : nptr + (int64_t)min(k, b - 1 - rn0) * stride;
                                    win[k] = static_cast<float>(*src);
                                }
                                // (the ring runs on into the next panel; past the block it re-reads the last SNP)
                                rmm[k] = *cptr(A.mu_mult, jj + kChainPrefetch);
                                rsv[k] = *cptr(A.shvt, jj + kChainPrefetch);
                                rul[k] = *cptr(A.u_logs, jj + kChainPrefetch);
                            }
                            const float lnp = rl(in.lnp, jj), eta_old = rl(in.eta_old, jj);
                            const float r = rl(rvec, jj);                                      // :505, formed in lane jj (below)
                            const float mu = cmm * r;                                          // :509
                            const float t = csv * mu;
                            float u = __builtin_fmaf(t, t, cul);                               // :511
                            u = (lane == K) ? lnp : u;
                            // max over lanes 0..K (order-free): prefix max inside each row of 16 lanes, rows combined
                            float mx = dpp_max_shr<1>(u);
                            mx = dpp_max_shr<2>(mx);
                            mx = dpp_max_shr<4>(mx);
                            mx = dpp_max_shr<8>(mx);
                            const float m_lo = rl(mx, 15), m_hi = rl(mx, K);
                            mx = (K >= 16) ? fmaxf(m_lo, m_hi) : m_hi;                         // c_max, :58-71
                            float e = expf_glibc_nonpos<kLookupPerLane>(u - mx, tab);
                            e = (lane <= K) ? e : 0.0f;                                        // neutral terms beyond the null component
                            // softmax denominator, :231-240: s = ((e_0 + e_1) + ...) + e_null, a scalar chain
                            // (8 reads at a time, pinned in SGPRs before the 8 additions that consume them: pin_sgprs)
                            static_assert((KMAX + 1) % 8 == 0, "chunks of 8 terms");
                            float ssum = 0.0f;
#pragma unroll
                            for (int c = 0; c < (KMAX + 1) / 8; ++c) {
                                int et[8];
#pragma unroll
                                for (int i = 0; i < 8; ++i) et[i] = __builtin_amdgcn_readlane(__float_as_int(e), 8 * c + i);
                                pin_sgprs(et);
#pragma unroll
                                for (int i = 0; i < 8; ++i) ssum = (c == 0 && i == 0) ? __int_as_float(et[0]) : ssum + __int_as_float(et[i]);
                            }
                            const float gam = comp ? e / ssum : 0.0f;                          // :239
                            // eta_diff, :519-523: d = fma(gam_k, mu_k, d) for k = 0 .. K-1 (gam = 0 beyond: exact no-ops; one more
                            // such term, lane KMAX, fills the last chunk)
                            float d = -eta_old;
#pragma unroll
                            for (int c = 0; c < (KMAX + 1) / 8; ++c) {
                                int gt[8];
                                float mb[8];
#pragma unroll
                                for (int i = 0; i < 8; ++i) {
                                    gt[i] = __builtin_amdgcn_readlane(__float_as_int(gam), 8 * c + i);
                                    mb[i] = rl(mu, 8 * c + i);
                                }
                                pin_sgprs(gt);
#pragma unroll
                                for (int i = 0; i < 8; ++i) d = __builtin_fmaf(__int_as_float(gt[i]), mb[i], d);
                            }
                            const bool livej = jj < nrows;                                     // wave-uniform
                            const float a = livej ? dq * d : 0.0f;
                            if (comp && livej && member == 0) {
                                const int64_t o = (s0 + r0 + jj) * K + lane;                   // (m, K) C-order
                                A.var_mu[o] = mu;
                                A.var_gamma[o] = gam;
                            }
                            int l = lane;
                            asm volatile("" : "+v"(l));
                            const bool me = (l == jj);
                            dvec = me ? d : dvec;
                            avec = me ? a : avec;
                            // (mirrored storage: row jj of the diagonal tile is non-zero left of jj too -- the lanes <= jj keep no
                            //  valid q in qc from here on, theirs is captured at their own step)
                            if (MIR) q_own = me ? qc : q_own;
                            qc = __builtin_fmaf(drow_jj, a, qc);
                            rvec = in.beta - qc;                                        // (as in the K <= 8 chain below)
                            if (!MIR) qc = (me && livej) ? qc - d : qc;                 // :527
                        }
                        }
                        if (!kMixLds) {
#pragma unroll
                            for (int k = 0; k < kChainPrefetch; ++k) dnext[k] = win[k];
                        }
                        if (member == 0 && live) {
                            A.eta_diff[j] = dvec;
                            if (TEAM) stage_store(A.eta_out + j, in.eta_old + dvec); else A.eta[j] = in.eta_old + dvec;   // :536
                        }
                        a_prev = avec;
                        if (SUMS) led[(p & 1) * kPanel + lane] = live ? dvec : 0.0f;
                    } else if constexpr (MODEL::kLaneParallel) {
                        // Mixture chain: the K components (and the null component, lane K) of ONE SNP
                        // are evaluated on K + 1 lanes -- one expf, one divide per SNP instead of K + 1
                        // and K; the ordered sums of the reference (softmax denominator, e_step.hpp:231-240;
                        // eta, :519-523) run over v_readlane values in component order.  Component
                        // inputs of the panel go through LDS as [SNP][k] (the (m, K) arrays' own layout).
                        const int K = in.K;
                        float* __restrict__ Lmm = lmx;
                        float* __restrict__ Lsv = lmx + kPanel * kPanelMaxK;
                        float* __restrict__ Lul = lmx + 2 * kPanel * kPanelMaxK;
                        float* __restrict__ Lmu = lmx + 3 * kPanel * kPanelMaxK;
                        float* __restrict__ Lga = lmx + 4 * kPanel * kPanelMaxK;
#pragma unroll
                        for (int k = 0; k < kPanelMaxK; ++k) {
                            if (k < K) {
                                Lmm[lane * K + k] = in.mm[k];
                                Lsv[lane * K + k] = in.sv[k];
                                Lul[lane * K + k] = in.ulog[k];
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                        const int kc = min(lane, K - 1);                   // lanes >= K: harmless copies
                        float cmm = Lmm[kc], csv = Lsv[kc], cul = Lul[kc];
                        float dvec = 0.0f, avec = 0.0f;                    // lane j: eta_diff / dq * eta_diff of SNP j
                        auto run_panel = [&](auto rounds_c, auto k4_c) {
                        constexpr int ROUNDS = decltype(rounds_c)::value;
                        // The two ordered sums of the step run as SCALAR chains over v_readlane values (ROUNDS + 1 and ROUNDS
                        // terms; the terms beyond K are exactly neutral: e = +0, gamma = 0) instead of lane-to-lane DPP
                        // rounds -- a DPP operand needs two wait states behind the VALU write it reads, every round.
                        // K == 4 exactly (BASELINE configs[3]): no masking of the terms either.
                        constexpr bool K4 = decltype(k4_c)::value;
                        float rvec = in.beta - qc;                         // lane j: beta_j - q_j, read by step j (:505)
                        // Rolled in 4 groups of kChainPrefetch = 16 steps (the fully unrolled mixture chain does
                        // not fit the instruction cache): the row consumed at step 16 g + k was loaded 16 steps
                        // earlier into the same register win[k]; the last group loads the first rows of the NEXT
                        // panel's diagonal tile, which are handed over through dnext.
                        static_assert(kChainPrefetch == 16 && kPanel == 64, "window indexing");
                        float win[kMixLds ? 1 : kChainPrefetch];
                        if (!kMixLds) {
#pragma unroll
                            for (int k = 0; k < kChainPrefetch; ++k) win[k] = drow[kMixLds ? 0 : k];
                        }
                        const float* __restrict__ Dt = lT + (p & 1) * kPanel * kPanel + lane;     // kMixLds: diagonal tile of the panel
#pragma unroll 1
                        for (int g = 0; g < kPanel / kChainPrefetch; ++g) {
#pragma unroll
                        for (int k = 0; k < kChainPrefetch; ++k) {
                            const int jj = kChainPrefetch * g + k;                             // wave-uniform
                            float drow_jj;
                            if (kMixLds) {
                                drow_jj = Dt[jj * kPanel];
                            } else {
                                drow_jj = win[kMixLds ? 0 : k];
                                const U* __restrict__ src = (g < kPanel / kChainPrefetch - 1)
                                    ? dptr + (int64_t)min(jj + kChainPrefetch, last) * stride
                                    : nptr + (int64_t)min(k, b - 1 - rn0) * stride;
                                win[kMixLds ? 0 : k] = static_cast<float>(*src);
                            }
                            const int jn = (jj + 1 < kPanel) ? jj + 1 : jj;
                            const float nmm = Lmm[jn * K + kc], nsv = Lsv[jn * K + kc], nul = Lul[jn * K + kc];
                            const float lnp = rl(in.lnp, jj), eta_old = rl(in.eta_old, jj);
                            const float r = rl(rvec, jj);                                      // :505, formed in lane jj (below)
                            const float mu = cmm * r;                                          // :509
                            const float t = csv * mu;
                            float u = __builtin_fmaf(t, t, cul);                               // :511
                            u = (lane == K) ? lnp : u;
                            // mu_k in every lane (operands of the eta chain at the end of the step: read here, long before)
                            float mub[ROUNDS];
#pragma unroll
                            for (int i = 0; i < ROUNDS; ++i) mub[i] = rl(mu, i);
                            // max over lanes 0..K (order-free): inclusive prefix max along the row, read at lane K
                            float mx;
                            if constexpr (K4) {
                                int ut[5];
#pragma unroll
                                for (int i = 0; i < 5; ++i) ut[i] = __builtin_amdgcn_readlane(__float_as_int(u), i);
                                pin_sgprs(ut);
                                mx = __int_as_float(ut[0]);
#pragma unroll
                                for (int i = 1; i <= 4; ++i) mx = fmaxf(mx, __int_as_float(ut[i]));
                            } else {
                                mx = dpp_max_shr<1>(u);
                                mx = dpp_max_shr<2>(mx);
                                mx = dpp_max_shr<4>(mx);
                                mx = dpp_max_shr<8>(mx);
                                mx = rl(mx, K);
                            }                                                                  // c_max, :58-71
                            const float e = softmax_exp<MODEL::kExact, kLookupPerLane>(u - mx, tab);
                            // softmax denominator, :231-240: s = ((e_0 + e_1) + ...) + e_null in this order.
                            const float e_t = (K4 || lane <= K) ? e : 0.0f;
                            // (all reads first, pinned by an empty asm: read-add-read-add costs two wait states per term
                            //  between the v_readlane that writes the SGPR and the add that reads it)
                            int et[ROUNDS + 1];
#pragma unroll
                            for (int i = 0; i <= ROUNDS; ++i) et[i] = __builtin_amdgcn_readlane(__float_as_int(e_t), i);
                            pin_sgprs(et);
                            float ssum = __int_as_float(et[0]);
#pragma unroll
                            for (int i = 1; i <= ROUNDS; ++i) ssum += __int_as_float(et[i]);
                            const float gam = softmax_div<MODEL::kExact>(e, ssum);             // :239
                            // eta_diff, :519-523: d_k = fma(gam_k, mu_k, d_{k-1}), d_{-1} = -eta_old
                            const float gam_t = (K4 || lane < K) ? gam : 0.0f;
                            int gt[ROUNDS];
#pragma unroll
                            for (int i = 0; i < ROUNDS; ++i) gt[i] = __builtin_amdgcn_readlane(__float_as_int(gam_t), i);
                            pin_sgprs(gt);
                            float d = -eta_old;
#pragma unroll
                            for (int i = 0; i < ROUNDS; ++i) d = __builtin_fmaf(__int_as_float(gt[i]), mub[i], d);
                            const bool livej = jj < nrows;                                     // wave-uniform
                            const float a = livej ? dq * d : 0.0f;
                            if (lane < K) {
                                Lmu[jj * K + lane] = mu;
                                Lga[jj * K + lane] = gam;
                            }
                            int l = lane;
                            asm volatile("" : "+v"(l));
                            const bool me = (l == jj);
                            dvec = me ? d : dvec;
                            avec = me ? a : avec;
                            // (mirrored storage: row jj of the diagonal tile is non-zero left of jj too -- the lanes <= jj keep no
                            //  valid q in qc from here on, theirs is captured at their own step)
                            if (MIR) q_own = me ? qc : q_own;
                            qc = __builtin_fmaf(drow_jj, a, qc);
                            // what the NEXT step starts from, beta - q of lane jj + 1, before anything else is done to q: the
                            // subtraction of the own term (lane jj only) stays off the path from a to the next step's mu
                            rvec = in.beta - qc;
                            if (!MIR) qc = (me && livej) ? qc - d : qc;                 // :527
                            cmm = nmm; csv = nsv; cul = nul;
                        }
                        }
                        if (!kMixLds) {
#pragma unroll
                            for (int k = 0; k < kChainPrefetch; ++k) dnext[k] = win[kMixLds ? 0 : k];
                        }
                        };
                        if (K == 4) run_panel(std::integral_constant<int, 4>{}, std::true_type{});
                        else if (K < 4) run_panel(std::integral_constant<int, 4>{}, std::false_type{});
                        else run_panel(std::integral_constant<int, kPanelMaxK>{}, std::false_type{});
                        __builtin_amdgcn_wave_barrier();
                        if (member == 0) {
                            // (m, K) C-order: the panel's var_mu / var_gamma are nrows * K contiguous floats
                            const int64_t o = (s0 + r0) * K;
                            for (int i = lane; i < nrows * K; i += 64) {
                                A.var_mu[o + i] = Lmu[i];
                                A.var_gamma[o + i] = Lga[i];
                            }
                            if (live) {
                                A.eta_diff[j] = dvec;
                                if (TEAM) stage_store(A.eta_out + j, in.eta_old + dvec); else A.eta[j] = in.eta_old + dvec;   // :536
                            }
                        }
                        a_prev = avec;
                        if (SUMS) led[(p & 1) * kPanel + lane] = live ? dvec : 0.0f;
                    } else {
                        // The 64 serial SNP updates.  Lane j carries SNP j (its own inputs, its own q[j]);
                        // every lane evaluates the update on its own values, but at step j only lane j's
                        // result is meaningful: it is broadcast with one v_readlane and applied to the whole
                        // panel through row j of the diagonal tile (LDS).  One wave per chain: every
                        // instruction of the step is on the critical path or competes for its issue slots,
                        // so the step is written for both --
                        //   * the next step's update reads q from the fma result `qf` directly; the own-lane
                        //     bookkeeping (q capture for the replay, the symmetric form's diagonal
                        //     subtraction e_step.hpp:427) hangs off the side through selects on an SGPR lane
                        //     mask that is shifted once per step (no per-step lane compare);
                        //   * skipped steps (|d| < eps, e_step.hpp:410) become a = 0: fma(R, 0, q) == q.
                        //     Lanes past a partial last panel carry all-zero inputs (MODEL::load), which makes
                        //     their d exactly 0: they skip by themselves, no `live` test inside the loop;
                        float qcap_v = 0.0f;   // lane j keeps the q_j its own update consumed
                        const float* __restrict__ Dt = lT + (p & 1) * kPanel * kPanel + lane;
                        unsigned long long lane_bit = 1ull;
                        float qf = qc;
    #pragma unroll
                        for (int jj = 0; jj < kPanel; ++jj) {
                            const float dr = Dt[jj * kPanel];
                            const float d = MODEL::template delta<kLookupLane>(in, qf, tab, jj);
                            const float dz = (MODEL::kHasSkip && fabsf(d) < Eps<float>::value) ? 0.0f : d;   // :410
                            // a = dq * d is formed per lane BEFORE the broadcast (same operands for the step's own lane, same
                            // bits): the dependent chain is select -> mul -> v_readlane -> fma with the SGPR as the fma's
                            // operand, not v_readlane -> v_mov -> mul -> fma; the second broadcast (d itself, for the
                            // symmetric form's diagonal subtraction) hangs off the side
                            const float az = dq * dz;
                            const float sa = rl(az, jj);
                            const float sdz = rl(dz, jj);
                            qcap_v = sel_mask(qcap_v, qc, lane_bit);
                            qf = __builtin_fmaf(dr, sa, qc);
                            // (mirrored storage: qc of the lanes <= jj is no longer a q after this -- theirs was captured in qcap_v)
                            qc = !MIR ? sel_mask(qf, qf - sdz, lane_bit) : qf;   // e_step.hpp:427 (own lane only)
                            asm volatile("s_lshl_b64 %0, %0, 1" : "+s"(lane_bit) : : "scc");
                        }
                        q_own = qcap_v;

                        PPROF(3, true);
                        // lane-parallel replay of the 64 updates (same operations, same inputs ->
                        // same bits) to produce the per-SNP outputs without serialising the stores
                        bool skipped_lane;
                        float d_lane = 0.0f;
                        a_prev = MODEL::template finish<TEAM>(A, j, in, qcap_v, tab, live, member == 0, skipped_lane, &d_lane);
                        my_skipped += __popcll(__ballot(skipped_lane));
                        if (SUMS) led[(p & 1) * kPanel + lane] = d_lane;
                    }
                    la[(p & 1) * kPanel + lane] = a_prev;
                    if (own(r0)) lq[loc(r0 + lane)] = MIR ? q_own : qc;
                    PPROF(4, true);
                }
            }
            __syncthreads();
        }
        } else {
        for (int p = 0; p < np + 1; ++p) {
            {
                // ================================ updaters ===================================
                const int uw = wave - 1;
                auto stage_tiles = [&]() {
                // stage the tile of the chain's next phase: the diagonal tile R[p+1, p+1] (lane-per-SNP models) or the
                // off-diagonal tile R[p, p+1] (mixture).  16 row groups of 4 rows dealt to the updater waves, all of a
                // wave's loads issued before its first LDS store (one memory round trip per phase)
                if (p + 1 < np) {
                    float* __restrict__ T = lT + ((p + 1) & 1) * kPanel * kPanel;
                    const int trow = lane >> 4, tcol = (lane & 15) * 4;
                    const int row_base = (kStageDiag ? p + 1 : p) * kPanel;
                    constexpr int kGroups = (kPanel / 4 + NW - 2) / (NW - 1);
                    float4 v[kGroups];
#pragma unroll
                    for (int g = 0; g < kGroups; ++g) {
                        const int row = 4 * min(uw + g * (NW - 1), kPanel / 4 - 1) + trow;
                        v[g] = load4<U>(base + (int64_t)min(row_base + row, b - 1) * stride + (p + 1) * kPanel + tcol);
                    }
                    float4 w[kGroups];
                    {
                        // ... and the off-diagonal tile R[p, p+1] its next phase STARTS with (rows of this panel; rows past a
                        // partial last panel are clamped: they only ever meet a = 0)
#pragma unroll
                        for (int g = 0; g < kGroups; ++g) {
                            const int row = 4 * min(uw + g * (NW - 1), kPanel / 4 - 1) + trow;
                            w[g] = load4<U>(base + (int64_t)min(p * kPanel + row, b - 1) * stride + (p + 1) * kPanel + tcol);
                        }
                    }
                    if (MIR && p > 0) {
                        // the buffer still holds the diagonal tile of panel p-1 until its rows' sums are done (below)
                        while (*reinterpret_cast<volatile int*>(&s_ddone) < p) __builtin_amdgcn_s_sleep(1);
                    }
#pragma unroll
                    for (int g = 0; g < kGroups; ++g) {
                        const int i = uw + g * (NW - 1);
                        if (i < kPanel / 4) *reinterpret_cast<float4*>(T + (4 * i + trow) * kPanel + tcol) = v[g];
                    }
                    {
                        // the single buffer still holds R[p-1, p] until the chain has applied it (first thing in its phase)
                        while (*reinterpret_cast<volatile int*>(&s_tdone) < p + 1) __builtin_amdgcn_s_sleep(1);
#pragma unroll
                        for (int g = 0; g < kGroups; ++g) {
                            const int i = uw + g * (NW - 1);
                            if (i < kPanel / 4) *reinterpret_cast<float4*>(lTo + (4 * i + trow) * kPanel + tcol) = w[g];
                        }
                    }
                }
                };
                // strips of this member are dealt round-robin to its updater waves; the strip that holds panel p+1 goes
                // first so that its q can be handed over early
                const int pp = p - 1;                       // panel whose a-vector is applied (p > 0)
                const int rr0 = pp * kPanel;
                const int last_row = min(kPanel, b - rr0) - 1;
                const float avec = p > 0 ? la[(pp & 1) * kPanel + lane] : 0.0f;
                const int nstrips = (bpad + kSW - 1) / kSW;
                const bool any_a = __ballot(avec != 0.0f) != 0;
                // mirrored upper form: eta_diff of panel pp, the multiplier of the second-pass sums (a = dq * eta_diff is the
                // trailing update's)
                const float edvec = (MIR && p > 0) ? led[(pp & 1) * kPanel + lane] : 0.0f;
                const bool any_ed = MIR && __ballot(edvec != 0.0f) != 0;
                const int s_pri = ((p + 1) * kPanel) / kSW;
                const int n_mine = (nstrips - member + TS - 1) / TS;       // strips member, member+TS, ...
                auto do_strip = [&](int k) {
                    int st = member + k * TS, kl = k;                     // kl: the strip's index in this member's LDS
                    if (TEAM) {
                        // rotate so that the priority strip (if this member owns it) is slot 0
                        const int k_pri = (s_pri % TS == member) ? (s_pri - member) / TS : 0;
                        kl = (k + k_pri) % n_mine;
                        st = member + kl * TS;
                    }
                    const int c = st * kSW + CPL * lane;
                    float* __restrict__ lq_c = lq + (TEAM ? kl * kSW + CPL * lane : c);
                    const int cp = c >> 6;
                    // symmetric form: every column except the chain's two panels (left of the
                    // chain = SNPs already visited, their q keeps accumulating for the next
                    // sweep); upper-triangular form: right of the chain only (the rest is the
                    // reference's second pass)
#if defined(PANEL_TIMING_NO_SECOND_PASS) && PANEL_TIMING_NO_SECOND_PASS + 0 >= 2     // (timing experiments only: wrong results)
                    const bool active = (c < b) && cp > p;
#else
                    const bool active = (c < b) && cp != pp && cp != p;
#endif
                    auto hand_off = [&]() {
                        if (TEAM && st == s_pri && p + 1 < np && p + 1 >= 2) {
                            // hand panel p+1 (now carrying a_0 .. a_{p-1}) to the other members
                            __builtin_amdgcn_wave_barrier();
                            const float v = lq[loc((p + 1) * kPanel + lane)];
                            const unsigned long long g =
                                ((unsigned long long)(A.tag_base + (unsigned)(p + 2)) << 32) | (unsigned long long)__float_as_uint(v);
                            __hip_atomic_store(gran + (int64_t)(p + 1) * kPanel + lane, g, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                        }
                    };
                    if constexpr (MIR) {
                        // rows of panel pp, lane = column: right of the chain q[c] = fma(R[j, c], a_j, q[c]) (the trailing update),
                        // left of it s[c] = fma(R[j, c], ed_j, s[c]) -- R[j, c] = R[c, j], the term of row c's second-pass sum
                        // (update_q_factor, e_step.hpp:331-337), rows j ascending.  A strip lies on one side of the chain (one
                        // wave-uniform multiplier vector, the per-lane target picks q or s) unless panels pp and p sit in its
                        // middle: then the multiplier is per lane too (strip_update<MIXED>).
                        const bool left = cp < pp;
                        float* __restrict__ tgt = left ? ls + (TEAM ? kl * kSW + CPL * lane : c) : lq_c;
                        const bool any_l = __ballot(active && left) != 0, any_r = __ballot(active && !left) != 0;
                        if (any_l && any_r) {
                            const float fvec = left ? 1.0f : dq;
                            if (any_ed && active) {
                                if (last_row == kPanel - 1)
                                    strip_update<U, CPL, true, kDepth, true>(base + (int64_t)rr0 * stride + c, stride, last_row, edvec, tgt, fvec);
                                else
                                    strip_update<U, CPL, false, kDepth, true>(base + (int64_t)rr0 * stride + c, stride, last_row, edvec, tgt, fvec);
                            }
                        } else {
                            // (pinned in a register under the full exec mask: strip_update reads it across lanes with v_readlane
                            //  inside the divergent region below, and a select the compiler sinks into that region would be
                            //  written for its active lanes only)
                            float mvec = any_l ? edvec : avec;
                            asm volatile("" : "+v"(mvec));
                            if ((any_l ? any_ed : any_a) && active) {
                                if (last_row == kPanel - 1)
                                    strip_update<U, CPL, true, kDepth>(base + (int64_t)rr0 * stride + c, stride, last_row, mvec, tgt);
                                else
                                    strip_update<U, CPL, false, kDepth>(base + (int64_t)rr0 * stride + c, stride, last_row, mvec, tgt);
                            }
                        }
                        if (k == uw) PPROF(6, wave == 1);
                        hand_off();
                    } else {
                        if (any_a && active) {
                            if (last_row == kPanel - 1)
                                strip_update<U, CPL, true, kDepth>(base + (int64_t)rr0 * stride + c, stride, last_row, avec, lq_c);
                            else
                                strip_update<U, CPL, false, kDepth>(base + (int64_t)rr0 * stride + c, stride, last_row, avec, lq_c);
                        }
                        if (k == uw) PPROF(6, wave == 1);
                        hand_off();
                    }
                };
                if (MIR && p > 0 && uw == NW - 2) {
                    // the diagonal tile of panel pp, still in LDS from the chain's last phase: its rows' sums over the later SNPs
                    // of the same panel -- by the member that owns the panel's strip, on its last updater wave (the first one
                    // carries the priority strip and the hand-off the rest of the team waits for).  The staging of the next
                    // diagonal tile into the same buffer waits for it (s_ddone; LDS executes a wave's operations in order).
#ifdef PANEL_TIMING_NO_SECOND_PASS           // (timing experiments only: wrong results)
                    if (false) {
#else
                    if (any_ed && own(rr0)) {
#endif
                        float* __restrict__ sp = ls + loc(rr0 + lane);
                        *sp = diag_lower_update(lT + (pp & 1) * kPanel * kPanel, edvec, *sp, lane);
                    }
                    if (lane == 0) *reinterpret_cast<volatile int*>(&s_ddone) = p;
                }
                stage_tiles();
                PPROF(5, wave == 1);
                if (p > 0) {
                    for (int k = uw; k < n_mine; k += NW - 1) do_strip(k);
                }
            }
            PPROF(7, wave == 1);
            __syncthreads();
        }
        }
#ifdef VIPRS_PANEL_PROFILE
        if (TEAM && item == 0 && member < 2 && tid == 0) {
            for (int p = 0; p < np && p < 64; p += (p < 4 || p > np - 4) ? 1 : 8) {
                const unsigned t = s_pprof[p][0];
                printf("m%d phase %2d @%7u: granule %4d T-apply %4d loop %5d replay %5d end %5d | upd: staged %5d pri %5d all %5d (x10ns)\n",
                       member, p, t - s_pprof[0][0], (int)(s_pprof[p][1] - t), (int)(s_pprof[p][2] - t), (int)(s_pprof[p][3] - t),
                       (int)(s_pprof[p][4] - t), (int)(s_pprof[p][4] - t), (int)(s_pprof[p][5] - t), (int)(s_pprof[p][6] - t),
                       (int)(s_pprof[p][7] - t));
            }
        }
#endif
        {   // teams: every member owns the final q of its own strips
            for (int li = tid; li < (TEAM ? n_own * kSW : b); li += NW * 64) {
                const int i = glob(li);
                if (i < b) {
                    const float v = SUMS ? lq[li] + A.dq * ls[li] : lq[li];   // upper form: q[j] += dq * dot (e_step.hpp:335)
                    if (TEAM) stage_store(A.q_out + s0 + i, v); else A.q[s0 + i] = v;
                }
            }
        }
        if (TEAM) {
            // eta / q are in-out and other members may still be reading the old values, so team blocks write to staging
            // buffers; the member that finishes the block LAST copies them into place (every member has then read what it
            // needed).  The staged values are written and read past the caches (agent-scope atomics: the members of a team
            // sit on different XCDs, each with its own L2), every wave waits for its own stores to complete before the
            // barrier, and the arrival is an agent-scope atomic -- no fence (__threadfence() here costs 8 % of the cfg3
            // sweep: its release writes back the whole L2).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) s_blk = (atomicAdd(A.arrive + item, 1) == TS - 1) ? 1 : 0;
            __syncthreads();
            const bool last = s_blk != 0;
            if (last) {
                for (int i = tid; i < b; i += NW * 64) {
                    A.eta[s0 + i] = __hip_atomic_load(A.eta_out + s0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    A.q[s0 + i] = __hip_atomic_load(A.q_out + s0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (tid == 0) atomicExch(A.arrive + item, 0);
            }
        }
        __syncthreads();
#ifdef VIPRS_SWEEP_TRACE
        if (tid == 0) {
            const unsigned k = atomicAdd(&g_sweep_trace_n, 1u);
            if (k < (1u << 15)) {
                g_sweep_trace[k][0] = ((unsigned long long)blockIdx.x << 32) | (unsigned)(TEAM ? 1 : 0);
                g_sweep_trace[k][1] = (unsigned long long)b;
                g_sweep_trace[k][2] = trace_t0;
                g_sweep_trace[k][3] = wall_clock64();
            }
        }
#endif
    }
    if (lane == 0 && my_skipped && member == 0) atomicAdd(A0.skipped, my_skipped);
}

// ---------------------------------------------------------------------------------------------
// The sweep over all dense blocks of a plan as ONE launch.  Workgroups 0 .. n_wg[0]-1 are the teams of
// the largest blocks, the next n_wg[1] the teams of the medium class, the rest small-block workers;
// a team workgroup that has finished its team's blocks carries on as a small-block worker.  One
// dispatch means: (i) the whole grid is resident by construction (the host sizes it to the
// occupancy of this kernel), so team members -- which wait on each other's hand-offs -- never depend
// on the launch order or timing of other kernels; (ii) workgroups are placed in index order, the
// critical path (the teams) first; (iii) no stream fork / join around the sweep.
// ---------------------------------------------------------------------------------------------
struct SweepArgs {
    EStepArgs<float> cls[3];     // per size class: block list, team geometry, queue counters
    int32_t qcap[3];             // LDS floats reserved for q per class (largest block of the class, padded)
    int32_t n_wg[2];             // team workgroups of class 0 / class 1 (0 = class empty)
    int32_t narrow0;             // class 0 works on the narrow team strips (panel_team_cols): teams of 8 and more members
    // launch bookkeeping, done by the workgroup that finishes LAST instead of by a prologue launch: the small-block queue
    // heads go back to 0 for the next sweep and the skip counter moves to `skipped_last` (what the host reads)
    int32_t* done;               // workgroups that have finished
    unsigned long long* skipped_last;
};

template <typename U, typename MODEL, int FORM, int NW, int CPL>
__global__ __launch_bounds__(NW * 64, PANEL_MIN_WAVES) void estep_sweep_kernel(SweepArgs S) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wg = blockIdx.x;
    const int team_cls = wg < S.n_wg[0] ? 0 : (wg < S.n_wg[0] + S.n_wg[1] ? 1 : 2);
    if (team_cls < 2) {
        // Narrow team strips (panel_team_cols) for the LARGE teams of the largest class in the UPPER-TRIANGULAR form only.
        // Measured, builds alternating on one box (EXPERIMENTS.md 4.2): upper fp32 cfg3max 1.09 -> 1.055 ms, int8 upper cfg3
        // 0.510 -> 0.5025; the SYMMETRIC form loses with them (cfg3 fp32 0.755 -> 0.768, cfg3max 0.99 -> 1.04, int8
        // +1 %), a block shared by 4 workgroups (the medium class; a populous large class) is bandwidth-bound and loses
        // 5-12 % with the 8-byte loads (tools/mixed_blocks_bench.py: 300 x 2 400 SNPs 1.65 -> 1.85 ms), and the mixture
        // chains, 3-4 x longer per step, never wait for their updaters (2.5 % slower): all of those keep the wide strips.
        // (mirrored upper form: integer LD -- a chain-bound sweep whose teams wait for the hand-off behind the priority strip --
        //  takes the narrow strips as well; fp32 LD, bound by the stream, keeps the 16-byte loads like the symmetric form)
        constexpr bool kNarrowForm = FORM == kFormMirror && sizeof(U) < 4;
        constexpr int CPL_N = (MODEL::kLaneParallel || !kNarrowForm) ? CPL : panel_team_cols<U>();
        const int wg_t = wg - (team_cls ? S.n_wg[0] : 0);
        if (CPL_N != CPL && team_cls == 0 && S.narrow0)
            panel_role<U, MODEL, FORM, NW, true, CPL_N>(S.cls[0], S.qcap[0], smem, wg_t);
        else
            panel_role<U, MODEL, FORM, NW, true, CPL>(S.cls[team_cls], S.qcap[team_cls], smem, wg_t);
        __syncthreads();
    }
    if (S.cls[2].n_blocks > 0) panel_role<U, MODEL, FORM, NW, false, CPL>(S.cls[2], S.qcap[2], smem, wg);
    __syncthreads();
    if (threadIdx.x == 0) {
        // (this workgroup's skip count and queue claims are atomics at L2, issued before this one)
        if (atomicAdd(S.done, 1) == (int)gridDim.x - 1) {
            S.cls[2].counter[0] = 0; S.cls[2].counter[1] = 0; S.cls[2].counter[2] = 0;
            atomicExch(S.skipped_last, atomicExch(S.cls[2].skipped, 0ull));      // this sweep's count; the running one back to 0
            atomicExch(S.done, 0);
        }
    }
}

}  // namespace viprs
