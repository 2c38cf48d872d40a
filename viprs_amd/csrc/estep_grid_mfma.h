// Batched grid E-step (e_step_grid, e_step.hpp:555-647) for up to 32 models at once: every LD row
// is read ONCE and applied to all models.
//
// For one SNP j the reference loops over the active models g and runs, per model, the scalar
// update (:613-620) followed by the axpy q[g, win(j)] = fma(R[j, .], dq * d_g, q[g, win(j)]) (:623).
// Models never interact, so the sweep can be re-ordered "all models per SNP" without changing any
// model's arithmetic.  Re-blocked into panels of 64 SNPs as in estep_panel.h:
//
//   wave 0 ("chain")   lane = (column half h, model g): the lane carries the 32 q values of model g for
//                      columns 32 h .. 32 h + 31 of the current panel in VGPRs.  Per SNP both halves
//                      evaluate the grid update of their model (inputs staged in LDS; the SNP's own q
//                      comes from the owning half through v_permlane32_swap) and apply row j of the
//                      diagonal tile -- staged in LDS as fp32, read as broadcast ds_read_b128 -- to
//                      their 32 columns.
//   waves 1..7         the trailing rank-64 update of the tiles RIGHT of the chain as a GEMM on the matrix
//                      cores:  Q[32 models x cols] += A[32 models x 64 rows] . R[64 rows x cols]  with
//                      v_mfma_f32_32x32x2_f32.  On gfx950 that instruction is bit-for-bit a k-ordered chain
//                      of fp32 fma (one rounding per product, no wider accumulator), i.e. exactly the
//                      reference's sequence q = fma(R[j][c], a_j, q) for j ascending -- so the batched
//                      kernel stays bit-identical to e_step_grid (tests/test_gpu_models.py and the golden
//                      fixtures hold the parity, both LD forms).  128-column tiles: one 16-byte load per
//                      lane is directly the B operand of four MFMAs (wtile_*); wave 1 additionally carries
//                      the 64-column tile the chain needs next across the barrier (tile_*).  They also
//                      stage the next panel's inputs / diagonal tile and flush the previous panel's outputs.
//
// q of a block lives in the accumulator registers of the updater waves for as long as the block is swept (resident form,
// grid_block_resident below): one workgroup per block of up to kGridResMaxCols SNPs (blocks pulled from a queue in
// descending size), a TEAM of workgroups with a migrating chain for larger blocks.  The columns LEFT of the chain keep
// receiving the later rows from the same accumulators -- the symmetric form's updates, and, over the MIRRORED storage of the
// upper-triangular form, the second-pass sums of update_q_factor_matrix (e_step.hpp:266-303).  (Rounds 2-5 also carried a
// streaming form with a lower-pass kernel and an epilogue kernel over the packed upper triangle; removed in round 6.)
#pragma once
#include "device_math.h"
#include "estep_panel.h"
#include "kernels_common.h"

namespace viprs {

#ifdef VIPRS_GRID_PROFILE
#define GPROF(slot, cond) do { if (blk == 0 && lane == 0 && p < 32 && (cond)) s_prof[p][slot] = (unsigned)wall_clock64(); } while (0)
#else
#define GPROF(slot, cond) do { } while (0)
#endif

constexpr int kGridIoPitch = kPanel + 1;           // LDS pitch of the per-panel input/output staging
// LDS carve (floats): io[2][4][32][65] | a[2][64][32] | diag[2][64][64] | carry[96][64] | qx[2][32][68]
constexpr int kGridIoArr = kGridModels * kGridIoPitch;
constexpr int kGridIoFloats = 4 * kGridIoArr;
constexpr int kGridAFloats = kPanel * kGridModels;
constexpr int kGridDiagFloats = kPanel * kPanel;
constexpr int kGridCarryFloats = (kPanel + 32) * 64;   // wave 1's tile (64 LD rows + 32 accumulators per lane) across a barrier
constexpr int kGridQxPitch = kPanel + 4;                  // 16-byte aligned rows, conflict-free 128-bit reads
constexpr int kGridQxFloats = kGridModels * kGridQxPitch;
constexpr int kGridLdsFloats = 2 * kGridIoFloats + 2 * kGridAFloats + 2 * kGridDiagFloats + kGridCarryFloats + 2 * kGridQxFloats;

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// ---- one 64-column tile on the matrix cores, split so that loads can run ahead of the MFMAs --------
// Q[model][c0 .. c0+63] += sum_k a[model][k] * R[row0 + k][c0 ..] for the 64 rows of a panel, k
// ascending, for all 32 models.

// this lane's column of the 64 rows of panel `rp` x 64-column tile `ct` (256 B per row and wave)
template <typename U>
__device__ __forceinline__ void tile_load_rows(float (&R)[kPanel], const U* __restrict__ base, int stride, int b,
                                               int rp, int ct, int lane) {
    const int row0 = rp * kPanel;
    const int last_row = min(kPanel, b - row0) - 1;
    const U* __restrict__ p = base + (int64_t)row0 * stride + ct * kPanel + lane;     // ct*64 + lane < stride
#pragma unroll
    for (int k = 0; k < kPanel; ++k) R[k] = static_cast<float>(p[(int64_t)min(k, last_row) * stride]);
    asm volatile("" ::: "memory");      // all 64 loads go out here (hipcc would sink each one to its MFMA)
}

// 64 MFMAs: rows (kr, kr + 1) of the panel per pair, k ascending
// (SCALED: the LDS array holds eta_diff, the A operand is scale * eta_diff -- the mirrored upper form, grid_block_resident)
template <bool SCALED = false>
__device__ __forceinline__ void tile_compute(f32x16& acc0, f32x16& acc1, const float (&R)[kPanel],
                                             const float* __restrict__ a_lds, int lane, float scale = 1.0f) {
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int kr = 0; kr < kPanel; kr += 2) {
        // [row kr : cols 0..31 | row kr+1 : cols 0..31]  and  [row kr : cols 32..63 | row kr+1 : cols 32..63]
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(R[kr]), __float_as_uint(R[kr + 1]), false, false);
        const float b0 = __uint_as_float(sw[0]), b1 = __uint_as_float(sw[1]);
        const float a_raw = a_lds[(kr + half) * kGridModels + l31];  // A[model = lane & 31][k = lane >> 5]
        const float aop = SCALED ? scale * a_raw : a_raw;
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aop, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aop, b1, acc1, 0, 0, 0);
    }
}

// ---- the same update on a 128-column tile: every lane fetches 4 consecutive columns (16 B for fp32) ----
// lane = (k slot = lane >> 5, n = lane & 31) loads columns c0 + 4n .. 4n+3 of row 2i + (lane >> 5): one
// load instruction covers 2 rows x 128 columns, which is directly the B operand of FOUR MFMAs (column
// 4n + j for j = 0..3) -- no cross-lane shuffle, 32 row loads + 16 accumulator loads in flight per tile
// (the memory pipeline takes at most 63 per wave, which throttles the 64-column version).
template <typename U> struct Vec4 { typedef U type __attribute__((ext_vector_type(4))); };
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // q columns start at arbitrary offsets

// accumulators acc[j][r]: model (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column c0 + 4 (lane & 31) + j.
// FULL: the tile lies inside the block (vector access); otherwise element-wise with column masks.
// (two halves, so that a tile's accumulators can be in flight while another tile is multiplied: `issue` only loads,
// `finish` applies the masks)
template <bool FULL>
__device__ __forceinline__ void wtile_load_acc_issue(f32x16 (&acc)[4], const EStepArgs<float>& A, const int* act, int64_t s0,
                                                     int b, int c0, bool lane_ok, int lane) {
    const int half = lane >> 5, c = c0 + 4 * (lane & 31);
    // raw loads straight into the accumulator registers, masks applied once everything is in flight
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int g = (r & 3) + 8 * (r >> 2) + 4 * half;
        const unsigned off = (unsigned)act[g] * (unsigned)A.m + (unsigned)s0;
        if (FULL) {
            const f32x4u v = *reinterpret_cast<const f32x4u*>(A.q + off + (lane_ok ? c : 0));
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j][r] = v[j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j][r] = A.q[off + ((lane_ok && c + j < b) ? c + j : 0)];
        }
    }
    asm volatile("" ::: "memory");
}
template <bool FULL>
__device__ __forceinline__ void wtile_load_acc_finish(f32x16 (&acc)[4], int b, int c0, bool lane_ok, int n_models, int lane) {
    const int half = lane >> 5, c = c0 + 4 * (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int g = (r & 3) + 8 * (r >> 2) + 4 * half;
        const bool ok = lane_ok && g < n_models;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j][r] = (ok && (FULL || c + j < b)) ? acc[j][r] : 0.0f;
    }
}
template <bool FULL>
__device__ __forceinline__ void wtile_load_acc(f32x16 (&acc)[4], const EStepArgs<float>& A, const int* act, int64_t s0,
                                               int b, int c0, bool lane_ok, int n_models, int lane) {
    wtile_load_acc_issue<FULL>(acc, A, act, s0, b, c0, lane_ok, lane);
    wtile_load_acc_finish<FULL>(acc, b, c0, lane_ok, n_models, lane);
}

template <bool FULL>
__device__ __forceinline__ void wtile_store_acc(const f32x16 (&acc)[4], const EStepArgs<float>& A, const int* act,
                                                int64_t s0, int b, int c0, bool lane_ok, int n_models, int lane) {
    const int half = lane >> 5, c = c0 + 4 * (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int g = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (lane_ok && g < n_models) {
            const unsigned off = (unsigned)act[g] * (unsigned)A.m + (unsigned)s0 + c;
            if (FULL) {
                *reinterpret_cast<f32x4u*>(A.q + off) = f32x4u{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < b) A.q[off + j] = acc[j][r];
            }
        }
    }
}

// q[model][column] += dq * acc (the second pass of the upper-triangular form, e_step.hpp:300: q[j] += dq * dot)
template <bool FULL>
__device__ __forceinline__ void wtile_add_to_q(const f32x16 (&acc)[4], const EStepArgs<float>& A, const int* act, int64_t s0,
                                               int b, int c0, bool lane_ok, int n_models, int lane, float dq) {
    const int half = lane >> 5, c = c0 + 4 * (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int g = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (lane_ok && g < n_models) {
            const unsigned off = (unsigned)act[g] * (unsigned)A.m + (unsigned)s0 + c;
            if (FULL) {
                const f32x4u v = *reinterpret_cast<const f32x4u*>(A.q + off);
                *reinterpret_cast<f32x4u*>(A.q + off) =
                    f32x4u{v[0] + dq * acc[0][r], v[1] + dq * acc[1][r], v[2] + dq * acc[2][r], v[3] + dq * acc[3][r]};
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < b) A.q[off + j] = A.q[off + j] + dq * acc[j][r];
            }
        }
    }
}

// ---- the chain wave's work for one panel: 64 serial SNP updates for all models ------------------------------
// (STORE_D: `la` receives eta_diff itself instead of a = dq * eta_diff -- the mirrored upper form, whose updater waves need
//  both: eta_diff for the second-pass sums, dq * eta_diff, formed again from the same operands, for the trailing updates)
template <bool SYM, bool EXACT = true, bool STORE_D = false>
__device__ __forceinline__ void grid_chain_panel(const EStepArgs<float>& A, float* io, float* la, float* dg, float* qx, int p, int b,
                                                 int64_t s0, int lane, int n_models, float dq, const ExpTab& tab
#ifdef VIPRS_GRID_PROFILE
                                                 , int blk, unsigned* prof
#endif
                                                 ) {
    const int cg = lane & 31, ch = lane >> 5;
    const bool has_model = cg < n_models;
#ifdef VIPRS_GRID_PROFILE
    unsigned (*s_prof)[12] = reinterpret_cast<unsigned (*)[12]>(prof);
#endif
    const int r0 = p * kPanel;
    const int nrows = min(kPanel, b - r0);
    float* iob = io + (p & 1) * kGridIoFloats + cg * kGridIoPitch;
    float* lap = la + (p & 1) * kGridAFloats + cg;
    const float* drow = dg + (p & 1) * kGridDiagFloats + 32 * ch;
    float* qmine = qx + (p & 1) * kGridQxFloats + cg * kGridQxPitch + 32 * ch;
    // this lane's 32 columns of its model's q
    // (register pairs: the row application below is v_pk_fma_f32, two columns per instruction)
    f32x2 qv[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(qmine + 4 * i);
        qv[2 * i] = has_model ? f32x2{v[0], v[1]} : f32x2{0.0f, 0.0f};
        qv[2 * i + 1] = has_model ? f32x2{v[2], v[3]} : f32x2{0.0f, 0.0f};
    }
    const float betav = A.std_beta[s0 + min(r0 + lane, b - 1)];
#ifdef VIPRS_GRID_PROFILE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    GPROF(2, true);
    // row jj of the diagonal tile for this half: register c <-> column 32 h + ((c + 16 g4) & 31)
    f32x2 rw[16];
    auto load_row = [&](f32x2 (&dst)[16], int jr, int g4) {
        const float* rp = drow + jr * kPanel;
        const int o0 = 16 * (g4 & 1), o1 = 16 * ((g4 + 1) & 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(rp + o0 + 4 * i);
            const f32x4 y = *reinterpret_cast<const f32x4*>(rp + o1 + 4 * i);
            dst[2 * i] = f32x2{x[0], x[1]};
            dst[2 * i + 1] = f32x2{x[2], x[3]};
            dst[8 + 2 * i] = f32x2{y[0], y[1]};
            dst[8 + 2 * i + 1] = f32x2{y[2], y[3]};
        }
    };
    load_row(rw, 0, 0);
    float mm = iob[0 * kGridIoArr], ulog = iob[1 * kGridIoArr], hvt = iob[2 * kGridIoArr],
          eta_old = iob[3 * kGridIoArr];
#pragma unroll 1
    for (int g4 = 0; g4 < kPanel / 16; ++g4) {
        const bool hi_owner = g4 >= 2;                                   // columns 32.. belong to half 1
        const bool owner = (ch == 1) == hi_owner;
        // (one LDS base per group of 16 SNPs: the SNP index inside the group is an immediate offset)
        float* const iog = iob + 16 * g4;
        float* const lag = lap + 16 * g4 * kGridModels;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int jj = 16 * g4 + k;                                  // wave-uniform
            const bool live = jj < nrows;
            // next SNP's inputs: in flight while this one is evaluated
            const int jn = min(jj + 1, kPanel - 1);
            const int kn = (k < 15) ? k + 1 : ((g4 < kPanel / 16 - 1) ? 16 : 15);   // = jn - 16 g4
            const float mm_n = iog[0 * kGridIoArr + kn], ulog_n = iog[1 * kGridIoArr + kn],
                        hvt_n = iog[2 * kGridIoArr + kn], eta_n = iog[3 * kGridIoArr + kn];
            // the SNP's own q from the half that owns its column
            const unsigned qbits = __float_as_uint(qv[k >> 1][k & 1]);
            auto sw = __builtin_amdgcn_permlane32_swap(qbits, qbits, false, false);
            const float qcur = __uint_as_float(hi_owner ? sw[1] : sw[0]);
            const float beta = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, betav), jj));
            const float mu = mm * (beta - qcur);                         // e_step.hpp:613
            const float u = ulog + hvt * mu * mu;                        // :616
            const float gamma = EXACT ? sigmoid_exact<kLookupPerLane>(u, tab) : sigmoid_fast(u);   // :617
            const float d = gamma * mu - eta_old;                        // :620
            const float a = (live && has_model) ? dq * d : 0.0f;
            // the next diagonal row goes out behind the sigmoid's table lookup (LDS is in-order).  (A scheduling barrier
            // stood here; without it the compiler still keeps the row's reads behind the lookup and places a few of the
            // fmas below into the sigmoid's shadows: configs[4] 1.986 -> 1.951 ms, tools/multi_ab.py.)
            f32x2 rn[16];
            load_row(rn, jn, (jj + 1) >> 4);
            const f32x2 a2 = {a, a};
#pragma unroll
            for (int c = 0; c < 16; ++c) {                               // :623, one IEEE fma per column
                qv[c] = __builtin_elementwise_fma(rw[c], a2, qv[c]);
                asm volatile("" : "+v"(qv[c]));      // apply now (hipcc would sink the chain to its use)
            }
            if (SYM) qv[k >> 1][k & 1] -= (live && owner && has_model) ? d : 0.0f;   // :629
            // Outputs and a: stored UNCONDITIONALLY by both column halves -- the two lanes of a model hold the same numbers
            // (same inputs, same q_j) and write them to the same address; slots of models >= n_models and of SNPs past a
            // partial last panel are never flushed (flush_outputs: g < n_models, j < b).  No exec-mask juggling, no
            // s_cbranch_execz on the serial chain (two masked regions per SNP before; int8 LD, whose sweep is bound by the
            // chain: 1.760 -> 1.719 ms; fp32 LD unchanged).
            iog[0 * kGridIoArr + k] = mu;
            iog[1 * kGridIoArr + k] = gamma;
            iog[2 * kGridIoArr + k] = d;
            iog[3 * kGridIoArr + k] = eta_old + d;                        // :633
            lag[k * kGridModels] = STORE_D ? ((live && has_model) ? d : 0.0f) : a;
            mm = mm_n; ulog = ulog_n; hvt = hvt_n; eta_old = eta_n;
#pragma unroll
            for (int c = 0; c < 16; ++c) rw[c] = rn[c];
            __builtin_amdgcn_sched_barrier(0);       // keep the prefetch distance at one SNP
        }
        // rotate the q registers by one group of 16 columns
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x2 t = qv[c];
            qv[c] = qv[8 + c];
            qv[8 + c] = t;
        }
    }
    GPROF(3, true);
    if (has_model) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<f32x4*>(qmine + 4 * i) = f32x4{qv[2 * i][0], qv[2 * i][1], qv[2 * i + 1][0], qv[2 * i + 1][1]};
    }
}

// ---- the chain wave's work for one panel, row application on the MATRIX CORE (round 3; -DVIPRS_GRID_MFMA_CHAIN) -------
// BUILT, MEASURED, NOT THE DEFAULT: bit-identical to the VALU chain above on every grid parity test, 81 instead of 97
// instructions per SNP -- and 9 % SLOWER (1 700 equal blocks of 650 SNPs, purely chain-bound: 1.86 against 1.71 ms; cfg3
// 2.50 against 2.34 ms), with the carry moved to the wave that shares the chain's SIMD or not.  The two MFMAs per SNP
// occupy the issuing wave for their 2 x 64 clocks: a wave's own VALU instructions do not run under its own MFMA, so the
// 16 v_pk_fma_f32 + 8 ds_read_b128 they replace were the cheaper way to spend those clocks.  (A v_mfma_f32_32x32x1_2b
// variant -- one MFMA per SNP, own column on the VALU -- would save half of that and land where the VALU chain is.)
// The VALU chain above spends most of its ~1 000 clocks per SNP on things that are not the serial dependency: the 32
// columns of the diagonal-tile row per lane come in as 8 ds_read_b128 (~27 clk of issue each) and go out as 16
// v_pk_fma_f32.  Here the panel's q lives in two 32x32 accumulator tiles, C[column i][model n] (lane = (half, model),
// register r <-> column (r & 3) + 8 (r >> 2) + 4 half), and the row of SNP jj is applied to all 64 columns of all 32
// models by TWO v_mfma_f32_32x32x2_f32:
//     k = 0:  A = D[jj][column],          B = a[model]            q = fma(D, a, q)          e_step.hpp:623
//     k = 1:  A = -1 at column jj, else 0, B = d[model] (SYM)      q[jj] = q[jj] - d         e_step.hpp:629
// (fma(-1, d, x) rounds as x - d does; fma(0, d, x) == x: q is never -0).  The A operand is ONE ds_read_b32 per lane (the
// row, lane = column) + one v_permlane32_swap.  The MFMA's ~64 clocks of latency stay off the critical path: the q the
// NEXT SNP needs (column jj+1) is read from the accumulators before this step's MFMAs are issued -- the previous step's
// have long completed -- and gets row jj by one VALU fma with D[jj][jj+1] from a v_readlane; the matrix core applies the
// same fma to the accumulator's copy.  Fully unrolled (the register <-> column map of the accumulators is fixed).
template <bool SYM, bool EXACT = true>
__device__ __forceinline__ void grid_chain_panel_mfma(const EStepArgs<float>& A, float* io, float* la, float* dg, float* qx, int p,
                                                      int b, int64_t s0, int lane, int n_models, float dq, const ExpTab& tab) {
    const int cg = lane & 31, ch = lane >> 5;
    const bool has_model = cg < n_models;
    const int r0 = p * kPanel;
    const int nrows = min(kPanel, b - r0);
    float* iob = io + (p & 1) * kGridIoFloats + cg * kGridIoPitch;
    float* lap = la + (p & 1) * kGridAFloats + cg;
    const float* drow = dg + (p & 1) * kGridDiagFloats + lane;             // D[jj][lane]
    float* qm = qx + (p & 1) * kGridQxFloats + cg * kGridQxPitch;          // this model's 64 columns
    f32x16 acc0, acc1;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(qm + 8 * g + 4 * ch);
        const f32x4 w = *reinterpret_cast<const f32x4*>(qm + 32 + 8 * g + 4 * ch);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc0[4 * g + e] = has_model ? v[e] : 0.0f;
            acc1[4 * g + e] = has_model ? w[e] : 0.0f;
        }
    }
    const float betav = A.std_beta[s0 + min(r0 + lane, b - 1)];
    float mm = iob[0 * kGridIoArr], ulog = iob[1 * kGridIoArr], hvt = iob[2 * kGridIoArr], eta_old = iob[3 * kGridIoArr];
    float rowv = drow[0];
    float qcur;
    {
        const unsigned x = __float_as_uint(acc0[0]);                      // column 0: half 0, register 0
        auto sx = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        qcur = __uint_as_float(sx[0]);
    }
#pragma unroll
    for (int jj = 0; jj < kPanel; ++jj) {
        const bool live = jj < nrows;
        const int jn = (jj + 1 < kPanel) ? jj + 1 : jj;
        // next SNP's inputs and diagonal-tile row: in flight while this one is evaluated
        const float mm_n = iob[0 * kGridIoArr + jn], ulog_n = iob[1 * kGridIoArr + jn], hvt_n = iob[2 * kGridIoArr + jn],
                    eta_n = iob[3 * kGridIoArr + jn];
        const float rowv_n = drow[jn * kPanel];
        const float beta = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, betav), jj));
        const float mu = mm * (beta - qcur);                         // e_step.hpp:613
        const float u = ulog + hvt * mu * mu;                        // :616
        const float gamma = EXACT ? sigmoid_exact<kLookupPerLane>(u, tab) : sigmoid_fast(u);   // :617
        const float d = gamma * mu - eta_old;                        // :620
        const float a = (live && has_model) ? dq * d : 0.0f;
        const float dsub = (SYM && live && has_model) ? d : 0.0f;
        __builtin_amdgcn_sched_barrier(0);      // (the accumulator reads below wait for the previous step's MFMAs: keep them here)
        // A operands: lanes 0..31 the row (k = 0), lanes 32..63 the diagonal selector (k = 1)
        auto sr = __builtin_amdgcn_permlane32_swap(__float_as_uint(rowv), __float_as_uint(rowv), false, false);
        const float row_lo = __uint_as_float(sr[0]), row_hi = __uint_as_float(sr[1]);      // D[jj][lane & 31], D[jj][32 + (lane & 31)]
        const float A0 = ch ? ((SYM && cg == jj) ? -1.0f : 0.0f) : row_lo;
        const float A1 = ch ? ((SYM && cg + 32 == jj) ? -1.0f : 0.0f) : row_hi;
        const float B = ch ? dsub : a;
        // the next SNP's q: its column out of the accumulators (rows < jj applied) + row jj on the VALU
        float qnext = 0.0f;
        if (jj + 1 < kPanel) {
            constexpr int kDummy = 0; (void)kDummy;
            const int c1 = jj + 1, i1 = c1 & 31, r1 = (i1 & 3) + 4 * (i1 >> 3), h1 = (i1 >> 2) & 1;
            const unsigned x = __float_as_uint((c1 >> 5) ? acc1[r1] : acc0[r1]);
            auto sx = __builtin_amdgcn_permlane32_swap(x, x, false, false);
            const float xq = __uint_as_float(h1 ? sx[1] : sx[0]);
            const float dn = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rowv), c1));   // D[jj][jj+1]
            qnext = __builtin_fmaf(dn, a, xq);                       // :623 for column jj+1
        }
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B, acc0, 0, 0, 0);      // :623 (+ :629) for all 64 columns x 32 models
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B, acc1, 0, 0, 0);
        if (has_model && live && ch == 0) {
            iob[0 * kGridIoArr + jj] = mu;
            iob[1 * kGridIoArr + jj] = gamma;
            iob[2 * kGridIoArr + jj] = d;
            iob[3 * kGridIoArr + jj] = eta_old + d;                   // :633
        }
        if (ch == 0) lap[jj * kGridModels] = a;
        mm = mm_n; ulog = ulog_n; hvt = hvt_n; eta_old = eta_n; rowv = rowv_n; qcur = qnext;
        __builtin_amdgcn_sched_barrier(0);
    }
    if (has_model) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<f32x4*>(qm + 8 * g + 4 * ch) = f32x4{acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]};
            *reinterpret_cast<f32x4*>(qm + 32 + 8 * g + 4 * ch) = f32x4{acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]};
        }
    }
}

constexpr int kGridWaves = 8;                      // 1 chain wave + 7 updater waves
constexpr int kGridNU = kGridWaves - 1;
constexpr int kGridPerWave = (kGridModels + kGridNU - 1) / kGridNU;     // model rows a wave stages / flushes
constexpr int kGridDiagPerWave = (kPanel + kGridNU - 1) / kGridNU;      // diagonal-tile rows a wave stages


// =====================================================================================================================
// Resident form (blocks of up to kGridResMaxCols SNPs): q of the WHOLE block lives in the accumulator registers of the
// updater waves for as long as the block is swept -- read from the state once, written once -- and every LD row of the
// block is read exactly once: the tiles LEFT of the chain receive the later rows from the same waves, in the same row order.
//
//   wave 0      chain (grid_chain_panel).
//   wave 4      carry: takes the 64 columns of panel p+1 out of their owner's registers one phase ahead (through LDS,
//               `cq`), applies a_{p-1} during phase p and a_p between the chain's two panels, hands them to the chain
//               (qx) -- the accumulators stay in ITS registers across the barriers (own loop over the phases).
//   the others  owners (waves 1, 2, 3, 5, 6, 7 = owner 0 .. 5): owner o holds the 128-column tiles T = o and T = o + 6 (32 models x 128 columns = 64
//               accumulator registers each).  Per phase p they apply the rows of panel p-1 (a_{p-1}) to their tiles --
//               LD rows in 4 chunks of 8 16-byte loads -- except to the panels that
//               are "away": p-1 (its own rows are the chain's diagonal tile), p and p+1 (with the carry).  The B operand
//               of a masked half is 0: fma(a, 0, acc) == acc (q is never -0: it starts at +0 and a sum is -0 only if
//               both addends are).  A panel comes back from the chain one phase after it was swept (qx).
// Same arithmetic per (model, column) as e_step_grid: the rows arrive in ascending order.
// =====================================================================================================================
// The carry is wave 4: the hardware puts waves w and w + 4 of a workgroup on the same SIMD, i.e. wave 4 shares the chain
// wave's matrix pipe -- and the chain now issues two MFMAs per SNP.  An owner wave there (128 back-to-back MFMAs per tile)
// makes each of them queue for up to two MFMA times (64 clocks each); the carry's few MFMAs mostly fall between the
// chain's panels.
constexpr int kGridCarryWave = 4;
// (kGridResOwners = 6, kGridResSlots = 2, kGridResMaxCols = 1 536: kernels_common.h -- the plan sorts by the same limit)

// rows of panel `pp` x the 128 columns from c0 applied to a tile whose accumulators are in registers; `on_l` / `on_r`: the
// left / right 64 columns take the update (a masked half gets B = 0).  LD rows in chunks of 32 VGPRs (fp32: 8 row pairs, one
// 16-byte load per lane = rows 2i, 2i+1 x 128 columns = the B operands of four MFMAs): a chunk's loads are not overlapped
// with this wave's own MFMAs -- the other owner waves of the SIMD fill the matrix pipe meanwhile (a deeper ring next to 128
// accumulator registers pushes loop-invariant addresses into scratch).

// Four consecutive columns of one LD row as loaded (one dword for int8, two for int16, four for fp32), converted with
// static_cast<float> (e_step.hpp:173) when consumed.  A masked half of a tile is masked on the raw dwords -- once per load
// instead of once per element (float(0) == 0).
// (Measured and withdrawn: one-instruction conversions, v_cvt_f32_i32 with an SDWA byte / word select and sign extension,
// as inline asm -- the instructions are right for all 2^16 inputs (tools/ubench/sdwa_cvt.hip) but an inline-asm result
// consumed as an MFMA operand gave wrong sums in the kernel: hipcc does not see the asm as a VALU instruction and leaves
// out whatever it keeps between a VALU write and the matrix core.  The plain casts cost the same time.)
template <typename U> struct RawCols4;
template <> struct RawCols4<float> {
    f32x4 v;
    static constexpr int kDwords = 4;
    __device__ __forceinline__ void mask(bool on) { v = on ? v : f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    // keep column j (of the four) iff row > col0 + j: the part of a mirrored diagonal tile BELOW the diagonal
    __device__ __forceinline__ void mask_below(int row, int col0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = row > col0 + j ? v[j] : 0.0f;
    }
    template <int J> __device__ __forceinline__ float get() const { return v[J]; }
};
template <> struct RawCols4<int8_t> {
    unsigned w;
    static constexpr int kDwords = 1;
    __device__ __forceinline__ void mask(bool on) { w = on ? w : 0u; }
    __device__ __forceinline__ void mask_below(int row, int col0) {
        unsigned m = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) m |= row > col0 + j ? 0xFFu << (8 * j) : 0u;
        w &= m;
    }
    template <int J> __device__ __forceinline__ float get() const { return static_cast<float>(static_cast<int8_t>(w >> (8 * J))); }
};
template <> struct RawCols4<int16_t> {
    unsigned w[2];
    static constexpr int kDwords = 2;
    __device__ __forceinline__ void mask(bool on) { w[0] = on ? w[0] : 0u; w[1] = on ? w[1] : 0u; }
    __device__ __forceinline__ void mask_below(int row, int col0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned m = (row > col0 + 2 * h ? 0xFFFFu : 0u) | (row > col0 + 2 * h + 1 ? 0xFFFF0000u : 0u);
            w[h] &= m;
        }
    }
    template <int J> __device__ __forceinline__ float get() const { return static_cast<float>(static_cast<int16_t>(w[J >> 1] >> (16 * (J & 1)))); }
};
template <typename U> __device__ __forceinline__ RawCols4<U> load_cols4(const char* p);
template <> __device__ __forceinline__ RawCols4<float> load_cols4<float>(const char* p) {
    return RawCols4<float>{*reinterpret_cast<const f32x4*>(p)};
}
template <> __device__ __forceinline__ RawCols4<int8_t> load_cols4<int8_t>(const char* p) {
    return RawCols4<int8_t>{*reinterpret_cast<const unsigned*>(p)};
}
template <> __device__ __forceinline__ RawCols4<int16_t> load_cols4<int16_t>(const char* p) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    return RawCols4<int16_t>{{t.x, t.y}};
}
// MIRV (mirrored upper form): the LDS array holds eta_diff and the A operand is scale * eta_diff (scale = dq: a trailing
// update; 1: a second-pass sum); `tri` = 0 / 1: the left / right 64 columns are the panel's OWN diagonal tile and take only
// its part below the diagonal (row > column: R[j, i] = R[i, j] for the SNPs i < j of the panel), -1: neither.
template <typename U, bool MIRV = false>
__device__ __forceinline__ void res_tile_apply(f32x16 (&acc)[4], const U* __restrict__ base, int stride, int pp, int c0,
                                               int lane, bool on_l, bool on_r, const float* __restrict__ a_lds,
                                               float scale = 1.0f, int tri = -1) {
    const int half = lane >> 5, l31 = lane & 31;
    const bool lane_on = (lane & 16) ? on_r : on_l;
    const bool lane_tri = MIRV && tri >= 0 && ((lane & 16) != 0) == (tri == 1);
    // first column of this lane inside its panel; opaque, so that the 32 x 4 row-against-column tests of the triangular half
    // are made where they are used and not hoisted out of the phase loop into as many mask registers
    int tri_col0 = 4 * (l31 & 15);
    if (MIRV) asm volatile("" : "+v"(tri_col0));
    int col = c0 + 4 * l31;
    if (col >= stride) col = c0;                      // right half of the last (odd) tile: masked by the caller
    const unsigned voff = (unsigned)((half * stride + (col - c0)) * (int)sizeof(U));
    const char* __restrict__ sb = reinterpret_cast<const char*>(base + (int64_t)pp * kPanel * stride + c0);
    const size_t step = (size_t)2 * stride * sizeof(U);
    // a chunk = 32 VGPRs of raw rows whatever the LD type: 8 row pairs of fp32, 16 of int16, all 32 of int8 -- integer LD
    // takes fewer memory round trips per tile, not fp32's four with a quarter of the bytes each
    constexpr int CH = 8 * 4 / RawCols4<U>::kDwords, NCH = kPanel / 2 / CH;
    RawCols4<U> ring[CH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int i = 0; i < CH; ++i) ring[i] = load_cols4<U>(sb + (c * CH + i) * step + voff);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            RawCols4<U> v = ring[i];
            v.mask(lane_on);
            if (MIRV && tri >= 0) { if (lane_tri) v.mask_below(2 * (c * CH + i) + half, tri_col0); }
            const float a_raw = a_lds[(2 * (c * CH + i) + half) * kGridModels + l31];    // A[model = lane & 31][k = lane >> 5]
            const float aop = MIRV ? scale * a_raw : a_raw;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aop, v.template get<0>(), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aop, v.template get<1>(), acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(aop, v.template get<2>(), acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(aop, v.template get<3>(), acc[3], 0, 0, 0);
        }
    }
}

// Teams (blocks beyond one workgroup's accumulator capacity, TS > 1): member m of a team of TS workgroups on TS CUs holds
// the tiles [12 m, 12 m + 12) of the block -- panels [24 m, 24 m + 24) -- resident in ITS owner waves' accumulators, and
// every member runs the same phase loop over all panels of the block.  The CHAIN MIGRATES: the member whose range
// holds panel p runs its 64 serial updates (its carry wave has the panel ready exactly as in a single-workgroup block),
// the others RECEIVE the panel's a-vector (64 x 32 floats = 16 KB, published by the holder's carry wave as generation-
// tagged 8-byte granules: relaxed agent-scope atomics, the data is the flag) and apply it to their own tiles one phase
// behind -- tiles right of the chain, and in the symmetric form the tiles left of it, which is what the lower-pass
// kernel did for such blocks.  No q value ever crosses a CU (the chain goes to where q lives), every LD row of the
// block is read once, q once in and once out, and no chain step is computed twice.
struct GridTeam {
    int member = 0, size = 1;
    unsigned long long* gran = nullptr;       // this block's a-vector granules: [panel][64 x 32]
    uint32_t tag_base = 0;                    // generation << 12; a granule of panel p carries tag_base + p + 1
};
constexpr int kGridTeamPanels = 2 * kGridResOwners * kGridResSlots;       // 24 panels (12 tiles) per member

// MIR (upper-triangular form over MIRRORED storage, abi_plan.hip: mirror_lower_kernel): the reference's second pass
//     q[g, i] += dq * sum_{j > i} R[i, j] eta_diff[g, j]                     (update_q_factor_matrix, e_step.hpp:266-303)
// runs inside the sweep, on the same accumulators: once a panel has gone to the chain, its owner's registers start again
// at 0 and collect the SUMS -- R[i, j] read as R[j, i] from the later rows j, in ascending order (the reference's dot, bit
// for bit), with eta_diff as the A operand where the trailing updates right of the chain take dq * eta_diff (`la` holds
// eta_diff; the owners form dq * eta_diff from the same two operands as the chain does).  The rows of a panel reach its own
// columns through the part of the diagonal tile below the diagonal (res_tile_apply `tri`).  At the end of the block
// q = q_chain + dq * sum (q_chain: what the chain's flush left in the state).  No epilogue kernel, every LD entry of the
// mirrored block read once.
template <typename U, bool SYM, bool EXACT>
__device__ __forceinline__ void grid_block_resident(const EStepArgs<float>& A, float* io, float* la, float* dg, f32x4* cy,
                                                    float* qx, const int* s_act, const BlockDesc& bd, int wave, int lane,
                                                    int n_models, float dq, const ExpTab& tab, const GridTeam tm = GridTeam()) {
    constexpr bool MIR = !SYM;                 // the upper-triangular arithmetic, over mirrored storage
    constexpr int NU = kGridNU;
    const int64_t s0 = bd.start;
    const int b = bd.size, stride = bd.stride;
    const U* __restrict__ base = static_cast<const U*>(A.ld_dense) + bd.ld_off;
    const int np = (b + kPanel - 1) / kPanel;
    float* cq = reinterpret_cast<float*>(cy + 16 * 64);          // [32 models][64 columns]: a panel on its way to the carry
    const int half = lane >> 5, n = lane & 31;
    // team geometry (a single workgroup: member 0 of 1, every panel its own)
    const bool team = tm.size > 1;
    const int p_lo = tm.member * kGridTeamPanels, p_hi = min(np, p_lo + kGridTeamPanels);
    const int T_lo = tm.member * kGridResOwners * kGridResSlots;
    auto mine = [&](int p) { return !team || (p >= p_lo && p < p_hi); };
    const int p_end = np;
    // the a-vector of panel p: published by the holder's carry wave, received by the other members' chain wave
    auto publish_a = [&](int p) {
        const float* src = la + (p & 1) * kGridAFloats;
        unsigned long long* g = tm.gran + (int64_t)p * kGridAFloats;
        const unsigned long long tag = (unsigned long long)(tm.tag_base + (unsigned)(p + 1)) << 32;
#pragma unroll 8
        for (int i = 0; i < kGridAFloats / 64; ++i)
            __hip_atomic_store(g + i * 64 + lane, tag | (unsigned long long)__float_as_uint(src[i * 64 + lane]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    };
    auto receive_a = [&](int p) {
        float* dst = la + (p & 1) * kGridAFloats;
        const unsigned long long* g = tm.gran + (int64_t)p * kGridAFloats;
        const unsigned want = tm.tag_base + (unsigned)(p + 1);
        for (int i0 = 0; i0 < kGridAFloats / 64; i0 += 8) {
            unsigned long long v[8];
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v[k] = __hip_atomic_load(g + (i0 + k) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (unsigned)(v[k] >> 32) == want;
                }
                if (__all(ok)) break;
                if (spins > (1u << 22)) {              // ~seconds: give up loudly, never hang
                    if (lane == 0) atomicExch(A.error, 1);
                    break;
                }
                // a wait has timed out somewhere (this wave's earlier chunk included): the launch is lost and reported; every
                // later wait returns at once instead of spinning out its own timeout (64 chunks x panels = minutes)
                if ((spins & 255u) == 0u && __hip_atomic_load(A.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                __builtin_amdgcn_s_sleep(16);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) dst[(i0 + k) * 64 + lane] = __uint_as_float((unsigned)v[k]);
        }
    };

    // the per-panel input / output staging (waves 1..7; sym: q of a panel is NOT written here)
    auto stage_inputs = [&](int pp, bool with_q) {
        float* dst = io + (pp & 1) * kGridIoFloats;
        const int r0 = pp * kPanel;
        const int j = r0 + lane;
        const bool ok = j < b;
        const unsigned jo = (unsigned)s0 + (ok ? j : 0);
        float v[kGridPerWave][5];
#pragma unroll
        for (int i = 0; i < kGridPerWave; ++i) {
            const int g = min(wave - 1 + NU * i, kGridModels - 1);
            const unsigned off = (unsigned)s_act[g] * (unsigned)A.m + jo;
            v[i][0] = A.mu_mult[off]; v[i][1] = A.u_logs[off]; v[i][2] = A.shvt[off]; v[i][3] = A.eta[off];
            v[i][4] = with_q ? A.q[off] : 0.0f;
        }
        const int nrows = min(kPanel, b - r0);
        const U* __restrict__ dp = base + (int64_t)r0 * stride + r0 + lane;
        float dv[kGridDiagPerWave];
#pragma unroll
        for (int i = 0; i < kGridDiagPerWave; ++i)
            dv[i] = static_cast<float>(dp[(int64_t)min(wave - 1 + NU * i, nrows - 1) * stride]);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < kGridPerWave; ++i) {
            const int g = wave - 1 + NU * i;
            if (g < n_models) {
#pragma unroll
                for (int a = 0; a < 4; ++a) dst[a * kGridIoArr + g * kGridIoPitch + lane] = ok ? v[i][a] : 0.0f;
                if (with_q) qx[(pp & 1) * kGridQxFloats + g * kGridQxPitch + lane] = ok ? v[i][4] : 0.0f;
            }
        }
        float* dd = dg + (pp & 1) * kGridDiagFloats;
#pragma unroll
        for (int i = 0; i < kGridDiagPerWave; ++i) {
            const int t = wave - 1 + NU * i;
            // (mirrored storage: the chain applies a row to the columns RIGHT of its SNP only)
            if (t < kPanel) dd[t * kPanel + lane] = (MIR && lane <= t) ? 0.0f : dv[i];
        }
    };
    auto flush_outputs = [&](int pp) {
        const float* src = io + (pp & 1) * kGridIoFloats;
        const float* qs = qx + (pp & 1) * kGridQxFloats;
        const int j = pp * kPanel + lane;
        if (j < b) {
#pragma unroll
            for (int i = 0; i < kGridPerWave; ++i) {
                const int g = wave - 1 + NU * i;
                if (g < n_models) {
                    const unsigned off = (unsigned)s_act[g] * (unsigned)A.m + (unsigned)(s0 + j);
                    A.var_mu[off] = src[0 * kGridIoArr + g * kGridIoPitch + lane];
                    A.var_gamma[off] = src[1 * kGridIoArr + g * kGridIoPitch + lane];
                    A.eta_diff[off] = src[2 * kGridIoArr + g * kGridIoPitch + lane];
                    A.eta[off] = src[3 * kGridIoArr + g * kGridIoPitch + lane];
                    // upper-triangular form: q as the chain left it (the owners add dq * the second-pass sums when the block is
                    // done); symmetric form: the panel goes back to its owner and is stored with the block
                    if (!SYM) A.q[off] = qs[g * kGridQxPitch + lane];
                }
            }
        }
    };

    if (wave == 0) {
        // ================================================= chain =========================================================
        __syncthreads();                                            // B0: inputs of panel 0 staged
        for (int p = 0; p <= p_end; ++p) {
            __syncthreads();                                        // mid: the carry has put panel p into qx
            if (p < np && mine(p)) {
#ifdef VIPRS_GRID_MFMA_CHAIN
                grid_chain_panel_mfma<SYM, EXACT>(A, io, la, dg, qx, p, b, s0, lane, n_models, dq, tab);
#else
                grid_chain_panel<SYM, EXACT, MIR>(A, io, la, dg, qx, p, b, s0, lane, n_models, dq, tab
#ifdef VIPRS_GRID_PROFILE
                                      , 1, nullptr
#endif
                                      );
#endif
            } else if (p < np) {
                receive_a(p);                                       // another member holds the chain: its a-vector -> la
            }
            __syncthreads();                                        // end
        }
    } else if (wave == kGridCarryWave) {
        // ================================================= carry =========================================================
        if (mine(0)) stage_inputs(0, true);
        __syncthreads();                                            // B0
        // the carried panel: during phase p it is panel p+1 (taking a_{p-1}), between the chain's panels it is finished
        // with a_p and handed to the chain; then the next one is picked up (ONE set of 32 accumulators, in registers
        // across the barriers).  Teams: only for the panels of this member's own range (the a-vectors it applies are
        // its own chain's or the ones its chain wave received).
        f32x16 c0v, c1v;
#pragma unroll
        for (int r = 0; r < 16; ++r) { c0v[r] = 0.0f; c1v[r] = 0.0f; }
        for (int p = 0; p <= p_end; ++p) {
            if (p > 0 && p < np && mine(p)) {
                float R1[kPanel];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const f32x4 v = cy[i * 64 + lane];
#pragma unroll
                    for (int e = 0; e < 4; ++e) R1[4 * i + e] = v[e];
                }
                tile_compute<MIR>(c0v, c1v, R1, la + ((p - 1) & 1) * kGridAFloats, lane, dq);
                float* qd = qx + (p & 1) * kGridQxFloats + n;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int g = (r & 3) + 8 * (r >> 2) + 4 * half;
                    qd[g * kGridQxPitch] = c0v[r];
                    qd[g * kGridQxPitch + 32] = c1v[r];
                }
            }
            if (p + 1 < np && mine(p + 1)) {
                // panel p+1 as its owner left it at the end of the previous phase (a_0 .. a_{p-2} applied)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int g = (r & 3) + 8 * (r >> 2) + 4 * half;
                    c0v[r] = cq[g * kPanel + n];
                    c1v[r] = cq[g * kPanel + 32 + n];
                }
            }
            __syncthreads();                                        // mid
            if (team && p > 0 && mine(p - 1)) publish_a(p - 1);     // (first: the other members' tiles wait for it)
            if (p > 0 && mine(p - 1)) flush_outputs(p - 1);
            if (p + 1 < np && mine(p + 1)) stage_inputs(p + 1, false);
            if (p + 1 < np && mine(p + 1)) {
                // one tile of LD rows at a time (this wave has the whole phase for two memory round trips; both tiles in
                // registers at once, 128 VGPRs, is what pushed loop-invariant addresses of every role into scratch)
                if (p > 0) {
                    float R0[kPanel];
                    tile_load_rows<U>(R0, base, stride, b, p - 1, p + 1, lane);
                    tile_compute<MIR>(c0v, c1v, R0, la + ((p - 1) & 1) * kGridAFloats, lane, dq);
                    asm volatile("" : "+v"(c0v), "+v"(c1v) :: "memory");
                }
                float R1[kPanel];
                tile_load_rows<U>(R1, base, stride, b, p, p + 1, lane);
#pragma unroll
                for (int i = 0; i < 16; ++i) cy[i * 64 + lane] = f32x4{R1[4 * i], R1[4 * i + 1], R1[4 * i + 2], R1[4 * i + 3]};
            }
            __syncthreads();                                        // end
        }
    } else {
        // ================================================= owners ========================================================
        const int ow = wave < kGridCarryWave ? wave - 1 : wave - 2;     // waves 1, 2, 3, 5, 6, 7
        f32x16 accA[4], accB[4];                                    // tiles T = T_lo + ow and T = T_lo + ow + kGridResOwners
        const int TA = T_lo + ow, TB = T_lo + ow + kGridResOwners;
        const int cA = TA * 2 * kPanel, cB = TB * 2 * kPanel;
        const bool hasA = cA < b, hasB = cB < b;
        auto load_tile = [&](f32x16 (&acc)[4], int c0, bool has) {
            if (!has) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
                return;
            }
            const bool lane_ok = (n < 16) || (c0 + kPanel < b);
            if (c0 + 2 * kPanel <= b) wtile_load_acc<true>(acc, A, s_act, s0, b, c0, lane_ok, n_models, lane);
            else wtile_load_acc<false>(acc, A, s_act, s0, b, c0, lane_ok, n_models, lane);
        };
        load_tile(accA, cA, hasA);
        load_tile(accB, cB, hasB);
        // the 64 columns of panel `e` out of the tile that holds them -> cq (for the carry); `e` belongs to this wave
        auto extract = [&](const f32x16 (&acc)[4], int e) {
            if ((n >= 16) == ((e & 1) != 0)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int g = (r & 3) + 8 * (r >> 2) + 4 * half;
                    *reinterpret_cast<f32x4*>(cq + g * kPanel + 4 * (n & 15)) = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                }
            }
        };
        // ... and back from the chain (qx of the panel's phase): the final value of the panel's own sweep
        auto take_back = [&](f32x16 (&acc)[4], int e) {
            if ((n >= 16) == ((e & 1) != 0)) {
                const float* qs = qx + (e & 1) * kGridQxFloats;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int g = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(qs + g * kGridQxPitch + 4 * (n & 15));
                    const bool okg = g < n_models;
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j][r] = okg ? v[j] : 0.0f;
                }
            }
        };
        // mirrored form: the 64 columns of panel `e` start again at 0 -- from here on they collect the panel's second-pass sums
        auto restart = [&](f32x16 (&acc)[4], int e) {
            if ((n >= 16) == ((e & 1) != 0)) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
            }
        };
        auto tile_of = [&](int panel) { return panel >> 1; };
        if (np > 1) {
            if (tile_of(1) == TA) extract(accA, 1);                 // (tile 0 is wave 2's first tile)
        }
        if (mine(0)) stage_inputs(0, true);
        __syncthreads();                                            // B0
        for (int p = 0; p <= p_end; ++p) {
            __syncthreads();                                        // mid
            if (p > 0 && mine(p - 1)) flush_outputs(p - 1);
            if (p + 1 < np && mine(p + 1)) stage_inputs(p + 1, false);
            if (p > 0) {
                const int pp = p - 1;
                if (SYM) {
                    if (tile_of(pp) == TA) take_back(accA, pp);
                    else if (tile_of(pp) == TB) take_back(accB, pp);
                }
                if (MIR) {
                    // (the panel left for the chain two phases ago -- panel 0 / 1: from the state / before the loop; nothing has
                    //  been applied to its columns since)
                    if (tile_of(pp) == TA) restart(accA, pp);
                    else if (tile_of(pp) == TB) restart(accB, pp);
                }
                const float* a_lds = la + (pp & 1) * kGridAFloats;
                auto panel_on = [&](int c) {
                    if (c >= np) return false;
                    // (mirrored form: the panel's own columns too -- below the diagonal of its tile)
                    return c < p - 1 || c > p + 1 || (MIR && c == p - 1);
                };
                auto apply = [&](f32x16 (&acc)[4], int T, int c0) {
                    const bool l = panel_on(2 * T), r = panel_on(2 * T + 1);
                    if (!(l || r)) return;
                    if constexpr (MIR) {
                        // a tile is never on both sides of the chain at once: left of it (or on it) -- sums, eta_diff as it is;
                        // right of it -- trailing updates, dq * eta_diff
                        const int tri = 2 * T == pp ? 0 : (2 * T + 1 == pp ? 1 : -1);
                        res_tile_apply<U, true>(acc, base, stride, pp, c0, lane, l, r, a_lds, 2 * T <= pp ? 1.0f : dq, tri);
                    } else {
                        res_tile_apply<U>(acc, base, stride, pp, c0, lane, l, r, a_lds);
                    }
                };
                if (hasA) apply(accA, TA, cA);
                if (hasB) apply(accB, TB, cB);
            }
            if (p + 2 < np) {
                const int e = p + 2;                                // (a_0 .. a_{p-1} applied) -> the carry's next panel
                if (tile_of(e) == TA) extract(accA, e);
                else if (tile_of(e) == TB) extract(accB, e);
            }
            __syncthreads();                                        // end
        }
        if (MIR) {
            // q = q_chain + dq * sum (e_step.hpp:300): q_chain is what the chain's flush wrote for the panel (this workgroup,
            // phases ago, barriers in between)
            auto add_tile = [&](const f32x16 (&acc)[4], int c0) {
                const bool lane_ok = (n < 16) || (c0 + kPanel < b);
                if (c0 + 2 * kPanel <= b) wtile_add_to_q<true>(acc, A, s_act, s0, b, c0, lane_ok, n_models, lane, dq);
                else wtile_add_to_q<false>(acc, A, s_act, s0, b, c0, lane_ok, n_models, lane, dq);
            };
            if (hasA) add_tile(accA, cA);
            if (hasB) add_tile(accB, cB);
        }
        if (SYM) {
            auto store_tile = [&](const f32x16 (&acc)[4], int c0) {
                const bool lane_ok = (n < 16) || (c0 + kPanel < b);
                if (c0 + 2 * kPanel <= b) wtile_store_acc<true>(acc, A, s_act, s0, b, c0, lane_ok, n_models, lane);
                else wtile_store_acc<false>(acc, A, s_act, s0, b, c0, lane_ok, n_models, lane);
            };
            if (hasA) store_tile(accA, cA);
            if (hasB) store_tile(accB, cB);
        }
    }
}

// Team blocks of a launch (host: launch_grid.inc): the first `n_wgs` workgroups are team members -- workgroup w is member
// `member[w]` of the team of block `block[w]` (`n_blocks` of the largest blocks of the size-sorted list, all of them beyond
// the resident form) -- and join the queue of the remaining blocks (from `queue_start`) when their team block is done.
struct GridTeams {
    int32_t n_wgs = 0, n_blocks = 0;
    int32_t queue_start = 0;                  // first block of the size-sorted list the queue hands out (= all team blocks of the plan)
    const int32_t* block = nullptr;           // [n_wgs]
    const int32_t* member = nullptr;          // [n_wgs]
    const int32_t* size = nullptr;            // [n_wgs] team size
    const int64_t* goff = nullptr;            // [n_blocks] granule offset of the block's a-vectors
    unsigned long long* gran = nullptr;
    uint32_t tag_base = 0;
};

// Every block of a launch takes the resident form or a team (the host deals the blocks beyond kGridResMaxCols to teams, in
// several launches when they do not fit the chip at once: launch_grid.inc).  SYM: the symmetric form; otherwise the
// upper-triangular arithmetic over the MIRRORED storage (abi_plan.hip: mirror_lower_kernel).
template <typename U, bool SYM, bool EXACT>
__global__ __launch_bounds__(64 * kGridWaves) void estep_grid_mfma_kernel(EStepArgs<float> A, GridTeams teams) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* io = smem;                                   // [2][4][32][65]: mm, ulog, hvt, eta -> mu, gamma, d, eta'
    float* la = smem + 2 * kGridIoFloats;               // [2][64][32] scaled eta_diff of a panel
    float* dg = la + 2 * kGridAFloats;                  // [2][64][64] diagonal tile of a panel, fp32
    f32x4* cy = reinterpret_cast<f32x4*>(dg + 2 * kGridDiagFloats);   // [24][64] x 16 B: the carry wave's tile
    float* qx = dg + 2 * kGridDiagFloats + kGridCarryFloats;          // [2][32][68] q of a panel: to / from the chain
    __shared__ int s_blk;
    __shared__ int s_act[kGridModels];                  // column of the (m, G) arrays for each model slot
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int n_models = A.n_active;
    const float dq = A.dq;
    ExpTab tab;
    tab.init();
    if (tid < kGridModels) s_act[tid] = A.active[min(tid, n_models - 1)];
    __syncthreads();

    if ((int)blockIdx.x < teams.n_wgs) {
        const int tb = teams.block[blockIdx.x];
        GridTeam tm;
        tm.member = teams.member[blockIdx.x];
        tm.size = teams.size[blockIdx.x];
        tm.gran = teams.gran + teams.goff[tb];
        tm.tag_base = teams.tag_base;
        grid_block_resident<U, SYM, EXACT>(A, io, la, dg, cy, qx, s_act, A.blocks[tb], wave, lane, n_models, dq, tab, tm);
        __syncthreads();
    }
    for (;;) {
        if (tid == 0) s_blk = teams.queue_start + atomicAdd(A.counter, 1);    // (the blocks beyond the resident form are the head of the list)
        __syncthreads();
        const int blk = s_blk;
        __syncthreads();
        if (blk >= A.n_blocks) break;
        const BlockDesc bd = A.blocks[blk];
        if (bd.size > kGridResMaxCols) {                 // (never: the host gives such blocks a team)
            // every thread stores (UNIFORM control flow: a divergent `if (tid == 0)` + `continue` here makes the compiler treat
            // the block descriptor's fields as divergent in the whole loop body -- address arithmetic on the VALU instead of
            // the SALU, +13 % VALU instructions, +17 % on the cfg3 sweep; EXPERIMENTS.md round 6)
            __hip_atomic_store(A.error, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        // q of the whole block fits the updater waves' accumulator registers
        grid_block_resident<U, SYM, EXACT>(A, io, la, dg, cy, qx, s_act, bd, wave, lane, n_models, dq, tab);
    }
}

}  // namespace viprs
