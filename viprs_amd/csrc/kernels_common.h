// Shared device-side descriptors for the E-step kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Experiment / instrumentation switches (tools/build_variant.sh, EXPERIMENTS.md).  Some of them produce WRONG results
// (timing experiments): none may reach a build by accident.  Every switch needs -DVIPRS_EXPERIMENTAL beside it, each
// translation unit records what it was built with, and viprs_build_flags() (C ABI) returns the union -- the empty
// string for the shipped library (tests/test_abi.py asserts that).
#if (defined(PANEL_TIMING_NO_HANDOFF_WAIT) || \
    defined(PANEL_TIMING_NO_SECOND_PASS) || \
    defined(PANEL_NW) || \
    defined(PANEL_STRIP_DEPTH) || \
    defined(PANEL_MIN_WAVES) || \
    defined(PANEL_TEAM_STRIP_DEPTH) || \
    defined(PANEL_TEAM_CPL_F32) || \
    defined(PANEL_CHAIN_PRIO) || \
    defined(VIPRS_GRID_MFMA_CHAIN) || \
    defined(VIPRS_GRID_PROFILE) || \
    defined(VIPRS_TILE_PROFILE) || \
    defined(VIPRS_SWEEP_TRACE) || \
    defined(VIPRS_PANEL_PROFILE)) && \
    !defined(VIPRS_EXPERIMENTAL)
#error "experiment switch defined without -DVIPRS_EXPERIMENTAL (see kernels_common.h)"
#endif
#ifdef VIPRS_EXPERIMENTAL
#define VIPRS_BF_VIPRS_EXPERIMENTAL " VIPRS_EXPERIMENTAL"
#else
#define VIPRS_BF_VIPRS_EXPERIMENTAL ""
#endif
#ifdef PANEL_TIMING_NO_HANDOFF_WAIT
#define VIPRS_BF_PANEL_TIMING_NO_HANDOFF_WAIT " PANEL_TIMING_NO_HANDOFF_WAIT"
#else
#define VIPRS_BF_PANEL_TIMING_NO_HANDOFF_WAIT ""
#endif
#ifdef PANEL_TIMING_NO_SECOND_PASS
#define VIPRS_BF_PANEL_TIMING_NO_SECOND_PASS " PANEL_TIMING_NO_SECOND_PASS"
#else
#define VIPRS_BF_PANEL_TIMING_NO_SECOND_PASS ""
#endif
#ifdef PANEL_NW
#define VIPRS_BF_PANEL_NW " PANEL_NW"
#else
#define VIPRS_BF_PANEL_NW ""
#endif
#ifdef PANEL_STRIP_DEPTH
#define VIPRS_BF_PANEL_STRIP_DEPTH " PANEL_STRIP_DEPTH"
#else
#define VIPRS_BF_PANEL_STRIP_DEPTH ""
#endif
#ifdef PANEL_MIN_WAVES
#define VIPRS_BF_PANEL_MIN_WAVES " PANEL_MIN_WAVES"
#else
#define VIPRS_BF_PANEL_MIN_WAVES ""
#endif
#ifdef PANEL_TEAM_STRIP_DEPTH
#define VIPRS_BF_PANEL_TEAM_STRIP_DEPTH " PANEL_TEAM_STRIP_DEPTH"
#else
#define VIPRS_BF_PANEL_TEAM_STRIP_DEPTH ""
#endif
#ifdef PANEL_TEAM_CPL_F32
#define VIPRS_BF_PANEL_TEAM_CPL_F32 " PANEL_TEAM_CPL_F32"
#else
#define VIPRS_BF_PANEL_TEAM_CPL_F32 ""
#endif
#ifdef PANEL_CHAIN_PRIO
#define VIPRS_BF_PANEL_CHAIN_PRIO " PANEL_CHAIN_PRIO"
#else
#define VIPRS_BF_PANEL_CHAIN_PRIO ""
#endif
#ifdef VIPRS_GRID_MFMA_CHAIN
#define VIPRS_BF_VIPRS_GRID_MFMA_CHAIN " VIPRS_GRID_MFMA_CHAIN"
#else
#define VIPRS_BF_VIPRS_GRID_MFMA_CHAIN ""
#endif
#ifdef VIPRS_GRID_PROFILE
#define VIPRS_BF_VIPRS_GRID_PROFILE " VIPRS_GRID_PROFILE"
#else
#define VIPRS_BF_VIPRS_GRID_PROFILE ""
#endif
#ifdef VIPRS_TILE_PROFILE
#define VIPRS_BF_VIPRS_TILE_PROFILE " VIPRS_TILE_PROFILE"
#else
#define VIPRS_BF_VIPRS_TILE_PROFILE ""
#endif
#ifdef VIPRS_SWEEP_TRACE
#define VIPRS_BF_VIPRS_SWEEP_TRACE " VIPRS_SWEEP_TRACE"
#else
#define VIPRS_BF_VIPRS_SWEEP_TRACE ""
#endif
#ifdef VIPRS_PANEL_PROFILE
#define VIPRS_BF_VIPRS_PANEL_PROFILE " VIPRS_PANEL_PROFILE"
#else
#define VIPRS_BF_VIPRS_PANEL_PROFILE ""
#endif
#define VIPRS_TU_BUILD_FLAGS VIPRS_BF_VIPRS_EXPERIMENTAL VIPRS_BF_PANEL_TIMING_NO_HANDOFF_WAIT VIPRS_BF_PANEL_TIMING_NO_SECOND_PASS VIPRS_BF_PANEL_NW VIPRS_BF_PANEL_STRIP_DEPTH VIPRS_BF_PANEL_MIN_WAVES VIPRS_BF_PANEL_TEAM_STRIP_DEPTH VIPRS_BF_PANEL_TEAM_CPL_F32 VIPRS_BF_PANEL_CHAIN_PRIO VIPRS_BF_VIPRS_GRID_MFMA_CHAIN VIPRS_BF_VIPRS_GRID_PROFILE VIPRS_BF_VIPRS_TILE_PROFILE VIPRS_BF_VIPRS_SWEEP_TRACE VIPRS_BF_VIPRS_PANEL_PROFILE

namespace viprs {
// abi_plan.hip: the registry behind viprs_build_flags(); every translation unit with kernels registers its own string
void register_build_flags(const char* flags);
struct BuildFlagsRegistrar { explicit BuildFlagsRegistrar(const char* f) { register_build_flags(f); } };
}  // namespace viprs

namespace viprs {

constexpr int kPanel = 64;          // SNPs per panel = lanes per wavefront on gfx950
constexpr int kStrip = 256;         // padding unit of the per-block q arrays (one fp32 updater strip: 64 lanes x 4 columns)

// ---- sizes the host-side schedule shares with the kernel headers ------------------------------------
// panel kernels (estep_panel.h): LDS carve (floats) q[qcap] | a[2][64] | T[2][64*64]
// (+ To[64*64], the off-diagonal tile of the chain's next phase, for the lane-per-SNP models: `offdiag_tile`)
__host__ __device__ constexpr int panel_lds_floats(int qcap, bool offdiag_tile = false) {
    return qcap + 2 * kPanel + 2 * kPanel * kPanel + (offdiag_tile ? kPanel * kPanel : 0);
}
// upper-triangular form over mirrored storage (kFormMirror): eta_diff of the last two panels and the sums s[qcap]
__host__ __device__ constexpr int panel_mirror_lds_floats(int qcap) { return 2 * kPanel + qcap; }
constexpr int kPanelMaxK = 8;       // mixture components the lane-parallel panel chain handles with its inputs staged in LDS
constexpr int kPanelWideMaxK = 31;  // ... and with scalar chains over v_readlane values (MixtureWideModel)
// LDS of the lane-parallel mixture chain: mu_mult | sqrt_half_var_tau | u_logs | var_mu | var_gamma, [64 SNPs][K]
constexpr int kMixLdsFloats = 5 * kPanel * kPanelMaxK;
constexpr int kMaxMixtureK = 64;    // generic mixture kernel (estep_generic.h): one lane per component
constexpr int kGridModels = 32;     // batched grid kernel (estep_grid_mfma.h): models per launch (one 32-row MFMA tile)
// ... its resident form: 6 owner waves x 2 tiles of 128 columns in accumulator registers = blocks of up to 1 536 SNPs
constexpr int kGridResOwners = 6, kGridResSlots = 2;
constexpr int kGridResMaxCols = kGridResOwners * kGridResSlots * 2 * kPanel;
constexpr int kBandMaxRingPanels = 256;   // band kernel (estep_band.h): 64 KB of q in the LDS ring

// One LD block as the device sees it.
struct BlockDesc {
    int32_t start;      // first SNP (index into the per-SNP vectors)
    int32_t size;       // SNPs in the block
    int32_t stride;     // row stride (elements) of the repacked dense block (multiple of 64)
    int32_t kind;       // viprs_block_kind
    int64_t ld_off;     // element offset of the repacked block inside the dense LD buffer
    int64_t gr_off;     // team kernels: first granule row (one row of 64 granules per panel) of this block
    int32_t band_left;  // windowed (ragged) components, estep_band.h: panels a row reaches to the left ...
    int32_t band_right; //   ... and to the right of its own panel (>= 1)
};

// Per-call argument pack for the spike-and-slab kernels (T = state float type).
template <typename T>
struct EStepArgs {
    const BlockDesc* blocks;     // schedule order (descending cost)
    int32_t n_blocks;
    int32_t* counter;            // work-queue head (zeroed before every launch)
    // Two-ended queue of the small-block class (bottom_mod > 0): counter[0] counts claims, counter[1] / counter[2] the
    // blocks taken from the large / the small end of the size-sorted list.  Workgroups with blockIdx % bottom_mod ==
    // bottom_mod - 1 take from the small end: at any time the chip then works on a MIX of bandwidth-bound (large)
    // and chain-bound (small) blocks instead of all workgroups moving from large to small blocks in lockstep.
    int32_t bottom_mod;
    unsigned long long* granules; // team kernels: {tag, value} hand-off granules; tag = tag_base + panel + 1
    uint32_t tag_base;           //   (tag_base = launch generation << 12: a granule of an earlier launch never matches,
                                 //    so the buffer needs no zeroing between launches)
    int32_t* arrive;             // team kernels: one arrival counter per (block, model) item -- the member that arrives
                                 //   LAST copies the item's staged eta / q into place and resets the counter
    int32_t* error;              // set to non-zero when a bounded spin gives up
    int32_t n_teams;             // team kernels: number of teams in the launch
    int32_t team_size;           // team kernels: workgroups per team
    int64_t granule_rows;        // team kernels: granule rows per model (grid: one set per active model)
    unsigned long long* skipped; // skip-branch counter (e_step.hpp:410-413)
    // row addressing for the generic kernels: row j holds rowlen[j] elements starting at element
    // rowstart[j] of ld_rows, covering columns lb[j] .. lb[j] + rowlen[j] - 1.  (Either the caller's
    // own concatenated layout or the repacked dense blocks, see abi_plan.hip.)
    const int32_t* lb;
    const int64_t* rowstart;
    const int32_t* rowlen;
    const void* ld_rows;
    // repacked dense blocks (panel kernels)
    const void* ld_dense;
    int64_t ld_zero_off;  // element offset (from ld_dense) of >= 64 zero elements, 16-byte aligned: the slack behind the last block
    // per-SNP vectors
    const T* std_beta;
    const T* u_logs;
    const T* shvt;        // sqrt_half_var_tau (spike-and-slab, mixture) or half_var_tau (grid)
    const T* mu_mult;
    const T* log_null_pi; // mixture only
    T* var_gamma;
    T* var_mu;
    T* eta;
    T* q;
    T* eta_diff;
    T* eta_out;           // team kernels: eta / q are in-out, but other members of a team still read the old
    T* q_out;             //   values, so member 0 writes here and commit_team_kernel copies into place
    T dq;
    int32_t low_memory;
    int32_t width;        // K (mixture) / G (grid); 1 otherwise
    int64_t m;            // SNPs in the plan (column stride of grid matrices)
    const int32_t* active; // grid: active model indices
    int32_t n_active;
};

// Grid launches run (block, model) work items: the same kernels, with every (m, G) column-major array
// offset to the model's column.  n_active == 0 means a plain (m,) state (one "model", no offset).
template <typename T>
__device__ __forceinline__ EStepArgs<T> select_model(const EStepArgs<T>& A0, int model_slot) {
    EStepArgs<T> A = A0;
    if (A0.n_active > 0) {
        const int64_t off = (int64_t)A0.active[model_slot] * A0.m;
        A.var_gamma += off; A.var_mu += off; A.eta += off; A.q += off; A.eta_diff += off;
        A.u_logs += off; A.shvt += off; A.mu_mult += off;
        if (A.eta_out) { A.eta_out += off; A.q_out += off; }
    }
    return A;
}

}  // namespace viprs
