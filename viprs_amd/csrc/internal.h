// Internal host-side declarations shared by the translation units of libviprs_hip.so:
// plan / state objects, error reporting, launcher entry points of the kernel families.
// (The C ABI itself is include/viprs_hip.h.)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/viprs_hip.h"
#include "kernels_common.h"
#include "planner.h"

namespace viprs {

int fail(int code, const std::string& msg);      // records the thread's error message, returns `code`

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return ::viprs::fail(VIPRS_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

inline size_t ld_elem_size(int ld_dtype) {
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
inline size_t float_size(int t) { return t == VIPRS_F32 ? 4 : (t == VIPRS_F64 ? 8 : 0); }

// workgroup-size classes of the panel kernel (waves per workgroup; wave 0 is the chain)
// Every class uses 4-wave workgroups (1 chain + 3 updaters) so that a team member needs exactly
// the same CU resources as a small-block workgroup; larger blocks get more CUs, not bigger groups.
struct SchedConfig {
    int large_block = 2304, medium_block = 1600;   // VIPRS_LARGE_BLOCK / VIPRS_MEDIUM_BLOCK
    int class_team[3] = {8, 4, 1};                 // workgroups (CUs) sharing one block of the class (0/1: teams)
    // mixture: the same team sizes (rounds 1-2 ran teams of 4 / single workgroups -- every member replicates the chain,
    // whose step is ~2.5x the spike-and-slab one; re-measured at the end of round 3, cfg3: K = 4 1.273 -> 1.230 ms
    // symmetric, 1.40 -> 1.36 upper; K = 10 2.11 -> 2.08, K = 20 2.91 -> 2.88)
    int class_team_mix[3] = {8, 4, 1};
    bool team_env = false;                         // VIPRS_TEAM0/1 given: they apply to every model
    // small-block queue: every `bottom_mod`-th workgroup pulls from the SMALL end of the size-sorted queue (0 = off)
    int bottom_mod = 0;                            // VIPRS_BOTTOM_MOD
};
SchedConfig& sched_config();                       // process-wide, read from the environment at plan creation
constexpr int kClassWaves[3] = {4, 4, 4};
constexpr int kPlanCounters = 32;                  // work-queue heads of a plan (d_counters), zeroed by every sweep's prologue

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

}  // namespace viprs

struct viprs_state;
struct viprs_comm;                                  // comm.hip
struct viprs_plan {
    int64_t m = 0;
    int64_t nnz = 0;
    int low_memory = 0;
    int mirror = 0;                  // upper-triangular form: the dense blocks currently hold the upper triangle mirrored into the lower one
                                     // (what the panel and grid kernels sweep; the float64 kernels read it with a zero lower triangle)
    int ld_dtype = 0;
    int device = 0;
    int n_cu = 0;
    int math_mode = VIPRS_MATH_EXACT;
    int math_used = 0;                      // kernels of the last sweep: bit 0 = exact arithmetic ran, bit 1 = fast arithmetic ran
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;             // float64 state: the big-block class of estep_tile.h runs beside the rest
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<viprs::Block> blocks;              // SNP order
    std::vector<viprs::BlockDesc> dense_h, ragged_h;  // schedule order (descending cost)
    // dense blocks are served by panel kernels of three workgroup sizes (more updater waves =
    // more row loads in flight = a larger share of HBM bandwidth for the larger blocks); class c
    // covers dense_h[class_begin[c] .. class_begin[c+1])
    int class_begin[4] = {0, 0, 0, 0};
    int team0 = 0;                          // team size of the largest class for this plan's largest block (0: the configured one)
    viprs::DevBuf<viprs::BlockDesc> d_dense, d_ragged;
    // viprs_plan_set_active_blocks: the lists above are what a sweep visits -- all blocks, or the active subset (the
    // lock-step fit of one model per chromosome drops the blocks of the chromosomes that have converged); the full lists:
    std::vector<viprs::BlockDesc> dense_all_h, ragged_all_h;
    int class_begin_all[4] = {0, 0, 0, 0};
    int max_dense_all = 0, max_ragged_all = 0;
    viprs::DevBuf<viprs::BlockDesc> d_dense_all;   // what ensure_upper_storage converts (every block, active or not)
    bool filtered = false;
    int64_t m_active = 0;                          // SNPs of the blocks a sweep visits
    viprs::DevBuf<unsigned long long> d_granules;  // team hand-off granules (one row of 64 per panel of a team block)
    int64_t n_granule_rows = 0;
    viprs::DevBuf<int32_t> d_error;
    int grid_mfma = -1;                     // batched grid E-step on the matrix cores: 1 always, 0 never (per-(block,
                                            // model) items), -1 when the plan has enough blocks to fill the CUs (VIPRS_GRID_MFMA)
    viprs::DevBuf<int32_t> d_lb;
    viprs::DevBuf<int64_t> d_ip;
    viprs::DevBuf<int32_t> d_rowlen;               // indptr[j+1] - indptr[j]
    viprs::DevBuf<int64_t> d_rowlist_dense, d_rowlist_ragged;   // float64 second pass (estep_tile.h): (block of the list) << 32 | first row, per group of 8 rows
    viprs::DevBuf<int64_t> d_rowstart_dense;       // row starts inside the repacked dense buffer (generic kernels)
    viprs::DevBuf<char> d_ld_raw;                  // kept only when ragged blocks exist
    viprs::DevBuf<char> d_ld_dense;
    int64_t dense_elems = 0;
    int max_dense = 0, max_ragged = 0;
    int max_band_panels = 0;                // ragged blocks: widest (band_left + band_right + 2), sizes the band kernel's q ring
    viprs::DevBuf<int32_t> d_counters;             // [0] dense queue head, [1] ragged queue head
    viprs::DevBuf<unsigned long long> d_skipped;   // [0] running count of the sweep in flight, [1] of the sweep kernel(s) that are done
    bool skip_count_in_place = false;              // the last sweep ran kernels that leave their count in [0] (no panel-sweep epilogue moved it)
    viprs::DevBuf<int32_t> d_arrive;               // panel sweep: arrival counters of the team (block, model) items; [last]: workgroups done
    uint32_t granule_gen = 0;                      // panel sweep launches so far (mod 2^20): generation of the hand-off tags
    // batched grid kernel, teams for the blocks beyond its resident form (launch_grid.inc): per team workgroup (block of the
    // size-sorted list, member, team size), per team block the offset of its a-vector granules
    // the split of the team budget between the two team classes (launch_panel.inc) depends on the plan and on these only
    struct TeamSplit { int64_t key[6] = {-1, -1, -1, -1, -1, -1}; int best[2] = {0, 0}; } team_split;
    bool grid_teams_built = false;
    int grid_team_blocks = 0, grid_team_wgs = 0;
    std::vector<int> grid_team_ts;                 // team size per team block (largest block first)
    viprs::DevBuf<int32_t> d_grid_team_block, d_grid_team_member, d_grid_team_size;
    viprs::DevBuf<int64_t> d_grid_team_goff;
    viprs::DevBuf<unsigned long long> d_grid_gran;
    uint32_t grid_gen = 0;
    // HIP-event ring: per sweep {sweep start, sweep end, panel start, panel end}, recorded on
    // the stream the kernels are launched on
    static constexpr int kRing = 256;
    std::vector<hipEvent_t> ev;             // 4 * kRing
    bool ev_dense_only[kRing] = {};         // ring slot: only [2] .. [3] were recorded (they bracket the whole sweep)
    double host_t0[kRing] = {}, host_t1[kRing] = {};   // host clock (ms) at the start event's record and behind the end event's: which = 2
    int64_t sweeps = 0;                     // sweeps recorded since the last timing reset
    viprs_state* scratch = nullptr;         // state used by the one-shot host-buffer calls
    // start event of the sweep being enqueued, recorded by the panel / grid launcher IMMEDIATELY in front of its first kernel
    // launch (record_start_event): with an empty stream the event is reached at once, and whatever the host does between the
    // record and the launch -- occupancy query, team split, launch gate, a scheduler hiccup -- would count as kernel time
    hipEvent_t pending_start_event = nullptr;

    ~viprs_plan();
};

struct viprs_state {
    viprs_plan* plan = nullptr;
    int device = 0;                         // (own copy: viprs_state_destroy must not read the plan -- a garbage collector may have destroyed it first)
    int float_dtype = VIPRS_F32;
    int model_kind = VIPRS_MODEL_SPIKE_SLAB;
    int width = 1;
    viprs::DevBuf<char> f[VIPRS_FIELD_COUNT];
    viprs::DevBuf<int32_t> d_active;               // grid: active model indices of the current call
    viprs::DevBuf<char> eta_out, q_out;            // team kernels' in-out staging (see kernels_common.h)
    viprs::DevBuf<double> d_n, d_var_tau, d_partials, d_sums;   // device-resident EM iteration
    viprs::DevBuf<double> d_weight;                // optional per-SNP weight of sum [0] (several chromosomes in one plan)
    viprs::DevBuf<double> d_log_var_tau0;          // mixture: the log var_tau of the initial state (the reference's ELBO never refreshes it)
    viprs::DevBuf<double> d_colparams, d_sumcols;  // grid: per-column parameters of the batched prep / of the batched sums
    std::vector<double> col_prep;                  // grid: (one_plus_lambda, sigma_eps, tau_beta) of every column's last prep (3 x width; NaN: none yet)
    int sums_cols = 0;                      // columns of the reduction in flight (grid: sums_columns_begin; groups: sums_groups_begin)
    // viprs_state_set_groups: contiguous SNP ranges with their own hyper-parameters and their own sums (one model per chromosome)
    int n_groups = 0, group_max_nb = 0;
    std::vector<int64_t> group_start;              // n_groups + 1 entries
    viprs::DevBuf<int64_t> d_group_start;
    viprs::DevBuf<double> d_group_prep, d_group_sumrows;   // per-launch parameter rows (6, mixture: 4 + 3 K / 2 doubles per listed group)
    double* h_gparams = nullptr;                   // pinned staging of both
    size_t h_sums_cap = 0;
    double* h_sums = nullptr;               // pinned landing buffer of the device sums
    bool sums_pending = false, sums_empty = false;
    viprs_comm* comm = nullptr;             // viprs_state_set_comm: the sums are all-rank sums (one all-gather per reduction)
    hipEvent_t ev_prep = nullptr;           // the last batched prep launch (it reads d_colparams)
    double* h_params = nullptr;             // pinned staging of the batched prep (6 x width) / sums (2 x width) parameters
    ~viprs_state() {
        if (h_sums) (void)hipHostFree(h_sums);
        if (h_params) (void)hipHostFree(h_params);
        if (h_gparams) (void)hipHostFree(h_gparams);
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

namespace viprs {

// abi_plan.hip: a plan whose LD rows (the caller's row-concatenated layout, `ip64[m]` elements) are written by `fill`
// into device memory instead of being uploaded from the host (synth.hip)
int plan_create_generated(viprs_plan** out, int64_t m, const int32_t* lb, const std::vector<int64_t>& ip64, int ld_dtype,
                          int low_memory, int device, const std::function<int(void*)>& fill);

// Launches whose workgroups wait for each other (team hand-offs: the panel sweep, the batched grid kernel) size their grid to
// the workgroups the WHOLE device can hold at once.  Two such launches in flight on one device (several plans of one
// process on their own streams: unmerged chromosomes, VIPRSGrid's per-plan sweeps) would each find only part of the device
// and could leave a team member undispatched while its peers spin.  `team_launch_gate` orders them: the stream of the plan
// waits for the previous gated launch on this device, `team_launch_done` records this one.  (Other PROCESSES on the
// same device cannot be ordered from here: the bounded spins report VIPRS_EDEVICE instead of hanging.)
int ensure_upper_storage(viprs_plan* P, bool mirrored);
int record_start_event(viprs_plan* P);
double host_clock_ms();
int team_launch_gate(viprs_plan* P);
int team_launch_done(viprs_plan* P);

// after a synchronisation point: did a team hand-off give up (bounded spin)?
int check_device_error(viprs_plan* P);

// comm.hip: all-gather + rank-ordered reduction of the device vector (n doubles, in place) on `stream`; the last
// element of every `group` is a maximum, the others are sums
int comm_reduce_on_stream(viprs_comm* C, double* d_vec, int n, int group, hipStream_t stream);

// ---- kernel families (one translation unit per family and LD element type) ------------------------
enum { kGenSpikeSlab = 0, kGenMixture = 1, kGenGrid = 2 };
enum { kPanelSpikeSlab = 0, kPanelGridColumn = 1, kPanelMixture = 2, kPanelMixtureWide = 3 };
enum { kBandSpikeSlab = 0, kBandGridColumn = 1, kBandMixture = 2 };

// generic kernels over one block list: `dense` = the repacked dense blocks, otherwise the ragged blocks
template <typename T, typename U> int launch_generic(viprs_plan* P, EStepArgs<T> A, int model, bool dense);
// float64 state: the panel-walking kernels (estep_tile.h) over one block list; falls back to launch_generic for the
// mixture and for blocks whose q does not fit the LDS
template <typename U> int launch_tile_f64(viprs_plan* P, EStepArgs<double> A, int model, bool dense);
// panel kernels of the three size classes on their own streams, joined into the plan's stream
template <typename U> int launch_panel(viprs_plan* P, EStepArgs<float> A, int model);
// band kernel for the windowed components
template <typename U> int launch_band(viprs_plan* P, EStepArgs<float> A, int model);
int band_ring_panels(const viprs_plan* P);
// batched grid E-step on the matrix cores
template <typename U> int launch_grid_mfma(viprs_plan* P, EStepArgs<float> A);

}  // namespace viprs
