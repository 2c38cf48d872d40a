// C ABI, part 3 (include/viprs_hip.h): the E-step entry points and the sweep schedule -- which kernel family
// serves which blocks of a plan, on which streams -- plus the per-sweep timing hooks.
#include "internal.h"

using namespace viprs;

namespace {

// zeroes the work-queue heads, the skip counter and the team hand-off granules in ONE launch
__global__ void sweep_prologue_kernel(int32_t* counters, int n_counters, unsigned long long* skipped,
                                      unsigned long long* granules, int64_t n_granules) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_counters) counters[i] = 0;
    if (i == 0) { skipped[0] = 0ull; skipped[1] = 0ull; }
    for (int64_t k = i; k < n_granules; k += (int64_t)gridDim.x * blockDim.x) granules[k] = 0ull;
}

// A sweep that is ONE panel launch has no prologue: its last workgroup moves the running skip count [0] to [1].  If the
// sweep before it on this plan ran kernels that count in place (float64 state, batched grid, generic / band kernels: they
// leave their total in [0]), that leftover must not be folded into this sweep's count: one 8-byte memset, only then.
hipError_t clear_stale_skip_count(viprs_plan* P) {
    if (!P->skip_count_in_place) return hipSuccess;
    P->skip_count_in_place = false;
    return hipMemsetAsync(P->d_skipped.p, 0, sizeof(unsigned long long), P->stream);
}

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
    A.ld_zero_off = P->dense_elems;          // (abi_plan.hip: zeroed slack of >= 4 * kStrip elements behind the last block)
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

int launch_tile_f64_u(viprs_plan* P, const EStepArgs<double>& A, int model, bool dense) {
    if (getenv("VIPRS_F64_ROW_BY_ROW")) return launch_generic_u<double>(P, A, model, dense);      // experiments / A-B
    switch (P->ld_dtype) {
        case VIPRS_LD_I8: return launch_tile_f64<int8_t>(P, A, model, dense);
        case VIPRS_LD_I16: return launch_tile_f64<int16_t>(P, A, model, dense);
        case VIPRS_LD_I32: return launch_tile_f64<int32_t>(P, A, model, dense);
        case VIPRS_LD_I64: return launch_tile_f64<int64_t>(P, A, model, dense);
        case VIPRS_LD_F32: return launch_tile_f64<float>(P, A, model, dense);
        case VIPRS_LD_F64: return launch_tile_f64<double>(P, A, model, dense);
        default: return fail(VIPRS_EINVAL, "bad LD dtype");
    }
}

int launch_panel_u(viprs_plan* P, const EStepArgs<float>& A, int model) {
    switch (P->ld_dtype) {
        case VIPRS_LD_F32: return launch_panel<float>(P, A, model);
        case VIPRS_LD_I8: return launch_panel<int8_t>(P, A, model);
        case VIPRS_LD_I16: return launch_panel<int16_t>(P, A, model);
        default: return fail(VIPRS_EINVAL, "dense schedule with unsupported LD dtype");
    }
}

bool use_band(const viprs_plan* P) {
    const char* f = getenv("VIPRS_BAND");
    if ((f && !atoi(f)) || P->ragged_h.empty()) return false;
    if (P->ld_dtype != VIPRS_LD_F32 && P->ld_dtype != VIPRS_LD_I8 && P->ld_dtype != VIPRS_LD_I16) return false;
    return band_ring_panels(P) <= kBandMaxRingPanels;
}

int launch_band_u(viprs_plan* P, const EStepArgs<float>& A, int model) {
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
    // two HIP events per sweep when the dense kernels are all there is (the common case): [2] .. [3] then also
    // stand for the whole sweep; every event record costs stream time
    const bool dense_only = S->float_dtype == VIPRS_F32 && !P->dense_h.empty() && P->ragged_h.empty();
    // (dense blocks only: the panel sweep is the only kernel of the call and keeps its own books -- generation-tagged
    // hand-off granules, queue heads reset and the skip count published by its last workgroup: no prologue launch)
    if (!dense_only) {
        const int64_t ng = P->n_granule_rows * kPanel;
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((ng + 255) / 256, 1024));
        sweep_prologue_kernel<<<grid, 256, 0, P->stream>>>(P->d_counters.p, kPlanCounters, P->d_skipped.p, P->d_granules.p, ng);
        HIP_TRY(hipGetLastError());
        P->skip_count_in_place = true;
    } else {
        HIP_TRY(clear_stale_skip_count(P));
    }
    P->math_used = 0;
    hipEvent_t* ev = P->ev.data() + 4 * (P->sweeps % viprs_plan::kRing);
    P->ev_dense_only[P->sweeps % viprs_plan::kRing] = dense_only;
    if (!dense_only) HIP_TRY(hipEventRecord(ev[0], P->stream));
    int rc = VIPRS_OK;
    if (S->float_dtype == VIPRS_F32) {
        EStepArgs<float> A = make_args<float>(S, dq);
        if (!P->dense_h.empty()) {
            P->pending_start_event = ev[2];              // recorded by the launcher, adjacent to the kernel launch
            switch (P->ld_dtype) {
                case VIPRS_LD_F32: rc = launch_panel<float>(P, A, kPanelSpikeSlab); break;
                case VIPRS_LD_I8: rc = launch_panel<int8_t>(P, A, kPanelSpikeSlab); break;
                case VIPRS_LD_I16: rc = launch_panel<int16_t>(P, A, kPanelSpikeSlab); break;
                default: rc = fail(VIPRS_EINVAL, "dense schedule with unsupported LD dtype"); break;
            }
            if (rc != VIPRS_OK) { P->pending_start_event = nullptr; return rc; }
            HIP_TRY(hipEventRecord(ev[3], P->stream));
            P->host_t1[P->sweeps % viprs_plan::kRing] = host_clock_ms();
        }
        if (use_band(P)) rc = launch_band_u(P, A, kBandSpikeSlab);
        else if (!P->ragged_h.empty()) rc = launch_generic_u<float>(P, A, kGenSpikeSlab, false);
    } else {
        // float64 state: the panel kernels specialise float; every block takes the panel-walking kernel of estep_tile.h
        EStepArgs<double> A = make_args<double>(S, dq);
        if (!P->dense_h.empty()) { P->host_t0[P->sweeps % viprs_plan::kRing] = host_clock_ms(); HIP_TRY(hipEventRecord(ev[2], P->stream)); }
        rc = launch_tile_f64_u(P, A, kGenSpikeSlab, true);
        if (rc == VIPRS_OK && !P->dense_h.empty()) { HIP_TRY(hipEventRecord(ev[3], P->stream)); P->host_t1[P->sweeps % viprs_plan::kRing] = host_clock_ms(); }
        if (rc == VIPRS_OK) rc = launch_tile_f64_u(P, A, kGenSpikeSlab, false);
    }
    if (rc != VIPRS_OK) return rc;
    if (!dense_only) HIP_TRY(hipEventRecord(ev[1], P->stream));
    P->sweeps++;
    return VIPRS_OK;
}

static int sweep_prologue(viprs_plan* P, int n_models = 1) {
    P->skip_count_in_place = true;
    const int64_t ng = P->n_granule_rows * kPanel * std::max(1, n_models);
    if (P->d_granules.n < (size_t)ng) {                      // grid launches: one granule set per active model
        HIP_TRY(hipStreamSynchronize(P->stream));
        HIP_TRY(P->d_granules.alloc((size_t)ng));
    }
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((ng + 255) / 256, 1024));
    sweep_prologue_kernel<<<grid, 256, 0, P->stream>>>(P->d_counters.p, kPlanCounters, P->d_skipped.p, P->d_granules.p, ng);
    HIP_TRY(hipGetLastError());
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
// everything else (ragged blocks, fp64 state, K > kPanelWideMaxK): the generic kernels.
int run_generic_model(viprs_state* S, double dq, int model, const int32_t* d_active, int n_active,
                      const int32_t* h_active) {
    viprs_plan* P = S->plan;
    HIP_TRY(hipSetDevice(P->device));
    // (the panel sweep keeps its own books, run_spike_slab: no prologue launch when it is the only kernel of the call)
    const bool panel_only = S->float_dtype == VIPRS_F32 && !P->dense_h.empty() && P->ragged_h.empty() &&
                            ((model == kGenMixture && S->width <= kPanelWideMaxK) || (model == kGenGrid && !use_grid_mfma(P, S->width)));
    int rc = panel_only ? VIPRS_OK : sweep_prologue(P, model == kGenGrid ? n_active : 1);
    if (rc != VIPRS_OK) return rc;
    if (panel_only) HIP_TRY(clear_stale_skip_count(P));
    P->math_used = 0;
    hipEvent_t* ev = P->ev.data() + 4 * (P->sweeps % viprs_plan::kRing);
    P->ev_dense_only[P->sweeps % viprs_plan::kRing] = true;      // [2] .. [3] bracket all kernels of the call
    {
        // the panel and batched-grid launchers record the start event themselves, adjacent to their (first) kernel launch
        const bool panel_ok0 = S->float_dtype == VIPRS_F32 && !P->dense_h.empty() && (model == kGenGrid || S->width <= kPanelWideMaxK);
        if (panel_ok0) P->pending_start_event = ev[2];
        else { P->host_t0[P->sweeps % viprs_plan::kRing] = host_clock_ms(); HIP_TRY(hipEventRecord(ev[2], P->stream)); }
    }
    if (S->float_dtype == VIPRS_F32) {
        EStepArgs<float> A = make_args<float>(S, dq);
        A.active = d_active;
        A.n_active = n_active;
        const bool panel_ok = !P->dense_h.empty() && (model == kGenGrid || S->width <= kPanelWideMaxK);
        if (panel_ok && model == kGenMixture) {
            rc = launch_panel_u(P, A, S->width <= kPanelMaxK ? kPanelMixture : kPanelMixtureWide);
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
        rc = launch_tile_f64_u(P, A, model, true);
        if (rc == VIPRS_OK) rc = launch_tile_f64_u(P, A, model, false);
    }
    if (rc != VIPRS_OK) { P->pending_start_event = nullptr; return rc; }
    { const int rc2 = record_start_event(P); if (rc2 != VIPRS_OK) return rc2; }      // (never pending here; belt and braces)
    HIP_TRY(hipEventRecord(ev[3], P->stream));
    P->host_t1[P->sweeps % viprs_plan::kRing] = host_clock_ms();
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
    return check_device_error(P);                      // a timed-out team hand-off must not return silently
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

static int sweep_ms(viprs_plan* P, int64_t sweep, int which, double* ms) {
    hipEvent_t* ev = P->ev.data() + 4 * (sweep % viprs_plan::kRing);
    float t = 0.f;
    if (which == 2) {           // host time between the record of the dominant kernel's start event and the record of its end event
        const double d = P->host_t1[sweep % viprs_plan::kRing] - P->host_t0[sweep % viprs_plan::kRing];
        *ms = d > 0.0 ? d : 0.0;
        return VIPRS_OK;
    }
    if (which == 1 || P->ev_dense_only[sweep % viprs_plan::kRing]) {
        if (which == 1 && P->dense_h.empty() && !P->ev_dense_only[sweep % viprs_plan::kRing]) { *ms = 0.0; return VIPRS_OK; }
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

int viprs_plan_last_math_modes(const viprs_plan* P, int* mask) {
    if (!P || !mask) return fail(VIPRS_EINVAL, "null argument");
    *mask = P->math_used;
    return VIPRS_OK;
}

int viprs_plan_last_skipped(viprs_plan* P, int64_t* n) {
    if (!P || !n) return fail(VIPRS_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(P->device));
    HIP_TRY(hipStreamSynchronize(P->stream));
    unsigned long long v[2] = {0, 0};       // [0]: kernels that count in place, [1]: moved there by the panel sweep's last workgroup
    HIP_TRY(hipMemcpy(v, P->d_skipped.p, sizeof(v), hipMemcpyDeviceToHost));
    *n = (int64_t)(v[0] + v[1]);
    return VIPRS_OK;
}

}  // extern "C"
