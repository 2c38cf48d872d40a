// C ABI, part 1 (include/viprs_hip.h): error reporting, device query, the LD plan -- validation, block
// discovery, upload and re-lay-out of the LD data ("load LD to memory", VIPRS.__init__, VIPRS.py:151-172).
#include "internal.h"
#include <chrono>

using namespace viprs;

namespace {
thread_local std::string g_err;
viprs::BuildFlagsRegistrar tu_build_flags_(VIPRS_TU_BUILD_FLAGS);
}

namespace viprs {
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
SchedConfig& sched_config() {
    static SchedConfig c;
    return c;
}
// experiment switches the translation units of this library were compiled with (kernels_common.h); "" for a clean build
static std::string& build_flags_registry() {
    static std::string s;
    return s;
}
void register_build_flags(const char* flags) {
    // `flags` = " NAME NAME ...": keep the union
    std::string& all = build_flags_registry();
    const std::string f(flags ? flags : "");
    size_t i = 0;
    while (i < f.size()) {
        while (i < f.size() && f[i] == ' ') ++i;
        size_t j = i;
        while (j < f.size() && f[j] != ' ') ++j;
        if (j > i) {
            const std::string name = f.substr(i, j - i);
            if ((" " + all + " ").find(" " + name + " ") == std::string::npos) all += (all.empty() ? "" : " ") + name;
        }
        i = j;
    }
}
}  // namespace viprs

namespace {

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

// Upper-triangular form: the lower triangle of every dense block (a padded square whose upper triangle the repack filled)
// becomes the mirror image of the upper one (fill = 1: what the panel kernels sweep, kFormMirror in estep_panel.h) or
// zeros again (fill = 0: what the batched grid kernel and the float64 kernels rely on).  One 64 x 64 tile pair per
// workgroup step, transposed through LDS; the diagonal stays 0.  Pure data movement.
template <typename U>
__global__ __launch_bounds__(256) void mirror_lower_kernel(U* __restrict__ dense, const BlockDesc* __restrict__ blocks, int fill) {
    __shared__ U t[kPanel][kPanel + 1];
    const BlockDesc bd = blocks[blockIdx.y];
    const int b = bd.size, np = (b + kPanel - 1) / kPanel;
    U* __restrict__ base = dense + bd.ld_off;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 64 columns x 4 rows per step
    // tile pairs (I <= J) of the block, I the row panel of the SOURCE tile
    const int n_pairs = np * (np + 1) / 2;
    for (int item = blockIdx.x; item < n_pairs; item += gridDim.x) {
        int I = 0, rem = item;
        while (rem >= np - I) { rem -= np - I; ++I; }
        const int J = I + rem;
        if (fill) {
            for (int r = ty; r < kPanel; r += 4) {
                const int row = I * kPanel + r, col = J * kPanel + tx;
                t[r][tx] = (row < b && col < b && col > row) ? base[(int64_t)row * bd.stride + col] : (U)0;
            }
        }
        __syncthreads();
        for (int r = ty; r < kPanel; r += 4) {
            const int row = J * kPanel + r, col = I * kPanel + tx;          // target (row > col only)
            if (row < b && col < row) base[(int64_t)row * bd.stride + col] = fill ? t[tx][r] : (U)0;
        }
        __syncthreads();
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

}  // namespace

namespace viprs { void team_launch_forget(const viprs_plan* P); }
viprs_plan::~viprs_plan() {
    viprs::team_launch_forget(this);           // (a later plan may get this address or this stream handle)
    delete scratch;
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (side_stream) (void)hipStreamDestroy(side_stream);
    if (stream) (void)hipStreamDestroy(stream);
}

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
                            const void* ld_data, int ld_dtype, int low_memory, int device, const ExpandSource* ex,
                            const std::function<int(void*)>* fill = nullptr) {
    const size_t es = ld_elem_size(ld_dtype);
    std::unique_ptr<viprs_plan> P(new viprs_plan());
    P->m = m;
    // environment switches (docs/EXPERIMENTS_r1-r3.md, "4.3 Environment switches"), re-read at every plan creation
    if (const char* f = getenv("VIPRS_GRID_MFMA")) P->grid_mfma = atoi(f);
    {
        SchedConfig c;                          // defaults
        // (an empty value means "not set": `VAR= command` from a shell loop must not turn into a team size of 1 or a class
        //  limit of 0)
        auto env = [](const char* name) -> const char* { const char* f = getenv(name); return (f && *f) ? f : nullptr; };
        if (const char* f = env("VIPRS_LARGE_BLOCK")) c.large_block = atoi(f);
        if (const char* f = env("VIPRS_MEDIUM_BLOCK")) c.medium_block = atoi(f);
        if (const char* f = env("VIPRS_TEAM0")) { c.class_team[0] = std::max(1, atoi(f)); c.team_env = true; }
        if (const char* f = env("VIPRS_TEAM1")) { c.class_team[1] = std::max(1, atoi(f)); c.team_env = true; }
        if (const char* f = getenv("VIPRS_BOTTOM_MOD")) c.bottom_mod = std::max(0, atoi(f));
        sched_config() = c;
    }
    P->low_memory = low_memory != 0;
    P->ld_dtype = ld_dtype;
    P->device = device;
    std::string err;
    int rc = plan_blocks(m, lb, ip64.data(), low_memory != 0, P->blocks, err);
    if (rc != VIPRS_OK) return fail(rc, err);
    P->nnz = m > 0 ? ip64[(size_t)m] : 0;
    if (P->nnz > 0 && !ld_data && !ex && !fill) return fail(VIPRS_EINVAL, "ld_data is null");

    HIP_TRY(hipSetDevice(device));
    // hipGetLastError() behind the launches below must report THOSE launches: an error left behind by an earlier, unchecked
    // call of this thread (another library's, a finalizer's) is cleared here
    (void)hipGetLastError();
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    P->n_cu = prop.multiProcessorCount;
    HIP_TRY(hipStreamCreateWithFlags(&P->stream, hipStreamNonBlocking));
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
        // the panel kernels keep q of a block (teams: of a member's strips) in LDS: a dense block beyond ~13 000 SNPs is
        // scheduled like a windowed component (band kernel if its ring fits, generic kernel otherwise).  (The limit as
        // rounds 3-5 derived it from the packed upper form's LDS carve; a team of up to 16 members fits any block below it.)
        constexpr int kMaxDenseBlock = 13184;
        static_assert(kMaxDenseBlock % kPanel == 0, "whole panels");
        static_assert(kMaxDenseBlock + kStrip + panel_lds_floats(kStrip, true) + kMixLdsFloats <= 160 * 1024 / 4, "the symmetric form fits one workgroup");
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
        while (i < n && P->dense_h[i].size >= sched_config().large_block) ++i;
        P->class_begin[1] = i;
        while (i < n && P->dense_h[i].size >= sched_config().medium_block) ++i;
        P->class_begin[2] = i;
        P->class_begin[3] = n;
        // Team size of the largest class: a member's updater waves pull ~16 GB/s under load, so the share of a block's
        // row panel per member has to stay near 100 KB for the phase to remain chain-bound (~9 us): 4 members per 1 500
        // SNPs of the largest block, 8..16 (cfg3, 3 619 SNPs: 12 -- 0.70-0.71 ms against 0.72-0.77 with 8 in the same
        // process; one 6 000-SNP block: 16 -- 1.05 against 1.17 ms).  VIPRS_TEAM0 overrides.
        // (bytes, not SNPs: int8 LD keeps 8 -- its sweep is bound by the number of chains in flight, 12 costs it 4 %)
        if (!sched_config().team_env && P->class_begin[1] > 0) {
            P->team0 = std::min(16, std::max(8, 4 * (int)(((int64_t)P->dense_h[0].size * (int64_t)es + 5999) / 6000)));
            // ... unless the class is populous: teams beyond the resident workgroups only queue up behind each other, and
            // more, smaller teams get through the class's chains sooner (300 blocks of 2 400 SNPs: 12 -> 4 members,
            // tools/mixed_blocks_bench.py)
            const int64_t room = (int64_t)P->n_cu * 2 * 5 / 8;
            while (P->team0 > 4 && (int64_t)P->class_begin[1] * P->team0 > room) P->team0 -= 4;
        }
        // hand-off granules for the blocks served by teams (classes 0 and 1)
        int64_t rows = 0;
        for (int k = 0; k < P->class_begin[2]; ++k) {
            P->dense_h[(size_t)k].gr_off = rows;
            rows += (P->dense_h[(size_t)k].size + kPanel - 1) / kPanel;
        }
        P->n_granule_rows = rows;
    }

    // ---- upload -------------------------------------------------------------------------------
    HIP_TRY(P->d_counters.alloc(kPlanCounters));
    // (the panel sweep leaves its queue heads at 0; hipMemset is asynchronous for device memory and the plan's stream does
    // not wait for the null stream: synchronise here)
    HIP_TRY(hipMemset(P->d_counters.p, 0, kPlanCounters * sizeof(int32_t)));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(P->d_error.alloc(1));
    HIP_TRY(hipMemset(P->d_error.p, 0, sizeof(int32_t)));
    HIP_TRY(P->d_skipped.alloc(2));
    HIP_TRY(hipMemset(P->d_skipped.p, 0, 2 * sizeof(unsigned long long)));
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
        if (fill) {
            // the rows are produced on the device (synth.hip), in the caller's row-concatenated layout
            rc = (*fill)(P->d_ld_raw.p);
            if (rc != VIPRS_OK) return rc;
        } else if (!ex) {
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
    if (P->n_granule_rows > 0) {
        HIP_TRY(P->d_granules.alloc((size_t)P->n_granule_rows * kPanel));
        HIP_TRY(hipMemset(P->d_granules.p, 0, sizeof(unsigned long long) * P->d_granules.n));    // tag 0 = no launch
        HIP_TRY(hipDeviceSynchronize());
    }
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
    }
    if (!P->ragged_h.empty()) {
        HIP_TRY(P->d_ragged.alloc(P->ragged_h.size()));
        HIP_TRY(hipMemcpy(P->d_ragged.p, P->ragged_h.data(), sizeof(BlockDesc) * P->ragged_h.size(), hipMemcpyHostToDevice));
    } else {
        // raw copy no longer needed: every block was repacked
        HIP_TRY(P->d_ld_raw.alloc(0));
    }
    // the full lists, kept beside the ones a sweep visits (viprs_plan_set_active_blocks)
    P->dense_all_h = P->dense_h;
    P->ragged_all_h = P->ragged_h;
    std::copy(P->class_begin, P->class_begin + 4, P->class_begin_all);
    P->max_dense_all = P->max_dense;
    P->max_ragged_all = P->max_ragged;
    P->m_active = m;
    if (!P->dense_h.empty()) {
        HIP_TRY(P->d_dense_all.alloc(P->dense_h.size()));
        HIP_TRY(hipMemcpy(P->d_dense_all.p, P->dense_h.data(), sizeof(BlockDesc) * P->dense_h.size(), hipMemcpyHostToDevice));
    }
    // every copy / memset above went through the null stream; the plan's own streams are non-blocking
    // (not ordered with it), so nothing may still be in flight when the first sweep is launched
    HIP_TRY(hipDeviceSynchronize());
    *out = P.release();
    return VIPRS_OK;
}

namespace viprs {
// The dense blocks of an upper-triangular plan in the form the coming launch sweeps (see mirror_lower_kernel); the
// conversion runs on the plan's stream, once per change of kernel family (a fit keeps to one).
int ensure_upper_storage(viprs_plan* P, bool mirrored) {
    // (every dense block of the plan, whatever subset the sweeps currently visit: the flag is per plan)
    if (!P->low_memory || P->dense_all_h.empty() || (P->mirror != 0) == mirrored) return VIPRS_OK;
    int max_np = 0;
    for (const BlockDesc& d : P->dense_all_h) max_np = std::max(max_np, (d.size + kPanel - 1) / kPanel);
    const dim3 grid((unsigned)std::min(max_np * (max_np + 1) / 2, 512), (unsigned)P->dense_all_h.size());
    switch (P->ld_dtype) {
        case VIPRS_LD_F32: mirror_lower_kernel<float><<<grid, 256, 0, P->stream>>>((float*)P->d_ld_dense.p, P->d_dense_all.p, mirrored ? 1 : 0); break;
        case VIPRS_LD_I8: mirror_lower_kernel<int8_t><<<grid, 256, 0, P->stream>>>((int8_t*)P->d_ld_dense.p, P->d_dense_all.p, mirrored ? 1 : 0); break;
        case VIPRS_LD_I16: mirror_lower_kernel<int16_t><<<grid, 256, 0, P->stream>>>((int16_t*)P->d_ld_dense.p, P->d_dense_all.p, mirrored ? 1 : 0); break;
        default: return fail(VIPRS_EINVAL, "dense blocks with unsupported LD dtype");
    }
    HIP_TRY(hipGetLastError());
    P->mirror = mirrored ? 1 : 0;
    return VIPRS_OK;
}

namespace {
std::mutex g_gate_mutex;
std::map<int, hipEvent_t> g_gate_event;        // per device: completion of the last launch with co-resident teams
std::map<int, const viprs_plan*> g_gate_plan;  // ... and the plan whose stream it went to (the same plan again: already
                                               //     ordered by its own stream; forgotten when that plan is destroyed)
}  // namespace
double host_clock_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
int record_start_event(viprs_plan* P) {
    if (P->pending_start_event) {
        P->host_t0[P->sweeps % viprs_plan::kRing] = host_clock_ms();
        HIP_TRY(hipEventRecord(P->pending_start_event, P->stream));
        P->pending_start_event = nullptr;
    }
    return VIPRS_OK;
}
int team_launch_gate(viprs_plan* P) {
    std::lock_guard<std::mutex> lock(g_gate_mutex);
    auto it = g_gate_event.find(P->device);
    auto last = g_gate_plan.find(P->device);
    if (it != g_gate_event.end() && (last == g_gate_plan.end() || last->second != P))
        HIP_TRY(hipStreamWaitEvent(P->stream, it->second, 0));
    return VIPRS_OK;
}
void team_launch_forget(const viprs_plan* P) {
    std::lock_guard<std::mutex> lock(g_gate_mutex);
    auto last = g_gate_plan.find(P->device);
    if (last != g_gate_plan.end() && last->second == P) g_gate_plan.erase(last);
}
int team_launch_done(viprs_plan* P) {
    std::lock_guard<std::mutex> lock(g_gate_mutex);
    auto it = g_gate_event.find(P->device);
    if (it == g_gate_event.end()) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        it = g_gate_event.emplace(P->device, e).first;
    }
    HIP_TRY(hipEventRecord(it->second, P->stream));
    g_gate_plan[P->device] = P;
    return VIPRS_OK;
}
int plan_create_generated(viprs_plan** out, int64_t m, const int32_t* lb, const std::vector<int64_t>& ip64, int ld_dtype,
                          int low_memory, int device, const std::function<int(void*)>& fill) {
    return plan_create_impl(out, m, lb, ip64, nullptr, ld_dtype, low_memory, device, nullptr, &fill);
}
}  // namespace viprs

extern "C" {

const char* viprs_last_error(void) { return g_err.c_str(); }
const char* viprs_version(void) { return "viprs_amd 0.1.0 (gfx950)"; }
const char* viprs_build_flags(void) { return build_flags_registry().c_str(); }

int viprs_device_count(int* count) {
    if (!count) return fail(VIPRS_EINVAL, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(VIPRS_EDEVICE, hipGetErrorString(e)); }
    *count = n;
    return VIPRS_OK;
}

// Content fingerprint of a host array (viprs_amd/vi/e_step_hip.py::plan_for: has the caller edited its LD arrays in place since
// they were uploaded?): the length, the first and last 4 KB and 256 evenly spaced 64-byte windows, mixed 8 bytes at a time --
// a few microseconds whatever the array size.  Pure host code.
int viprs_host_fingerprint(const void* data, int64_t n_bytes, uint64_t* out) {
    if (!out || n_bytes < 0 || (n_bytes > 0 && !data)) return fail(VIPRS_EINVAL, "bad argument");
    const unsigned char* b = static_cast<const unsigned char*>(data);
    uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t)n_bytes;
    auto mix = [&](const unsigned char* p, int64_t n) {
        int64_t i = 0;
        for (; i + 8 <= n; i += 8) {
            uint64_t w;
            std::memcpy(&w, p + i, 8);
            h = (h ^ w) * 0xff51afd7ed558ccdull;
            h ^= h >> 32;
        }
        for (; i < n; ++i) { h = (h ^ p[i]) * 0x100000001b3ull; }
    };
    constexpr int64_t kEdge = 4096, kSamples = 256, kWin = 64;
    if (n_bytes <= 2 * kEdge + kSamples * kWin) {
        mix(b, n_bytes);
    } else {
        mix(b, kEdge);
        mix(b + n_bytes - kEdge, kEdge);
        const int64_t step = (n_bytes - kWin) / kSamples;
        for (int64_t k = 0; k < kSamples; ++k) mix(b + k * step, kWin);
    }
    *out = h;
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

// Which LD blocks the following sweeps visit: `active` = one byte per block in SNP order (viprs_plan_get_blocks), NULL =
// every block again.  The size-sorted block lists the kernels work through are rebuilt from the full ones (a few thousand
// descriptors: microseconds) and every schedule derived from them is forgotten; LD and per-SNP arrays do not move.
int viprs_plan_set_active_blocks(viprs_plan* P, const uint8_t* active, int64_t n_blocks) {
    if (!P) return fail(VIPRS_EINVAL, "null plan");
    if (active && n_blocks != (int64_t)P->blocks.size()) return fail(VIPRS_EINVAL, "one flag per LD block of the plan is needed");
    if (!active && !P->filtered) return VIPRS_OK;
    HIP_TRY(hipSetDevice(P->device));
    HIP_TRY(hipStreamSynchronize(P->stream));            // sweeps in flight read the lists
    if (P->side_stream) HIP_TRY(hipStreamSynchronize(P->side_stream));
    auto is_active = [&](const BlockDesc& d) {
        if (!active) return true;
        // P->blocks is in SNP order: the block that starts at d.start
        auto it = std::lower_bound(P->blocks.begin(), P->blocks.end(), (int64_t)d.start,
                                   [](const Block& b, int64_t s) { return b.start < s; });
        return active[(size_t)(it - P->blocks.begin())] != 0;
    };
    P->dense_h.clear();
    P->ragged_h.clear();
    P->max_dense = P->max_ragged = 0;
    P->m_active = 0;
    int cb[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < P->dense_all_h.size(); ++i) {
        const BlockDesc& d = P->dense_all_h[i];
        if (!is_active(d)) continue;
        const int cls = (int)i < P->class_begin_all[1] ? 0 : ((int)i < P->class_begin_all[2] ? 1 : 2);
        for (int c = cls + 1; c < 4; ++c) ++cb[c];
        P->dense_h.push_back(d);                          // (the full list is in descending order of size: so is this one)
        P->max_dense = std::max(P->max_dense, d.size);
        P->m_active += d.size;
    }
    std::copy(cb, cb + 4, P->class_begin);
    for (const BlockDesc& d : P->ragged_all_h) {
        if (!is_active(d)) continue;
        P->ragged_h.push_back(d);
        P->max_ragged = std::max(P->max_ragged, d.size);
        P->m_active += d.size;
    }
    if (!P->dense_h.empty())
        HIP_TRY(hipMemcpy(P->d_dense.p, P->dense_h.data(), sizeof(BlockDesc) * P->dense_h.size(), hipMemcpyHostToDevice));
    if (!P->ragged_h.empty())
        HIP_TRY(hipMemcpy(P->d_ragged.p, P->ragged_h.data(), sizeof(BlockDesc) * P->ragged_h.size(), hipMemcpyHostToDevice));
    P->filtered = active != nullptr && (P->dense_h.size() != P->dense_all_h.size() || P->ragged_h.size() != P->ragged_all_h.size());
    // schedules derived from the lists: rebuilt on the next launch that needs them
    P->team_split = viprs_plan::TeamSplit();
    P->grid_teams_built = false;
    HIP_TRY(P->d_rowlist_dense.alloc(0));
    HIP_TRY(P->d_rowlist_ragged.alloc(0));
    return VIPRS_OK;
}

int viprs_plan_set_math_mode(viprs_plan* P, int mode) {
    if (!P) return fail(VIPRS_EINVAL, "null plan");
    if (mode != VIPRS_MATH_EXACT && mode != VIPRS_MATH_FAST) return fail(VIPRS_EINVAL, "bad math mode");
    P->math_mode = mode;
    return VIPRS_OK;
}

}  // extern "C"
