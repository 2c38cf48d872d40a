// Synthetic "longrange" LD blocks (viprs_amd/utils/synthetic.py::_longrange_block) generated ON THE DEVICE, straight into
// a plan: measurement support for bench.py and the tests (a rank of an N-GPU run gets its 1.1 M-SNP / 3.8 GB workload in
// milliseconds instead of 20 s of host work + a PCIe upload).  Not on the E-step path.
//
// One entry costs five float32 operations, each rounded on its own exactly as the NumPy code does them (-ffp-contract=off):
//   R[i, j] = (uf0[i] uf0[j] + uf1[i] uf1[j]) + (pw[|i - j|] sa[i]) sf[j]     (i != j),     R[i, i] = 1
// integer LD: rint(R * qmax) (float32 product, round-half-even = np.round), stored as int8 / int16.
// The host version of the same function fills the caller's array: tests/test_synth_device.py checks it == make_ld bit for bit.
#include "internal.h"

#include <cmath>
#include <functional>

using namespace viprs;

namespace {

struct SynthBlock {
    int64_t start;      // first SNP of the block (index into the per-SNP parameter vectors)
    int64_t off;        // first element of the block in the row-concatenated output
    int32_t size;
    int32_t pad;
};

__host__ __device__ __forceinline__ float synth_entry(const float* __restrict__ pw, const float* __restrict__ uf0,
                                                      const float* __restrict__ uf1, const float* __restrict__ sa,
                                                      const float* __restrict__ sf, int64_t s, int i, int j) {
    if (i == j) return 1.0f;
    const float m0 = uf0[s + i] * uf0[s + j];
    const float m1 = uf1[s + i] * uf1[s + j];
    const float mm = m0 + m1;
    const int k = i > j ? i - j : j - i;
    float t = pw[s + k] * sa[s + i];
    t = t * sf[s + j];
    return mm + t;
}

template <typename U> __host__ __device__ __forceinline__ U synth_store(float v, float qmax) {
    if constexpr (sizeof(U) == 4) {
        return v;
    } else {
#if defined(__HIP_DEVICE_COMPILE__)
        return static_cast<U>(__builtin_rintf(v * qmax));
#else
        return static_cast<U>(std::nearbyintf(v * qmax));
#endif
    }
}

// row r of a block: symmetric form b entries at off + r b; upper form b - 1 - r entries at off + r (b - 1) - r (r - 1) / 2
__host__ __device__ __forceinline__ int64_t synth_row_off(const SynthBlock& B, int r, bool upper) {
    return upper ? B.off + (int64_t)r * (B.size - 1) - (int64_t)r * (r - 1) / 2 : B.off + (int64_t)r * B.size;
}

template <typename U>
__global__ void synth_longrange_kernel(const SynthBlock* __restrict__ blocks, const float* __restrict__ pw,
                                       const float* __restrict__ uf0, const float* __restrict__ uf1,
                                       const float* __restrict__ sa, const float* __restrict__ sf, U* __restrict__ out,
                                       int upper, float qmax) {
    const SynthBlock B = blocks[blockIdx.y];
    for (int r = blockIdx.x; r < B.size; r += gridDim.x) {
        U* __restrict__ row = out + synth_row_off(B, r, upper != 0);
        const int c0 = upper ? r + 1 : 0;
        for (int c = c0 + threadIdx.x; c < B.size; c += blockDim.x)
            row[c - c0] = synth_store<U>(synth_entry(pw, uf0, uf1, sa, sf, B.start, r, c), qmax);
    }
}

template <typename U>
void synth_host(const std::vector<SynthBlock>& blocks, const float* pw, const float* uf0, const float* uf1, const float* sa,
                const float* sf, U* out, bool upper, float qmax) {
    for (const SynthBlock& B : blocks)
        for (int r = 0; r < B.size; ++r) {
            U* row = out + synth_row_off(B, r, upper);
            const int c0 = upper ? r + 1 : 0;
            for (int c = c0; c < B.size; ++c) row[c - c0] = synth_store<U>(synth_entry(pw, uf0, uf1, sa, sf, B.start, r, c), qmax);
        }
}

// sizes -> blocks, m, nnz; the index arrays of the reference's layout (e_step_cpp.pyx:91-93) when asked for
int synth_layout(int64_t n_blocks, const int64_t* sizes, bool upper, std::vector<SynthBlock>& blocks, int64_t& m, int64_t& nnz,
                 std::vector<int32_t>* lb, std::vector<int64_t>* ip) {
    if (n_blocks < 0 || (n_blocks > 0 && !sizes)) return fail(VIPRS_EINVAL, "bad block list");
    m = 0;
    nnz = 0;
    blocks.clear();
    for (int64_t b = 0; b < n_blocks; ++b) {
        if (sizes[b] < 1 || sizes[b] > (1 << 20)) return fail(VIPRS_EINVAL, "block size out of range");
        SynthBlock B{m, nnz, (int32_t)sizes[b], 0};
        blocks.push_back(B);
        m += sizes[b];
        nnz += upper ? sizes[b] * (sizes[b] - 1) / 2 : sizes[b] * sizes[b];
    }
    if (m > INT32_MAX) return fail(VIPRS_EINVAL, "m out of range");
    if (lb) {
        lb->resize((size_t)m);
        ip->assign((size_t)m + 1, 0);
        for (const SynthBlock& B : blocks)
            for (int r = 0; r < B.size; ++r) {
                const int64_t j = B.start + r;
                (*lb)[(size_t)j] = (int32_t)(upper ? j + 1 : B.start);
                (*ip)[(size_t)j + 1] = synth_row_off(B, r, upper) + (upper ? B.size - 1 - r : B.size);
            }
    }
    return VIPRS_OK;
}

float synth_qmax(int ld_dtype) { return ld_dtype == VIPRS_LD_I8 ? 127.0f : (ld_dtype == VIPRS_LD_I16 ? 32767.0f : 1.0f); }

}  // namespace

extern "C" {

int viprs_plan_create_synthetic(viprs_plan** out, int64_t n_blocks, const int64_t* sizes, const float* pw, const float* uf0,
                                const float* uf1, const float* sa, const float* sf, int ld_dtype, int low_memory, int device) {
    if (!out) return fail(VIPRS_EINVAL, "plan output is null");
    *out = nullptr;
    if (ld_dtype != VIPRS_LD_F32 && ld_dtype != VIPRS_LD_I8 && ld_dtype != VIPRS_LD_I16)
        return fail(VIPRS_EINVAL, "synthetic LD: float32, int8 or int16");
    if (n_blocks > 65535) return fail(VIPRS_EINVAL, "synthetic LD: at most 65535 blocks");
    std::vector<SynthBlock> blocks;
    std::vector<int32_t> lb;
    std::vector<int64_t> ip;
    int64_t m = 0, nnz = 0;
    int rc = synth_layout(n_blocks, sizes, low_memory != 0, blocks, m, nnz, &lb, &ip);
    if (rc != VIPRS_OK) return rc;
    if (m > 0 && (!pw || !uf0 || !uf1 || !sa || !sf)) return fail(VIPRS_EINVAL, "null parameter vector");
    const float qmax = synth_qmax(ld_dtype);
    std::function<int(void*)> fill = [&](void* d_raw) -> int {
        if (m == 0 || nnz == 0) return VIPRS_OK;
        DevBuf<SynthBlock> d_blocks;
        DevBuf<float> d_par;
        HIP_TRY(d_blocks.alloc(blocks.size()));
        HIP_TRY(d_par.alloc(5 * (size_t)m));
        HIP_TRY(hipMemcpy(d_blocks.p, blocks.data(), sizeof(SynthBlock) * blocks.size(), hipMemcpyHostToDevice));
        const float* src[5] = {pw, uf0, uf1, sa, sf};
        for (int k = 0; k < 5; ++k)
            HIP_TRY(hipMemcpy(d_par.p + (size_t)k * m, src[k], sizeof(float) * (size_t)m, hipMemcpyHostToDevice));
        const float *d0 = d_par.p, *d1 = d0 + m, *d2 = d1 + m, *d3 = d2 + m, *d4 = d3 + m;
        dim3 grid(64, (unsigned)blocks.size());
        switch (ld_dtype) {
            case VIPRS_LD_F32: synth_longrange_kernel<float><<<grid, 256>>>(d_blocks.p, d0, d1, d2, d3, d4, (float*)d_raw, low_memory, qmax); break;
            case VIPRS_LD_I8:  synth_longrange_kernel<int8_t><<<grid, 256>>>(d_blocks.p, d0, d1, d2, d3, d4, (int8_t*)d_raw, low_memory, qmax); break;
            default:           synth_longrange_kernel<int16_t><<<grid, 256>>>(d_blocks.p, d0, d1, d2, d3, d4, (int16_t*)d_raw, low_memory, qmax); break;
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        return VIPRS_OK;
    };
    return plan_create_generated(out, m, lb.data(), ip, ld_dtype, low_memory, device, fill);
}

int viprs_synthetic_ld_host(int64_t n_blocks, const int64_t* sizes, const float* pw, const float* uf0, const float* uf1,
                            const float* sa, const float* sf, int ld_dtype, int low_memory, void* out, int64_t capacity) {
    if (ld_dtype != VIPRS_LD_F32 && ld_dtype != VIPRS_LD_I8 && ld_dtype != VIPRS_LD_I16)
        return fail(VIPRS_EINVAL, "synthetic LD: float32, int8 or int16");
    std::vector<SynthBlock> blocks;
    int64_t m = 0, nnz = 0;
    int rc = synth_layout(n_blocks, sizes, low_memory != 0, blocks, m, nnz, nullptr, nullptr);
    if (rc != VIPRS_OK) return rc;
    if (capacity < nnz || (nnz > 0 && !out)) return fail(VIPRS_EINVAL, "output too small");
    if (m > 0 && (!pw || !uf0 || !uf1 || !sa || !sf)) return fail(VIPRS_EINVAL, "null parameter vector");
    const float qmax = synth_qmax(ld_dtype);
    const bool up = low_memory != 0;
    switch (ld_dtype) {
        case VIPRS_LD_F32: synth_host<float>(blocks, pw, uf0, uf1, sa, sf, (float*)out, up, qmax); break;
        case VIPRS_LD_I8:  synth_host<int8_t>(blocks, pw, uf0, uf1, sa, sf, (int8_t*)out, up, qmax); break;
        default:           synth_host<int16_t>(blocks, pw, uf0, uf1, sa, sf, (int16_t*)out, up, qmax); break;
    }
    return VIPRS_OK;
}

}  // extern "C"
