// Band kernel (estep_band.h) launchers: windowed (ragged) components, fp32 state, f32 / int8 / int16 LD.
#include "internal.h"
#include "estep_band.h"

namespace viprs {

int band_ring_panels(const viprs_plan* P) {
    int rp = 4;
    while (rp < P->max_band_panels) rp *= 2;
    return rp;
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
    if (model == kBandGridColumn) kfn = exact ? BK(GridColumnModel<true>) : BK(GridColumnModel<false>);
    else if (model == kBandMixture) kfn = exact ? BK(MixtureSerialModel<true>) : BK(MixtureSerialModel<false>);
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

template int launch_band<float>(viprs_plan*, EStepArgs<float>, int);
template int launch_band<int8_t>(viprs_plan*, EStepArgs<float>, int);
template int launch_band<int16_t>(viprs_plan*, EStepArgs<float>, int);

}  // namespace viprs
