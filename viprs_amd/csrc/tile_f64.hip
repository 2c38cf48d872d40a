// float64 state: the panel-walking kernels of estep_tile.h for every LD element type (spike-and-slab and the grid's
// (block, model) items), over the repacked dense blocks or the ragged blocks in the caller's own layout.
#include "internal.h"
#include "estep_tile.h"

namespace viprs {

template <typename U>
int launch_tile_f64(viprs_plan* P, EStepArgs<double> A, int model, bool dense) {
    const std::vector<BlockDesc>& list = dense ? P->dense_h : P->ragged_h;
    if (list.empty()) return VIPRS_OK;
    const int max_b = dense ? P->max_dense : P->max_ragged;
    const int qcap = (max_b + 3) / 4 * 4;
    const size_t shmem = tile_lds_bytes(qcap, sizeof(U));
    // q of a block lives in LDS; blocks beyond that (and the mixture) keep the row-by-row kernels
    if (shmem > 150 * 1024 || model == kGenMixture) return launch_generic<double, U>(P, A, model, dense);
    A.blocks = dense ? P->d_dense.p : P->d_ragged.p;
    A.n_blocks = (int)list.size();
    A.counter = P->d_counters.p + (dense ? 2 : 1);
    const void* kfn = nullptr;
    if (model == kGenGrid) kfn = dense ? (const void*)estep_tile_f64_kernel<U, TileGridColumn, true>
                                       : (const void*)estep_tile_f64_kernel<U, TileGridColumn, false>;
    else kfn = dense ? (const void*)estep_tile_f64_kernel<U, TileSpikeSlab, true>
                     : (const void*)estep_tile_f64_kernel<U, TileSpikeSlab, false>;
    if (shmem > 48 * 1024) HIP_TRY(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, kTileThreads, shmem));
    per_cu = std::max(1, per_cu);
    const int64_t n_items = (int64_t)A.n_blocks * std::max(1, A.n_active);
    const int grid = (int)std::min<int64_t>(n_items, (int64_t)P->n_cu * per_cu);
    int qc = qcap;
    void* params[] = {(void*)&A, (void*)&qc};
    HIP_TRY(hipLaunchKernel(kfn, dim3(grid), dim3(kTileThreads), params, shmem, P->stream));
    if (P->low_memory) {
        // second pass: the rows of all blocks of the list over the whole device
        DevBuf<int64_t>& rows = dense ? P->d_rowlist_dense : P->d_rowlist_ragged;      // groups of kTileGroupRows rows
        if (rows.n == 0) {
            std::vector<int64_t> h;
            for (size_t i = 0; i < list.size(); ++i)
                for (int r = 0; r < list[i].size; r += kTileGroupRows) h.push_back((int64_t)i << 32 | (int64_t)r);
            HIP_TRY(rows.alloc(h.size()));
            HIP_TRY(hipMemcpy(rows.p, h.data(), sizeof(int64_t) * h.size(), hipMemcpyHostToDevice));
        }
        const int64_t* pp = rows.p;
        int64_t n_rows = (int64_t)rows.n;
        void* params2[] = {(void*)&A, (void*)&pp, (void*)&n_rows};
        const void* k2 = dense ? (const void*)tile_f64_second_pass_kernel<U, true> : (const void*)tile_f64_second_pass_kernel<U, false>;
        HIP_TRY(hipLaunchKernel(k2, dim3(P->n_cu * 8), dim3(kTileThreads), params2, 0, P->stream));
    }
    return VIPRS_OK;
}

template int launch_tile_f64<int8_t>(viprs_plan*, EStepArgs<double>, int, bool);
template int launch_tile_f64<int16_t>(viprs_plan*, EStepArgs<double>, int, bool);
template int launch_tile_f64<int32_t>(viprs_plan*, EStepArgs<double>, int, bool);
template int launch_tile_f64<int64_t>(viprs_plan*, EStepArgs<double>, int, bool);
template int launch_tile_f64<float>(viprs_plan*, EStepArgs<double>, int, bool);
template int launch_tile_f64<double>(viprs_plan*, EStepArgs<double>, int, bool);

}  // namespace viprs
