// float64 state: the panel-walking kernels of estep_tile.h for every LD element type (spike-and-slab and the grid's
// (block, model) items), over the repacked dense blocks or the ragged blocks in the caller's own layout.
#include "internal.h"
#include "estep_tile.h"

namespace viprs {

// One launch of the panel-walking kernel over blocks [first, first + count) of a list, NW waves per workgroup.
template <typename U, int NW>
static int launch_tile_class(viprs_plan* P, EStepArgs<double> A, int model, bool dense, const BlockDesc* d_blocks, int count,
                             int max_b, int32_t* counter, hipStream_t stream, int cus) {
    if (count == 0) return VIPRS_OK;
    const int qcap = (max_b + 3) / 4 * 4;
    const size_t shmem = tile_lds_bytes(qcap, sizeof(U));
    A.blocks = d_blocks;
    A.n_blocks = count;
    A.counter = counter;
    const void* kfn = nullptr;
    if (model == kGenGrid) kfn = dense ? (const void*)estep_tile_f64_kernel<U, TileGridColumn, true, NW>
                                       : (const void*)estep_tile_f64_kernel<U, TileGridColumn, false, NW>;
    else if (model == kGenMixture && A.width <= kTileMixK)
        kfn = dense ? (const void*)estep_tile_f64_kernel<U, TileMixture<kTileMixK>, true, NW>
                    : (const void*)estep_tile_f64_kernel<U, TileMixture<kTileMixK>, false, NW>;
    else if (model == kGenMixture)
        kfn = dense ? (const void*)estep_tile_f64_kernel<U, TileMixture<kTileMixWideK>, true, NW>
                    : (const void*)estep_tile_f64_kernel<U, TileMixture<kTileMixWideK>, false, NW>;
    else kfn = dense ? (const void*)estep_tile_f64_kernel<U, TileSpikeSlab, true, NW>
                     : (const void*)estep_tile_f64_kernel<U, TileSpikeSlab, false, NW>;
    if (shmem > 48 * 1024) HIP_TRY(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, NW * 64, shmem));
    per_cu = std::max(1, per_cu);
    const int64_t n_items = (int64_t)count * std::max(1, A.n_active);
    const int grid = (int)std::min<int64_t>(n_items, (int64_t)cus * per_cu);      // persistent workgroups on `cus` CUs
    int qc = qcap;
    void* params[] = {(void*)&A, (void*)&qc};
    HIP_TRY(hipLaunchKernel(kfn, dim3(grid), dim3(NW * 64), params, shmem, stream));
    return VIPRS_OK;
}

// Blocks at least this large take 8-wave workgroups (1 chain + 7 waves for the row pass): with 3 row-pass waves the
// conversion + fma of 64 x n LD elements per panel outlasts the chain's 64 steps from ~1 800 columns on
// (tools/fp64_block_bench.py).  The two classes run side by side on two streams.
constexpr int kTileBigBlock = 1792;

template <typename U>
int launch_tile_f64(viprs_plan* P, EStepArgs<double> A, int model, bool dense) {
    const std::vector<BlockDesc>& list = dense ? P->dense_h : P->ragged_h;
    if (list.empty()) return VIPRS_OK;
    const int max_b = dense ? P->max_dense : P->max_ragged;
    // q of a block lives in LDS; blocks beyond that (and mixtures of more than kTileMixWideK components) keep the row-by-row kernels
    if (tile_lds_bytes((max_b + 3) / 4 * 4, sizeof(U)) > 150 * 1024 || (model == kGenMixture && A.width > kTileMixWideK))
        return launch_generic<double, U>(P, A, model, dense);
    const BlockDesc* d_blocks = dense ? P->d_dense.p : P->d_ragged.p;
    // the lists are in descending order of size: the big class is a prefix
    int n_big = 0;
    while (n_big < (int)list.size() && list[(size_t)n_big].size >= kTileBigBlock) ++n_big;
    const int n_small = (int)list.size() - n_big;
    int32_t* counter = P->d_counters.p + (dense ? 2 : 1);

    // second pass of the upper form (update_q_factor): groups of kTileGroupRows rows in list order -- the big class's
    // groups first -- over the whole device
    DevBuf<int64_t>& groups = dense ? P->d_rowlist_dense : P->d_rowlist_ragged;
    int64_t n_groups_big = 0;
    for (int i = 0; i < n_big; ++i) n_groups_big += (list[(size_t)i].size + kTileGroupRows - 1) / kTileGroupRows;
    if (P->low_memory && groups.n == 0) {
        std::vector<int64_t> h;
        for (size_t i = 0; i < list.size(); ++i)
            for (int r = 0; r < list[i].size; r += kTileGroupRows) h.push_back((int64_t)i << 32 | (int64_t)r);
        HIP_TRY(groups.alloc(h.size()));
        HIP_TRY(hipMemcpy(groups.p, h.data(), sizeof(int64_t) * h.size(), hipMemcpyHostToDevice));
    }
    auto second_pass = [&](int64_t first, int64_t count, hipStream_t stream, int cus) -> int {
        if (!P->low_memory || count == 0) return VIPRS_OK;
        EStepArgs<double> A2 = A;
        A2.blocks = d_blocks;
        A2.n_blocks = (int)list.size();
        const int64_t* pp = groups.p + first;
        void* params2[] = {(void*)&A2, (void*)&pp, (void*)&count};
        const int64_t n_items = count * std::max(1, A.n_active);
        const void* k2 = dense ? (const void*)tile_f64_second_pass_dense_kernel<U> : (const void*)tile_f64_second_pass_kernel<U, false>;
        // (the dense kernel: as many workgroups as are resident at a time, each wave a pipeline over its share of the groups)
        int per_cu = 8;
        if (dense) {
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k2, kTileThreads, 0));
            per_cu = std::max(1, per_cu);
            if (const char* f = getenv("VIPRS_F64_PASS2_WGS")) if (*f) per_cu = std::max(1, atoi(f));       // experiments
        }
        const int grid = (int)std::min<int64_t>((n_items + kTileThreads / 64 - 1) / (kTileThreads / 64), (int64_t)cus * per_cu);
        HIP_TRY(hipLaunchKernel(k2, dim3(grid), dim3(kTileThreads), params2, 0, stream));
        return VIPRS_OK;
    };
    const int64_t n_groups = P->low_memory ? (int64_t)groups.n : 0;

    if (n_big > 0 && n_small > 0) {
        if (!P->side_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&P->side_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&P->ev_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&P->ev_join, hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(P->ev_fork, P->stream));
        HIP_TRY(hipStreamWaitEvent(P->side_stream, P->ev_fork, 0));
        // an 8-wave workgroup fills a CU; the persistent workgroups of the other class leave those CUs alone (they would
        // not drain before their queue is empty)
        const int64_t big_items = (int64_t)n_big * std::max(1, A.n_active);
        // how many CUs for the big class: the split under which the later of the two classes ends earliest -- a panel
        // costs the chain's 64 steps (~19 us on an 8-wave workgroup, ~25 us on a 4-wave one sharing its CU with another),
        // a class cannot end before its largest block
        int big_cus = 1;
        {
            double panels_big = 0.0, panels_small = 0.0;
            for (int i = 0; i < (int)list.size(); ++i)
                (i < n_big ? panels_big : panels_small) += (double)((list[(size_t)i].size + kPanel - 1) / kPanel);
            const double nm = (double)std::max(1, A.n_active);
            const double floor_big = (double)((list[0].size + kPanel - 1) / kPanel) * 19e-6;
            const double floor_small = (double)((list[(size_t)n_big].size + kPanel - 1) / kPanel) * 25e-6;
            double best = 1e30;
            const int max_cus = (int)std::min<int64_t>(big_items, P->n_cu - 1);
            for (int c = 1; c <= max_cus; ++c) {
                const double tb = std::max(floor_big, panels_big * nm / c * 19e-6);
                const double ts = std::max(floor_small, panels_small * nm / ((P->n_cu - c) * 2.0) * 25e-6);
                const double t = std::max(tb, ts);
                if (t < best * (1.0 - 1e-9)) { best = t; big_cus = c; }
            }
        }
        // the big class goes FIRST and on the plan's own stream (it starts the moment the work before it ends; the other
        // class has to come through the fork event and finds those CUs taken)
        // (after the fork every exit joins the side stream again: a kernel of the side stream must never be left running
        //  behind a caller that was told the call failed and may reuse or free the state)
        auto forked = [&]() -> int {
            int rc = launch_tile_class<U, 8>(P, A, model, dense, d_blocks, n_big, max_b, P->d_counters.p + (dense ? 20 : 21), P->stream, big_cus);
            if (rc != VIPRS_OK) return rc;
            rc = launch_tile_class<U, 4>(P, A, model, dense, d_blocks + n_big, n_small, list[(size_t)n_big].size, counter, P->side_stream,
                                         P->n_cu - big_cus);
            if (rc != VIPRS_OK) return rc;
            // each class's second pass behind its own sweep: the small class's runs while the largest blocks are still
            // walking their panels
            rc = second_pass(n_groups_big, n_groups - n_groups_big, P->side_stream, P->n_cu - big_cus);
            if (rc != VIPRS_OK) return rc;
            return second_pass(0, n_groups_big, P->stream, P->n_cu);
        };
        const int rc = forked();
        const hipError_t e1 = hipEventRecord(P->ev_join, P->side_stream);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(P->stream, P->ev_join, 0) : e1;
        if (e2 != hipSuccess) (void)hipStreamSynchronize(P->side_stream);       // last resort: never leave it unjoined
        if (rc != VIPRS_OK) return rc;
        HIP_TRY(e2);
    } else {
        int rc = n_big > 0 ? launch_tile_class<U, 8>(P, A, model, dense, d_blocks, n_big, max_b, counter, P->stream, P->n_cu)
                           : launch_tile_class<U, 4>(P, A, model, dense, d_blocks, n_small, max_b, counter, P->stream, P->n_cu);
        if (rc != VIPRS_OK) return rc;
        rc = second_pass(0, n_groups, P->stream, P->n_cu);
        if (rc != VIPRS_OK) return rc;
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
