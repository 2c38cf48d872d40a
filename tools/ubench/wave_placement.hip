// Which SIMD does wave w of a workgroup land on when several workgroups share a CU?
// (The chain waves of the E-step kernels are issue-bound: two of them on one SIMD halve each other.)
// Build: hipcc -O2 --offload-arch=gfx950 wave_placement.hip -o wave_placement ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>

__global__ void probe(unsigned* out, int spin) {
    extern __shared__ char lds[];
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the workgroup resident for a while so that the next ones must co-reside
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = hw;
        out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = xcc;
    }
}

int main(int argc, char** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 4;
    const int wg_per_cu = argc > 2 ? atoi(argv[2]) : 2;
    const int lds = argc > 3 ? atoi(argv[3]) : 40 * 1024;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    const int grid = n_cu * wg_per_cu;
    unsigned* d;
    hipMalloc(&d, sizeof(unsigned) * 2 * grid * waves);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    probe<<<grid, 64 * waves, lds>>>(d, 20000000);     // ~0.2 s at 100 MHz
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * grid * waves);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // per physical CU: which SIMD got wave 0 of each resident workgroup
    std::map<unsigned, std::vector<std::pair<int, int>>> by_cu;
    int hist[4][8] = {};
    for (int b = 0; b < grid; ++b)
        for (int w = 0; w < waves; ++w) {
            const unsigned hw = h[(b * waves + w) * 2], xcc = h[(b * waves + w) * 2 + 1] & 0xf;
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            if (w < 8) hist[simd][w]++;
            if (w == 0) by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back({b, (int)simd});
        }
    printf("%d CUs, %d workgroups of %d waves, %d B LDS each\n", n_cu, grid, waves, lds);
    printf("wave index -> SIMD histogram:\n");
    for (int w = 0; w < waves && w < 8; ++w)
        printf("  wave %d: SIMD0 %d  SIMD1 %d  SIMD2 %d  SIMD3 %d\n", w, hist[0][w], hist[1][w], hist[2][w], hist[3][w]);
    int shown = 0, collide = 0, multi = 0;
    for (auto& kv : by_cu) {
        if (kv.second.size() > 1) {
            ++multi;
            bool c = false;
            for (size_t i = 0; i < kv.second.size(); ++i)
                for (size_t j = i + 1; j < kv.second.size(); ++j) c |= kv.second[i].second == kv.second[j].second;
            collide += c;
        }
        if (shown < 6) {
            printf("  CU %05x:", kv.first);
            for (auto& e : kv.second) printf("  wg %d wave0->SIMD%d", e.first, e.second);
            printf("\n");
            ++shown;
        }
    }
    printf("%d physical CUs seen, %d host several workgroups, on %d of them two wave-0s share a SIMD\n",
           (int)by_cu.size(), multi, collide);
    return 0;
}
