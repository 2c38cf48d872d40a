// Development tool: where do the waves of co-resident workgroups land?  Every wave records its HW_ID (SIMD, CU, SE)
// and XCC_ID; the host prints, per CU, the SIMD of each workgroup's wave 0..3.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/wave_placement.hip -o /tmp/wp && /tmp/wp [waves_per_wg] [lds_kb]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void probe(unsigned* out, int spin) {
    extern __shared__ float smem[];
    const int wave = threadIdx.x >> 6;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * (blockDim.x >> 6) + wave) * 2] = hw;
        out[(blockIdx.x * (blockDim.x >> 6) + wave) * 2 + 1] = xcc;
    }
    // stay resident long enough for the whole grid to be placed
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) smem[threadIdx.x] += 1.0f;
    __syncthreads();
}

int main(int argc, char** argv) {
    const int nw = argc > 1 ? atoi(argv[1]) : 4, lds_kb = argc > 2 ? atoi(argv[2]) : 72;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount, grid = 2 * n_cu;
    unsigned* out; hipMalloc(&out, grid * nw * 8);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    probe<<<grid, nw * 64, lds_kb * 1024>>>(out, 200000);     // 2 ms
    hipDeviceSynchronize();
    std::vector<unsigned> h(grid * nw * 2);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> by_cu;
    for (int w = 0; w < grid; ++w) {
        const unsigned hw = h[(w * nw) * 2], xcc = h[(w * nw) * 2 + 1] & 15;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        by_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(w);
    }
    int shown = 0, same = 0, pairs = 0;
    for (auto& kv : by_cu) {
        if (kv.second.size() >= 2) {
            ++pairs;
            const unsigned s0 = (h[(kv.second[0] * nw) * 2] >> 4) & 3, s1 = (h[(kv.second[1] * nw) * 2] >> 4) & 3;
            same += s0 == s1;
        }
        if (shown++ < 12) {
            printf("xcc %u se %u cu %2u:", kv.first >> 16, (kv.first >> 8) & 7, kv.first & 15);
            for (int w : kv.second) {
                printf("  wg %3d simd", w);
                for (int v = 0; v < nw; ++v) printf(" %u", (h[(w * nw + v) * 2] >> 4) & 3);
            }
            printf("\n");
        }
    }
    printf("%zu CUs seen, %d with two workgroups, wave 0 of both on the same SIMD in %d\n", by_cu.size(), pairs, same);
    return 0;
}
