// Development micro-benchmark: does a partially filled EXEC mask shorten a dependent VALU chain on gfx950?
// (The chain wave's sigmoid is needed for ONE lane per step; if inactive rows were skipped, running it under a
// one-row mask would shorten the step.)   hipcc -O3 --offload-arch=gfx950 exec_mask_latency.hip -o exec_mask_latency
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kIters = 4096;

template <int MODE>
__global__ void bench(float* out, unsigned long long* wall, int active) {
    const int lane = threadIdx.x & 63;
    float acc = lane;
    double dacc = 1.0 + 1e-9 * lane;
    const unsigned long long w0 = wall_clock64();
    if (lane < active) {
        if (MODE == 0) {
            for (int it = 0; it < kIters; ++it) {
#pragma unroll
                for (int k = 0; k < 32; ++k) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
            }
        } else if (MODE == 1) {
            for (int it = 0; it < kIters; ++it) {
#pragma unroll
                for (int k = 0; k < 32; ++k) dacc = __builtin_fma(dacc, 1.00000000001, 1e-12);
            }
        } else if (MODE == 2) {
            for (int it = 0; it < kIters; ++it) {
#pragma unroll
                for (int k = 0; k < 8; ++k) dacc = __builtin_amdgcn_rcp(dacc) + 1.0;
            }
        } else if (MODE == 3) {     // independent f32 (8 chains)
            float a[8];
            for (int k = 0; k < 8; ++k) a[k] = acc + k;
            for (int it = 0; it < kIters; ++it) {
#pragma unroll
                for (int k = 0; k < 32; ++k) a[k & 7] = __builtin_fmaf(a[k & 7], 1.0000001f, 1e-9f);
            }
            for (int k = 0; k < 8; ++k) acc += a[k];
        }
    }
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *wall = w1 - w0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (float)dacc;
}

template <int MODE>
void run(const char* name, int per_iter) {
    float* out; unsigned long long* wall;
    hipMalloc(&out, 256 * 64 * 4); hipMalloc(&wall, 8);
    for (int active : {64, 32, 16, 4, 1}) {
        for (int rep = 0; rep < 2; ++rep) bench<MODE><<<1, 64>>>(out, wall, active);
        hipDeviceSynchronize();
        unsigned long long w;
        hipMemcpy(&w, wall, 8, hipMemcpyDeviceToHost);
        printf("%-28s active lanes %2d  %7.2f ns/op\n", name, active, w * 10.0 / kIters / per_iter);
    }
    hipFree(out); hipFree(wall);
}

int main() {
    run<0>("dependent f32 fma", 32);
    run<3>("independent f32 fma", 32);
    run<1>("dependent f64 fma", 32);
    run<2>("dependent v_rcp_f64 + add", 8);
    return 0;
}
