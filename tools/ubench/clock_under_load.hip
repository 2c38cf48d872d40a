// Development micro-benchmark: does the shader clock drop while the chip streams from HBM?
// A pure-ALU dependent-fma chain runs on 16 workgroups, alone and next to a 240-workgroup streaming kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/clock_under_load.hip -o /tmp/cul && /tmp/cul
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void alu_chain(float* out, unsigned long long* cyc, unsigned long long* wall, int iters) {
    float acc = threadIdx.x;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64; ++k) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { cyc[blockIdx.x] = c1 - c0; wall[blockIdx.x] = w1 - w0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void streamer(const float4* __restrict__ src, size_t n, float* out, int passes) {
    float acc = 0.f;
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            const float4 v = src[i];
            acc += v.x + v.y + v.z + v.w;
        }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    const size_t n = (size_t)1 << 28;                               // 4 GiB of float4
    float4* src; float *o1, *o2; unsigned long long *cyc, *wall;
    hipMalloc(&src, n * 16); hipMemset(src, 0, n * 16);
    hipMalloc(&o1, 1 << 20); hipMalloc(&o2, 4 << 20); hipMalloc(&cyc, 8 * 64); hipMalloc(&wall, 8 * 64);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const int iters = 200000;                                        // ~ 30 ms of dependent fma
    for (int loaded = 0; loaded < 2; ++loaded) {
        for (int rep = 0; rep < 2; ++rep) {
            if (loaded) streamer<<<960, 256, 0, s2>>>(src, n, o2, 40);
            alu_chain<<<16, 64, 0, s1>>>(o1, cyc, wall, iters);
            hipStreamSynchronize(s1);
            hipDeviceSynchronize();
        }
        unsigned long long c[16], w[16];
        hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost); hipMemcpy(w, wall, sizeof(w), hipMemcpyDeviceToHost);
        double ns = 0, cy = 0;
        for (int i = 0; i < 16; ++i) { ns += w[i] * 10.0; cy += c[i]; }
        printf("%-22s dependent fma: %.3f ns/op, %.2f clk64/op, clock64 rate %.0f MHz\n",
               loaded ? "next to HBM streaming" : "alone", ns / 16 / iters / 64, cy / 16 / iters / 64, cy / ns * 1e3);
    }
    return 0;
}
