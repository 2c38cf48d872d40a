// Development micro-benchmark: what an int8 -> double conversion costs next to the double fma it feeds (the row pass and
// the second pass of estep_tile.h spend one of each per LD element).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/cvt_f64.hip -o /tmp/cvt_f64 && /tmp/cvt_f64
// Modes (per wave, 32 independent accumulators, clocks per element):
//   0  v_fma_f64 alone                       1  v_cvt_f64_i32 + v_fma_f64 (static_cast, what the kernels did)
//   2  byte -> double through the exponent trick: (2^52 + 2^51 + 128 + b) - (2^52 + 2^51 + 128), b = byte ^ 0x80,
//      then v_fma_f64                        3  v_cvt_f32_i32 (SDWA-able) + v_cvt_f64_f32 + v_fma_f64
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int kIters = 2048;

__device__ inline double magic_byte(uint32_t dw_flipped, int k) {
    const uint32_t b = (dw_flipped >> (8 * k)) & 0xffu;
    const uint64_t bits = 0x4338000000000000ull | (uint64_t)b;
    return __builtin_bit_cast(double, bits) - 6755399441055872.0;       // 2^52 + 2^51 + 128
}

template <int MODE>
__global__ void bench(double* out, unsigned long long* cyc, const uint32_t* in) {
    const int lane = threadIdx.x & 63;
    double a[32];
    for (int k = 0; k < 32; ++k) a[k] = 1e-9 * (lane + k);
    const double e = 1.0 + 1e-12 * lane;
    uint32_t w[8];
    for (int k = 0; k < 8; ++k) w[k] = in[(threadIdx.x + 64 * k) & 1023];
    __syncthreads();
    const unsigned long long c0 = clock64();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            w[k] = w[k] * 1664525u + 1013904223u;        // (2 integer ops per 4 elements in every mode)
            const uint32_t d = w[k];
            const uint32_t f = d ^ 0x80808080u;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                double v;
                if (MODE == 0) v = __builtin_bit_cast(double, (uint64_t)d << 20 | 0x3ff0000000000000ull);
                else if (MODE == 1) v = static_cast<double>(static_cast<int8_t>(d >> (8 * x)));
                else if (MODE == 2) v = magic_byte(f, x);
                else v = static_cast<double>(static_cast<float>(static_cast<int8_t>(d >> (8 * x))));
                a[k * 4 + x] = __builtin_fma(v, e, a[k * 4 + x]);
            }
        }
    }
    const unsigned long long c1 = clock64();
    double s = 0;
    for (int k = 0; k < 32; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = c1 - c0;
}

template <int MODE>
void run(const char* name, double* out, unsigned long long* cyc, const uint32_t* in) {
    for (int waves : {1, 4, 8}) {          // waves per workgroup = per CU (1: one SIMD alone; 4: one per SIMD; 8: two per SIMD)
        hipLaunchKernelGGL(bench<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, in);
        hipLaunchKernelGGL(bench<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, in);
        hipDeviceSynchronize();
        unsigned long long c = 0;
        hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        std::vector<double> h(64);
        hipMemcpy(h.data(), out, 64 * sizeof(double), hipMemcpyDeviceToHost);
        printf("%-44s %d wave(s)/CU: %6.2f clk (s_memtime units) per element per wave   [check %.6g]\n", name, waves,
               (double)c / (kIters * 32.0), h[5]);
    }
}

int main() {
    double* out;
    unsigned long long* cyc;
    uint32_t* in;
    hipMalloc(&out, 8 * 64 * 8 * sizeof(double));
    hipMalloc(&cyc, 8);
    hipMalloc(&in, 4096);
    std::vector<uint32_t> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 2654435761u * (i + 1);
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    run<0>("fma_f64 alone", out, cyc, in);
    run<1>("cvt_f64_i32 + fma_f64", out, cyc, in);
    run<2>("exponent trick (bfe + add_f64) + fma_f64", out, cyc, in);
    run<3>("cvt_f32_i32 + cvt_f64_f32 + fma_f64", out, cyc, in);
    // the two conversions must agree for every byte
    return 0;
}
