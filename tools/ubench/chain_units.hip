// Development micro-benchmark: single-wave instruction timings that the chain waves depend on
// (LDS broadcast reads of 4/8/16 bytes, dependent f32/f64 fma chains, v_rcp_f64, ds_bpermute).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/chain_units.hip -o tools/ubench/chain_units && ./chain_units
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

constexpr int kIters = 4096;

template <int MODE>
__global__ void bench(float* out, unsigned long long* cyc, unsigned long long* wall, int stride) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 1.0f + 1e-7f * i;
    __syncthreads();
    const int half = lane >> 5;
    float acc = lane;
    double dacc = 1.0 + 1e-9 * lane;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    if (MODE == 0) {            // ds_read_b32, one address per half (broadcast)
        for (int it = 0; it < kIters; ++it) {
            const float* p = lds + ((it & 63) * 64 + 32 * half) * 1;
#pragma unroll
            for (int k = 0; k < 32; ++k) acc += p[k];
        }
    } else if (MODE == 1) {     // ds_read_b64 broadcast
        for (int it = 0; it < kIters; ++it) {
            const f32x2* p = reinterpret_cast<const f32x2*>(lds + (it & 63) * 64 + 32 * half);
#pragma unroll
            for (int k = 0; k < 16; ++k) { f32x2 v = p[k]; acc += v[0]; acc += v[1]; }
        }
    } else if (MODE == 2) {     // ds_read_b128 broadcast
        for (int it = 0; it < kIters; ++it) {
            const f32x4* p = reinterpret_cast<const f32x4*>(lds + (it & 63) * 64 + 32 * half);
#pragma unroll
            for (int k = 0; k < 8; ++k) { f32x4 v = p[k]; acc += v[0]; acc += v[1]; acc += v[2]; acc += v[3]; }
        }
    } else if (MODE == 3) {     // ds_read_b128, per-lane distinct address (conflict-free layout)
        for (int it = 0; it < kIters; ++it) {
            const f32x4* p = reinterpret_cast<const f32x4*>(lds + (it & 3) * 2048) + lane;
#pragma unroll
            for (int k = 0; k < 8; ++k) { f32x4 v = p[k * 64]; acc += v[0]; acc += v[1]; acc += v[2]; acc += v[3]; }
        }
    } else if (MODE == 4) {     // 32 dependent f32 fma
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int k = 0; k < 32; ++k) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
        }
    } else if (MODE == 5) {     // 32 dependent f64 fma
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int k = 0; k < 32; ++k) dacc = __builtin_fma(dacc, 1.00000000001, 1e-12);
        }
    } else if (MODE == 6) {     // 32 independent f32 fma (8 chains)
        float a[8];
        for (int k = 0; k < 8; ++k) a[k] = acc + k;
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int k = 0; k < 32; ++k) a[k & 7] = __builtin_fmaf(a[k & 7], 1.0000001f, 1e-9f);
        }
        for (int k = 0; k < 8; ++k) acc += a[k];
    } else if (MODE == 7) {     // 8 dependent v_rcp_f64
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) dacc = __builtin_amdgcn_rcp(dacc) + 1.0;
        }
    } else if (MODE == 8) {     // 8 dependent ds_bpermute
        int v = lane;
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v = __builtin_amdgcn_ds_bpermute(((v + 1) & 63) << 2, v);
        }
        acc += v;
    } else if (MODE == 9) {     // 8 independent f64 chains
        double a[8];
        for (int k = 0; k < 8; ++k) a[k] = dacc + k;
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int k = 0; k < 32; ++k) a[k & 7] = __builtin_fma(a[k & 7], 1.00000000001, 1e-12);
        }
        for (int k = 0; k < 8; ++k) dacc += a[k];
    } else if (MODE == 10) {    // cvt f32->f64->f32 dependent chain, 8 round trips
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc = (float)((double)acc * 1.0000001);
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { *cyc = c1 - c0; *wall = w1 - w0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (float)dacc;
}

template <int MODE>
void run(const char* name, int per_iter, int waves) {
    float* out; unsigned long long *cyc, *wall;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8); hipMalloc(&wall, 8);
    for (int rep = 0; rep < 2; ++rep) bench<MODE><<<256, 64 * waves>>>(out, cyc, wall, 1);
    hipDeviceSynchronize();
    unsigned long long c, w;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&w, wall, 8, hipMemcpyDeviceToHost);
    const double ns = w * 10.0;
    printf("%-44s waves/WG=%d  %8.2f ns/op  %8.2f clk64/op  (%.0f MHz by clock64/wall)\n", name, waves,
           ns / kIters / per_iter, (double)c / kIters / per_iter, c / ns * 1e3);
    hipFree(out); hipFree(cyc); hipFree(wall);
}

int main() {
    for (int waves : {1, 4}) {
        run<0>("ds_read_b32 broadcast (per op)", 32, waves);
        run<1>("ds_read_b64 broadcast (per op)", 16, waves);
        run<2>("ds_read_b128 broadcast (per op)", 8, waves);
        run<3>("ds_read_b128 distinct (per op)", 8, waves);
        run<4>("dependent f32 fma", 32, waves);
        run<6>("independent f32 fma", 32, waves);
        run<5>("dependent f64 fma", 32, waves);
        run<9>("independent f64 fma", 32, waves);
        run<7>("dependent v_rcp_f64 + add", 8, waves);
        run<8>("dependent ds_bpermute", 8, waves);
        run<10>("cvt f32->f64, mul, cvt ->f32", 8, waves);
    }
    return 0;
}
