// Development micro-benchmark: HBM streaming rate of the panel kernel's access pattern (each updater wave
// reads 64 rows x 1 KB of a row-major b x b block, 16 loads in flight) against a contiguous stream of the
// same bytes.   hipcc -O3 --offload-arch=gfx950 tools/ubench/stream_pattern.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>   // 0: row-major strips (the kernel's pattern); 1: tile-major (each strip-panel contiguous); 2: fully linear
__global__ __launch_bounds__(256) void stream(const float* __restrict__ ld, int b, int n_blocks, int* counter, float* out) {
    __shared__ int s_blk;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    for (;;) {
        if (threadIdx.x == 0) s_blk = atomicAdd(counter, 1);
        __syncthreads();
        const int blk = s_blk;
        __syncthreads();
        if (blk >= n_blocks) break;
        const float* base = ld + (size_t)blk * b * b;
        const int np = b / 64, ns = b / 256;                       // b multiple of 256
        for (int p = 0; p < np; ++p) {
            for (int s = wave; s < ns; s += 4) {
                float4 buf[16];
                auto addr = [&](int row) -> const float4* {
                    if (MODE == 0) return reinterpret_cast<const float4*>(base + (size_t)(p * 64 + row) * b + s * 256 + lane * 4);
                    if (MODE == 1) return reinterpret_cast<const float4*>(base + ((size_t)(p * ns + s) * 64 + row) * 256 + lane * 4);
                    return reinterpret_cast<const float4*>(base + ((size_t)(p * ns + s) * 64 + row) * 256 + lane * 4);
                };
#pragma unroll
                for (int k = 0; k < 16; ++k) buf[k] = *addr(k);
#pragma unroll 1
                for (int g = 0; g < 3; ++g) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const float4 v = buf[k];
                        buf[k] = *addr(16 * (g + 1) + k);
                        acc += v.x + v.y + v.z + v.w;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) acc += buf[k].x + buf[k].y + buf[k].z + buf[k].w;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const int b = 768, n_blocks = 1536;                              // 3.6 GB
    const size_t n = (size_t)n_blocks * b * b;
    float *ld, *out; int* counter;
    hipMalloc(&ld, n * 4); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&counter, 4);
    hipMemset(ld, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {256, 512, 768, 1024}) {
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipMemset(counter, 0, 4);
                hipEventRecord(e0);
                if (mode == 0) stream<0><<<grid, 256>>>(ld, b, n_blocks, counter, out);
                else stream<1><<<grid, 256>>>(ld, b, n_blocks, counter, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep) best = ms < best ? ms : best;
            }
            printf("grid %4d (%.1f WG/CU) %-10s %.3f ms  %.2f TB/s\n", grid, grid / 256.0, mode == 0 ? "row-major" : "tile-major", best, n * 4 / best / 1e9);
        }
    }
    return 0;
}
