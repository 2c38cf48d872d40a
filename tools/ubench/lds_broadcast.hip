// Development tool: LDS cost of the read patterns of the batched grid chain -- clocks per ds_read for one wave,
// eight reads in flight, addresses (a) the same 16 B for all lanes, (b) one 16 B block per half of the wave,
// (c) 16 B per lane, consecutive (conflict-free), for b128 / b64 / b32 reads.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/lds_broadcast.hip -o /tmp/lb && /tmp/lb
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE, int WIDTH>
__global__ void bench(float* out, unsigned long long* clk, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 8 + 64];
    for (int i = threadIdx.x; i < 64 * 4 * 8 + 64; i += 64) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x;
    const int base = MODE == 0 ? 0 : MODE == 1 ? (lane >> 5) * 64 : lane * 4;      // floats
    float acc = 0.0f;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int off = base + ((MODE == 2) ? j * 256 : j * 4);
            if (WIDTH == 4) { const f4 v = *reinterpret_cast<const f4*>(lds + off); acc += v[0] + v[3]; }
            else if (WIDTH == 2) { const f2 v = *reinterpret_cast<const f2*>(lds + off); acc += v[0] + v[1]; }
            else acc += lds[off];
        }
        asm volatile("" : "+v"(acc));
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) *clk = t1 - t0;
    out[threadIdx.x] = acc;
}

template <int MODE, int WIDTH>
void run(const char* name, float* out, unsigned long long* clk) {
    const int iters = 20000;
    bench<MODE, WIDTH><<<1, 64>>>(out, clk, iters);
    bench<MODE, WIDTH><<<1, 64>>>(out, clk, iters);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("%-44s %6.1f clk per read\n", name, (double)c / iters / 8);
}

int main() {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 256); hipMalloc(&clk, 8);
    run<0, 4>("b128, one address for the whole wave", out, clk);
    run<1, 4>("b128, one address per half", out, clk);
    run<2, 4>("b128, 16 B per lane, consecutive", out, clk);
    run<0, 2>("b64, one address for the whole wave", out, clk);
    run<1, 2>("b64, one address per half", out, clk);
    run<0, 1>("b32, one address for the whole wave", out, clk);
    run<1, 1>("b32, one address per half", out, clk);
    run<2, 1>("b32, stride 16 B per lane", out, clk);
    return 0;
}
