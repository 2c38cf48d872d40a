// Development check: what s_getreg_b32 hwreg(HW_REG_XCC_ID) returns per workgroup, next to blockIdx % 8.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/xcc_id.hip -o build/xcc_id && build/xcc_id
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
    const int n = 2048;
    unsigned* d;
    hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d);
    std::vector<unsigned> h(n);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    printf("raw values of blocks 0..15:");
    for (int i = 0; i < 16; ++i) printf(" 0x%x", h[i]);
    printf("\n");
    int hist[8][8] = {};
    for (int i = 0; i < n; ++i) hist[i % 8][h[i] & 7]++;
    for (int r = 0; r < 8; ++r) {
        printf("blockIdx %% 8 = %d: low 3 bits histogram", r);
        for (int c = 0; c < 8; ++c) printf(" %4d", hist[r][c]);
        printf("\n");
    }
    return 0;
}
