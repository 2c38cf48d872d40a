// Development tool: checks the one-instruction int8 / int16 -> float conversions (v_cvt_f32_i32 with an SDWA byte / word
// select and sign extension) against static_cast<float> for every byte / word value, alone and feeding an MFMA.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/sdwa_cvt.hip -o build/ubench/sdwa_cvt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int J> __device__ __forceinline__ float cvt8(unsigned w) {
    float f;
    if constexpr (J == 0) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(f) : "v"(w));
    else if constexpr (J == 1) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(f) : "v"(w));
    else if constexpr (J == 2) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(f) : "v"(w));
    else asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(f) : "v"(w));
    return f;
}
template <int J> __device__ __forceinline__ float cvt16(unsigned w) {
    float f;
    if constexpr (J == 0) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(f) : "v"(w));
    else asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f) : "v"(w));
    return f;
}

__global__ void k(const unsigned* in, float* out, float* ref, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned w = in[i];
    out[6 * i + 0] = cvt8<0>(w); out[6 * i + 1] = cvt8<1>(w); out[6 * i + 2] = cvt8<2>(w); out[6 * i + 3] = cvt8<3>(w);
    out[6 * i + 4] = cvt16<0>(w); out[6 * i + 5] = cvt16<1>(w);
    ref[6 * i + 0] = (float)(int8_t)(w); ref[6 * i + 1] = (float)(int8_t)(w >> 8); ref[6 * i + 2] = (float)(int8_t)(w >> 16);
    ref[6 * i + 3] = (float)(int8_t)(w >> 24); ref[6 * i + 4] = (float)(int16_t)(w); ref[6 * i + 5] = (float)(int16_t)(w >> 16);
}

int main() {
    const int n = 1 << 16;
    std::vector<unsigned> h(n);
    for (int i = 0; i < n; ++i) h[i] = (unsigned)i * 2654435761u ^ ((unsigned)i << 16) ^ (unsigned)i;
    for (int i = 0; i < 65536; ++i) h[i] = (h[i] & 0xFFFF0000u) | (unsigned)i;          // every low word (hence every low byte)
    unsigned* d; float *o, *r;
    hipMalloc(&d, n * 4); hipMalloc(&o, n * 24); hipMalloc(&r, n * 24);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(d, o, r, n);
    std::vector<float> ho(6 * n), hr(6 * n);
    hipMemcpy(ho.data(), o, n * 24, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), r, n * 24, hipMemcpyDeviceToHost);
    long bad[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) for (int j = 0; j < 6; ++j) if (ho[6 * i + j] != hr[6 * i + j]) { if (bad[j]++ < 3) printf("j=%d w=%08x got %g want %g\n", j, h[i], ho[6 * i + j], hr[6 * i + j]); }
    printf("mismatches BYTE_0..3 WORD_0..1: %ld %ld %ld %ld %ld %ld of %d\n", bad[0], bad[1], bad[2], bad[3], bad[4], bad[5], n);
    return 0;
}
