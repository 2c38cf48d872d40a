// Development tool: (1) exhaustive bit-check of cheaper formulations of the chain's exact sigmoid
// (e_step.hpp:245-261 with glibc's expf) against the shipped sigmoid_exact over ALL 2^32 float
// inputs, (2) ns per dependent evaluation of each formulation on a single wave (the serial chain's
// situation: one wave per SIMD, every instruction on the critical path).
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -I viprs_amd/csrc tools/ubench/sigmoid_variants.hip -o /tmp/sv && /tmp/sv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#include "device_math.h"

using namespace viprs;

// ---- candidate pieces ---------------------------------------------------------------------------------
// expf for x <= 0 without the final underflow select: the argument is clamped to -104 and the double
// result 2^-150.04 rounds to 0.0f by itself (glibc returns 0 below -103.972).
template <int LOOKUP>
__device__ __forceinline__ float expf_noselect(float x, const ExpTab& tab, int sel = 0) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32;
    const double xd = (double)fmaxf(x, -104.0f);
    const double z = InvLn2N * xd;
    const double kd = rint(z);
    const double r = fma(InvLn2N, xd, -kd);
    const int ki = (int)kd;
    int tlo, thi;
    if (LOOKUP == kLookupLane) {
        const int sidx = __builtin_amdgcn_readlane(ki, sel);      // lane select uses the low 6 bits; lanes l and l+32 hold the same entry
        tlo = __builtin_amdgcn_readlane(tab.lo, sidx);
        thi = __builtin_amdgcn_readlane(tab.hi, sidx);
    } else {
        tlo = __builtin_amdgcn_ds_bpermute(ki << 2, tab.lo);
        thi = __builtin_amdgcn_ds_bpermute(ki << 2, tab.hi);
    }
    thi += (int)((unsigned)ki << 15);
    const uint64_t t = ((uint64_t)(uint32_t)thi << 32) | (uint32_t)tlo;
    return expf_glibc_finish(r, t);
}

// the same with glibc's own round-to-integer (e_expf.c without TOINT_INTRINSICS): kd = z + 0x1.8p52, ki = the low
// word of kd's bits, kd -= 0x1.8p52 -- the table index is ready one conversion earlier
template <int LOOKUP>
__device__ __forceinline__ float expf_shift(float x, const ExpTab& tab, int sel = 0) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32;
    const double Shift = 0x1.8p52;
    const double xd = (double)fmaxf(x, -104.0f);
    const double z = InvLn2N * xd;
    double kd = z + Shift;
    int ki = (int)(uint32_t)(uint64_t)__double_as_longlong(kd);
    kd = kd - Shift;
    const double r = fma(InvLn2N, xd, -kd);
    int tlo, thi;
    if (LOOKUP == kLookupLane) {
        asm volatile("" : "+v"(ki));
        const int sidx = __builtin_amdgcn_readlane(ki, sel);
        tlo = __builtin_amdgcn_readlane(tab.lo, sidx);
        thi = __builtin_amdgcn_readlane(tab.hi, sidx);
        thi += (int)((unsigned)sidx << 15);
    } else {
        tlo = __builtin_amdgcn_ds_bpermute(ki << 2, tab.lo);
        thi = __builtin_amdgcn_ds_bpermute(ki << 2, tab.hi);
        thi += (int)((unsigned)ki << 15);
    }
    const uint64_t t = ((uint64_t)(uint32_t)thi << 32) | (uint32_t)tlo;
    return expf_glibc_finish(r, t);
}

// divide variants: num / den, den in [1, 2], num in [0, 1]
__device__ __forceinline__ double div_v1(double num, double den) {      // one Newton step + residual correction
    double r = __builtin_amdgcn_rcp(den);
    const double e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q0 = num * r;
    const double rem = __builtin_fma(-den, q0, num);
    return __builtin_fma(rem, r, q0);
}
__device__ __forceinline__ double div_v2(double num, double den) {      // two residual corrections on the raw reciprocal
    const double r = __builtin_amdgcn_rcp(den);
    double q = num * r;
    double rem = __builtin_fma(-den, q, num);
    q = __builtin_fma(rem, r, q);
    rem = __builtin_fma(-den, q, num);
    return __builtin_fma(rem, r, q);
}
__device__ __forceinline__ double div_v3(double num, double den) {      // one residual correction on the raw reciprocal
    const double r = __builtin_amdgcn_rcp(den);
    const double q = num * r;
    const double rem = __builtin_fma(-den, q, num);
    return __builtin_fma(rem, r, q);
}
__device__ __forceinline__ double div_v4(double num, double den) {      // f32 reciprocal seed, one Newton step + correction
    double r = (double)__builtin_amdgcn_rcpf((float)den);
    const double e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q0 = num * r;
    const double rem = __builtin_fma(-den, q0, num);
    return __builtin_fma(rem, r, q0);
}

// EARLY SEED (round 5): an approximate reciprocal of the denominator from the hardware units, computed from x itself while the
// exact expf runs -- v_exp_f32 of x log2(e), 1 + e in float, v_rcp_f32: ~2^-21 relative.  The exact denominator then only
// corrects it: no v_rcp_f64 (20 clocks) on the dependent path.
__device__ __forceinline__ double early_seed(float x) {
    const float ea = __builtin_amdgcn_exp2f(-fabsf(x) * 0x1.715476p+0f);
    return (double)__builtin_amdgcn_rcpf(1.0f + ea);
}
__device__ __forceinline__ double div_seed_newton(double num, double den, double r) {    // seed, one Newton step, quotient, one correction
    const double e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q0 = num * r;
    const double rem = __builtin_fma(-den, q0, num);
    return __builtin_fma(rem, r, q0);
}
__device__ __forceinline__ double div_seed_two_corr(double num, double den, double r) {  // seed, quotient, two residual corrections
    double q = num * r;
    double rem = __builtin_fma(-den, q, num);
    q = __builtin_fma(rem, r, q);
    rem = __builtin_fma(-den, q, num);
    return __builtin_fma(rem, r, q);
}

template <int V, int LOOKUP>
__device__ __forceinline__ float sigmoid_var(float x, const ExpTab& tab, int sel = 0) {
    if (V == 0) return sigmoid_exact<LOOKUP>(x, tab, sel);
    if (V == 7 || V == 8) {
        const double r0 = early_seed(x);
        const float e = expf_noselect<LOOKUP>(-fabsf(x), tab, sel);
        const double ed = (double)e;
        const double den = 1.0 + ed;
        const double num = (x < 0.0f) ? ed : 1.0;
        return (float)(V == 7 ? div_seed_newton(num, den, r0) : div_seed_two_corr(num, den, r0));
    }
    const float e = (V == 6) ? expf_shift<LOOKUP>(-fabsf(x), tab, sel) : expf_noselect<LOOKUP>(-fabsf(x), tab, sel);
    const double ed = (double)e;
    const double den = 1.0 + ed;
    const double num = (x < 0.0f) ? ed : 1.0;
    double q;
    if (V == 1) q = div_unit_range(num, den);
    else if (V == 2 || V == 6) q = div_v1(num, den);
    else if (V == 3) q = div_v2(num, den);
    else if (V == 4) q = div_v3(num, den);
    else q = div_v4(num, den);
    return (float)q;
}

constexpr int kVariants = 9;

__global__ void check_all(unsigned long long* mism, unsigned* first_bad) {
    ExpTab tab;
    tab.init();
    const uint64_t n = 1ull << 32;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)i);
        const bool nan = (x != x);
        const float ref = sigmoid_exact<kLookupPerLane>(nan ? 0.0f : x, tab);
        // also the IEEE divide itself (the reference's `/`)
        {
            const float e = expf_glibc_nonpos<kLookupPerLane>(-fabsf(nan ? 0.0f : x), tab);
            const float numf = (x < 0.0f) ? e : 1.0f;
            const float g = (float)((double)numf / (1.0 + (double)e));
            if (!nan && __float_as_uint(g) != __float_as_uint(ref)) atomicAdd(&mism[0], 1ull);
        }
#define CHK(V)                                                                                     \
        {                                                                                          \
            const float g = sigmoid_var<V, kLookupPerLane>(nan ? 0.0f : x, tab);                   \
            if (!nan && __float_as_uint(g) != __float_as_uint(ref)) {                              \
                if (atomicAdd(&mism[V], 1ull) == 0) first_bad[V] = (unsigned)i;                    \
            }                                                                                      \
        }
        CHK(1) CHK(2) CHK(3) CHK(4) CHK(5) CHK(6) CHK(7) CHK(8)
#undef CHK
    }
}

template <int V>
__global__ void time_chain(float* out, unsigned long long* wall, int iters, float x0) {
    ExpTab tab;
    tab.init();
    const int lane = threadIdx.x & 63;
    float x = x0 + 0.01f * lane;
    const unsigned long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float g = sigmoid_var<V, kLookupLane>(x, tab, k);
            x = __builtin_fmaf(g, 7.5f, -3.7f);          // keeps x in (-3.7, 3.8): both branches taken over time
        }
    }
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *wall = w1 - w0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

int main() {
    unsigned long long* mism; unsigned* first_bad;
    hipMalloc(&mism, 8 * kVariants); hipMalloc(&first_bad, 4 * kVariants);
    hipMemset(mism, 0, 8 * kVariants); hipMemset(first_bad, 0, 4 * kVariants);
    check_all<<<256 * 16, 256>>>(mism, first_bad);
    hipDeviceSynchronize();
    unsigned long long hm[kVariants]; unsigned hb[kVariants];
    hipMemcpy(hm, mism, sizeof(hm), hipMemcpyDeviceToHost); hipMemcpy(hb, first_bad, sizeof(hb), hipMemcpyDeviceToHost);
    const char* names[kVariants] = {"IEEE '/' vs shipped Newton divide", "no underflow select (clamp only), shipped divide",
                                    "+ one Newton step dropped", "+ rcp, two residual corrections", "+ rcp, one residual correction",
                                    "+ f32 rcp seed, one Newton step", "one Newton step + glibc shift-trick k",
                                    "EARLY f32 seed (v_exp_f32, v_rcp_f32 of x) + Newton + corr", "EARLY f32 seed + two residual corrections"};
    for (int v = 0; v < kVariants; ++v)
        printf("variant %d  %-52s mismatches over all 2^32 inputs: %llu  (first bad bits 0x%08x)\n", v, names[v], hm[v], hb[v]);

    float* out; unsigned long long* wall;
    hipMalloc(&out, 256 * 64 * 4); hipMalloc(&wall, 8);
    const int iters = 4096;
#define TIME(V)                                                                                   \
    {                                                                                             \
        for (int rep = 0; rep < 2; ++rep) time_chain<V><<<256, 64>>>(out, wall, iters, 0.3f);      \
        hipDeviceSynchronize();                                                                   \
        unsigned long long w; hipMemcpy(&w, wall, 8, hipMemcpyDeviceToHost);                      \
        printf("variant %d: %.1f ns per dependent sigmoid (+1 fma), one wave per workgroup\n", V, w * 10.0 / iters / 16); \
    }
    TIME(0) TIME(1) TIME(2) TIME(3) TIME(4) TIME(5) TIME(6) TIME(7) TIME(8)
    return 0;
}
