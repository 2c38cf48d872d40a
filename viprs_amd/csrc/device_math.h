// Device-side scalar math of the per-SNP posterior update, written so that every IEEE operation
// matches the reference's C++ (e_step.hpp) operation for operation.  Build with
// -ffp-contract=off: fma is used exactly where the reference calls std::fma and nowhere else.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "exp_glibc_f64_tab.h"

namespace viprs {

// ---------------------------------------------------------------------------------------------
// expf, bit-identical to the host libm the reference links against (glibc 2.35,
// sysdeps/ieee754/flt-32/e_expf.c, the FMA ifunc variant selected on every x86-64 CPU with FMA).
// glibc evaluates expf in double: k = round(x * 32/ln2), r = x*32/ln2 - k (contracted to one
// fma), 2^(k/32) from a 32-entry table, a cubic in r, one final rounding to float.  Those are
// exactly-specified IEEE double operations, so executing the same sequence on the device yields
// the same bits.  oracle/estep_oracle.c holds the same model; tests sweep it against expf() over
// all 2.2e9 floats in [-104, 88.7] (0 mismatches) and the device copy against it on the GPU.
// ---------------------------------------------------------------------------------------------
__device__ __constant__ const uint64_t kExp2fTab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

// The table spread over the lanes of a wave (lane l holds entry l & 31), so a lookup is two
// v_readlane (wave-uniform index) or two ds_bpermute (per-lane index) instead of a memory load
// on the serial chain.
struct ExpTab {
    int lo, hi;
    __device__ __forceinline__ void init() {
        const uint64_t t = kExp2fTab[threadIdx.x & 31];
        lo = (int)(uint32_t)t;
        hi = (int)(uint32_t)(t >> 32);
    }
};

// Polynomial + scaling shared by both lookup flavours.  `t` = table bits + (k << 47).
__device__ __forceinline__ float expf_glibc_finish(double r, uint64_t t) {
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32;
    const double C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32;
    const double C2 = 0x1.62e42ff0c52d6p-1 / 32;
    const double s = __longlong_as_double((long long)t);
    const double zz = fma(C0, r, C1);
    const double r2 = r * r;
    double y = fma(C2, r, 1.0);
    y = fma(zz, r2, y);
    y = y * s;
    return (float)y;
}

// Table lookup flavours.
//   kLookupPerLane : every lane looks up its own index (two ds_bpermute; all 64 lanes active).
//   kLookupLane    : only lane `sel` (wave-uniform) needs a correct result -- the serial chain,
//                    where lane j carries SNP j and the other lanes compute don't-care values.
//                    The index is read from lane `sel` and the entry fetched with two v_readlane,
//                    which keeps the LDS crossbar off the chain.
enum { kLookupPerLane = 0, kLookupLane = 1 };

// x <= 0 is all the E-step needs (sigmoid / softmax arguments are -|x| or u - max(u)).
// glibc returns 0 (__math_uflowf) below -0x1.9fe368p6 = -103.97; here the argument is clamped to -104, where the
// double result 2^-150.04 rounds to 0.0f by itself (it is below half the smallest denormal) -- no select, and no
// out-of-range table index either.  tools/ubench/sigmoid_variants.hip holds the exhaustive bit-check of this and of
// the divide below over all 2^32 inputs of the sigmoid.
template <int LOOKUP>
__device__ __forceinline__ float expf_glibc_nonpos(float x, const ExpTab& tab, int sel = 0) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32;
    // (a select, not fmaxf: v_max_f32 would turn a NaN argument into -104 and the sigmoid of a NaN into 1.0 -- a NaN in
    // the summary statistics has to come out as NaN, as it does in the reference)
    const double xd = (double)((x < -104.0f) ? -104.0f : x);
    const double z = InvLn2N * xd;
    const double kd = rint(z);                 // == (z + 0x1.8p52) - 0x1.8p52 in round-to-nearest
    const double r = fma(InvLn2N, xd, -kd);
    int ki = (int)kd;                          // |kd| < 2^13
    int tlo, thi;
    if (LOOKUP == kLookupLane) {
        // only lane `sel` matters: its k is read once (v_readlane uses the low 6 bits of the lane select, and lanes
        // l and l + 32 hold the same table entry: no mask), the table words and the exponent shift are scalar
        asm volatile("" : "+v"(ki));           // (a per-lane convert and ONE v_readlane: hipcc would read kd's two words)
        const int sk = __builtin_amdgcn_readlane(ki, sel);
        tlo = __builtin_amdgcn_readlane(tab.lo, sk);
        thi = __builtin_amdgcn_readlane(tab.hi, sk) + (int)((unsigned)sk << 15);
    } else {
        tlo = __builtin_amdgcn_ds_bpermute(ki << 2, tab.lo);      // (ds_bpermute takes address bits 7:2)
        thi = __builtin_amdgcn_ds_bpermute(ki << 2, tab.hi) + (int)((unsigned)ki << 15);
    }
    // t = tab + (ki << 47): only the high word changes (ki << 15), two's complement wraps as
    // the 64-bit add in glibc does (tab low word is untouched: 47 >= 32).
    const uint64_t t = ((uint64_t)(uint32_t)thi << 32) | (uint32_t)tlo;
    return expf_glibc_finish(r, t);
}

// ---------------------------------------------------------------------------------------------
// sigmoid<T> (e_step.hpp:245-261).  With T = float the literal `1.` there is a double, so the add
// and the divide happen in double and the quotient is rounded to float once.
// ---------------------------------------------------------------------------------------------
// num / den for den in [1, 2], 0 <= num <= 1, rounded to float by the caller: v_rcp_f64 (2^-23 relative), ONE
// Newton step (2^-46), the quotient and one residual correction -- the double quotient is the IEEE one except in
// ties that never decide the float rounding: over all 2^32 arguments of the sigmoid the float result equals
// (float)(num / den) bit for bit (0 mismatches; with the raw reciprocal and one correction: 1 025).  No
// v_div_scale / v_div_fmas / v_div_fixup: they only rescale operands near the exponent limits and patch
// inf / nan / 0, none of which can occur for these ranges.
__device__ __forceinline__ double div_unit_range(double num, double den) {
    double r = __builtin_amdgcn_rcp(den);
    const double e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q0 = num * r;
    const double rem = __builtin_fma(-den, q0, num);
    return __builtin_fma(rem, r, q0);
}

template <int LOOKUP>
__device__ __forceinline__ float sigmoid_exact(float x, const ExpTab& tab, int sel = 0) {
    const float e = expf_glibc_nonpos<LOOKUP>(-fabsf(x), tab, sel);
    const double ed = (double)e;
    const double num = (x < 0.0f) ? ed : 1.0;
    return (float)div_unit_range(num, 1.0 + ed);
}

// Hardware-transcendental variants (math_mode = fast, VIPRS_MATH_FAST): exp(x) for x <= 0 as v_exp_f32 of a
// COMPENSATED x * log2(e) product -- p = fl(t * L_hi) goes to v_exp_f32 at once, the product's rounding error and the
// low part of log2(e) come back in as a first-order correction e0 * (1 + ln2 * perr), whose factor is computed while
// the transcendental unit works: TWO dependent instructions besides v_exp_f32 itself (the product and one fma).
// Error: v_exp_f32's 1 ulp + 2^-23 -- a few 1e-7 relative for every argument (an uncompensated product is off by
// |x| * 1e-7, 1e-5 at x = -100), against 1e-5 allowed.  Arguments below -126 / log2(e) give denormal or zero
// results like expf's (v_exp_f32 flushes at 2^-126: below the resolution of anything the E-step adds them to).
__device__ __forceinline__ float expf_fast_nonpos(float t) {
    const float L_hi = 0x1.715476p+0f;         // log2(e) rounded to float
    const float L_lo = 0x1.4ae0cp-26f;         // log2(e) - L_hi
    const float p = t * L_hi;
    const float perr = __builtin_fmaf(t, L_lo, __builtin_fmaf(t, L_hi, -p));
    const float k = perr * 0x1.62e43p-1f;      // ln2 * perr (off the critical path)
    const float e0 = __builtin_amdgcn_exp2f(p);
    return __builtin_fmaf(e0, k, e0);
}

// sigmoid (e_step.hpp:245-261) on the hardware units: v_exp_f32 + v_rcp_f32 (1 ulp each), fp32 throughout -- a few
// ulp from sigmoid_exact (which follows the reference's double add and divide), inside the 1e-5 tolerance.
__device__ __forceinline__ float sigmoid_fast(float x) {
    const float e = expf_fast_nonpos(-fabsf(x));
    const float num = (x < 0.0f) ? e : 1.0f;
    return num * __builtin_amdgcn_rcpf(1.0f + e);
}

// ---------------------------------------------------------------------------------------------
// double state (float_precision='float64'): exp for x <= 0, bit-identical to the host libm's (glibc 2.35,
// sysdeps/ieee754/dbl-64/e_exp.c, the FMA ifunc variant): exp(x) = 2^(k/128) exp(r) with 2^(k/128) = scale (1 + tail) from
// a 2 x 128-entry table and a degree-5 polynomial.  The operation sequence (which products are fused is part of the
// contract) and the constants -- read from the library itself by tools/extract_glibc_exp.py -- are in
// exp_glibc_f64_tab.h; oracle/estep_oracle.c holds the same model and tests/test_oracle_vs_ref.py compares it with the
// host's exp() over 27 million arguments, subnormal results included (0 mismatches).
//   exp_glibc_f64_core  : everything after the table lookup (shared by the per-lane and the wave-uniform lookups)
//   exp_glibc_f64_nonpos: per-lane arguments, table in constant memory (row-by-row kernels of estep_generic.h)
// (estep_tile.h holds the wave-uniform flavour of the chain: table across the lanes, two v_readlane per word)
// ---------------------------------------------------------------------------------------------
__device__ __constant__ const uint64_t kExp64Tab[256] = VIPRS_EXP64_TAB_INIT;

struct Exp64Reduced { double r; long long ki; unsigned abstop; };
__device__ __forceinline__ Exp64Reduced exp_glibc_f64_reduce(double x) {
    Exp64Reduced o;
    o.abstop = (unsigned)((unsigned long long)(__double_as_longlong(x) & 0x7fffffffffffffffll) >> 52);
    const double kds = __builtin_fma(x, VIPRS_EXP64_INVLN2N, VIPRS_EXP64_SHIFT);
    o.ki = __double_as_longlong(kds);
    const double kd = kds - VIPRS_EXP64_SHIFT;
    double r = __builtin_fma(kd, VIPRS_EXP64_NEGLN2HIN, x);
    o.r = __builtin_fma(kd, VIPRS_EXP64_NEGLN2LON, r);
    return o;
}
// `tail`, `sbits0` = T[2 (ki % 128)], T[2 (ki % 128) + 1].  The common range 2^-54 <= |x| < 512 runs straight through;
// everything else (UNIFORM = the argument is wave-uniform: one scalar branch; otherwise selects) is patched afterwards.
template <bool UNIFORM>
__device__ __forceinline__ double exp_glibc_f64_core(double x, const Exp64Reduced& q, double tail, unsigned long long sbits0) {
    const double r = q.r;
    const unsigned long long sbits = sbits0 + ((unsigned long long)q.ki << 45);
    const double p23 = __builtin_fma(r, VIPRS_EXP64_C3, VIPRS_EXP64_C2);
    const double t0 = r + tail;
    const double r2 = r * r;
    const double p45 = __builtin_fma(r, VIPRS_EXP64_C5, VIPRS_EXP64_C4);
    double tmp = __builtin_fma(p23, r2, t0);
    tmp = __builtin_fma(r2 * r2, p45, tmp);
    const double scale = __longlong_as_double((long long)sbits);
    double y = __builtin_fma(scale, tmp, scale);          // |x| in [2^-54, 512): the scale is a normal number
    auto special = [&]() {
        double z;
        if (q.abstop < 0x3c9u) {
            z = 1.0 + x;                                  // |x| < 2^-54
        } else if (q.abstop >= 0x409u) {
            z = (x != x) ? 1.0 + x : 0.0;                 // nan; x = -inf, x <= -1024 (underflow)
        } else {
            // specialcase(), k < 0: x in (-1024, -512], subnormal results
            const double sc = __longlong_as_double((long long)(sbits + (1022ull << 52)));
            const double st = sc * tmp;
            z = sc + st;
            if (z < 1.0) {
                double lo = sc - z + st;
                const double hi = 1.0 + z;
                lo = 1.0 - hi + z + lo;
                z = (hi + lo) - 1.0;
                if (z == 0.0) z = 0.0;
            }
            z = 0x1p-1022 * z;
        }
        return z;
    };
    const bool rare = q.abstop - 0x3c9u > 0x3eu;          // |x| < 2^-54 or |x| >= 512 (inf, nan)
    if (UNIFORM) {
        if (__builtin_expect(__builtin_amdgcn_readfirstlane((int)rare) != 0, 0)) y = special();
    } else if (__builtin_amdgcn_ballot_w64(rare) != 0) {
        const double z = special();
        y = rare ? z : y;
    }
    return y;
}
// num / den for den in [1, 2], 0 <= num <= 1 as an IEEE divide: the reciprocal refined to the last place by two Newton
// steps, the quotient, its exact remainder and ONE correction whose rounding is the quotient's (Markstein) -- the sequence
// the compiler emits for `/`, without v_div_scale / v_div_fmas / v_div_fixup: those only rescale operands near the
// exponent limits and patch inf / nan / 0, none of which occurs here.  tools/ubench/div_f64_check.hip compares it with `/` on
// 2^32 (num, 1 + e) pairs over every binade of e, subnormals included (0 mismatches); tests/test_gpu_float64.py holds the `==`
// tests against the host.
__device__ __forceinline__ double div_unit_range_f64(double num, double den) {
    double r = __builtin_amdgcn_rcp(den);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    const double q0 = num * r;
    const double rem = __builtin_fma(-den, q0, num);
    return __builtin_fma(rem, r, q0);
}
__device__ __forceinline__ double exp_glibc_f64_nonpos(double x) {
    const Exp64Reduced q = exp_glibc_f64_reduce(x);
    const unsigned idx = 2u * ((unsigned)q.ki & 127u);
    return exp_glibc_f64_core<false>(x, q, __longlong_as_double((long long)kExp64Tab[idx]), kExp64Tab[idx + 1]);
}

// sigmoid<double> (e_step.hpp:245-261): glibc's exp, IEEE add and divide -- bit-identical to the reference on the host.
__device__ __forceinline__ double sigmoid_f64(double x) {
    const double e = exp_glibc_f64_nonpos(-fabs(x));
    const double num = (x < 0.0) ? e : 1.0;
    return num / (1.0 + e);
}

template <typename T> struct Eps;
template <> struct Eps<float> { static constexpr float value = 1.1920928955078125e-07f; };   // max(FLT_EPSILON, 1e-8f)
template <> struct Eps<double> { static constexpr double value = 1e-8; };                     // max(DBL_EPSILON, 1e-8)

template <typename T> __device__ __forceinline__ T fma_t(T a, T b, T c);
template <> __device__ __forceinline__ float fma_t<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <> __device__ __forceinline__ double fma_t<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename T> __device__ __forceinline__ T abs_t(T a);
template <> __device__ __forceinline__ float abs_t<float>(float a) { return fabsf(a); }
template <> __device__ __forceinline__ double abs_t<double>(double a) { return fabs(a); }

// One spike-and-slab posterior update (e_step.hpp:401-408) from the current q_j.
//   mu = fma(mm, beta, -(mm*q)); u = s*mu; gamma = sigmoid(fma(u,u,ulog)); d = fma(gamma, mu, -eta)
template <bool EXACT, int LOOKUP>
__device__ __forceinline__ void snp_update(float mm, float beta, float s, float ulog, float eta_old,
                                           float qj, const ExpTab& tab, float& mu, float& gamma,
                                           float& d, int sel = 0) {
    const float p = mm * qj;
    mu = __builtin_fmaf(mm, beta, -p);
    const float u = s * mu;
    const float x = __builtin_fmaf(u, u, ulog);
    gamma = EXACT ? sigmoid_exact<LOOKUP>(x, tab, sel) : sigmoid_fast(x);
    d = __builtin_fmaf(gamma, mu, -eta_old);
}

template <bool EXACT, int LOOKUP>
__device__ __forceinline__ void snp_update(double mm, double beta, double s, double ulog,
                                           double eta_old, double qj, const ExpTab&, double& mu,
                                           double& gamma, double& d, int = 0) {
    const double p = mm * qj;
    mu = __builtin_fma(mm, beta, -p);
    const double u = s * mu;
    const double x = __builtin_fma(u, u, ulog);
    gamma = sigmoid_f64(x);
    d = __builtin_fma(gamma, mu, -eta_old);
}

}  // namespace viprs
