/* ============================================================================================
 * TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C) of the reference's coordinate-ascent
 * E-step kernels, used as the parity oracle by tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py.  The product path (viprs_amd/) never imports, links or calls
 * anything in this directory.
 *
 * Restates viprs/model/vi/e_step.hpp (paths relative to /root/reference) for threads = 1:
 *   e_step :343-442, e_step_mixture :447-551, e_step_grid :555-647, update_q_factor(_matrix)
 *   :266-338, dot :82-104, axpy :157-175, softmax :222-241, sigmoid :245-261.
 * Arithmetic contract: fma exactly where the reference calls std::fma,
 * every other operation rounded individually (build with -ffp-contract=off), the sigmoid divide
 * in double (F6), exp from the host libm as in the reference.
 *
 * Pinning: tests/test_oracle_vs_ref.py checks this restatement bit-for-bit against
 * oracle/_ref/libviprs_ref.so (the reference's own e_step.hpp compiled by oracle/Makefile), and
 * against the committed golden fixtures in tests/golden/ (generated from oracle/_ref by
 * tests/golden/make_golden.py).
 * ============================================================================================ */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ORC_F32 = 0, ORC_F64 = 1 };
enum { ORC_LD_I8 = 0, ORC_LD_I16 = 1, ORC_LD_I32 = 2, ORC_LD_I64 = 3, ORC_LD_F32 = 4, ORC_LD_F64 = 5 };

#define T float
#define EXP_T expf
#define FMA_T fmaf
#define ABS_T fabsf
#define EPS_T FLT_EPSILON
#define U int8_t
#define SUF f32_i8
#include "estep_oracle_impl.h"
#undef U
#undef SUF
#define U int16_t
#define SUF f32_i16
#include "estep_oracle_impl.h"
#undef U
#undef SUF
#define U int32_t
#define SUF f32_i32
#include "estep_oracle_impl.h"
#undef U
#undef SUF
#define U int64_t
#define SUF f32_i64
#include "estep_oracle_impl.h"
#undef U
#undef SUF
#define U float
#define SUF f32_f32
#include "estep_oracle_impl.h"
#undef U
#undef SUF
#define U double
#define SUF f32_f64
#include "estep_oracle_impl.h"
#undef U
#undef SUF
#undef T
#undef EXP_T
#undef FMA_T
#undef ABS_T
#undef EPS_T

#define T double
#define EXP_T exp
#define FMA_T fma
#define ABS_T fabs
#define EPS_T DBL_EPSILON
#define U int8_t
#define SUF f64_i8
#include "estep_oracle_impl.h"
#undef U
#undef SUF
#define U int16_t
#define SUF f64_i16
#include "estep_oracle_impl.h"
#undef U
#undef SUF
#define U int32_t
#define SUF f64_i32
#include "estep_oracle_impl.h"
#undef U
#undef SUF
#define U int64_t
#define SUF f64_i64
#include "estep_oracle_impl.h"
#undef U
#undef SUF
#define U float
#define SUF f64_f32
#include "estep_oracle_impl.h"
#undef U
#undef SUF
#define U double
#define SUF f64_f64
#include "estep_oracle_impl.h"
#undef U
#undef SUF
#undef T
#undef EXP_T
#undef FMA_T
#undef ABS_T
#undef EPS_T

#define CALL(SUFX, TT, UU, WHAT, ...) oracle_##WHAT##_##SUFX(__VA_ARGS__)

#define SWITCH_U(TT, TSUF, BODY)                                                 \
    switch (ucode) {                                                             \
        case ORC_LD_I8:  { typedef int8_t UU;  BODY(TSUF##_i8, TT, UU) } break;  \
        case ORC_LD_I16: { typedef int16_t UU; BODY(TSUF##_i16, TT, UU) } break; \
        case ORC_LD_I32: { typedef int32_t UU; BODY(TSUF##_i32, TT, UU) } break; \
        case ORC_LD_I64: { typedef int64_t UU; BODY(TSUF##_i64, TT, UU) } break; \
        case ORC_LD_F32: { typedef float UU;   BODY(TSUF##_f32, TT, UU) } break; \
        case ORC_LD_F64: { typedef double UU;  BODY(TSUF##_f64, TT, UU) } break; \
        default: return -2;                                                      \
    }

/* indptr is always int64 here: the Python loader widens int32 indptr (pure integer work). */
int oracle_e_step(int tcode, int ucode, int64_t m, const int32_t* lb, const int64_t* ip,
                  const void* ld, const void* std_beta, void* var_gamma, void* var_mu, void* eta,
                  void* q, void* eta_diff, const void* u_logs, const void* shvt, const void* mu_mult,
                  double dq, int low_memory) {
#define BODY(S, TT, UU)                                                                         \
    oracle_e_step_##S(m, lb, ip, (const UU*)ld, (const TT*)std_beta, (TT*)var_gamma, (TT*)var_mu, \
                      (TT*)eta, (TT*)q, (TT*)eta_diff, (const TT*)u_logs, (const TT*)shvt,      \
                      (const TT*)mu_mult, (TT)dq, low_memory);
    if (tcode == ORC_F32) { SWITCH_U(float, f32, BODY) return 0; }
    if (tcode == ORC_F64) { SWITCH_U(double, f64, BODY) return 0; }
#undef BODY
    return -1;
}

int oracle_e_step_mixture(int tcode, int ucode, int64_t m, int K, const int32_t* lb,
                          const int64_t* ip, const void* ld, const void* std_beta, void* var_gamma,
                          void* var_mu, void* eta, void* q, void* eta_diff, const void* log_null_pi,
                          const void* u_logs, const void* shvt, const void* mu_mult, double dq,
                          int low_memory) {
    void* scratch = malloc((size_t)(K + 1) * sizeof(double));
    if (!scratch) return -4;
#define BODY(S, TT, UU)                                                                          \
    oracle_e_step_mixture_##S(m, K, lb, ip, (const UU*)ld, (const TT*)std_beta, (TT*)var_gamma,  \
                              (TT*)var_mu, (TT*)eta, (TT*)q, (TT*)eta_diff,                      \
                              (const TT*)log_null_pi, (const TT*)u_logs, (const TT*)shvt,        \
                              (const TT*)mu_mult, (TT)dq, low_memory, (TT*)scratch);
    int rc = -1;
    if (tcode == ORC_F32) { SWITCH_U(float, f32, BODY) rc = 0; }
    else if (tcode == ORC_F64) { SWITCH_U(double, f64, BODY) rc = 0; }
#undef BODY
    free(scratch);
    return rc;
}

int oracle_e_step_grid(int tcode, int ucode, int64_t m, int n_active, const int32_t* active,
                       const int32_t* lb, const int64_t* ip, const void* ld, const void* std_beta,
                       void* var_gamma, void* var_mu, void* eta, void* q, void* eta_diff,
                       const void* u_logs, const void* hvt, const void* mu_mult, double dq,
                       int low_memory) {
#define BODY(S, TT, UU)                                                                       \
    oracle_e_step_grid_##S(m, n_active, active, lb, ip, (const UU*)ld, (const TT*)std_beta,   \
                           (TT*)var_gamma, (TT*)var_mu, (TT*)eta, (TT*)q, (TT*)eta_diff,      \
                           (const TT*)u_logs, (const TT*)hvt, (const TT*)mu_mult, (TT)dq,     \
                           low_memory);
    if (tcode == ORC_F32) { SWITCH_U(float, f32, BODY) return 0; }
    if (tcode == ORC_F64) { SWITCH_U(double, f64, BODY) return 0; }
#undef BODY
    return -1;
}

/* Double-precision model of the host libm's expf (glibc 2.35, sysdeps/ieee754/flt-32/e_expf.c,
 * FMA ifunc variant): the same sequence of IEEE double operations the HIP kernels execute on
 * the device (viprs_amd/csrc/expf_glibc.h).  tests/test_oracle_vs_ref.py sweeps it against
 * expf() here; the device copy is checked against this one on the GPU. */
static const uint64_t EXP2F_TAB[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51,
    0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1,
    0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
    0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585,
    0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
    0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
    0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069,
    0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540};

float oracle_expf_model(float x) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32;
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32;
    const double C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32;
    const double C2 = 0x1.62e42ff0c52d6p-1 / 32;
    if (x < -104.0f) return 0.0f;
    double xd = (double)x;
    double z = InvLn2N * xd;
    double kd = z + 0x1.8p52;
    uint64_t ki;
    memcpy(&ki, &kd, 8);
    kd -= 0x1.8p52;
    double r = fma(InvLn2N, xd, -kd);
    uint64_t t = EXP2F_TAB[ki % 32] + (ki << 47);
    double s;
    memcpy(&s, &t, 8);
    double zz = fma(C0, r, C1);
    double r2 = r * r;
    double y = fma(C2, r, 1.0);
    y = fma(zz, r2, y);
    y = y * s;
    return (float)y;
}

/* Model of the host libm's DOUBLE exp for x <= 0 (glibc 2.35, sysdeps/ieee754/dbl-64/e_exp.c, the FMA ifunc variant; the
 * operation sequence and the constants are in exp_glibc_f64_tab.h, generated from this host's libm by
 * tools/extract_glibc_exp.py): what the device executes for a float64 state (viprs_amd/csrc/device_math.h).
 * tests/test_oracle_vs_ref.py compares it with exp() here, bit for bit. */
#include "exp_glibc_f64_tab.h"
static const uint64_t kExp64Tab[256] = VIPRS_EXP64_TAB_INIT;
static inline uint64_t d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
double oracle_exp_model(double x) {
    const uint64_t ax = d2u(x) & 0x7fffffffffffffffull;
    const unsigned abstop = (unsigned)(ax >> 52);
    if (abstop - 0x3c9u > 0x3eu) {                       /* |x| < 2^-54, |x| >= 512, inf, nan */
        if (abstop < 0x3c9u) return 1.0 + x;
        if (abstop >= 0x409u) {                          /* |x| >= 1024 */
            if (d2u(x) == 0xfff0000000000000ull) return 0.0;
            if (abstop >= 0x7ffu) return 1.0 + x;        /* nan (+inf does not occur: x <= 0) */
            return 0.0;                                   /* __math_uflow: x <= -1024 */
        }
    }
    const double kds = fma(x, VIPRS_EXP64_INVLN2N, VIPRS_EXP64_SHIFT);
    const uint64_t ki = d2u(kds);
    const double kd = kds - VIPRS_EXP64_SHIFT;
    double r = fma(kd, VIPRS_EXP64_NEGLN2HIN, x);
    r = fma(kd, VIPRS_EXP64_NEGLN2LON, r);
    const unsigned idx = 2u * (unsigned)(ki & 127u);
    const double tail = u2d(kExp64Tab[idx]);
    uint64_t sbits = kExp64Tab[idx + 1] + (ki << 45);
    const double p23 = fma(r, VIPRS_EXP64_C3, VIPRS_EXP64_C2);
    const double t0 = r + tail;
    const double r2 = r * r;
    const double p45 = fma(r, VIPRS_EXP64_C5, VIPRS_EXP64_C4);
    double tmp = fma(p23, r2, t0);
    tmp = fma(r2 * r2, p45, tmp);
    if (abstop < 0x408u) {                               /* |x| < 512: the scale is a normal number */
        const double scale = u2d(sbits);
        return fma(scale, tmp, scale);
    }
    /* specialcase(), k < 0: x in (-1024, -512] -- the result may be subnormal */
    sbits += 1022ull << 52;
    const double scale = u2d(sbits);
    const double st = scale * tmp;
    double y = scale + st;
    if (y < 1.0) {
        double lo = scale - y + st;
        const double hi = 1.0 + y;
        lo = 1.0 - hi + y + lo;
        y = (hi + lo) - 1.0;
        if (y == 0.0) y = 0.0;                           /* no -0 */
    }
    return 0x1p-1022 * y;
}
/* bit mismatches against exp() over n doubles x_i = -(lo + i * step) and over their neighbours in the last place */
int64_t oracle_exp_model_mismatches(double lo, double step, int64_t n) {
    int64_t bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const double x0 = -(lo + (double)i * step);
        for (int d = -1; d <= 1; ++d) {
            const double x = u2d(d2u(x0) + (uint64_t)(int64_t)d);
            const double a = exp(x), b = oracle_exp_model(x);
            if (d2u(a) != d2u(b) && !(a != a && b != b)) ++bad;
        }
    }
    return bad;
}

/* Sweep helper: counts bit mismatches of oracle_expf_model vs libm expf over the float bit
 * patterns [lo_bits, hi_bits] (inclusive) with the given stride. */
int64_t oracle_expf_model_mismatches(uint32_t lo_bits, uint32_t hi_bits, uint32_t stride) {
    int64_t mm = 0;
    for (uint64_t u = lo_bits; u <= hi_bits; u += stride) {
        uint32_t uu = (uint32_t)u;
        float x, a, b;
        memcpy(&x, &uu, 4);
        a = expf(x);
        b = oracle_expf_model(x);
        if (memcmp(&a, &b, 4)) mm++;
    }
    return mm;
}
