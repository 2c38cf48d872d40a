// Development tool: ns per serial SNP update of candidate formulations of the panel kernel's chain step
// (estep_panel.h), one wave per workgroup, and a bit-check of every candidate against the shipped step.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -I viprs_amd/csrc tools/ubench/chain_step.hip -o build/ubench/chain_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>

#include "device_math.h"

using namespace viprs;

constexpr int kP = 64;
constexpr float kEps = 1.1920928955078125e-07f;

__device__ __forceinline__ float rlf(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// r = mask[lane] ? b : a, mask in an SGPR pair
__device__ __forceinline__ float sel(float a, float b, unsigned long long m) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
    return r;
}

// ---- the new sigmoid: clamp instead of the underflow select, one Newton step + residual correction -----
__device__ __forceinline__ float expf_lane(float x, const ExpTab& tab, int sel_lane) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32;
    const double xd = (double)fmaxf(x, -104.0f);
    const double z = InvLn2N * xd;
    const double kd = rint(z);
    const double r = fma(InvLn2N, xd, -kd);
    int ki = (int)kd;
    asm volatile("" : "+v"(ki));                              // (per-lane convert, then ONE readlane: hipcc would read kd's two words)
    const int sidx = __builtin_amdgcn_readlane(ki, sel_lane);
    int tlo = __builtin_amdgcn_readlane(tab.lo, sidx);
    int thi = __builtin_amdgcn_readlane(tab.hi, sidx);
    thi += (int)((unsigned)sidx << 15);                       // scalar: the table word and the shift of lane `sel_lane`
    const uint64_t t = ((uint64_t)(uint32_t)thi << 32) | (uint32_t)tlo;
    return expf_glibc_finish(r, t);
}
__device__ __forceinline__ float expf_lane_shift(float x, const ExpTab& tab, int sel_lane) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32;
    const double Shift = 0x1.8p52;
    const double xd = (double)fmaxf(x, -104.0f);
    const double z = InvLn2N * xd;
    double kd = z + Shift;
    int ki = (int)(uint32_t)(uint64_t)__double_as_longlong(kd);
    kd = kd - Shift;
    const double r = fma(InvLn2N, xd, -kd);
    asm volatile("" : "+v"(ki));
    const int sidx = __builtin_amdgcn_readlane(ki, sel_lane);
    int tlo = __builtin_amdgcn_readlane(tab.lo, sidx);
    int thi = __builtin_amdgcn_readlane(tab.hi, sidx);
    thi += (int)((unsigned)sidx << 15);
    const uint64_t t = ((uint64_t)(uint32_t)thi << 32) | (uint32_t)tlo;
    return expf_glibc_finish(r, t);
}
template <bool SHIFT = false>
__device__ __forceinline__ float sigmoid_new(float x, const ExpTab& tab, int sel_lane) {
    const float e = SHIFT ? expf_lane_shift(-fabsf(x), tab, sel_lane) : expf_lane(-fabsf(x), tab, sel_lane);
    const double ed = (double)e;
    const double den = 1.0 + ed;
    const double num = (x < 0.0f) ? ed : 1.0;
    double r = __builtin_amdgcn_rcp(den);
    const double e1 = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e1, r);
    const double q0 = num * r;
    const double rem = __builtin_fma(-den, q0, num);
    return (float)__builtin_fma(rem, r, q0);
}

struct In { float mm, beta, sv, ulog, eta_old; };

// V = 0 shipped step (global prefetch of the diagonal rows, VALU skip test, lane-compare selects)
// V = 1 new: sigmoid_new, scalar skip test, next step fed from the un-selected fma result, SGPR lane mask, rows from LDS
// V = 2 as 1 with the shipped global prefetch
// V = 3 as 1 with the VALU skip test
// V = 5 as 3 with glibc's shift trick for the table index and no multiply by dq (valid for dq == 1 only)
// V = 4 as 1 without the own-lane bookkeeping (q capture, diagonal subtraction): lower bound, NOT the same result
template <int V>
__global__ __launch_bounds__(64) void chain(const float* __restrict__ tile_g, const In* __restrict__ ins, float* __restrict__ out,
                                            unsigned long long* wall, int panels, float dq) {
    __shared__ float T[kP * kP];
    const int lane = threadIdx.x;
    ExpTab tab;
    tab.init();
    for (int i = lane; i < kP * kP; i += 64) T[i] = tile_g[i];
    __syncthreads();
    In in = ins[lane];
    float qc = 0.001f * lane, acc_a = 0.0f, acc_q = 0.0f;
    const float* __restrict__ dptr = tile_g + (size_t)blockIdx.x * 0 + lane;
    float dnext[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) dnext[k] = dptr[k * kP];
    const unsigned long long w0 = wall_clock64();
    for (int p = 0; p < panels; ++p) {
        float qcap = 0.0f, avec = 0.0f;
        if (V == 0 || V == 2) {
            float drow[kP];
#pragma unroll
            for (int k = 0; k < 16; ++k) drow[k] = dnext[k];
            if (V == 0) {
#pragma unroll
                for (int jj = 0; jj < kP; ++jj) {
                    if (jj + 16 < kP) drow[jj + 16] = dptr[(jj + 16) * kP];
                    else dnext[jj + 16 - kP] = dptr[(jj + 16 - kP) * kP];
                    float mu, gamma, d;
                    snp_update<true, kLookupLane>(in.mm, in.beta, in.sv, in.ulog, in.eta_old, qc, tab, mu, gamma, d, jj);
                    const bool upd = !(fabsf(d) < kEps);
                    const float a_lane = upd ? dq * d : 0.0f;
                    int l = lane;
                    asm volatile("" : "+v"(l));
                    const bool me = (l == jj);
                    qcap = me ? qc : qcap;
                    const float a = rlf(a_lane, jj);
                    avec = me ? a : avec;
                    qc = __builtin_fmaf(drow[jj], a, qc);
                    qc = (me && upd) ? qc - d : qc;
                }
            } else {
                unsigned long long mask = 1ull;
                float qf = qc;
#pragma unroll
                for (int jj = 0; jj < kP; ++jj) {
                    if (jj + 16 < kP) drow[jj + 16] = dptr[(jj + 16) * kP];
                    else dnext[jj + 16 - kP] = dptr[(jj + 16 - kP) * kP];
                    const float p_ = in.mm * qf;
                    const float mu = __builtin_fmaf(in.mm, in.beta, -p_);
                    const float u = in.sv * mu;
                    const float x = __builtin_fmaf(u, u, in.ulog);
                    const float gamma = sigmoid_new(x, tab, jj);
                    const float d = __builtin_fmaf(gamma, mu, -in.eta_old);
                    const int sd = __builtin_amdgcn_readlane(__float_as_int(d), jj);
                    const bool upd = (unsigned)(sd & 0x7fffffff) >= 0x34000000u;
                    const float sdz = __int_as_float(upd ? sd : 0);
                    const float sa = dq * sdz;
                    qcap = sel(qcap, qc, mask);
                    avec = sel(avec, sa, mask);
                    qf = __builtin_fmaf(drow[jj], sa, qc);
                    qc = sel(qf, qf - sdz, mask);
                    asm volatile("s_lshl_b64 %0, %0, 1" : "+s"(mask) : : "scc");
                }
            }
        } else {
            unsigned long long mask = 1ull;
            float qf = qc;
#pragma unroll
            for (int jj = 0; jj < kP; ++jj) {
                const float drow = T[jj * kP + lane];
                const float p_ = in.mm * qf;
                const float mu = __builtin_fmaf(in.mm, in.beta, -p_);
                const float u = in.sv * mu;
                const float x = __builtin_fmaf(u, u, in.ulog);
                const float gamma = (V == 5) ? sigmoid_new<true>(x, tab, jj) : sigmoid_new<false>(x, tab, jj);
                const float d = __builtin_fmaf(gamma, mu, -in.eta_old);
                float sa, sdz;
                if (V == 3 || V == 5) {
                    const bool upd = !(fabsf(d) < kEps);
                    const float dz = upd ? d : 0.0f;
                    sdz = rlf(dz, jj);
                    sa = (V == 5) ? sdz : dq * sdz;          // V = 5: dq == 1 (fp32 LD), the multiply by 1 is dropped
                } else {
                    const int sd = __builtin_amdgcn_readlane(__float_as_int(d), jj);
                    const bool upd = (unsigned)(sd & 0x7fffffff) >= 0x34000000u;
                    sdz = __int_as_float(upd ? sd : 0);
                    sa = dq * sdz;
                }
                if (V != 4) {
                    qcap = sel(qcap, qc, mask);
                    avec = sel(avec, sa, mask);
                }
                qf = __builtin_fmaf(drow, sa, qc);
                if (V != 4) {
                    qc = sel(qf, qf - sdz, mask);
                    asm volatile("s_lshl_b64 %0, %0, 1" : "+s"(mask) : : "scc");
                } else {
                    qc = qf;
                }
            }
        }
        acc_a += avec;
        acc_q += qcap;
        // keep the values bounded and the panels dependent
        qc = qc * 0.5f + 0.001f * lane;
        in.eta_old = in.eta_old * 0.5f;
    }
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *wall = w1 - w0;
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += dnext[k];
    out[((size_t)blockIdx.x * 64 + lane) * 4 + 0] = qc;
    out[((size_t)blockIdx.x * 64 + lane) * 4 + 1] = acc_a;
    out[((size_t)blockIdx.x * 64 + lane) * 4 + 2] = acc_q;
    out[((size_t)blockIdx.x * 64 + lane) * 4 + 3] = s;
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 256;
    const int panels = 512;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    std::vector<float> tile(kP * kP);
    for (int r = 0; r < kP; ++r)
        for (int c = 0; c < kP; ++c) tile[r * kP + c] = r == c ? 1.0f : powf(0.6f, (float)abs(r - c));
    std::vector<In> ins(64);
    for (auto& i : ins) i = In{0.9f + 0.01f * nd(rng), 0.004f * nd(rng), 250.0f + nd(rng), -5.2f + 0.05f * nd(rng), 1e-4f * nd(rng)};
    float *d_tile, *d_out; In* d_ins; unsigned long long* d_wall;
    hipMalloc(&d_tile, tile.size() * 4); hipMalloc(&d_ins, sizeof(In) * 64); hipMalloc(&d_out, (size_t)grid * 64 * 16); hipMalloc(&d_wall, 8);
    hipMemcpy(d_tile, tile.data(), tile.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_ins, ins.data(), sizeof(In) * 64, hipMemcpyHostToDevice);
    std::vector<float> ref;
    for (float dq : {1.0f, 1.0f / 127.0f}) {
#define RUN(V)                                                                                                      \
        {                                                                                                           \
            for (int rep = 0; rep < 2; ++rep) chain<V><<<grid, 64>>>(d_tile, d_ins, d_out, d_wall, panels, dq);      \
            hipDeviceSynchronize();                                                                                 \
            unsigned long long w; hipMemcpy(&w, d_wall, 8, hipMemcpyDeviceToHost);                                  \
            std::vector<float> o(64 * 4); hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost);           \
            if (V == 0) ref = o;                                                                                    \
            int bad = 0;                                                                                            \
            for (int i = 0; i < 64; ++i) for (int k = 0; k < 3; ++k) bad += memcmp(&o[i * 4 + k], &ref[i * 4 + k], 4) != 0; \
            printf("dq=%.4f variant %d: %.1f ns per chain step   (bitwise differences from variant 0: %d of 192)\n", dq, V, \
                   w * 10.0 / panels / kP, bad);                                                                    \
        }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    }
    return 0;
}
