// TEST INFRASTRUCTURE ONLY -- never linked, imported or called by the product path.
//
// C-ABI wrapper that instantiates the *reference's own* header-only E-step templates
// (viprs/model/vi/e_step.hpp, compiled from where it lies under /root/reference via -I;
// no reference source is copied into this repository) for every (T, U, I) combination that
// the reference's Cython boundary exposes (viprs/model/vi/e_step_cpp.pxd:7-17: indptr_type
// {int32,int64} x noncomplex_numeric {int8,int16,int32,int64,float32,float64} x floating
// {float,double}).  The entry points mirror e_step_cpp.pyx:91-195 with dtype codes in place
// of Cython fused types.
//
// Built by oracle/Makefile into oracle/_ref/libviprs_ref.so with the reference's flags
// (setup.py:202-221: -O3 -std=c++17 -fopenmp, no -march, HAVE_CBLAS undefined).
#include <cstdint>
#include <vector>
#include "e_step.hpp"   // found through -I/root/reference/viprs/model/vi

namespace {

enum { F32 = 0, F64 = 1 };
enum { LD_I8 = 0, LD_I16 = 1, LD_I32 = 2, LD_I64 = 3, LD_F32 = 4, LD_F64 = 5 };
enum { IP_I32 = 0, IP_I64 = 1 };

template <typename T, typename U, typename I>
void run_e_step(int m, int* lb, void* ip, void* ld, void* std_beta, void* var_gamma, void* var_mu,
                void* eta, void* q, void* eta_diff, void* u_logs, void* shvt, void* mu_mult,
                double dq_scale, int threads, int low_memory) {
    e_step<T, U, I>(m, lb, (I*)ip, (U*)ld, (T*)std_beta, (T*)var_gamma, (T*)var_mu, (T*)eta, (T*)q,
                    (T*)eta_diff, (T*)u_logs, (T*)shvt, (T*)mu_mult, (T)dq_scale, threads,
                    low_memory != 0);
}

template <typename T, typename U, typename I>
void run_e_step_mixture(int m, int K, int* lb, void* ip, void* ld, void* std_beta, void* var_gamma,
                        void* var_mu, void* eta, void* q, void* eta_diff, void* log_null_pi,
                        void* u_logs, void* shvt, void* mu_mult, double dq_scale, int threads,
                        int low_memory) {
    e_step_mixture<T, U, I>(m, K, lb, (I*)ip, (U*)ld, (T*)std_beta, (T*)var_gamma, (T*)var_mu,
                            (T*)eta, (T*)q, (T*)eta_diff, (T*)log_null_pi, (T*)u_logs, (T*)shvt,
                            (T*)mu_mult, (T)dq_scale, threads, low_memory != 0);
}

template <typename T, typename U, typename I>
void run_e_step_grid(int m, int n_active, int* active, int* lb, void* ip, void* ld, void* std_beta,
                     void* var_gamma, void* var_mu, void* eta, void* q, void* eta_diff,
                     void* u_logs, void* hvt, void* mu_mult, double dq_scale, int threads,
                     int low_memory) {
    e_step_grid<T, U, I>(m, n_active, active, lb, (I*)ip, (U*)ld, (T*)std_beta, (T*)var_gamma,
                         (T*)var_mu, (T*)eta, (T*)q, (T*)eta_diff, (T*)u_logs, (T*)hvt,
                         (T*)mu_mult, (T)dq_scale, threads, low_memory != 0);
}

// The reference's coarse, EXACT parallelism (joblib over chromosomes, bin/viprs_fit:1080-1086) at LD-block grain: one
// e_step<T,U,I>(threads = 1) call per LD block, blocks handed to the OpenMP threads dynamically in the caller's order
// (largest first).  Blocks share no q entries, so the result equals ONE threads = 1 call over all SNPs bit for bit.
// bench.py's CPU-baseline variant 3.
template <typename T, typename U, typename I>
void run_e_step_blocks(int n_blocks, const int64_t* block_start, const int32_t* order, int* lb, void* ip, void* ld,
                       void* std_beta, void* var_gamma, void* var_mu, void* eta, void* q, void* eta_diff,
                       void* u_logs, void* shvt, void* mu_mult, double dq_scale, int n_threads, int low_memory) {
    #pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int k = 0; k < n_blocks; ++k) {
        const int b = order[k];
        const int64_t s = block_start[b], e = block_start[b + 1];
        std::vector<int> lb_local((size_t)(e - s));
        for (int64_t j = s; j < e; ++j) lb_local[(size_t)(j - s)] = lb[j] - (int)s;     // window starts relative to the block
        e_step<T, U, I>((int)(e - s), lb_local.data(), (I*)ip + s, (U*)ld, (T*)std_beta + s, (T*)var_gamma + s,
                        (T*)var_mu + s, (T*)eta + s, (T*)q + s, (T*)eta_diff + s, (T*)u_logs + s, (T*)shvt + s,
                        (T*)mu_mult + s, (T)dq_scale, 1, low_memory != 0);
    }
}

#define DISPATCH_I(FN, T, U, ...)                                   \
    do {                                                            \
        if (icode == IP_I32) { FN<T, U, int32_t>(__VA_ARGS__); return 0; } \
        if (icode == IP_I64) { FN<T, U, int64_t>(__VA_ARGS__); return 0; } \
        return -3;                                                  \
    } while (0)

#define DISPATCH_U(FN, T, ...)                                      \
    do {                                                            \
        switch (ucode) {                                            \
            case LD_I8:  DISPATCH_I(FN, T, int8_t, __VA_ARGS__);    \
            case LD_I16: DISPATCH_I(FN, T, int16_t, __VA_ARGS__);   \
            case LD_I32: DISPATCH_I(FN, T, int32_t, __VA_ARGS__);   \
            case LD_I64: DISPATCH_I(FN, T, int64_t, __VA_ARGS__);   \
            case LD_F32: DISPATCH_I(FN, T, float, __VA_ARGS__);     \
            case LD_F64: DISPATCH_I(FN, T, double, __VA_ARGS__);    \
            default: return -2;                                     \
        }                                                           \
    } while (0)

#define DISPATCH(FN, ...)                                           \
    do {                                                            \
        if (tcode == F32) DISPATCH_U(FN, float, __VA_ARGS__);       \
        if (tcode == F64) DISPATCH_U(FN, double, __VA_ARGS__);      \
        return -1;                                                  \
    } while (0)

}  // namespace

extern "C" {

int ref_blas_supported() { return blas_supported() ? 1 : 0; }
int ref_omp_supported() { return omp_supported() ? 1 : 0; }

int ref_e_step(int tcode, int ucode, int icode, int m, int* lb, void* ip, void* ld, void* std_beta,
               void* var_gamma, void* var_mu, void* eta, void* q, void* eta_diff, void* u_logs,
               void* shvt, void* mu_mult, double dq_scale, int threads, int low_memory) {
    DISPATCH(run_e_step, m, lb, ip, ld, std_beta, var_gamma, var_mu, eta, q, eta_diff, u_logs, shvt,
             mu_mult, dq_scale, threads, low_memory);
}

int ref_e_step_blocks(int tcode, int ucode, int icode, int n_blocks, const int64_t* block_start, const int32_t* order,
                      int* lb, void* ip, void* ld, void* std_beta, void* var_gamma, void* var_mu, void* eta, void* q,
                      void* eta_diff, void* u_logs, void* shvt, void* mu_mult, double dq_scale, int n_threads,
                      int low_memory) {
    DISPATCH(run_e_step_blocks, n_blocks, block_start, order, lb, ip, ld, std_beta, var_gamma, var_mu, eta, q,
             eta_diff, u_logs, shvt, mu_mult, dq_scale, n_threads, low_memory);
}

int ref_e_step_mixture(int tcode, int ucode, int icode, int m, int K, int* lb, void* ip, void* ld,
                       void* std_beta, void* var_gamma, void* var_mu, void* eta, void* q,
                       void* eta_diff, void* log_null_pi, void* u_logs, void* shvt, void* mu_mult,
                       double dq_scale, int threads, int low_memory) {
    DISPATCH(run_e_step_mixture, m, K, lb, ip, ld, std_beta, var_gamma, var_mu, eta, q, eta_diff,
             log_null_pi, u_logs, shvt, mu_mult, dq_scale, threads, low_memory);
}

int ref_e_step_grid(int tcode, int ucode, int icode, int m, int n_active, int* active, int* lb,
                    void* ip, void* ld, void* std_beta, void* var_gamma, void* var_mu, void* eta,
                    void* q, void* eta_diff, void* u_logs, void* hvt, void* mu_mult, double dq_scale,
                    int threads, int low_memory) {
    DISPATCH(run_e_step_grid, m, n_active, active, lb, ip, ld, std_beta, var_gamma, var_mu, eta, q,
             eta_diff, u_logs, hvt, mu_mult, dq_scale, threads, low_memory);
}

}  // extern "C"
