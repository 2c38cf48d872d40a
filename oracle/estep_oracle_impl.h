/* TEST INFRASTRUCTURE ONLY (see estep_oracle.c).  Included once per (T, U) pair with
 *   T       floating state type (float | double)
 *   U       LD storage type
 *   SUF     function-name suffix
 *   EXP_T   exp for T (expf | exp), FMA_T (fmaf | fma), ABS_T (fabsf | fabs), EPS_T
 * Every function cites the reference lines it restates (paths relative to /root/reference). */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* viprs/model/vi/e_step.hpp:157-175 `axpy`: x[i] = fma((T)y[i], alpha, x[i]) in index order. */
static void FN(axpy_)(T* x, const U* y, T alpha, int64_t size) {
    for (int64_t i = 0; i < size; ++i) x[i] = FMA_T((T)y[i], alpha, x[i]);
}

/* viprs/model/vi/e_step.hpp:82-104 `dot`: s = fma((T)y[i], x[i], s) from s = 0, in index order
 * (the `omp simd` pragma there carries no reduction clause and std::fma is a libm call under the
 * reference's flags, so the loop stays a serial chain; pinned against oracle/_ref). */
static T FN(dot_)(const T* x, const U* y, int64_t size) {
    T s = (T)0.;
    for (int64_t i = 0; i < size; ++i) s = FMA_T((T)y[i], x[i], s);
    return s;
}

/* viprs/model/vi/e_step.hpp:245-261 `sigmoid`: the literal `1.` is a double, so with T = float the
 * add and the divide are carried out in double and rounded to T once. */
static T FN(sigmoid_)(T x) {
    if (x < 0) {
        T e = EXP_T(x);
        return (T)((double)e / (1. + (double)e));
    } else {
        T e = EXP_T(-x);
        return (T)(1. / (1. + (double)e));
    }
}

/* viprs/model/vi/e_step.hpp:307-338 `update_q_factor` (upper-triangular epilogue). */
static void FN(update_q_factor_)(int64_t m, const int32_t* lb, const int64_t* ip, const U* ld,
                                 const T* eta, T* q, T dq) {
    for (int64_t j = 0; j < m; ++j) {
        int64_t s = ip[j], e = ip[j + 1];
        q[j] += dq * FN(dot_)(eta + lb[j], ld + s, e - s);
    }
}

/* viprs/model/vi/e_step.hpp:343-442 `e_step` (spike-and-slab), threads = 1 semantics. */
void FN(oracle_e_step_)(int64_t m, const int32_t* lb, const int64_t* ip, const U* ld,
                        const T* std_beta, T* var_gamma, T* var_mu, T* eta, T* q, T* eta_diff,
                        const T* u_logs, const T* shvt, const T* mu_mult, T dq, int low_memory) {
    const T eps = (EPS_T > (T)1e-8) ? EPS_T : (T)1e-8; /* :382 */
    for (int64_t j = 0; j < m; ++j) {
        int64_t ls = ip[j], le = ip[j + 1];
        int64_t start = lb[j], len = le - ls;
        T mu = FMA_T(mu_mult[j], std_beta[j], -mu_mult[j] * q[j]); /* :401 */
        T u = shvt[j] * mu;                                         /* :404 */
        T gamma = FN(sigmoid_)(FMA_T(u, u, u_logs[j]));            /* :405 */
        T d = FMA_T(gamma, mu, -eta[j]);                           /* :408 */
        if (ABS_T(d) < eps) {                                      /* :410-413 */
            eta_diff[j] = (T)0.;
        } else {
            var_mu[j] = mu;                                        /* :416-418 */
            var_gamma[j] = gamma;
            eta_diff[j] = d;
            FN(axpy_)(q + start, ld + ls, dq * d, len);            /* :421 */
            if (!low_memory) q[j] -= d;                            /* :423-428 */
            eta[j] += d;                                           /* :431 */
        }
    }
    if (low_memory) FN(update_q_factor_)(m, lb, ip, ld, eta_diff, q, dq); /* :435-440 */
}

/* viprs/model/vi/e_step.hpp:447-551 `e_step_mixture` + :222-241 `softmax` + :58-71 `c_max`.
 * (m, K) arrays are C-ordered (index j*K + k, :508).  `u` is caller scratch of K + 1 values. */
void FN(oracle_e_step_mixture_)(int64_t m, int K, const int32_t* lb, const int64_t* ip, const U* ld,
                                const T* std_beta, T* var_gamma, T* var_mu, T* eta, T* q,
                                T* eta_diff, const T* log_null_pi, const T* u_logs, const T* shvt,
                                const T* mu_mult, T dq, int low_memory, T* u) {
    for (int64_t j = 0; j < m; ++j) {
        int64_t ls = ip[j], le = ip[j + 1];
        int64_t start = lb[j], len = le - ls;
        T r = std_beta[j] - q[j];                                  /* :505 */
        for (int k = 0; k < K; ++k) {                              /* :507-512 */
            int64_t idx = j * K + k;
            var_mu[idx] = mu_mult[idx] * r;
            T t = shvt[idx] * var_mu[idx];
            u[k] = FMA_T(t, t, u_logs[idx]);
        }
        u[K] = log_null_pi[j];                                     /* :515 */
        /* softmax(u, var_gamma + j*K, K + 1), :231-240 */
        T mx = u[0];
        for (int i = 1; i < K + 1; ++i) if (mx < u[i]) mx = u[i];
        T s = (T)0.;
        for (int i = 0; i < K + 1; ++i) { u[i] = EXP_T(u[i] - mx); s += u[i]; }
        for (int i = 0; i < K; ++i) var_gamma[j * K + i] = u[i] / s;
        T d = -eta[j];                                             /* :519-524 */
        for (int k = 0; k < K; ++k) d = FMA_T(var_gamma[j * K + k], var_mu[j * K + k], d);
        eta_diff[j] = d;
        FN(axpy_)(q + start, ld + ls, dq * d, len);                /* :527 */
        if (!low_memory) q[j] -= d;                                /* :529-534 */
        eta[j] += d;                                               /* :536 */
    }
    if (low_memory) FN(update_q_factor_)(m, lb, ip, ld, eta_diff, q, dq); /* :543-549 */
}

/* viprs/model/vi/e_step.hpp:555-647 `e_step_grid` + :266-303 `update_q_factor_matrix`.
 * (m, G) arrays are column-major (index g*m + j, :610); `active` may be any subset/order. */
void FN(oracle_e_step_grid_)(int64_t m, int n_active, const int32_t* active, const int32_t* lb,
                             const int64_t* ip, const U* ld, const T* std_beta, T* var_gamma,
                             T* var_mu, T* eta, T* q, T* eta_diff, const T* u_logs, const T* hvt,
                             const T* mu_mult, T dq, int low_memory) {
    for (int64_t j = 0; j < m; ++j) {
        int64_t ls = ip[j], le = ip[j + 1];
        int64_t start = lb[j], len = le - ls;
        for (int a = 0; a < n_active; ++a) {
            int64_t g = active[a];
            int64_t idx = g * m + j;
            var_mu[idx] = mu_mult[idx] * (std_beta[j] - q[idx]);            /* :613 */
            T uj = u_logs[idx] + hvt[idx] * var_mu[idx] * var_mu[idx];      /* :616 */
            var_gamma[idx] = FN(sigmoid_)(uj);                              /* :617 */
            eta_diff[idx] = var_gamma[idx] * var_mu[idx] - eta[idx];        /* :620 */
            FN(axpy_)(q + (g * m + start), ld + ls, dq * eta_diff[idx], len); /* :623 */
            if (!low_memory) q[idx] -= eta_diff[idx];                       /* :625-630 */
            eta[idx] += eta_diff[idx];                                      /* :633 */
        }
    }
    if (low_memory) {                                                       /* :637-645, :294-302 */
        for (int64_t j = 0; j < m; ++j) {
            int64_t s = ip[j], e = ip[j + 1];
            for (int a = 0; a < n_active; ++a) {
                int64_t off = (int64_t)active[a] * m;
                q[off + j] += dq * FN(dot_)(eta_diff + (off + lb[j]), ld + s, e - s);
            }
        }
    }
}

#undef FN
#undef CAT
#undef CAT_
