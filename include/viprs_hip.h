/*
 * viprs_hip.h -- C ABI of libviprs_hip.so: the MI355X (gfx950) implementation of viprs's
 * coordinate-ascent variational E-step over LD-matrix blocks.
 *
 * This header is the drop-in boundary.  Every entry point names the reference interface it
 * replaces (paths relative to the shz9/viprs tree, v0.1.4):
 *
 *   reference (Cython -> C++ templates)                         this library
 *   ----------------------------------------------------------  -------------------------------
 *   e_step_cpp.pyx:91-122   cpp_e_step          -> e_step.hpp:343-442   viprs_e_step
 *   e_step_cpp.pyx:125-159  cpp_e_step_mixture  -> e_step.hpp:447-551   viprs_e_step_mixture
 *   e_step_cpp.pyx:161-195  cpp_e_step_grid     -> e_step.hpp:555-647   viprs_e_step_grid
 *   e_step_cpp.pyx:71-76    check_blas_support / check_omp_support      viprs_check_*_support
 *   VIPRS.py:151-172        "load LD matrices to memory" (VIPRS.__init__) viprs_plan_create
 *   VIPRS.py:393-422        per-chromosome loop in VIPRS.e_step()       viprs_state_* (resident)
 *
 * Conventions
 *   - plain pointers + sizes + dtype codes; no C++/torch types cross this boundary;
 *   - every function returns 0 on success, a negative VIPRS_E* code otherwise;
 *     viprs_last_error() returns a thread-local message for the last failure;
 *   - host buffers stay caller-owned; device mirrors are owned by the opaque handles;
 *   - results always follow the reference's `threads = 1` (exact serial Gauss-Seidel)
 *     semantics; the `threads` argument is accepted for signature parity and ignored
 *     (the reference's threads > 1 path is a racy Hogwild loop, e_step.hpp:384-387).
 */
#ifndef VIPRS_HIP_H
#define VIPRS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- dtype codes (mirror the Cython fused types, e_step_cpp.pxd:7-17) ------------------- */
enum viprs_float_dtype { VIPRS_F32 = 0, VIPRS_F64 = 1 };               /* `floating`          */
enum viprs_ld_dtype {                                                   /* noncomplex_numeric  */
    VIPRS_LD_I8 = 0, VIPRS_LD_I16 = 1, VIPRS_LD_I32 = 2, VIPRS_LD_I64 = 3,
    VIPRS_LD_F32 = 4, VIPRS_LD_F64 = 5
};
enum viprs_indptr_dtype { VIPRS_IP_I32 = 0, VIPRS_IP_I64 = 1 };        /* indptr_type         */

/* ---- status codes ----------------------------------------------------------------------- */
enum viprs_status {
    VIPRS_OK = 0,
    VIPRS_EINVAL = -1,     /* bad argument / dtype code                                        */
    VIPRS_ELAYOUT = -2,    /* LD index arrays violate the contiguous-window contract            */
    VIPRS_EDEVICE = -3,    /* HIP runtime error (message in viprs_last_error)                   */
    VIPRS_ENOMEM = -4,
    VIPRS_EUNSUPPORTED = -5
};

/* ---- E-step numerics mode ---------------------------------------------------------------- */
enum viprs_math_mode {
    /* Bit-for-bit the reference's arithmetic (glibc-2.35 expf reproduced in double on the
     * device, sigmoid divide in double as e_step.hpp:254-260 does).  Default.                 */
    VIPRS_MATH_EXACT = 0,
    /* Hardware v_exp_f32 / v_rcp_f32 sigmoid (<= 2 ulp from EXACT per evaluation; well inside
     * the 1e-5 relative parity tolerance).  Shorter serial chain per SNP.                     */
    VIPRS_MATH_FAST = 1
};

/* ---- block kinds reported by the planner ------------------------------------------------- */
enum viprs_block_kind {
    VIPRS_BLOCK_DENSE_SYM = 0,    /* every row's window == the whole block (symmetric form)    */
    VIPRS_BLOCK_DENSE_UPPER = 1,  /* row j's window == (j, block_end)      (upper-tri form)    */
    VIPRS_BLOCK_RAGGED = 2        /* anything else (banded / thresholded windows)              */
};

typedef struct viprs_plan viprs_plan;     /* device-resident LD + block schedule               */
typedef struct viprs_state viprs_state;   /* device-resident variational state bound to a plan */

/* ---- library / device ------------------------------------------------------------------- */
const char* viprs_last_error(void);
const char* viprs_version(void);
/* Experiment / instrumentation switches (-DPANEL_TIMING_*, -DVIPRS_*_PROFILE, ...: kernels_common.h) the translation
 * units of this library were compiled with, space separated.  The shipped library returns "": some of those switches
 * change results, a build that carries any is not the product (they do not compile without -DVIPRS_EXPERIMENTAL). */
const char* viprs_build_flags(void);
int viprs_device_count(int* count);
/* Mirrors check_blas_support() / check_omp_support() (e_step_cpp.pyx:71-76): neither BLAS
 * nor OpenMP is involved on the device path, both report 0.                                   */
int viprs_check_blas_support(void);
int viprs_check_omp_support(void);

/* Content fingerprint of a HOST array: length + first / last 4 KB + 256 evenly spaced 64-byte windows (pure host code, a few
 * microseconds).  The Cython entry points read the caller's LD arrays on every call (e_step_cpp.pyx:91-122); a binding that
 * keeps them resident on the device uses this to notice an in-place edit (viprs_amd/vi/e_step_hip.py::plan_for).            */
int viprs_host_fingerprint(const void* data, int64_t n_bytes, uint64_t* fingerprint);

/* ---- planner (pure host code; usable without a GPU) -------------------------------------- */
/* Validates the LD index arrays (bit-exact integer checks: indptr[0] == 0, indptr monotone,
 * 0 <= left_bound[j], left_bound[j] + len_j <= m; symmetric form: window contains j;
 * upper form: left_bound[j] == j + 1) and partitions SNPs 0..m-1 into independent LD blocks
 * (= connected components of the row windows; the reference has no explicit block loop,
 * independence is implicit in (left_bound, indptr): e_step.hpp:389-392).
 *
 *   block_start  out, capacity m + 1: block b covers SNPs [block_start[b], block_start[b+1])
 *   block_kind   out, capacity m (may be NULL)
 */
int viprs_plan_blocks(int64_t m, const int32_t* ld_left_bound, const void* ld_indptr,
                      int indptr_dtype, int low_memory, int64_t* n_blocks, int64_t* block_start,
                      int32_t* block_kind);

/* ---- plan: upload LD once (replaces "load LD to memory", VIPRS.py:151-172) ---------------- */
int viprs_plan_create(viprs_plan** plan, int64_t m, const int32_t* ld_left_bound,
                      const void* ld_indptr, int indptr_dtype, const void* ld_data, int ld_dtype,
                      int low_memory, int device);
/* Symmetric plan (the `low_memory = False` arithmetic, e_step.hpp:421-428) built from the compact
 * UPPER-TRIANGULAR store: row j of `upper_data` holds the correlations with SNPs j+1 .. j+len_j
 * (`upper_indptr`, m + 1 entries), as LD matrices are kept on disk.  The store is uploaded once and
 * mirrored into the symmetric windows on the device; `diag_value` fills the diagonal (1 for float
 * LD, the quantisation maximum -- 127 / 32767 -- for integer LD).  Replaces the host-side
 * `ld_mat.load(return_symmetric=True)` of VIPRS.py:167-172: the symmetric copy never exists in host
 * memory and half the bytes cross PCIe.  The right ends j + len_j must not decrease (otherwise the
 * mirrored rows would not be contiguous windows): VIPRS_EINVAL.
 * viprs_plan_get_windows returns the (left_bound, indptr) arrays of the plan's rows (any plan). */
int viprs_plan_create_expanded(viprs_plan** plan, int64_t m, const void* upper_indptr, int indptr_dtype,
                               const void* upper_data, int ld_dtype, double diag_value, int device);
int viprs_plan_get_windows(const viprs_plan* plan, int32_t* left_bound, int64_t* indptr);
int viprs_plan_destroy(viprs_plan* plan);

enum viprs_plan_info_key {
    VIPRS_INFO_M = 0, VIPRS_INFO_NNZ = 1, VIPRS_INFO_N_BLOCKS = 2, VIPRS_INFO_N_DENSE = 3,
    VIPRS_INFO_N_RAGGED = 4, VIPRS_INFO_MAX_BLOCK = 5, VIPRS_INFO_LD_BYTES_DEVICE = 6,
    VIPRS_INFO_LD_ELEM_SIZE = 7, VIPRS_INFO_DEVICE = 8, VIPRS_INFO_LOW_MEMORY = 9,
    VIPRS_INFO_N_CU = 10
};
int viprs_plan_info(const viprs_plan* plan, int key, int64_t* value);
/* Copies the planner's block boundaries (n_blocks + 1 entries) */
int viprs_plan_get_blocks(const viprs_plan* plan, int64_t* block_start, int32_t* block_kind);
int viprs_plan_set_math_mode(viprs_plan* plan, int math_mode);
/* Which LD blocks the following sweeps visit: one byte per block in SNP order (the order of viprs_plan_get_blocks), non-zero =
 * swept; NULL = every block again.  The reference fits one model per chromosome by default (bin/viprs_fit:232-238, fan-out
 * :1079-1086) and each of those fits stops at its own iteration (VIPRS.py:1046-1094); with the chromosomes' blocks in ONE
 * plan (viprs_state_set_groups below) a converged chromosome's blocks leave the sweep this way -- its state stays as its
 * last E-step left it.  Host-side list surgery over a few thousand descriptors; LD and per-SNP arrays do not move.
 * Not supported by the batched grid kernel (VIPRS_EUNSUPPORTED from viprs_state_e_step on a grid state). */
int viprs_plan_set_active_blocks(viprs_plan* plan, const uint8_t* active, int64_t n_blocks);

/* ---- one-shot calls on host buffers: the drop-ins for the Cython entry points ------------ */
/* Argument order and meaning follow e_step_cpp.pyx:91-122 after the plan handle (which stands
 * for ld_left_bound / ld_indptr / ld_data).  var_gamma, var_mu, eta, q, eta_diff are updated
 * in place exactly as e_step.hpp:343-442 does; all float buffers share `float_dtype`.         */
int viprs_e_step(viprs_plan* plan, int float_dtype, const void* std_beta, void* var_gamma,
                 void* var_mu, void* eta, void* q, void* eta_diff, const void* u_logs,
                 const void* sqrt_half_var_tau, const void* mu_mult, double dq_scale, int threads,
                 int low_memory);

/* e_step_cpp.pyx:125-159: (m, K) arrays are C-ordered.                                        */
int viprs_e_step_mixture(viprs_plan* plan, int float_dtype, int K, const void* std_beta,
                         void* var_gamma, void* var_mu, void* eta, void* q, void* eta_diff,
                         const void* log_null_pi, const void* u_logs,
                         const void* sqrt_half_var_tau, const void* mu_mult, double dq_scale,
                         int threads, int low_memory);

/* e_step_cpp.pyx:161-195: (m, G) arrays are column-major; active_model_idx is contiguous here
 * (the ctypes shim gathers a strided int[:]).                                                 */
int viprs_e_step_grid(viprs_plan* plan, int float_dtype, int G, const void* std_beta,
                      void* var_gamma, void* var_mu, void* eta, void* q, void* eta_diff,
                      const void* u_logs, const void* half_var_tau, const void* mu_mult,
                      double dq_scale, const int32_t* active_model_idx, int n_active, int threads,
                      int low_memory);

/* ---- device-resident state: what VIPRS.e_step()'s per-iteration loop keeps alive --------- */
enum viprs_model_kind { VIPRS_MODEL_SPIKE_SLAB = 0, VIPRS_MODEL_MIXTURE = 1, VIPRS_MODEL_GRID = 2 };

enum viprs_field {
    /* inputs */
    VIPRS_FIELD_STD_BETA = 0, VIPRS_FIELD_U_LOGS = 1, VIPRS_FIELD_SQRT_HALF_VAR_TAU = 2,
    VIPRS_FIELD_MU_MULT = 3, VIPRS_FIELD_LOG_NULL_PI = 4,
    /* in/out state */
    VIPRS_FIELD_VAR_GAMMA = 5, VIPRS_FIELD_VAR_MU = 6, VIPRS_FIELD_ETA = 7, VIPRS_FIELD_Q = 8,
    VIPRS_FIELD_ETA_DIFF = 9,
    VIPRS_FIELD_COUNT = 10
};

/* `width` = 1 (spike-and-slab), K (mixture) or G (grid).                                      */
int viprs_state_create(viprs_state** state, viprs_plan* plan, int float_dtype, int model_kind,
                       int width);
int viprs_state_destroy(viprs_state* state);
/* Whole-field copies; sizes follow the reference shapes ((m,), (m,K) C-order, (m,G) F-order).  */
int viprs_state_upload(viprs_state* state, int field, const void* host);
int viprs_state_download(viprs_state* state, int field, void* host);
/* Device-side re-initialisation to the reference's standard start (VIPRS.py:344-358):
 * var_gamma = pi, var_mu = eta = q = eta_diff = 0.                                            */
int viprs_state_reset(viprs_state* state, double pi);
/* One E-step sweep over every LD block with the resident inputs; asynchronous on the plan's
 * stream unless `sync` != 0.                                                                  */
int viprs_state_e_step(viprs_state* state, double dq_scale, const int32_t* active_model_idx,
                       int n_active, int sync);
int viprs_state_synchronize(viprs_state* state);

/* ---- device-resident EM iteration (spike-and-slab): host prep, zeta and the M-step / ELBO sums ---
 * Replaces the O(m) NumPy passes of VIPRS.e_step / compute_zeta / m_step / elbo
 * (VIPRS.py:400-418, :888-897, :426-471, :497-581) so that fit() moves only scalars per iteration. */
/* n_per_snp (float64, (m,)), uploaded once.                                                      */
int viprs_state_set_n_per_snp(viprs_state* state, const double* n_per_snp);
/* On-device VIPRS.py:400-418 in float64, rounded to the state precision at the end:
 *   var_tau = n one_plus_lambda / sigma_epsilon + tau_beta
 *   mu_mult = n / (var_tau sigma_epsilon);  sqrt_half_var_tau = sqrt(var_tau / 2)
 *   u_logs  = logit_pi + 0.5 (log_tau_beta - log var_tau)
 * (the scalars logit_pi = log(pi) - log(1 - pi), log_tau_beta and one_plus_lambda = 1 + lambda_min
 * are evaluated by the caller so that the reference's scalar dtype promotion -- float32 scalars in
 * VIPRS.py:400-406 -- can be reproduced exactly).                                                 */
int viprs_state_prep(viprs_state* state, double logit_pi, double log_tau_beta, double sigma_epsilon,
                     double tau_beta, double one_plus_lambda);
/* Deterministic (fixed-order) float64 reductions over the plan's SNPs with the hyper-parameters of
 * the last viprs_state_prep; `out` receives VIPRS_N_SUMS doubles on the HOST:
 *   [0] sum gamma  [1] sum zeta  [2] sum(one_plus_lambda zeta + q eta)  [3] sum std_beta eta  [4] sum eta^2
 *   [5] sum g log g  [6] sum (1-g) log(1-g)  [7] sum g  [8] sum (1-g)   (g clipped to [1e-15, 1-1e-15])
 *   [9] sum g log var_tau  [10] max |eta_diff|
 * zeta = gamma (mu^2 + 1/var_tau) in float64 (VIPRS.py:896).  Synchronises the plan's stream.      */
#define VIPRS_N_SUMS 11
/* Optional per-SNP weights (m doubles; NULL clears them) for sum [0]: with several chromosomes merged into one
 * plan, w_j = 1 / (SNPs of j's chromosome) makes sum [0] the reference's sum of per-chromosome means
 * (update_pi, VIPRS.py:446-453). */
int viprs_state_set_snp_weights(viprs_state* state, const double* weights);
int viprs_state_sums(viprs_state* state, double one_plus_lambda, double* out);
/* The same in two halves, so that several plans (chromosomes) reduce concurrently: `begin` enqueues the
 * reduction and an asynchronous copy on the plan's stream, `end` waits for it and returns the sums. */
int viprs_state_sums_begin(viprs_state* state, double one_plus_lambda);
int viprs_state_sums_end(viprs_state* state, double* out);
/* The same two operations on ONE model (column `g`) of a grid state ((m, G) column-major arrays),
 * for the batched grid fit: e_step_grid takes half_var_tau = var_tau / 2 (e_step.hpp:616) where
 * e_step takes its square root, otherwise the formulas are those above.  (A grid state does not keep
 * var_tau -- m x G doubles written per prep and read back per reduction: the sums form it again from n_j
 * and the scalars of the column's last prep, the same expression, the same bits.)                   */
int viprs_state_prep_column(viprs_state* state, int g, double logit_pi, double log_tau_beta, double sigma_epsilon,
                            double tau_beta, double one_plus_lambda);
int viprs_state_sums_column(viprs_state* state, int g, double one_plus_lambda, double* out);
/* ---- device-resident EM iteration of the mixture model (K <= 8), VIPRSMix.py:169-260 -------------------
 * prep    : var_tau[j,k] = n_j * one_plus_lambda / sigma_epsilon + tau_beta[k] and the E-step inputs of the state
 *           (u_logs, sqrt_half_var_tau, mu_mult, log_null_pi).  The K-vectors logit_pi = log(pi) - log(1 - pi),
 *           log_tau_beta and tau_beta and the scalar log_null_pi are evaluated by the caller (reference dtype
 *           semantics).  Needs viprs_state_set_n_per_snp.
 * log_var_tau : (m, K) doubles, C order -- log of the INITIAL var_tau, which the reference's ELBO keeps using.
 * sums    : 7 + 6 K doubles: sum zeta | sum((1+lambda) zeta + q eta) | sum std_beta eta | sum eta^2 |
 *           sum null_gamma log null_gamma | sum null_gamma | then K-vectors sum gamma | sum gamma (mu^2 + 1/var_tau) |
 *           sum gamma_c log gamma_c | sum gamma_c | sum gamma_c log_var_tau | sum gamma_c (mu^2 + 1/var_tau)
 *           (gamma_c = gamma clipped to [1e-15, 1 - 1e-15]) | max |eta_diff|. */
int viprs_state_set_log_var_tau(viprs_state* state, const double* log_var_tau);
int viprs_state_prep_mixture(viprs_state* state, const double* logit_pi, const double* log_tau_beta,
                             const double* tau_beta, double log_null_pi, double sigma_epsilon, double one_plus_lambda);
int viprs_state_sums_mixture_begin(viprs_state* state, double one_plus_lambda);
int viprs_state_sums_mixture_end(viprs_state* state, double* out);

/* Several models per launch (the batched grid fit touches every active model in every EM iteration):
 *   params : n rows of 6 doubles  (column, logit_pi, log_tau_beta, sigma_epsilon, tau_beta, one_plus_lambda)
 *   cols   : n rows of 2 doubles  (column, one_plus_lambda)
 *   out    : n rows of VIPRS_N_SUMS doubles, in the order of `cols`
 * `begin` is asynchronous on the plan's stream, `end` waits for it. */
int viprs_state_prep_columns(viprs_state* state, int n, const double* params);
int viprs_state_sums_columns_begin(viprs_state* state, int n, const double* cols);
int viprs_state_sums_columns_end(viprs_state* state, double* out);
/* Per-column re-initialisation of a grid state: var_gamma[:, g] = pi_g, everything else 0.         */
int viprs_state_reset_column(viprs_state* state, int g, double pi);

/* ---- SNP groups: one spike-and-slab model per chromosome, all chromosomes in ONE plan -----------------------------
 * The reference's default mode (bin/viprs_fit:232-238: `split_by_chromosome()` unless --genomewide) fits an independent
 * VIPRS model per chromosome, each with its own (pi, tau_beta, sigma_epsilon), M-step sums, ELBO and stopping iteration.
 * A chromosome-sized fit cannot fill the device (its sweep is bound by the chain of its largest LD block), 22 of them in
 * one plan cost one genome-wide sweep.  A GROUP is a contiguous SNP range made of whole LD blocks:
 *   viprs_state_set_groups        group g = SNPs [group_start[g], group_start[g+1]); n_groups + 1 entries covering 0 .. m;
 *                                 n_groups = 0 removes the groups.  Spike-and-slab and mixture (K <= 8) states.
 *   viprs_state_prep_groups       viprs_state_prep with per-group scalars: n rows of 6 doubles
 *                                 (group, logit_pi, log_tau_beta, sigma_epsilon, tau_beta, one_plus_lambda); only the listed
 *                                 groups' SNPs are rewritten (a converged group keeps the inputs of its last E-step).
 *   viprs_state_sums_groups_begin n rows of 2 doubles (group, one_plus_lambda): the VIPRS_N_SUMS sums of viprs_state_sums
 *                                 per listed group, one launch; [0] is the plain sum of gamma (no per-SNP weights).
 *   viprs_state_sums_groups_end   n rows of VIPRS_N_SUMS doubles in the order of the rows given to `begin`.
 * A group's inputs and sums are bit-identical to those of a plan that holds only that group's blocks (the reduction over
 * a group uses the workgroup count and element order a plan of that size would use).  With viprs_state_set_comm the rows
 * are all-rank sums (one all-gather for all groups). */
int viprs_state_set_groups(viprs_state* state, int n_groups, const int64_t* group_start);
int viprs_state_prep_groups(viprs_state* state, int n, const double* params);
int viprs_state_sums_groups_begin(viprs_state* state, int n, const double* rows);
int viprs_state_sums_groups_end(viprs_state* state, double* out);
/* The same for a mixture state (one VIPRSMix model per chromosome; K = the state's width):
 *   viprs_state_prep_mixture_groups       viprs_state_prep_mixture with per-group parameters: n rows of 4 + 3 K doubles
 *                                         (group, log_null_pi, sigma_epsilon, one_plus_lambda, logit_pi[K], log_tau_beta[K],
 *                                         tau_beta[K])
 *   viprs_state_sums_mixture_groups_begin n rows of 2 doubles (group, one_plus_lambda)
 *   viprs_state_sums_mixture_groups_end   n rows of 7 + 6 K doubles, the layout of viprs_state_sums_mixture_end
 * viprs_state_set_log_var_tau takes the (m, K) array of all groups as before. */
int viprs_state_prep_mixture_groups(viprs_state* state, int n, const double* params);
int viprs_state_sums_mixture_groups_begin(viprs_state* state, int n, const double* rows);
int viprs_state_sums_mixture_groups_end(viprs_state* state, double* out);

/* ---- multi-GPU: RCCL over xGMI for the scalar reductions of the EM iteration -------------------------
 * One process per GPU; LD blocks are sharded over the ranks (independent units: within one E-step call
 * the hyper-parameters are fixed and blocks share no q entries, so the data path has NO collective).
 * What the ranks must agree on per EM iteration are the ~10-300 float64 partial sums of the M-step, the
 * ELBO and the stopping rules (VIPRS.py:426-484, :497-581, :997) -- the reference itself has no collective
 * at all (one process, OpenMP inside the kernel, joblib over chromosomes: bin/viprs_fit:1080-1086).
 *
 * The communicator wraps an RCCL (ncclComm_t) communicator; librccl is opened at run time, so processes
 * that never create a communicator do not load it.  Every reduction is ONE ncclAllGather of the small
 * vector followed by a rank-ordered reduction on the device: sums are added in rank order 0..n-1 (so every
 * rank gets bit-identical results, run to run), the last element of every `group` consecutive elements is
 * reduced with max (that slot carries max |eta_diff|).
 *
 *   viprs_comm_unique_id   rank 0: fills `id` (VIPRS_COMM_ID_BYTES) -- hand it to the other ranks out of band
 *                          (viprs_amd.parallel does it through the file system / the launcher's environment)
 *   viprs_comm_create      collective over all `world_size` processes; `device` = this rank's HIP device
 *   viprs_comm_allreduce   host vector in/out (control plane: hyper-parameter broadcast, bench timing);
 *                          group = 0: plain sum, group = -1: plain max, group > 0: see above
 *   viprs_comm_barrier     all ranks' devices idle + all ranks arrived
 *   viprs_state_set_comm   from now on viprs_state_sums*_begin/_end of this state return the ALL-RANK sums:
 *                          the all-gather + ordered reduction run on the plan's stream between the local
 *                          reduction kernels and the copy to the host -- one collective per EM iteration,
 *                          no host round trip in front of it.  NULL detaches.  A rank whose plan is empty
 *                          still takes part (it contributes zeros).                                          */
#define VIPRS_COMM_ID_BYTES 128
typedef struct viprs_comm viprs_comm;
int viprs_comm_unique_id(void* id);
int viprs_comm_create(viprs_comm** comm, const void* id, int rank, int world_size, int device);
int viprs_comm_destroy(viprs_comm* comm);
int viprs_comm_rank(const viprs_comm* comm, int* rank, int* world_size);
int viprs_comm_allreduce(viprs_comm* comm, double* vec, int n, int group);
/* bulk exchange at the END of a fit (the posterior of each rank's SNPs, BayesPRSModel.py:333-410 wants all of them):
 * every rank sends `n` doubles, `recv` receives world_size x n in rank order -- one ncclAllGather.                    */
int viprs_comm_allgather(viprs_comm* comm, const double* send, int64_t n, double* recv);
int viprs_comm_barrier(viprs_comm* comm);
int viprs_state_set_comm(viprs_state* state, viprs_comm* comm);
/* hipDeviceSynchronize() on `device` (what the bench brackets its timed region with).                   */
int viprs_device_synchronize(int device);

/* ---- measurement hooks (bench.py) --------------------------------------------------------- */
/* HIP-event time (ms) of the kernels of the last viprs_state_e_step / viprs_e_step* call on
 * this plan, measured on the stream they were launched on.  `which`: 0 = all kernels of the
 * sweep, 1 = dominant (panel) kernel only, 2 = HOST time the library spent between recording that
 * kernel's start event and recording its end event (the launch call: a start event recorded into
 * an empty stream is reached at once, so host time spent there counts into the event bracket). */
int viprs_plan_last_kernel_ms(viprs_plan* plan, int which, double* ms);
/* Every sweep records its HIP events into a ring of 256 entries.  `timing_reset` forgets them;
 * `timing_history` returns the durations (ms) of the most recent min(capacity, 256, recorded)
 * sweeps, oldest first.                                                                       */
int viprs_plan_timing_reset(viprs_plan* plan);
int viprs_plan_timing_history(viprs_plan* plan, int which, double* ms, int capacity, int* n);
/* Arithmetic the kernels of the last sweep on this plan really ran in: bit 0 = exact, bit 1 = fast.
 * viprs_plan_set_math_mode(FAST) applies where a fast instantiation exists (spike-and-slab, grid, mixtures of up to 8
 * components, fp32 state); mixtures of 9+ components, the mixture / grid kernels of windowed blocks with K > 8 and
 * every float64 state run exact whatever was asked for -- this is how a caller (bench.py) finds out.          */
int viprs_plan_last_math_modes(const viprs_plan* plan, int* mask);
/* Number of SNPs of the last sweep that took the skip branch (e_step.hpp:410-413).            */
int viprs_plan_last_skipped(viprs_plan* plan, int64_t* n_skipped);

/* ---- measurement support: synthetic LD generated on the device (bench.py, tests) --------------
 * The "longrange" LD blocks of viprs_amd/utils/synthetic.py (the workload of BASELINE.json's configs, SURVEY.md 8d: the
 * reference gets its LD from magenpy stores, VIPRS.py:151-172, none of which exists here) written straight into a plan's
 * device memory: a rank of a multi-GPU run has its genome-scale workload in milliseconds instead of generating 3.8 GB on
 * the host and uploading it.  `sizes`: n_blocks block sizes; pw / uf0 / uf1 / sa / sf: per-SNP float32 parameter
 * vectors (m = sum of sizes entries each; synthetic.longrange_device_params): entry (i, j) of a block is
 *   (uf0[i] uf0[j] + uf1[i] uf1[j]) + (pw[|i - j|] sa[i]) sf[j]  off the diagonal, 1 on it,
 * every operation rounded to float32; int8 / int16 LD stores rint(127 x) / rint(32767 x).  Layout, validation, block
 * discovery and re-lay-out are those of viprs_plan_create.  viprs_synthetic_ld_host runs the SAME function on the host
 * into the caller's row-concatenated array of `capacity` elements (the CPU test of the generator). */
int viprs_plan_create_synthetic(viprs_plan** plan, int64_t n_blocks, const int64_t* sizes, const float* pw,
                                const float* uf0, const float* uf1, const float* sa, const float* sf, int ld_dtype,
                                int low_memory, int device);
int viprs_synthetic_ld_host(int64_t n_blocks, const int64_t* sizes, const float* pw, const float* uf0, const float* uf1,
                            const float* sa, const float* sf, int ld_dtype, int low_memory, void* out, int64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* VIPRS_HIP_H */
