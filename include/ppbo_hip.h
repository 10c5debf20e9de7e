/*
 * ppbo_hip.h -- C-ABI of libppbo_hip.so: the MI355X (gfx950) GP-surrogate and
 * acquisition engine behind PPBO's GPModel / next_query() / Hsampler surface.
 *
 * The reference (AaltoPML/PPBO) has no FFI layer: its boundary is a Python
 * object surface.  Each entry point below names the reference expression it
 * replaces (file:line relative to the reference repository).  INTEGRATION.md
 * shows the ctypes stub a PPBO maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types.
 *   - every function returns int: 0 ok, <0 invalid argument, >0 HIP error code
 *     (PPBO_ERR_NOT_PD = 1001 is the only non-HIP positive code).  Nothing throws.
 *   - pointers named d_* are DEVICE pointers owned by the caller (row-major,
 *     C-contiguous float64 unless stated); h_* are HOST pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *     Functions with host outputs synchronise that stream before returning.  Calls WITHOUT host outputs
 *     (ppbo_gram, ppbo_cross_cov, ppbo_rff_project, ppbo_predict / ppbo_rff_score with NULL h_best_*, ...)
 *     neither synchronise nor -- after their first call at a given size -- allocate, so a sequence of them can be
 *     captured in a HIP graph on that stream and replayed (tests/test_gpu_concurrent.py).
 *   - a ppbo_ctx is bound to ONE device; every entry point runs on that device and restores
 *     the caller's current device on return.  Several contexts (on the same or different
 *     devices) may live in one process; a single ctx is not re-entrant across threads.
 *     A ctx owns only private workspaces; the library has no process-global mutable state.
 */
#ifndef PPBO_HIP_H
#define PPBO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPBO_ABI_VERSION 6
#define PPBO_ERR_NOT_PD 1001

/* The library is built with -fvisibility=hidden: the entry points declared here are its ONLY dynamic symbols
 * (tests/test_abi.py checks `nm -D --defined-only` against this header). */
#if defined(__GNUC__) || defined(__clang__)
#define PPBO_API __attribute__((visibility("default")))
#else
#define PPBO_API
#endif

typedef struct ppbo_ctx ppbo_ctx;

/* kernel ids: src/kernels.py:19 (SE), :27 (RQ, alpha=2), :36 (camphor-copper, D must be 6) */
enum { PPBO_KERNEL_SE = 0, PPBO_KERNEL_RQ = 1, PPBO_KERNEL_CAMPHOR = 2 };

/* candidate score folded into the running argmax */
enum {
  PPBO_SCORE_MEAN = 0,         /* mu(x): what mu_star maximises, src/gp_model.py:422 */
  PPBO_SCORE_POINTWISE_EI = 1, /* (mu-mu*)Phi(z)+s phi(z): G=1 closed form of src/acquisition.py:72-81 */
  PPBO_SCORE_VARIANCE = 2      /* sigma^2(x): G=1 form of varmax, src/acquisition.py:170-178 */
};

PPBO_API int ppbo_abi_version(void);
PPBO_API int ppbo_ctx_create(int device, ppbo_ctx** out);
PPBO_API int ppbo_ctx_destroy(ppbo_ctx* ctx);
/* copies the last error text of this ctx into buf (NUL terminated) */
PPBO_API int ppbo_last_error(ppbo_ctx* ctx, char* buf, size_t n);

/* ---- per-kernel event timing (the reference only prints time.time() deltas when verbose,
 * src/gp_model.py:110-132).  When enabled, the named hot kernels are bracketed
 * by hipEvents on the caller's stream; ppbo_profile_read synchronises those events and
 * returns the accumulated duration and launch count since the last reset.
 * names: "gram", "kstar", "quadform", "score", "rff_project", "rff_score", "potrf", "line_kstar", "line_y",
 * "line_cov", "line_mc", "fused_score" (the one-launch scoring kernel of models with up to 1024 rows). */
PPBO_API int ppbo_profile_enable(ppbo_ctx* ctx, int on);
PPBO_API int ppbo_profile_reset(ppbo_ctx* ctx);
PPBO_API int ppbo_profile_read(ppbo_ctx* ctx, const char* name, double* h_total_ms, int* h_count);

/* ---- K1: Gram matrix with the closed-form shrinkage --------------------
 * replaces GPModel.create_Gramian (src/gp_model.py:147-151) =
 * kernel(X,X,theta) (src/kernels.py:19-53) + regularize_covariance
 * (src/misc.py:71-88; SVD round trip == identity, shrink == (1-s)K + s tr(K)/N I).
 * d_X[N,D] -> d_Sigma[N,N]. */
PPBO_API int ppbo_gram(ppbo_ctx* ctx, int kernel_id, const double* d_X, int N, int D,
              const double h_theta[3], double shrink, double* d_Sigma, void* stream);

/* a-3 as its own operator: regularize_covariance(X, reg_level, pos_diag=True, jitter) of src/misc.py:71-88 applied
 * IN PLACE to any device matrix d_K[N, ldk] (gp_model.py:150 calls it on the raw Gramian; ppbo_gram fuses the same
 * closed form): negative diagonal entries -> jitter when pos_diag != 0, then K <- (1 - reg_level) K + reg_level tr(K)/N I
 * (sklearn.covariance.shrunk_covariance).  The SVD round trip of misc.py:79-80 is the identity and is not executed. */
PPBO_API int ppbo_regularize_covariance(ppbo_ctx* ctx, double* d_K, int N, int ldk, double reg_level, int pos_diag,
                               double jitter, void* stream);

/* measurement probe, not part of the path: a write-only pass over d_S[N,N] (32 x 128 tiles, 16-byte write-through
 * stores, the fastest store shape tools/store_floor.hip found): the ceiling of ANY Gram kernel at that N on this
 * chip.  bench.py times it beside ppbo_gram (`write_only_floor_*`) instead of quoting constants. */
PPBO_API int ppbo_store_floor(ppbo_ctx* ctx, double* d_S, int N, void* stream);

/* ---- K2: raw cross-covariance ------------------------------------------
 * replaces GPModel.create_Gramian_nonsquare (src/gp_model.py:153-155).
 * d_X1[n1,D], d_X2[n2,D] -> d_K[n1,n2] (row stride ldk >= n2). */
PPBO_API int ppbo_cross_cov(ppbo_ctx* ctx, int kernel_id, const double* d_X1, int n1,
                   const double* d_X2, int n2, int D, const double h_theta[3],
                   double* d_K, int ldk, void* stream);

/* ---- K6: dense SPD factor / inverse (fp64, hand-written blocked kernels) --
 * ppbo_potrf: in-place lower Cholesky of d_A[N,N] (upper triangle untouched).
 *   *h_info = 0 on success, k>0 if the leading minor of order k is not PD
 *   (LAPACK convention); return value PPBO_ERR_NOT_PD in that case.
 * ppbo_pd_inverse replaces misc.pd_inverse (src/misc.py:96-100): d_Ainv = A^-1
 *   (full symmetric matrix written). */
PPBO_API int ppbo_potrf(ppbo_ctx* ctx, double* d_A, int N, int lda, int* h_info, void* stream);
PPBO_API int ppbo_pd_inverse(ppbo_ctx* ctx, const double* d_A, int N, double* d_Ainv, int* h_info, void* stream);

/* ppbo_pd_inverse that also hands out L^-1 (A = L L^T, full matrix, zeros above the diagonal); d_Linv may be NULL */
PPBO_API int ppbo_pd_inverse_factors(ppbo_ctx* ctx, const double* d_A, int N, double* d_Ainv, double* d_Linv, int* h_info,
                            void* stream);
/* ... and, when d_L is not NULL, the Cholesky factor itself: d_L[N,N], lower triangle = L, the strict upper triangle
 * is a copy of A's and is never read by this library.  L is what ppbo_fit_fmap_whitened iterates with, and
 * L z (ppbo_dgemv, lower = 1) is the prior draw of src/gp_model.py:374,381.  d_L and d_Linv may each be NULL. */
PPBO_API int ppbo_pd_inverse_ex(ppbo_ctx* ctx, const double* d_A, int N, double* d_Ainv, double* d_L, double* d_Linv,
                       int* h_info, void* stream);

/* ---- f-4: the inverse after one query has been appended --------------------------------
 * FeedbackProcessing.update_X (src/feedback_processing.py:133-154) only ever APPENDS the m+1 rows of the
 * new query, so Sigma_new = [[Sigma_old, B], [B^T, C]] with Sigma_old unchanged.  Instead of the reference's
 * full refactorisation (update_Sigma_inv -> pd_inverse, src/gp_model.py:161-162) the Cholesky factor is
 * bordered and the inverse follows from it -- through L^-1 (condition sqrt(cond Sigma)), never through
 * Sigma_old^-1 B (bordering the explicit inverse loses cond(Sigma) * eps relative to the tiny Schur complement):
 *   Y = L11^-1 B,  S = C - Y^T Y = L22 L22^T,  X = L22^-1 Y^T L11^-1  (k x N1, k = N - N1 <= 64),
 *   L_new^-1 = [[L11^-1, 0], [-X, L22^-1]],
 *   Sigma_new^-1 = [[Sigma_old^-1 + X^T X, -X^T L22^-1], [-L22^-T X, L22^-T L22^-1]].
 * d_A[N,N] is the new matrix; d_A11inv / d_L11inv [N1,N1] (row stride N1) the inverse of its leading block and of
 * that block's Cholesky factor (from ppbo_pd_inverse_factors or a previous append); d_Ainv / d_Linv [N,N] the
 * results.  PPBO_ERR_NOT_PD (info = failing column of A) when S is not positive definite. */
PPBO_API int ppbo_pd_inverse_append(ppbo_ctx* ctx, const double* d_A, int N, const double* d_A11inv, const double* d_L11inv,
                           int N1, double* d_Ainv, double* d_Linv, int* h_info, void* stream);
/* ... that also borders the factor itself: d_L11[N1,N1] (row stride N1, lower triangle = factor of the leading block,
 * as ppbo_pd_inverse_ex or a previous call returned it) -> d_L[N,N] = [[L11, 0], [Y^T, L22]], so the appended design
 * can go straight into ppbo_fit_fmap_whitened.  d_L11 and d_L are both NULL or both given. */
PPBO_API int ppbo_pd_inverse_append_ex(ppbo_ctx* ctx, const double* d_A, int N, const double* d_A11inv, const double* d_L11inv,
                              const double* d_L11, int N1, double* d_Ainv, double* d_Linv, double* d_L, int* h_info,
                              void* stream);

/* ---- K5: Laplace terms of the projective-preference likelihood -----------
 * replaces sum_Phi/sum_Phi_vec (src/gp_model.py:176-218), the likelihood part of
 * T (:221-226), beta of T_grad (:234-238) and create_Lambda (:249-274).
 * Design contract (src/feedback_processing.py:110-130): N = n_q (m+1); row
 * q(m+1) is the observation of query q, the next m rows its pseudo-observations.
 * Outputs (any may be NULL): d_Tlik[1] = -(1/m) sum_q sum_j Phi(Delta_qj/sqrt2);
 * d_beta[N]; d_lam_diag[N], d_lam_off[N] = Lambda in star-graph form
 * (off[j] = Lambda[obs(j), j] for pseudo rows, 0 on observation rows). */
PPBO_API int ppbo_laplace_terms(ppbo_ctx* ctx, const double* d_f, int N, int m, double sigma,
                       double* d_Tlik, double* d_beta, double* d_lam_diag,
                       double* d_lam_off, void* stream);

/* sum_Phi_vec(order_of_derivative, f, sigma) of src/gp_model.py:206-218 (sum_Phi :176-204 is one element of it):
 * d_out[q] for the n_q = N/(m+1) queries, order 0 = sum_j Phi(Delta_qj/sqrt2) (closed form of the Gauss-Hermite
 * integral at :192), 1 = sum_j var2_normal_pdf(Delta_qj), 2 = sum_j -Delta_qj/2 var2_normal_pdf(Delta_qj);
 * any other order is an argument error (the reference prints and returns None). */
PPBO_API int ppbo_sum_phi(ppbo_ctx* ctx, const double* d_f, int N, int m, double sigma, int order, double* d_out,
                 void* stream);

/* ---- a-8: f_MAP by trust-region Newton -----------------------------------
 * replaces GPModel.update_fMAP's scipy.optimize.minimize(method='trust-exact')
 * (src/gp_model.py:354-389) for ONE start vector d_f_init[N].
 * Radius rules follow SciPy's trust-region driver (initial 1, max 1000,
 * eta 0.15); the subproblem is a More-Sorensen iteration on device Cholesky
 * factors.  Stops when |grad T|_2 < gtol or after maxiter outer iterations. */
typedef struct ppbo_fit_opts {
  double gtol;    /* reference default 1e-4 (SciPy); 100 during initialisation (src/gp_model.py:365-366) */
  int maxiter;    /* <=0: 200*N like SciPy */
  int verbose;
  double initial_radius;   /* first trust radius; <= 0: SciPy's default 1.0 (what the reference runs with) */
  int lbfgs_max_evals;     /* ppbo_fit_fmap_whitened: evaluation budget of the whitened pre-phase; <= 0: 4000 */
  int judge_by_gradient_below_noise;  /* 0 (what ppbo_fit_fmap's callers pass): SciPy trust-exact's acceptance and
                            * radius rules throughout (actual / predicted decrease).  1 (set internally by
                            * ppbo_fit_fmap_whitened for its finishing phase): a step whose predicted decrease is below
                            * the rounding noise of the objective difference is accepted when it lowers |grad| --
                            * next to the optimum actual / predicted is a random number */
  int start_is_whitened;   /* ppbo_gp_fit only: d_f_init holds z0 (e.g. the standard-normal draw itself) and the search
                            * starts at f = L z0 -- the reference's prior draw N(0, Sigma) (src/gp_model.py:374,381)
                            * without forming it and whitening it again */
} ppbo_fit_opts;
typedef struct ppbo_fit_stats {
  int iterations;   /* outer trust-region iterations */
  int n_cholesky;   /* factorizations attempted */
  int converged;    /* 1 if |grad| < gtol */
  double T;         /* T(f_MAP) (src/gp_model.py:221-226) */
  double gradnorm;  /* |grad T(f_MAP)|_2 */
  int lbfgs_iterations; /* ppbo_fit_fmap_whitened: accepted quasi-Newton steps of the whitened pre-phase (else 0) */
  int lbfgs_evals;      /* ... its objective/gradient evaluations (each O(N^2): two products with L, one with Sigma^-1) */
  int lbfgs_status;     /* ... how it ended: 1 |grad T| < gtol, 2 rounding floor, 3 line search failed, 4 non-finite
                         * start, 5 budget spent, 6 the factor does not exist (Sigma not positive definite: ppbo_gp_fit
                         * returns PPBO_ERR_NOT_PD); -1 when the pre-phase did not run */
} ppbo_fit_stats;
PPBO_API int ppbo_fit_fmap(ppbo_ctx* ctx, const double* d_Sigma_inv, int N, int m, double sigma,
                  const double* d_f_init, const ppbo_fit_opts* opts, double* d_fMAP,
                  ppbo_fit_stats* h_stats, void* stream);

/* ---- a-8, the production path: the same minimiser found in the prior-whitened variable ------------------
 * replaces the same call (src/gp_model.py:354-389, scipy trust-exact on f) for ONE start vector.
 * With Sigma = L L^T (d_L from ppbo_pd_inverse_ex, row stride ldl) and f = L z,
 *   -T(L z) = 1/2 |z|^2 + (1/m) sum_q sum_j Phi(Delta_qj / sqrt2)
 * has the Hessian I - L^T Lambda L, free of cond(Sigma) ~ 1e7: a device-resident L-BFGS (history 8, Armijo /
 * approximate-Wolfe acceptance) needs tens of O(N^2) evaluations where the trust-region Newton on f needs
 * tens of O(N^3) factorizations.  It stops on the reference's own rule |grad_f T|_2 < gtol; whatever is left
 * (a request below the rounding floor of the whitened iteration, a stalled line search) is finished by
 * ppbo_fit_fmap from the point reached, so the result satisfies exactly what ppbo_fit_fmap's does.
 * Same optimum as the reference on every golden fixture; the PATH (and hence, on a multi-modal posterior, which
 * local maximum is found) is not SciPy's. */
PPBO_API int ppbo_fit_fmap_whitened(ppbo_ctx* ctx, const double* d_L, int ldl, const double* d_Sigma_inv, int N, int m,
                           double sigma, const double* d_f_init, const ppbo_fit_opts* opts, double* d_fMAP,
                           ppbo_fit_stats* h_stats, void* stream);

/* ---- a-1 ... a-8 + the posterior in ONE call: what GPModel.update_model does between update_data and mu_star ----
 * replaces update_Sigma + update_Sigma_inv + update_fMAP (one trial from d_f_init) + create_Lambda + the posterior
 * covariance (src/gp_model.py:91-117) -- i.e. ppbo_gram, ppbo_pd_inverse_ex, ppbo_fit_fmap_whitened and
 * ppbo_posterior behind one entry, with what that allows:
 *   - the start is whitened with the factor's inverse that is at hand anyway (z0 = L^-1 f_init: one triangular product);
 *   - no host wait between the phases: the two factorizations' info words and the search's state are read once, at
 *     the end (the search itself is steered through a host-mapped progress word, not through stream synchronisation);
 *   - with start_is_whitened and N >= 1024 the triangular inverse and Sigma^-1 -- which the search does not need until
 *     its |grad_f T| rule is armed -- are formed on a second stream owned by the ctx beside the first evaluations; the
 *     search's stream joins it at a fixed slot (PPBO_FIT_GF_FROM, default 8), the first evaluation that may apply the
 *     |grad_f T| rule.  That slot -- not the stream layout -- is what the result depends on: with start_is_whitened and
 *     N >= 1024 the rule is armed from the same evaluation in the one-stream form (PPBO_FIT_OVERLAP=0) too, so the two
 *     forms agree bit for bit for every start; without start_is_whitened, or below N = 1024, the rule is armed from
 *     the first evaluation.  Everything (both streams) is complete when the call returns.
 *   - a Sigma that is not positive definite ends the search before its first evaluation (the factorization's info word
 *     is read on the device by the search's first launch): the search performs no evaluation and the call returns
 *     PPBO_ERR_NOT_PD without a posterior phase.  The inverse pipeline that was enqueued behind the factorization
 *     (triangular inverse, Sigma^-1, the start product) still runs to completion on the half-factored matrix; its
 *     outputs (d_Sigma_inv, d_L, d_Linv, d_fMAP) are undefined in that case.
 * Outputs (device, caller-owned): d_Sigma [N,N] (NULL to skip), d_Sigma_inv [N,N], d_L [N,N] (Cholesky factor of
 * Sigma, lower triangle valid), d_Linv [N,N] (NULL: kept in a workspace), d_fMAP [N], and -- all four or none --
 * d_alpha, d_lam_diag, d_lam_off [N], d_G [N,N] as ppbo_posterior defines them.
 * Returns PPBO_ERR_NOT_PD with *h_info = 1 when Sigma is not positive definite, *h_info = 2 when Sigma^-1 - Lambda_MAP
 * is not (f_MAP and the factors of Sigma are valid then; the reference prints its '---!!!---' line and keeps the
 * previous posterior, src/gp_model.py:118-120). */
PPBO_API int ppbo_gp_fit(ppbo_ctx* ctx, int kernel_id, const double* d_X, int N, int D, const double theta[3], double shrink,
                int m, const double* d_f_init, const ppbo_fit_opts* opts, double* d_Sigma, double* d_Sigma_inv,
                double* d_L, double* d_Linv, double* d_fMAP, double* d_alpha, double* d_lam_diag, double* d_lam_off,
                double* d_G, ppbo_fit_stats* h_stats, int* h_info, void* stream);

/* T(f) and grad T(f) for a given f (src/gp_model.py:221-240); h_T / d_grad may be NULL */
PPBO_API int ppbo_T_and_grad(ppbo_ctx* ctx, const double* d_Sigma_inv, const double* d_f, int N, int m,
                    double sigma, double* h_T, double* d_grad, void* stream);

/* ---- posterior state for prediction ---------------------------------------
 * replaces update_model's tail (src/gp_model.py:111-117) and the per-call
 * A = Sigma^-1 - Sigma^-1 P Sigma^-1 of mu_Sigma_pred (:449).  With
 * W = -Lambda_MAP, B = Sigma^-1 + W = L_B L_B^T, R = L_B^-1:
 *   d_alpha[N] = Sigma^-1 f_MAP;  d_G[N,N] = R W (block lower triangular);
 *   k*^T A k* = k*^T W k* - |G k*|^2   (Woodbury; identical operator).
 * d_P (optional, may be NULL) = posterior_covariance = B^-1 (src/gp_model.py:117).
 * Returns PPBO_ERR_NOT_PD when B is not positive definite (the reference prints
 * '---!!!--- Posterior covariance matrix is not PSD ---!!!---' and continues). */
PPBO_API int ppbo_posterior(ppbo_ctx* ctx, const double* d_Sigma_inv, const double* d_fMAP, int N, int m,
                   double sigma, double* d_alpha, double* d_lam_diag, double* d_lam_off,
                   double* d_G, double* d_P, int* h_info, void* stream);

/* ---- K2+K3+K4: batched candidate scoring with on-device argmax ------------
 * replaces mu_pred (src/gp_model.py:454-458) / diag of mu_Sigma_pred (:441-452)
 * called once per candidate by mu_star's differential evolution (:415-437) and
 * by EI/varmax (src/acquisition.py:72-81,170-178).
 * d_Xc[M,D] candidates in [0,1]^D.  Outputs (NULL to skip): d_mu[M], d_var[M],
 * d_score[M]; h_best_val / h_best_idx = max score and its FIRST index
 * (np.argmax semantics).  d_G may be NULL when only the mean is wanted.
 * Any N = n_q (m + 1), any m, any M: where N is not a multiple of the 128-row tile / 16-deep chunk of the variance
 * contraction (m = 25, the reference's default) the call works on a zero-framed copy of G in a ctx workspace (from
 * 2048 candidates on; ~3 N^2 x 8 bytes moved per call), and K* is always padded to whole 128-candidate tiles. */
typedef struct ppbo_model {
  int kernel_id, N, D, m;
  double theta[3];
  const double* d_X;        /* [N,D] */
  const double* d_alpha;    /* [N]   */
  const double* d_lam_diag; /* [N]  Lambda_MAP diagonal   */
  const double* d_lam_off;  /* [N]  Lambda_MAP star edges */
  const double* d_G;        /* [N,N] R W, see ppbo_posterior.  Block lower triangular, and stored that way: every
                             * entry right of the last star that reaches into its row must be an explicit ZERO
                             * (ppbo_posterior / ppbo_gp_fit write them): the contractions round their K ranges up to
                             * whole 16-column chunks and read up to 15 of those zeros per row */
  int kstar_fp32;           /* 0: everything fp64 (the product path).  1: K* entries evaluated in fp32 from direct
                             * differences, all accumulation fp64 -- BASELINE config 5's "fp32 tolerance" variant;
                             * its error against the fp64 path is REPORTED (bench.py), it does not meet 1e-5 */
  const double* d_Gt;       /* optional (ABI 6): the TRANSPOSE of d_G as ppbo_transposed_G writes it.  Models of up to ~500
                             * rows are scored by one launch (csrc/fused.hip) whose matrix-core loop reads G transposed;
                             * NULL: the library forms the transpose in a workspace on every call (2-4 us) */
} ppbo_model;

/* The transpose of G = R Lambda in the layout the one-launch scoring kernel reads: d_Gt[rows][ld] with
 * rows = (N rounded up to 16) + 16, ld = (N rounded up to 32) + 32 (ppbo_transposed_G_shape), d_Gt[k][i] = d_G[i][k] for
 * i, k < N and zero elsewhere.  Form it once per fit and hand it over as ppbo_model.d_Gt.  No reference counterpart (G
 * itself replaces the dense A = Sigma^-1 - Sigma^-1 P Sigma^-1 of src/gp_model.py:449). */
PPBO_API int ppbo_transposed_G_shape(int N, int* rows, int* ld);
PPBO_API int ppbo_transposed_G(ppbo_ctx* ctx, const double* d_G, int N, double* d_Gt, void* stream);
PPBO_API int ppbo_predict(ppbo_ctx* ctx, const ppbo_model* model, const double* d_Xc, int64_t M,
                 int score_kind, double mustar, double* d_mu, double* d_var, double* d_score,
                 double* h_best_val, int64_t* h_best_idx, void* stream);

/* The same scoring passes for ONE SHARD of a sharded candidate search: nothing comes back to the
 * host and nothing synchronises.  d_record[2] (device) receives (best score, index_offset + its FIRST row index as
 * a double, exact below 2^53) -- (NaN, -1) when no candidate has a non-NaN score -- i.e. exactly the 16-byte record
 * that ppbo_argmax_allgather_record / torch.distributed all-gather (written by the one-workgroup argmax launch that
 * ends the pass). */
PPBO_API int ppbo_predict_record(ppbo_ctx* ctx, const ppbo_model* model, const double* d_Xc, int64_t M,
                        int score_kind, double mustar, int64_t index_offset, double* d_record, void* stream);

/* full predictive covariance of small sets (the G=70 line grid of EI):
 * d_cov[M,M] = (1-s)K(Xc,Xc) + s sigma_f^2 I - K*^T A K*  (src/gp_model.py:447-450) */
PPBO_API int ppbo_predict_cov(ppbo_ctx* ctx, const ppbo_model* model, const double* d_Xc, int M,
                     double shrink, double* d_mu, double* d_cov, void* stream);

/* ---- f-2: posterior mean with its analytic gradient ----------------------------
 * d_mu[M] = K*^T alpha (src/gp_model.py:454-458), d_grad[M,D] = d mu / d x in the model's scaled
 * coordinates.  The reference has no gradient: mu_star (src/gp_model.py:415-437) maximises mu_pred by
 * differential evolution; the drop-in refines the best candidates of the batched search by a
 * multi-start ascent on these gradients instead.  Only kernel_id, N, D, theta, d_X, d_alpha of the
 * model are read. */
PPBO_API int ppbo_mean_grad(ppbo_ctx* ctx, const ppbo_model* model, const double* d_Xc, int64_t M,
                   double* d_mu, double* d_grad, void* stream);

/* ---- f-2, device-resident: the maximiser of the posterior mean ---------------------------------------
 * replaces GPModel.mu_star's differential evolution (src/gp_model.py:415-437: ~2 k ... 17 k sequential mu_pred
 * calls per trial) by one enqueue: score the M candidates (ppbo_predict, mean only), thin them to <= 4096 group
 * winners, pick the K best that are pairwise more than `sep` apart, and run a projected Barzilai-Borwein gradient
 * ascent from each -- the WHOLE iteration inside one kernel, one workgroup per start, no host round trip.
 * d_x[K,D] / d_mu[K]: the refined maxima (rows >= *h_found: mu = -inf).  h_found may be NULL (then nothing
 * synchronises).  ppbo_mean_ascent is the last stage alone, from caller-chosen starts (d_iters[K] optional).
 * ppbo_shift_points: d_out = frac(d_in + h_shift[D]) row-wise -- a rotation of a RESIDENT uniform candidate pool,
 * so that repeated searches see fresh candidates without regenerating and uploading M x D numbers. */
PPBO_API int ppbo_mean_search(ppbo_ctx* ctx, const ppbo_model* model, const double* d_cand, int64_t M, int K, double sep,
                     int iters, double tol, double* d_x, double* d_mu, int* h_found, void* stream);
PPBO_API int ppbo_mean_ascent(ppbo_ctx* ctx, const ppbo_model* model, const double* d_starts, int K, int iters, double tol,
                     double* d_x, double* d_mu, int* d_iters, void* stream);
PPBO_API int ppbo_shift_points(ppbo_ctx* ctx, const double* d_in, int64_t M, int D, const double* h_shift, double* d_out,
                      void* stream);

/* ALL trials of one mu_star call in one enqueue (src/gp_model.py:415-437: `mustar_finding_trials` differential-evolution
 * runs, 3 per iteration, 20 on the last).  Trial t scores the resident uniform pool d_pool[M,D] through its own rotation
 * frac(pool + h_shifts[t]) -- formed on the fly, nothing is written out -- and trial 0 also E_rows extra points
 * d_extra[E_rows,D] (NULL with E_rows = N: the model's own design points) and, when h_xprev[D] is given, the previous
 * x*; then, for all trials together: one thinning launch, one start-selection launch (a workgroup per trial), ONE ascent
 * launch of T K workgroups.  Against T calls of ppbo_mean_search on three contexts / streams: 3 trials at C3 2.2 -> 1.1 ms.
 * screen_fp32 = 1: the candidates are RANKED by a mean whose kernel values are evaluated in fp32 (packed fp32 math,
 * v_exp_f32; accumulated in fp64; relative error ~1e-6) -- every reported value and point comes from the fp64 ascent;
 * 0: ranked by the fp64 mean of ppbo_predict, as ppbo_mean_search does (bit-identical results to T such calls).
 * d_x[T,K,D], d_mu[T,K]: the refined maxima per trial (rows that found no start: mu = -inf).  Nothing synchronises. */
PPBO_API int ppbo_mean_search_multi(ppbo_ctx* ctx, const ppbo_model* model, const double* d_pool, int64_t M,
                           const double* h_shifts, int T, const double* d_extra, int E_rows, const double* h_xprev,
                           int K, double sep, int iters, double tol, int screen_fp32, double* d_x, double* d_mu,
                           void* stream);

/* ---- K10: Monte-Carlo line acquisition -------------------------------------
 * replaces EI / varmax (src/acquisition.py:72-81, 170-178) for B lines of G points
 * with stored standard-normal draws d_z[S,G]: f = mu + chol(cov) z.
 * d_grid[B,G,D] -> d_ei[B], d_varmax[B] (either may be NULL). */
PPBO_API int ppbo_line_acq(ppbo_ctx* ctx, const ppbo_model* model, const double* d_grid, int B, int G,
                  double shrink, const double* d_z, int S, double mustar, double jitter,
                  double* d_ei, double* d_varmax, void* stream);
/* The same with the grid points formed ON THE DEVICE: line b is {alpha[g] * d_xi[b,:] + d_x[b,:]}, g < G -- what
 * FeedbackProcessing.xi_grid(xi, x, 'equispaced', m = 70, is_scaled = True) (src/feedback_processing.py:57-107) returns
 * for the abscissae alpha the caller has drawn (the reference: 70 noisy-equispaced values in [0, 1] per EI call,
 * src/acquisition.py:72-75).  d_alpha[G] when alpha_per_line == 0 (one abscissa vector shared by all lines: common
 * random numbers), d_alpha[B,G] otherwise.  Nothing of size B x G x D crosses the host boundary: (xi, x, alpha) are
 * B x (2 D + G) numbers (EI-EXT at D = 20: 1000 lines, 110 KB instead of an 11 MB grid built by 1000 Python calls). */
PPBO_API int ppbo_line_acq_xi(ppbo_ctx* ctx, const ppbo_model* model, const double* d_xi, const double* d_x,
                     const double* d_alpha, int alpha_per_line, int B, int G, double shrink, const double* d_z, int S,
                     double mustar, double jitter, double* d_ei, double* d_varmax, void* stream);


/* standard normal draws for the Monte-Carlo acquisitions, generated on the device: d_out[n] = a pure function of
 * (seed, index) (Philox-4x32-10 + Box-Muller), i.e. reproducible and independent of the launch geometry.  Replaces
 * the np.random.multivariate_normal / standard_normal draws of src/acquisition.py:76,174 where the caller does not
 * need NumPy's own stream (the batched searches; EI() / varmax() called with explicit draws keep the caller's z). */
PPBO_API int ppbo_randn(ppbo_ctx* ctx, uint64_t seed, double* d_out, int64_t n, void* stream);

/* ---- K7/K8/K9: random Fourier features --------------------------------------
 * ppbo_rff_project replaces Hsampler.phiVec/update_phi_X
 *   (src/random_fourier_sampler.py:45-47,57-58): d_Phi[F,N] = sqrt(2 sf^2/F) cos(W X^T + b).
 * ppbo_rff_score replaces phi(x)^T omega (:166,170) batched over M candidates with
 *   an on-device argmax (Phi(Xc) never materialised).
 * ppbo_rff_terms replaces S, S_grad, diag(S_hessian) (:106-122); outputs may be NULL. */
PPBO_API int ppbo_rff_project(ppbo_ctx* ctx, const double* d_X, int N, int D, const double* d_W, int F,
                     const double* d_b, double sigma_f, double* d_Phi, void* stream);
PPBO_API int ppbo_rff_score(ppbo_ctx* ctx, const double* d_Xc, int64_t M, int D, const double* d_W, int F,
                   const double* d_b, double sigma_f, const double* d_omega, double* d_score,
                   double* h_best_val, int64_t* h_best_idx, void* stream);
PPBO_API int ppbo_rff_terms(ppbo_ctx* ctx, const double* d_Phi, int F, int N, int m, double sigma,
                   const double* d_omega, double* h_S, double* d_grad, double* d_hdiag, void* stream);

/* Hsampler.update_omega_MAP (src/random_fourier_sampler.py:124-132): maximise S from the start vector in d_omega[F]
 * (in: start, out: omega_MAP).  The reference hands -S to SciPy's trust-exact; S_hessian is diagonal, so the trust
 * region subproblem has the closed form s_i = g_i / (max(-h_i, 1e-12) + lam) with lam = 0 when the Newton step fits
 * and the More-Sorensen root of |s(lam)| = radius otherwise, under SciPy's radius rules (x1/4 below rho = 1/4, x2
 * above 3/4 on the boundary, accept above 0.15).  The whole loop is device-resident: omega, gradient, Hessian diagonal
 * AND the trust region's state stay on the device, one workgroup judges each trial and forms the next (four launches per
 * iteration), the host only keeps a few iterations enqueued ahead of a host-mapped progress word and reads S, |grad S|
 * and the iteration count once, at the end.  d_omega is written on `stream` (ordered for later work on that stream).
 * Stops on |grad S| < gtol (h_gradnorm belongs to the point returned), maxiter, or a collapsed radius. */
PPBO_API int ppbo_rff_omega_map(ppbo_ctx* ctx, const double* d_Phi, int F, int N, int m, double sigma, double* d_omega,
                       int maxiter, double gtol, double* h_S, double* h_gradnorm, int* h_iterations, void* stream);

/* the maximiser of ONE posterior sample phi(x)^T omega, device-resident: replaces Hsampler.return_xstar's 5-30
 * L-BFGS-B starts on NumPy phi / Dphi (src/random_fourier_sampler.py:143-176).  Scores the M candidates
 * (ppbo_rff_score), keeps the K best that are > sep apart and runs the whole projected Barzilai-Borwein ascent of
 * each inside one kernel with the analytic gradient -a sum_f omega_f sin(w_f.x + b_f) w_f (:51-53).
 * d_x[K,D] / d_val[K]: refined maxima (rows >= *h_found: value -inf). */
PPBO_API int ppbo_rff_search(ppbo_ctx* ctx, const double* d_cand, int64_t M, int D, const double* d_W, int F,
                    const double* d_b, double sigma_f, const double* d_omega, int K, double sep, int iters,
                    double tol, double* d_x, double* d_val, int* h_found, void* stream);

/* ---- generic fp64 MFMA GEMM (exposed for tests and host-side composition) ----
 * C[M,N] = alpha op(A) op(B) + beta C.  transA/transB: 0 = as stored, 1 = transposed. */
PPBO_API int ppbo_dgemm(ppbo_ctx* ctx, int transA, int transB, int M, int N, int K, double alpha,
               const double* d_A, int lda, const double* d_B, int ldb, double beta,
               double* d_C, int ldc, void* stream);

/* ---- a-9: the determinant term of the Laplace evidence ---------------------------------
 * ppbo_lu_slogdet replaces  P,L,U = scipy.linalg.lu(M); slogdet(P), slogdet(L), slogdet(U)
 * (src/gp_model.py:303-310): LU with partial pivoting (LAPACK pivot rule) IN PLACE on d_A;
 * returns *h_u_sign = prod sign(u_ii) and *h_u_logdet = sum log|u_ii|.  P and L contribute
 * sign*0, so the reference's sum of sign*logdet is exactly (*h_u_sign) * (*h_u_logdet).
 * *h_info = k>0 if u_kk == 0.
 * ppbo_laplace_logdet forms M = I + Sigma*Lambda (src/gp_model.py:301-302, plus sign as in the
 * reference) from the star-form Lambda and calls ppbo_lu_slogdet on it. */
PPBO_API int ppbo_lu_slogdet(ppbo_ctx* ctx, double* d_A, int N, int lda, double* h_u_sign, double* h_u_logdet,
                    int* h_info, void* stream);
PPBO_API int ppbo_laplace_logdet(ppbo_ctx* ctx, const double* d_Sigma, const double* d_lam_diag,
                        const double* d_lam_off, int N, int m, double* h_u_sign, double* h_u_logdet,
                        int* h_info, void* stream);

/* ---- (e) the path's one collective, over RCCL / xGMI ---------------------------------------------------
 * Candidate rows are sharded over one process per GPU; every rank scores its shard with
 * ppbo_predict and contributes (best score, GLOBAL row index).  ppbo_argmax_allgather is ONE ncclAllGather of a
 * 16-byte record per rank followed by a local reduction: largest value, ties to the smallest index
 * (np.argmax first-occurrence semantics), NaN or index < 0 never win.  librccl is dlopen'ed on first use.
 *   rank 0:  ppbo_dist_unique_id(ctx, id)  -> ship the 128 bytes to the other ranks by any means (file, MPI, env)
 *   all:     ppbo_dist_init(ctx, id, rank, world)   (collective; the ctx's device is the rank's GPU)
 *   search:  ppbo_argmax_allgather(ctx, local_val, local_idx + shard_offset, &val, &idx, stream)
 * Return codes 2000 + ncclResult_t for RCCL failures. */
PPBO_API int ppbo_dist_unique_id(ppbo_ctx* ctx, void* h_id128);
PPBO_API int ppbo_dist_init(ppbo_ctx* ctx, const void* h_id128, int rank, int world);
PPBO_API int ppbo_dist_destroy(ppbo_ctx* ctx);
PPBO_API int ppbo_argmax_allgather(ppbo_ctx* ctx, double local_val, int64_t local_global_idx, double* h_best_val,
                          int64_t* h_best_idx, void* stream);
/* the same exchange fed from DEVICE memory (d_record[2] as written by ppbo_predict_record): no host value is
 * uploaded first; all-gather, reduction and the 16-byte read-back run behind each other on `stream`. */
PPBO_API int ppbo_argmax_allgather_record(ppbo_ctx* ctx, const double* d_record, double* h_best_val, int64_t* h_best_idx,
                                 void* stream);
/* one whole sharded search step in ONE call: ppbo_predict_record on this rank's M rows (global row index =
 * index_offset + local row), ncclAllGather of the records, reduction, one 16-byte read-back, ONE host wait.
 * Every rank returns the job-wide (best score, global index).  Without ppbo_dist_init (a single-process search)
 * the collective is skipped.  Replaces the sequential search of mu_star (src/gp_model.py:415-437) over a sharded
 * candidate set; the reference itself is process-per-run (ppbo_numerical_main.py:192-193). */
PPBO_API int ppbo_search_sharded(ppbo_ctx* ctx, const ppbo_model* model, const double* d_Xc, int64_t M, int score_kind,
                        double mustar, int64_t index_offset, double* h_best_val, int64_t* h_best_idx, void* stream);
/* the reduction alone, for callers that run the all-gather themselves (torch.distributed in ppbo_amd/dist.py):
 * d_records[world][2] = (value, global index as a double) per rank, already gathered in device memory; one
 * single-wavefront kernel applies the rule above and ONE 16-byte record is copied back. */
PPBO_API int ppbo_argmax_combine(ppbo_ctx* ctx, const double* d_records, int world, double* h_best_val, int64_t* h_best_idx,
                        void* stream);

/* y = op(A) x for a square fp64 matrix; lower != 0 reads only the lower triangle (A is then
 * treated as lower-triangular).  Used for alpha = Sigma^-1 f_MAP (src/gp_model.py:445) and
 * prior draws L z (src/gp_model.py:374). */
PPBO_API int ppbo_dgemv(ppbo_ctx* ctx, int trans, int lower, int N, const double* d_A, int lda,
               const double* d_x, double* d_y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PPBO_HIP_H */
