// K5 + a-8: Laplace terms of the projective-preference likelihood, f_MAP by
// trust-region Newton, and the posterior state used by prediction.
//   reference: gp_model.py:176-274 (sum_Phi, T, T_grad, create_Lambda),
//              :354-389 (update_fMAP -> scipy trust-exact), :111-117 (posterior covariance).
//
// Likelihood terms: Phi(Delta/sqrt2) = erfc(-Delta/2)/2 is the closed form of the
// reference's Gauss-Hermite-200 quadrature (gp_model.py:192; agrees to 3e-16).
// One wavefront per query (star): lanes = pseudo-observations, wave-shuffle sums.
//
// Trust region: radius rules of SciPy's driver (initial 1, max 1000, eta 0.15,
// x1/4 when rho<1/4, x2 when rho>3/4 on the boundary); subproblem by the
// More-Sorensen iteration on lam (Conn/Gould/Toint Alg. 7.3.4 without hard-case
// refinement): each trial lam = one device Cholesky (potrf) + triangular inverse
// (trtri) + three GEMVs.  The host loop only reads a handful of scalars per trial.
#include <chrono>

#include "linalg.h"

namespace {

constexpr double INV_SQRT_4PI = 0.28209479177387814347;

__global__ __launch_bounds__(256) void laplace_kernel(const double* __restrict__ f, int N, int mblk, int n_q,
                                                      double sigma, double* __restrict__ tq,
                                                      double* __restrict__ beta, double* __restrict__ lam_diag,
                                                      double* __restrict__ lam_off) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= n_q) return;
  const int m = mblk - 1;
  const int i = q * mblk;
  const double f0 = f[i];
  const double c = 1.0 / ((double)m * (sigma * sigma));
  const double bsc = sigma * (double)m;
  double sphi = 0.0, sp2 = 0.0, sw = 0.0;
  for (int r = 1 + lane; r <= m && i + r < N; r += 64) {
    const int j = i + r;
    const double delta = (f[j] - f0) / sigma;
    sphi += 0.5 * erfc(-0.5 * delta);
    const double p2 = INV_SQRT_4PI * exp(-0.25 * (delta * delta));
    const double w = 0.5 * c * delta * p2;
    sp2 += p2;
    sw += w;
    if (beta) beta[j] = -p2 / bsc;
    if (lam_diag) lam_diag[j] = w;
    if (lam_off) lam_off[j] = -w;
  }
  sphi = wave_sum(sphi);
  sp2 = wave_sum(sp2);
  sw = wave_sum(sw);
  if (lane == 0) {
    if (tq) tq[q] = sphi;
    if (beta) beta[i] = sp2 / bsc;
    if (lam_diag) lam_diag[i] = sw;
    if (lam_off) lam_off[i] = 0.0;
  }
}

// sum_Phi_vec of gp_model.py:206-218 as its own operator: out[q] = sum_j Phi^(order)((f[q(m+1)+j] - f[q(m+1)]) / sigma)
// in the reference's conventions (gp_model.py:176-204): order 0 = Phi(Delta/sqrt2) (closed form of the
// Gauss-Hermite cross-correlation integral), 1 = var2_normal_pdf(Delta), 2 = -Delta/2 var2_normal_pdf(Delta).
__global__ __launch_bounds__(256) void sum_phi_kernel(const double* __restrict__ f, int N, int mblk, int n_q, double sigma,
                                                      int order, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= n_q) return;
  const int i = q * mblk;
  const double f0 = f[i];
  double acc = 0.0;
  for (int r = 1 + lane; r < mblk && i + r < N; r += 64) {
    const double delta = (f[i + r] - f0) / sigma;
    if (order == 0) acc += 0.5 * erfc(-0.5 * delta);
    else {
      const double p2 = INV_SQRT_4PI * exp(-0.25 * (delta * delta));
      acc += (order == 1) ? p2 : -0.5 * delta * p2;
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) out[q] = acc;
}

// out[0] = -(1/m) sum_q tq[q]
__global__ __launch_bounds__(1024) void tlik_reduce_kernel(const double* __restrict__ tq, int n_q, int m,
                                                           double* __restrict__ out) {
  __shared__ double sh[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < n_q; i += 1024) s += tq[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += sh[w];
    *out = -t / (double)m;
  }
}

// g = v - beta (gradient of phi = -T);  out[0] = f.v, out[1] = |g|^2 ; optional grad_T = -g
__global__ __launch_bounds__(1024) void grad_kernel(const double* __restrict__ f, const double* __restrict__ v,
                                                    const double* __restrict__ beta, int N,
                                                    double* __restrict__ g, double* __restrict__ gradT,
                                                    double* __restrict__ out) {
  __shared__ double sh[2][16];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < N; i += 1024) {
    const double gi = v[i] - beta[i];
    if (g) g[i] = gi;
    if (gradT) gradT[i] = -gi;
    a += f[i] * v[i];
    b += gi * gi;
  }
  a = wave_sum(a);
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0.0, sb = 0.0;
    for (int w = 0; w < 16; ++w) { sa += sh[0][w]; sb += sh[1][w]; }
    out[0] = sa;
    out[1] = sb;
  }
}

// H = Sinv - Lambda + lam I; optional per-row Gershgorin statistics of the UNSHIFTED H
__global__ __launch_bounds__(256) void form_shifted_kernel(const double* __restrict__ Sinv, int N, int mblk,
                                                           const double* __restrict__ lam_diag,
                                                           const double* __restrict__ lam_off, double lam,
                                                           double* __restrict__ H, double* __restrict__ rowstats) {
  __shared__ double sh[2][4];
  __shared__ double shd[4];
  const int i = blockIdx.x;
  const int q0 = (i / mblk) * mblk;      // observation row of i's star
  const bool is_obs = (i == q0);
  const double* src = Sinv + (size_t)i * N;
  double* dst = H + (size_t)i * N;
  double rs = 0.0, sq = 0.0, dg = 0.0;
  for (int j = threadIdx.x; j < N; j += 256) {
    double h = src[j];
    if (j == i) h -= lam_diag[i];
    else if (is_obs && j > q0 && j < q0 + mblk) h -= lam_off[j];
    else if (!is_obs && j == q0) h -= lam_off[i];
    sq += h * h;
    if (j == i) { dg = h; h += lam; }
    else rs += fabs(h);
    store_through(dst + j, h);
  }
  if (rowstats) {
    rs = wave_sum(rs);
    sq = wave_sum(sq);
    dg = wave_sum(dg);  // exactly one lane holds the diagonal, the rest 0
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = rs; sh[1][w] = sq; shd[w] = dg; }
    __syncthreads();
    if (threadIdx.x == 0) {
      rowstats[i] = shd[0] + shd[1] + shd[2] + shd[3];
      rowstats[N + i] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
      rowstats[2 * N + i] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    }
  }
}

// out: [0] min diag, [1] max(diag+rs), [2] min(diag-rs), [3] fro^2, [4] max(|diag|+rs)
__global__ __launch_bounds__(1024) void stats_reduce_kernel(const double* __restrict__ rowstats, int N,
                                                            double* __restrict__ out) {
  __shared__ double sh[5][16];
  double mn = INFINITY, gmax = -INFINITY, gmin = INFINITY, fro = 0.0, inf = 0.0;
  for (int i = threadIdx.x; i < N; i += 1024) {
    const double d = rowstats[i], rs = rowstats[N + i];
    mn = fmin(mn, d);
    gmax = fmax(gmax, d + rs);
    gmin = fmin(gmin, d - rs);
    fro += rowstats[2 * N + i];
    inf = fmax(inf, fabs(d) + rs);
  }
  mn = -wave_max(-mn);
  gmax = wave_max(gmax);
  gmin = -wave_max(-gmin);
  fro = wave_sum(fro);
  inf = wave_max(inf);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[0][w] = mn; sh[1][w] = gmax; sh[2][w] = gmin; sh[3][w] = fro; sh[4][w] = inf; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 16; ++k) {
      sh[0][0] = fmin(sh[0][0], sh[0][k]);
      sh[1][0] = fmax(sh[1][0], sh[1][k]);
      sh[2][0] = fmin(sh[2][0], sh[2][k]);
      sh[3][0] += sh[3][k];
      sh[4][0] = fmax(sh[4][0], sh[4][k]);
    }
    for (int k = 0; k < 5; ++k) out[k] = sh[k][0];
  }
}

// p = -u ; fn = f + p ; out: [0]=|p|^2 [1]=g.p   (u = H_lam^-1 g)
__global__ __launch_bounds__(1024) void step_kernel(const double* __restrict__ u, const double* __restrict__ g,
                                                    const double* __restrict__ f, int N, double* __restrict__ p,
                                                    double* __restrict__ fn, double* __restrict__ out) {
  __shared__ double sh[2][16];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < N; i += 1024) {
    const double pi = -u[i];
    p[i] = pi;
    fn[i] = f[i] + pi;
    a += pi * pi;
    b += g[i] * pi;
  }
  a = wave_sum(a);
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0.0, sb = 0.0;
    for (int w = 0; w < 16; ++w) { sa += sh[0][w]; sb += sh[1][w]; }
    out[0] = sa;
    out[1] = sb;
  }
}

// out[0] = sum_i p_i (v1_i + v2_i): the change of the quadratic term f'Sf between f and f+p is
// p'(Sf) + 1/2 p'Sp = 1/2 p'(v1+v2) -- evaluated from SMALL quantities instead of subtracting two
// O(1e1) numbers that each carry cond(Sigma)*eps of summation noise.
__global__ __launch_bounds__(1024) void pdot2_kernel(const double* __restrict__ p, const double* __restrict__ v1,
                                                     const double* __restrict__ v2, int N, double* __restrict__ out) {
  __shared__ double sh[16];
  double a = 0.0;
  for (int i = threadIdx.x; i < N; i += 1024) a += p[i] * (v1[i] + v2[i]);
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += sh[w];
    *out = t;
  }
}

// G[i][j] = (R Lambda)[i][j] for the star-structured Lambda.  One thread per entry, consecutive threads on
// consecutive columns of one row (coalesced; the first version gave each thread a whole star of one row, i.e. a
// 16 KB stride between neighbouring lanes: 74 us for 2 x 33 MB).  Pseudo-observation column j of star q0:
// r_j lam_jj + r_q0 lam_q0j; observation column q0: r_q0 lam_q0q0 + sum_k r_k lam_kq0 over its star.
__global__ __launch_bounds__(256) void g_build_kernel(const double* __restrict__ R, int N, int mblk,
                                                      const double* __restrict__ lam_diag,
                                                      const double* __restrict__ lam_off, double* __restrict__ G) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const double* r = R + (size_t)blockIdx.y * N;
  // R is lower triangular and only its 64 x 64 diagonal blocks carry written zeros above the diagonal (the triangular
  // inverse skipped the memset of the rest): right of this row's block everything counts as zero, unread
  const int lim = (((int)blockIdx.y >> 6) + 1) << 6;
  if (mblk == 32 && (N & 31) == 0) {
    // a star is exactly one aligned half-wavefront: the observation column's sum over its star comes from the
    // other 31 lanes by shuffles instead of a 31-iteration loop in one lane (which held the whole wavefront:
    // 125 us for 2 x 33 MB at N = 2048)
    const bool in = j < N;            // N % 32 == 0: a half-wavefront is entirely inside or entirely outside
    const double rj = (in && j < lim) ? r[j] : 0.0, ld = in ? lam_diag[j] : 0.0, lo = in ? lam_off[j] : 0.0;
    const bool obs = (j & 31) == 0;
    double t = obs ? rj * ld : rj * lo;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    const double r0 = __shfl(rj, (threadIdx.x & 63) & ~31, 64);
    if (in) G[(size_t)blockIdx.y * N + j] = obs ? t : rj * ld + r0 * lo;
    return;
  }
  auto rr = [&](int k) { return k < lim ? r[k] : 0.0; };
  // any other star size up to 64 rows (m = 25, the reference's default: 26): the products r_k lam_off_k of this
  // workgroup's 256 columns and of the 64 behind them go through LDS once, so that an observation column adds up its
  // star from there in the same order (the loop over global memory below, kept for larger stars, held its whole wavefront for
  // mblk dependent round trips: ~100 us per posterior at N = 2080)
  __shared__ double sp[256 + 64];
  const bool via_lds = mblk <= 64;
  if (via_lds) {
    const int jb = blockIdx.x * 256;
    for (int t = threadIdx.x; t < 256 + 64; t += 256) {
      const int k = jb + t;
      sp[t] = (k < N) ? rr(k) * lam_off[k] : 0.0;
    }
    __syncthreads();
  }
  if (j >= N) return;
  const int q0 = (j / mblk) * mblk;
  double acc = rr(j) * lam_diag[j];
  if (j != q0) {
    acc += rr(q0) * lam_off[j];
  } else {
    const int end = (q0 + mblk < N) ? q0 + mblk : N;
    if (via_lds) {
      const double* spj = sp + threadIdx.x;
      for (int k = 1; k < end - q0; ++k) acc += spj[k];
    } else {
      for (int k = q0 + 1; k < end; ++k) acc += rr(k) * lam_off[k];
    }
  }
  G[(size_t)blockIdx.y * N + j] = acc;
}

struct Vecs {  // one evaluation point
  double *f, *v, *beta, *ld, *lo, *g;
  double fSf, gn2, Tlik;
  double pv;   // p.(v_prev + v_this) when evaluated as a trial point
};

// p given (device): fn = f + p ; out: [0]=|p|^2 [1]=g.p
__global__ __launch_bounds__(1024) void restep_kernel(const double* __restrict__ p, const double* __restrict__ g,
                                                      const double* __restrict__ f, int N, double* __restrict__ fn,
                                                      double* __restrict__ out) {
  __shared__ double sh[2][16];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < N; i += 1024) {
    const double pi = p[i];
    fn[i] = f[i] + pi;
    a += pi * pi;
    b += g[i] * pi;
  }
  a = wave_sum(a);
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0.0, sb = 0.0;
    for (int w = 0; w < 16; ++w) { sa += sh[0][w]; sb += sh[1][w]; }
    out[0] = sa;
    out[1] = sb;
  }
}

// ---------------------------------------------------------------------------------------------------
// Whitened quasi-Newton pre-phase of the f_MAP search (ppbo_fit_fmap_whitened).
//
// With Sigma = L L^T and f = L z the objective phi(z) = -T(L z) = 1/2 |z|^2 + (1/m) sum Phi(Delta/sqrt2) has
// the Hessian I - L^T Lambda L: cond(Sigma) ~ 1e7 (what forces the reference, and ppbo_fit_fmap, into an exact
// trust-region Newton with one N^3/3 factorization per trial shift) is gone from it, and L-BFGS reaches the
// optimum in tens of O(N^2) evaluations:  f = L z,  beta(f),  grad_z phi = z - L^T beta.
// The whole recurrence is device-resident.  One SLOT = {f = L zt; Laplace terms; u = L^T beta; v = Sigma^-1 f
// (gated); lbfgs_step_kernel}; the step kernel (ONE workgroup) judges the trial point, updates the pair history,
// forms the next direction by the vector-free two-loop recursion on the Gram matrix of the basis
// {s_0..s_{H-1}, y_0..y_{H-1}, g} and writes the next trial point.  The host enqueues several slots per
// read-back; every kernel of a slot is gated on the state's status word, so the slots behind the one that
// finishes are no-ops.  The stopping rule is the reference's own: |grad_f T|_2 = |Sigma^-1 f - beta| < gtol
// (src/gp_model.py:382-384, SciPy's default gtol), evaluated only once |grad_z| is small enough for it to be
// possible (|grad_f| >= |grad_z| / |L|_F).
constexpr int LB_H = 8;                 // history pairs
constexpr int LB_NB = 2 * LB_H + 1;     // basis vectors: s ring [0,H), y ring [H,2H), current gradient 2H
constexpr int LB_T = 1024;              // threads of the step workgroup: two elements each at N = 2048
constexpr int LB_MAX_BACKTRACK = 12;

struct WhState {
  int status;      // 0 running, 1 converged (|grad_f| < gtol), 2 stagnated at the rounding floor, 3 line search
                   // failed along steepest descent, 4 non-finite objective at the start, 5 evaluation budget spent,
                   // 6 the factor L does not exist (its factorization's info word was non-zero)
  int evals, iters, hist, head, first, ls, stall, need_gf, max_evals;
  int gf_avail;    // 0: Sigma^-1 f is not available to the judgement (|grad_f| cannot be asked for); always 1 today
  int gf_from;     // the |grad_f| rule is armed from this evaluation on (ppbo_gp_fit with the side stream: Sigma^-1 is
                   // guaranteed from that slot on -- a fixed number, so the result does not depend on timing)
  double phi, dphi, alpha, gz2, gf2, gate, gtol2, gzbest;
  double B[LB_NB * LB_NB];
  double delta[LB_NB];   // coefficients (over the basis) of the direction behind the current trial point
};

// rowsq[i] = sum_{k <= i} L[i][k]^2 (= Sigma_ii); their sum is |L|_F^2 >= lambda_max(Sigma)
// (also zeroes `nzero` doubles at `zero`: the search's state and basis -- as a memset of its own that was a 7-us blit
// kernel and an 11-us gap in front of it, in every fit)
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const double* __restrict__ L, int N, int ldl,
                                                         double* __restrict__ rowsq, double* __restrict__ zero,
                                                         size_t nzero) {
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < nzero; k += (size_t)gridDim.x * 256) zero[k] = 0.0;
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const double* row = L + (size_t)i * ldl;
  double s = 0.0;
  for (int k = lane; k <= i; k += 64) s += row[k] * row[k];
  s = wave_sum(s);
  if (lane == 0) rowsq[i] = s;
}

// K sums over the NT threads of the workgroup, each thread holding K partials.  Quad sums by DPP (VALU only), one
// lane per quad parks them in LDS ([K][NT/4]), 32 workers per value add NT/128 consecutive partials, one thread per
// value the 32 results.  The first version reduced every value with six shuffle steps per wavefront: 16 waves x 25
// values x 6 ds_bpermute pairs through ONE LDS pipe took 12 of the judgement kernel's 21 us.
template <int K, int NT>
__device__ __forceinline__ void lb_block_sum(double (&v)[K], double* __restrict__ red, double* __restrict__ red2,
                                             double* __restrict__ out) {
  constexpr int P = NT / 4, C = 32, PER = P / C;
  static_assert(P % C == 0 && K * C <= NT, "block-sum geometry");
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = quad_sum_dpp(v[k]);
  if ((tid & 3) == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) red[k * P + (tid >> 2)] = v[k];
  }
  __syncthreads();
  if (tid < K * C) {
    const int k = tid / C, c = tid % C;
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < PER; ++i) t += red[k * P + c * PER + i];
    red2[k * C + c] = t;
  }
  __syncthreads();
  if (tid < K) {
    double t = 0.0;
#pragma unroll
    for (int c = 0; c < C; ++c) t += red2[tid * C + c];
    out[tid] = t;
  }
  __syncthreads();
}

// The scalar head of WhState (everything before B), as the step works on it in LDS: global memory is read once at
// the start and written once at the end -- a thread that loads, computes, stores and loads again from global memory
// pays ~1 us per dependent access, which made the first version of this function take 25-45 us.
struct WhHead {
  int status, evals, iters, hist, head, first, ls, stall, need_gf, max_evals, gf_avail, gf_from;
  double phi, dphi, alpha, gz2, gf2, gate, gtol2, gzbest;
};
// Progress of the search as the host sees it: the step kernel stores (status << 32 | evals) into host-mapped memory
// after every judgement (system scope), and the scalar head once the search has ended (before the final progress
// word).  The host polls the word between enqueues: it never runs more than a few slots ahead of the device, stops
// enqueueing as soon as the search is over, and needs neither a device-to-host copy nor a stream synchronisation.
struct WhProgress {
  unsigned long long* word;   // device view of the host-mapped progress word
  int* head;                  // device view of the host-mapped copy of WhHead (written when status != 0)
};
__device__ __forceinline__ void wh_publish(const WhProgress& pr, const WhHead& hs) {
  if (!pr.word) return;
  if (hs.status != 0) {
    constexpr int HWI = (int)(sizeof(WhHead) / 4);
    const int* src = reinterpret_cast<const int*>(&hs);
    for (int k = 0; k < HWI; ++k) pr.head[k] = src[k];
  }
  __hip_atomic_store(pr.word, ((unsigned long long)(unsigned)hs.status << 32) | (unsigned)hs.evals, __ATOMIC_RELEASE,
                     __HIP_MEMORY_SCOPE_SYSTEM);
}
static_assert(sizeof(WhHead) == offsetof(WhState, B), "WhHead mirrors the head of WhState");

__global__ __launch_bounds__(LB_T) void lbfgs_init_kernel(WhState* __restrict__ st, const double* __restrict__ rowsq,
                                                          int N, double gtol, int max_evals, int gf_avail, int gf_from,
                                                          const int* __restrict__ factor_info, WhProgress prog,
                                                          const double* __restrict__ z0, double* __restrict__ zt) {
  __shared__ double sh[LB_T / 64];
  double s = 0.0;
  for (int i = threadIdx.x; i < N; i += LB_T) {
    s += rowsq[i];
    if (z0) zt[i] = z0[i];              // a start handed over in the whitened variable: the first trial point (was a copy of its own)
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < LB_T / 64; ++w) t += sh[w];
    st->status = 0; st->evals = 0; st->iters = 0; st->hist = 0; st->head = 0; st->first = 1; st->ls = 0;
    st->stall = 0; st->need_gf = 0; st->max_evals = max_evals; st->gf_avail = gf_avail; st->gf_from = gf_from;
    st->phi = 0.0; st->dphi = 0.0; st->alpha = 0.0; st->gz2 = 0.0; st->gf2 = -1.0;
    st->gate = gtol * sqrt(t);          // |grad_f| < gtol needs |grad_z| < gtol |L|_F
    st->gtol2 = gtol * gtol;
    st->gzbest = INFINITY;
    // the factor the search walks on does not exist (potrf stopped at a non-positive pivot and left its info word on
    // the device): end the search before its first slot, so that the host hears of it at once instead of running the
    // whole pipeline on a half-factored matrix
    if (factor_info && factor_info[0] != 0) {
      st->status = 6;
      WhHead hs = *reinterpret_cast<const WhHead*>(st);
      wh_publish(prog, hs);
    }
  }
}


// One judgement of the trial point zt (its products u = L^T beta(L zt), v = Sigma^-1 L zt and the per-query
// likelihood sums tq are already in memory).  tests/probes/whitened_lbfgs_proto.py is the NumPy statement of the
// recurrence.  TWO passes over the N-vectors and one block reduction:
//   pass 1   objective, |grad_f|^2, the pair's curvature numbers s.y, s.s, y.y, the slope g_t.d, and the 17 + 1
//            inner products of the trial gradient with the basis and with itself;
//   scalars  accept / backtrack; on acceptance the Gram matrix of the NEW basis follows from those 18 products and
//            the old matrix by algebra (s_new = alpha d is a combination of the old basis: s_new.b = alpha (B delta),
//            y_new.b = g_t.b - g.b) -- "vector-free" L-BFGS; then the two-loop recursion on coefficients;
//   pass 2   commit z, g and the pair, form the next direction and the next trial point.
template <int NT>
__device__ __forceinline__ void lbfgs_step(WhState* __restrict__ st, int N, int m, int n_q, double* __restrict__ z,
                                           double* __restrict__ zt, double* __restrict__ d, const double* u,
                                           const double* v, const double* beta, const double* tq,
                                           double* __restrict__ basis, long long* dbg = nullptr,
                                           WhProgress prog = WhProgress{nullptr, nullptr},
                                           const double* __restrict__ dots = nullptr, int n_parts = 0) {
  // verbose >= 2: wall_clock64 stamps of the phases of evaluations 3..9 (tools/fit_whitened.py <cfg> <gtol> 2)
#define LSTAMP(k) do { if (dbg && threadIdx.x == 0) dbg[k] = wall_clock64(); } while (0)
  constexpr int HW = (int)(sizeof(WhHead) / 4);
  constexpr int NA = 7 + LB_NB + 1;      // zz, tsum, gf2, sy, ss, yy, gt.d | gt.b_l | gt.gt
  __shared__ double shB[LB_NB * LB_NB];
  __shared__ double red[NA * (NT / 4)];
  __shared__ double red2[NA * 32];
  __shared__ double out[NA];
  __shared__ double delta[LB_NB];        // coefficients of the direction that produced the trial point (old basis)
  __shared__ double bdel[LB_NB];         // B delta
  __shared__ double shs[4];
  __shared__ int act[4];
  __shared__ WhHead hs;
  const int tid = threadIdx.x;
  if (tid < HW) reinterpret_cast<int*>(&hs)[tid] = reinterpret_cast<const int*>(st)[tid];
  for (int i = tid; i < LB_NB * LB_NB; i += NT) shB[i] = st->B[i];
  if (tid < LB_NB) delta[tid] = st->delta[tid];
  __syncthreads();
  if (hs.status != 0) return;
  if (dbg) dbg += 16 * (hs.evals < 15 ? hs.evals : 15);
  LSTAMP(1);
  const int first = hs.first, need_gf = hs.need_gf;
  double* gcur = basis + (size_t)(2 * LB_H) * N;
  const double c1 = 1e-4, c2 = 0.9, eps_f = 1e-13;
  // ---- pass 1: the inner products.  When the launch that finished u left them as rows of partial sums (PpboDotsOut:
  // one row per workgroup of that launch), they are only added up here -- 32 lanes per value, fixed order -- instead
  // of being formed by this ONE workgroup from 23 N-vectors (4.6 + 2.6 us of the kernel's ~17 at N = 2048)
  if (dots && n_parts > 0) {
    static_assert(8 + LB_NB <= 31 && NT == 1024, "32 values x 32 lanes");
    __shared__ double stage[32];
    const int k = tid >> 5, l32 = tid & 31;
    double sacc = 0.0;
    if (k < 8 + LB_NB) {
      for (int p = l32; p < n_parts; p += 32) sacc += dots[(size_t)p * PPBO_DOTS_STRIDE + k];
    } else if (k == 31) {
      for (int q = l32; q < n_q; q += 32) sacc += tq[q];
    }
    const double rs = dpp_add(dpp_add(dpp_add(dpp_add(sacc, 0), 1), 2), 3);       // every lane: the sum of its row of 16
    double r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      r[q] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rs), 16 * q),
                              __builtin_amdgcn_readlane(__double2loint(rs), 16 * q));
    if ((tid & 63) == 0) { stage[2 * (tid >> 6)] = r[0] + r[1]; stage[2 * (tid >> 6) + 1] = r[2] + r[3]; }
    __syncthreads();
    if (tid < NA) {
      // out: zz, tsum, gf2, sy, ss, yy, gt.d | gt.b_l | gt.gt   <-   row: zz, gf2, sy, ss, yy, gt.d, gt.gt, -, gt.b_l
      int src;
      if (tid == 0) src = 0;
      else if (tid == 1) src = 31;
      else if (tid < 7) src = tid - 1;
      else if (tid < 7 + LB_NB) src = 8 + (tid - 7);
      else src = 6;
      out[tid] = stage[src];
    }
    __syncthreads();
    LSTAMP(2);
    LSTAMP(3);
  } else {
  double acc[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = 0.0;
  for (int i = tid; i < N; i += NT) {
    const double zi = zt[i], gt = zi - u[i];
    acc[0] += zi * zi;
    if (need_gf) { const double gf = v[i] - beta[i]; acc[2] += gf * gf; }
    if (!first) {
      const double s = zi - z[i], y = gt - gcur[i];
      acc[3] += s * y; acc[4] += s * s; acc[5] += y * y; acc[6] += gt * d[i];
#pragma unroll
      for (int l = 0; l < LB_NB; ++l) acc[7 + l] += gt * basis[(size_t)l * N + i];
    }
    acc[7 + LB_NB] += gt * gt;
  }
  for (int q = tid; q < n_q; q += NT) acc[1] += tq[q];
  LSTAMP(2);
  lb_block_sum<NA, NT>(acc, red, red2, out);
  LSTAMP(3);
  }
  // ---- the judgement (thread 0)
  if (tid == 0) {
    const double phi_t = 0.5 * out[0] + out[1] / (double)m;
    const double phi = hs.phi, dphi = hs.dphi, alpha = hs.alpha;
    hs.evals += 1;
    int a = 1;           // 0 backtrack, 1 accept, 2 restart along steepest descent, 3 stop
    int pair_ok = 0;
    const bool finite = isfinite(phi_t);
    if (!first) {
      const double dphi_t = out[6];
      const bool armijo = phi_t <= phi + c1 * alpha * dphi;
      const bool approx_wolfe = phi_t <= phi + eps_f * fmax(1.0, fabs(phi)) && (2.0 * c1 - 1.0) * dphi >= dphi_t &&
                                dphi_t >= c2 * dphi;
      if (!(finite && (armijo || approx_wolfe))) a = 0;
    } else if (!finite) {
      hs.status = 4;
      a = 3;
    }
    if (a == 0) {
      hs.ls += 1;
      if (hs.evals >= hs.max_evals) { hs.status = 5; a = 3; }
      else if (hs.ls > LB_MAX_BACKTRACK) {
        if (hs.hist > 0) {            // forget the history, retry from the accepted point along -g
          hs.hist = 0; hs.ls = 0;
          const double gz2 = shB[(2 * LB_H) * LB_NB + 2 * LB_H];
          hs.dphi = -gz2;
          hs.alpha = fmin(1.0, 1.0 / sqrt(gz2));
          shs[0] = hs.alpha;
          a = 2;
        } else { hs.status = 3; a = 3; }
      } else {
        double an = finite ? -dphi * alpha * alpha / (2.0 * (phi_t - phi - dphi * alpha)) : 0.0;
        if (!(an >= 0.1 * alpha)) an = 0.1 * alpha;      // also catches NaN
        if (an > 0.5 * alpha) an = 0.5 * alpha;
        hs.alpha = an;
        shs[0] = an;
      }
    } else if (a == 1) {
      if (!first) {
        pair_ok = out[3] > 1e-10 * sqrt(out[4] * out[5]);
        // stagnation = the objective has stopped moving AND the gradient has stopped shrinking: next to the optimum
        // phi changes by |g|^2 ~ 1e-15 per step, far below its own rounding, while |g| still falls by factors
        const double gtgt = out[7 + LB_NB];
        if (gtgt < 0.25 * hs.gzbest) { hs.gzbest = gtgt; hs.stall = 0; }
        else hs.stall = (phi - phi_t <= 1e-14 * fmax(1.0, fabs(phi))) ? hs.stall + 1 : 0;
        hs.iters += 1;
      }
      hs.phi = phi_t;
      hs.ls = 0;
      if (need_gf) hs.gf2 = out[2];
      shs[1] = need_gf ? out[2] : -1.0;
      shs[2] = alpha;                 // the step length that produced the accepted point
    }
    act[0] = a; act[1] = pair_ok; act[2] = hs.head;
  }
  __syncthreads();
  LSTAMP(4);
  const int a = act[0];
  if (a != 1) {
    if (a == 0) {                     // shorter step along the same direction
      const double an = shs[0];
      for (int i = tid; i < N; i += NT) zt[i] = z[i] + an * d[i];
    } else if (a == 2) {              // steepest descent from the accepted point
      const double an = shs[0];
      for (int i = tid; i < N; i += NT) { const double di = -gcur[i]; d[i] = di; zt[i] = z[i] + an * di; }
      if (tid < LB_NB) st->delta[tid] = (tid == 2 * LB_H) ? -1.0 : 0.0;
    }
    if (tid < HW) reinterpret_cast<int*>(st)[tid] = reinterpret_cast<const int*>(&hs)[tid];
    if (tid == 0) wh_publish(prog, hs);
    return;
  }
  // ---- accepted: the Gram matrix of the new basis, by algebra
  const int pair_ok = act[1], r = act[2];
  if (tid < LB_NB) {                   // B delta (old matrix, old coefficients)
    double t = 0.0;
    for (int mm = 0; mm < LB_NB; ++mm) t += shB[mm * LB_NB + tid] * delta[mm];
    bdel[tid] = t;
  }
  __syncthreads();
  if (tid < LB_NB) {
    const int l = tid;
    const double al = shs[2];
    const double Gl = out[7 + l];                        // g_t . b_l (old basis; l = 2H: g_t . g_old)
    const double gtgt = out[7 + LB_NB];
    const double Gold = shB[(2 * LB_H) * LB_NB + l];      // g_old . b_l
    double rs = 0.0, ry = 0.0, rg = Gl;                  // rows of s_new, y_new, g_new against entry l
    if (pair_ok) {
      rs = al * bdel[l];
      ry = Gl - Gold;
      if (l == r) { rs = out[4]; ry = out[3]; rg = al * out[6]; }                     // against s_new itself
      else if (l == LB_H + r) { rs = out[3]; ry = out[5]; rg = gtgt - out[7 + 2 * LB_H]; }   // against y_new
      else if (l == 2 * LB_H) { rs = al * out[6]; ry = gtgt - out[7 + 2 * LB_H]; rg = gtgt; }   // against g_new
    } else if (l == 2 * LB_H) rg = gtgt;
    shs[3] = 0.0;
    // all reads of the old matrix are done (bdel, Gold): write the new rows / columns
    if (pair_ok) {
      shB[r * LB_NB + l] = rs;             shB[l * LB_NB + r] = rs;
      shB[(LB_H + r) * LB_NB + l] = ry;    shB[l * LB_NB + LB_H + r] = ry;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    shB[(2 * LB_H) * LB_NB + l] = rg;      shB[l * LB_NB + 2 * LB_H] = rg;
  }
  __syncthreads();
  LSTAMP(5);
  // ---- stopping tests (thread 0) and the next direction: vector-free two-loop, lane l of wavefront 0 = coefficient l
  if (tid == 0) {
    int hist = hs.hist, head = hs.head;
    if (pair_ok) { head = (head + 1) % LB_H; hist = hist < LB_H ? hist + 1 : LB_H; }
    const double gz2 = shB[(2 * LB_H) * LB_NB + 2 * LB_H];
    hs.gz2 = gz2;
    int stop = 0;
    if (shs[1] >= 0.0 && shs[1] < hs.gtol2) { hs.status = 1; stop = 1; }
    else if (hs.stall >= 5) { hs.status = 2; stop = 1; }     // (3 until round 4: C4's stored start then stopped ONE evaluation short
                                                              // of the |grad_f| rule and paid the finisher's factorization: 4.0 instead of 3.5 ms)
    else if (hs.evals >= hs.max_evals) { hs.status = 5; stop = 1; }
    else if (!(gz2 > 0.0)) { hs.status = isfinite(gz2) ? 1 : 4; stop = 1; }
    hs.need_gf = hs.gf_avail && hs.evals >= hs.gf_from && sqrt(gz2) < hs.gate;
    hs.hist = hist; hs.head = head; hs.first = 0;
    act[3] = stop;
  }
  __syncthreads();
  LSTAMP(6);
  if (!act[3] && tid < 64) {
    const int l = tid;                                    // all 64 lanes take part in the reductions, l >= LB_NB adds 0
    const int hist = hs.hist, head = hs.head;
    const bool live = l < LB_NB;
    double dl = (l == 2 * LB_H) ? -1.0 : 0.0;
    double al[LB_H];
#pragma unroll
    for (int k = LB_H - 1; k >= 0; --k) {                 // newest to oldest (unrolled: al[] stays in registers)
      al[k] = 0.0;
      if (k >= hist) continue;
      const int rr = (head - hist + k + 2 * LB_H) % LB_H;
      double t = wave_sum_dpp(live ? dl * shB[l * LB_NB + rr] : 0.0);
      t /= shB[rr * LB_NB + LB_H + rr];
      al[k] = t;
      if (l == LB_H + rr) dl -= t;
    }
    if (hist > 0) {
      const int rn = (head - 1 + LB_H) % LB_H;
      dl *= shB[rn * LB_NB + LB_H + rn] / shB[(LB_H + rn) * LB_NB + LB_H + rn];
    }
#pragma unroll
    for (int k = 0; k < LB_H; ++k) {                      // oldest to newest
      if (k >= hist) continue;
      const int rr = (head - hist + k + 2 * LB_H) % LB_H;
      double t = wave_sum_dpp(live ? dl * shB[l * LB_NB + LB_H + rr] : 0.0);
      t /= shB[rr * LB_NB + LB_H + rr];
      if (l == rr) dl += al[k] - t;
    }
    double dphi = wave_sum_dpp(live ? dl * shB[l * LB_NB + 2 * LB_H] : 0.0);
    const double gz2 = shB[(2 * LB_H) * LB_NB + 2 * LB_H];
    if (!(dphi < 0.0)) {                                  // not a descent direction: drop the history
      dl = (l == 2 * LB_H) ? -1.0 : 0.0;
      dphi = -gz2;
      if (l == 0) hs.hist = 0;
    }
    if (live) delta[l] = dl;
    if (l == 0) {
      hs.dphi = dphi;
      // the first move is the unit step too: in the whitened variable the Hessian is I + (a correction), so -g is a
      // Newton-like step; the customary 1 / |g| start cost 2 of 19 evaluations at C3, 8 of 70 at C2, 10 of 97 at C4
      // (tests/probes/whitened_lbfgs_proto.py), Armijo backtracking catches the overshoots
      hs.alpha = 1.0;
      shs[0] = hs.alpha;
    }
  }
  __syncthreads();
  LSTAMP(7);
  for (int i = tid; i < LB_NB * LB_NB; i += NT) st->B[i] = shB[i];
  if (tid < LB_NB) st->delta[tid] = delta[tid];
  if (tid < HW) reinterpret_cast<int*>(st)[tid] = reinterpret_cast<const int*>(&hs)[tid];
  // ---- pass 2: commit the point, the gradient and the pair; next direction and trial point
  const int stop = act[3];
  // a search that goes on is announced BEFORE pass 2 (the host may enqueue the next slot meanwhile); one that has
  // ended only after z has been committed: the host reads z (through f = L z) as soon as it sees the end
  if (tid == 0 && !stop) wh_publish(prog, hs);
  const double an = shs[0];
  for (int i = tid; i < N; i += NT) {
    const double zi = zt[i], gt = zi - u[i];
    double bv[LB_NB];
#pragma unroll
    for (int l = 0; l < LB_NB; ++l) bv[l] = basis[(size_t)l * N + i];
    if (pair_ok) {
      const double sn = zi - z[i], yn = gt - bv[2 * LB_H];
      basis[(size_t)r * N + i] = sn;
      basis[(size_t)(LB_H + r) * N + i] = yn;
#pragma unroll
      for (int l = 0; l < 2 * LB_H; ++l)
        if (l == r) bv[l] = sn; else if (l == LB_H + r) bv[l] = yn;
    }
    bv[2 * LB_H] = gt;
    gcur[i] = gt;
    z[i] = zi;
    double di = 0.0;
    if (!stop) {
#pragma unroll
      for (int l = 0; l < LB_NB; ++l) di += delta[l] * bv[l];
      d[i] = di;
      zt[i] = zi + an * di;
    }
  }
  LSTAMP(8);
  if (stop) {
    __syncthreads();
    __threadfence();
    if (tid == 0) wh_publish(prog, hs);
  }
#undef LSTAMP
}

__global__ __launch_bounds__(LB_T) void lbfgs_step_kernel(WhState* __restrict__ st, int N, int m, int n_q,
                                                          double* __restrict__ z, double* __restrict__ zt,
                                                          double* __restrict__ d, const double* __restrict__ u,
                                                          const double* __restrict__ v,
                                                          const double* __restrict__ beta,
                                                          const double* __restrict__ tq, double* __restrict__ basis,
                                                          long long* dbg, WhProgress prog,
                                                          const double* __restrict__ dots, int n_parts) {
  lbfgs_step<LB_T>(st, N, m, n_q, z, zt, d, u, v, beta, tq, basis, dbg, prog, dots, n_parts);
}


// beta(f) and u = L^T beta in ONE launch: every workgroup rebuilds beta for itself in LDS (N exponentials -- cheaper
// than a dependent launch), then each wavefront owns one row of the row-major copy U of L^T (zero left of the
// diagonal): u_i = sum_{k >= i} U[i][k] beta[k].  Workgroup 0 publishes beta (the judgement needs it for |grad_f|),
// the last workgroup the per-query likelihood sums.  Replaces laplace_kernel + gemvT_partial + sum_slabs in a slot.
__global__ __launch_bounds__(256) void beta_lt_kernel(const int* __restrict__ status, const double* __restrict__ ft,
                                                      int N, int mblk, int n_q, double sigma,
                                                      const double* __restrict__ U, double* __restrict__ beta,
                                                      double* __restrict__ tq, double* __restrict__ u,
                                                      const int* __restrict__ need_gf,
                                                      const double* __restrict__ Sinv, double* __restrict__ v) {
  extern __shared__ __attribute__((aligned(16))) double blds[];     // f [N] | beta [N]
  if (*status != 0) return;
  double* sf = blds;
  double* sb = blds + N;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int m = mblk - 1;
  const bool liker = blockIdx.x == gridDim.x - 1;
  for (int i = threadIdx.x; i < N; i += 256) sf[i] = ft[i];
  __syncthreads();
  const double bsc = sigma * (double)m;
  for (int q = wv; q < n_q; q += 4) {
    const int i = q * mblk;
    const double f0 = sf[i];
    double sp2 = 0.0, sphi = 0.0;
    for (int r = 1 + lane; r <= m; r += 64) {
      const double delta = (sf[i + r] - f0) / sigma;
      const double p2 = INV_SQRT_4PI * exp(-0.25 * (delta * delta));
      if (liker) sphi += 0.5 * erfc(-0.5 * delta);
      sp2 += p2;
      sb[i + r] = -p2 / bsc;
    }
    sp2 = wave_sum_dpp(sp2);
    if (liker) sphi = wave_sum_dpp(sphi);
    if (lane == 0) { sb[i] = sp2 / bsc; if (liker) tq[q] = sphi; }
  }
  __syncthreads();
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < N; i += 256) beta[i] = sb[i];
  const int i = blockIdx.x * 4 + wv;
  const bool row_ok = i < N;
  double su = 0.0, sv = 0.0;
  const int want_v = *need_gf;
  if (row_ok) {
    const double* ur = U + (size_t)i * N;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int k = (i & ~1) + 2 * lane;          // rows are 16-byte aligned (N even); entries left of the diagonal are zero
    for (; k + 128 < N; k += 256) {
      const double2 u0 = *reinterpret_cast<const double2*>(ur + k), b0 = *reinterpret_cast<const double2*>(sb + k);
      const double2 u1 = *reinterpret_cast<const double2*>(ur + k + 128), b1 = *reinterpret_cast<const double2*>(sb + k + 128);
      a0 += u0.x * b0.x; a1 += u0.y * b0.y; a2 += u1.x * b1.x; a3 += u1.y * b1.y;
    }
    for (; k < N; k += 128) {
      const double2 u0 = *reinterpret_cast<const double2*>(ur + k), b0 = *reinterpret_cast<const double2*>(sb + k);
      a0 += u0.x * b0.x; a1 += u0.y * b0.y;
    }
    su = wave_sum_dpp((a0 + a1) + (a2 + a3));
    if (lane == 0) u[i] = su;
    if (want_v) {
      // v_i = (Sigma^-1 f)_i for the |grad_f| rule, wanted only near the end: f is in LDS already, and a launch of its
      // own costs its 4-5 us in EVERY slot, gated off or not
      const double* sr = Sinv + (size_t)i * N;
      a0 = a1 = a2 = a3 = 0.0;
      k = 2 * lane;
      for (; k + 128 < N; k += 256) {
        const double2 u0 = *reinterpret_cast<const double2*>(sr + k), b0 = *reinterpret_cast<const double2*>(sf + k);
        const double2 u1 = *reinterpret_cast<const double2*>(sr + k + 128), b1 = *reinterpret_cast<const double2*>(sf + k + 128);
        a0 += u0.x * b0.x; a1 += u0.y * b0.y; a2 += u1.x * b1.x; a3 += u1.y * b1.y;
      }
      for (; k < N; k += 128) {
        const double2 u0 = *reinterpret_cast<const double2*>(sr + k), b0 = *reinterpret_cast<const double2*>(sf + k);
        a0 += u0.x * b0.x; a1 += u0.y * b0.y;
      }
      sv = wave_sum_dpp((a0 + a1) + (a2 + a3));
      if (lane == 0) v[i] = sv;
    }
  }
}

// U = L^T (upper triangle, zero below; row pitch N); 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_lower_kernel(const double* __restrict__ L, int N, int ldl,
                                                              double* __restrict__ U) {
  __shared__ double t[32][33];
  const int bi = blockIdx.y, bj = blockIdx.x;        // tile of L at rows bi*32.., cols bj*32..
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int i = bi * 32 + r, j = bj * 32 + tx;
    t[r][tx] = (i < N && j < N && j <= i) ? L[(size_t)i * ldl + j] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int j = bj * 32 + r, i = bi * 32 + tx;     // U[j][i] = L[i][j]
    if (i < N && j < N) U[(size_t)j * N + i] = t[tx][r];
  }
}

struct FitWork {
  ppbo_ctx* ctx;
  hipStream_t s;
  const double* Sinv;
  int N, m, mblk, n_q;
  double sigma;
  double *H, *Linv;       // N x N each
  double *tq, *rowstats, *sc, *u, *w, *p, *q, *tmpv;
  double* hsc;            // pinned host scalars
  int* d_info;
};

int eval_point(FitWork& W, Vecs& P, const double* step = nullptr, const double* v_prev = nullptr) {
  // v = Sinv f ; laplace terms ; g ; scalars -> host
  if (int rc = ppbo_gemv_async(W.ctx, W.Sinv, W.N, W.N, P.f, P.v, 0, 0, W.s)) return rc;
  if (step) pdot2_kernel<<<1, 1024, 0, W.s>>>(step, v_prev, P.v, W.N, W.sc + 3);
  laplace_kernel<<<(W.n_q + 3) / 4, 256, 0, W.s>>>(P.f, W.N, W.mblk, W.n_q, W.sigma, W.tq, P.beta, P.ld, P.lo);
  tlik_reduce_kernel<<<1, 1024, 0, W.s>>>(W.tq, W.n_q, W.m, W.sc + 2);
  grad_kernel<<<1, 1024, 0, W.s>>>(P.f, P.v, P.beta, W.N, P.g, nullptr, W.sc);
  PPBO_LAUNCH_CHECK(W.ctx);
  PPBO_HIP_CHECK(W.ctx, hipMemcpyAsync(W.hsc, W.sc, 4 * sizeof(double), hipMemcpyDeviceToHost, W.s));
  PPBO_HIP_CHECK(W.ctx, hipStreamSynchronize(W.s));
  P.fSf = W.hsc[0];
  P.gn2 = W.hsc[1];
  P.Tlik = W.hsc[2];
  P.pv = step ? W.hsc[3] : 0.0;
  return 0;
}

inline double phi_of(const Vecs& P) { return 0.5 * P.fSf - P.Tlik; }  // phi = -T

int carve_fit_work(ppbo_ctx* ctx, FitWork& W, Vecs* pts, int npts) {
  const int N = W.N;
  const size_t nn = (size_t)N * N;
  W.H = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_LINALG, 2 * nn * sizeof(double));
  if (!W.H) return (int)hipErrorOutOfMemory;
  W.Linv = W.H + nn;
  const size_t nvec = (size_t)npts * 6 + 5 + 3 + 1;
  double* v = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_SCRATCH, (nvec * N + W.n_q + 64) * sizeof(double) + 256);
  if (!v) return (int)hipErrorOutOfMemory;
  for (int k = 0; k < npts; ++k) {
    pts[k].f = v; v += N; pts[k].v = v; v += N; pts[k].beta = v; v += N;
    pts[k].ld = v; v += N; pts[k].lo = v; v += N; pts[k].g = v; v += N;
  }
  W.u = v; v += N; W.w = v; v += N; W.p = v; v += N; W.q = v; v += N; W.tmpv = v; v += N;
  W.rowstats = v; v += 3 * (size_t)N;
  W.tq = v; v += W.n_q;
  W.sc = v; v += 32;
  W.d_info = (int*)v;
  W.hsc = (double*)ppbo_pinned(ctx, 64 * sizeof(double));
  if (!W.hsc) return ppbo_set_error(ctx, (int)hipErrorOutOfMemory, "pinned staging");
  return 0;
}

}  // namespace

extern "C" {

int ppbo_laplace_terms(ppbo_ctx* ctx, const double* d_f, int N, int m, double sigma, double* d_Tlik,
                       double* d_beta, double* d_lam_diag, double* d_lam_off, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_f && N > 0 && m >= 1 && sigma > 0, "arguments");
  PPBO_REQUIRE(ctx, N % (m + 1) == 0, "N must be n_q*(m+1) (feedback_processing.py:110-130)");
  hipStream_t s = (hipStream_t)stream;
  const int mblk = m + 1, n_q = N / mblk;
  double* tq = nullptr;
  if (d_Tlik) {
    tq = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_VEC, (size_t)n_q * sizeof(double));
    if (!tq) return (int)hipErrorOutOfMemory;
  }
  laplace_kernel<<<(n_q + 3) / 4, 256, 0, s>>>(d_f, N, mblk, n_q, sigma, tq, d_beta, d_lam_diag, d_lam_off);
  if (d_Tlik) tlik_reduce_kernel<<<1, 1024, 0, s>>>(tq, n_q, m, d_Tlik);
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

int ppbo_sum_phi(ppbo_ctx* ctx, const double* d_f, int N, int m, double sigma, int order, double* d_out, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_f && d_out && N > 0 && m >= 1 && sigma > 0, "arguments");
  PPBO_REQUIRE(ctx, N % (m + 1) == 0, "N must be n_q*(m+1) (feedback_processing.py:110-130)");
  PPBO_REQUIRE(ctx, order >= 0 && order <= 2, "The derivatives of an order higher than 2 are not needed! (gp_model.py:203)");
  const int n_q = N / (m + 1);
  sum_phi_kernel<<<(n_q + 3) / 4, 256, 0, (hipStream_t)stream>>>(d_f, N, m + 1, n_q, sigma, order, d_out);
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

int ppbo_T_and_grad(ppbo_ctx* ctx, const double* d_Sigma_inv, const double* d_f, int N, int m, double sigma,
                    double* h_T, double* d_grad, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_Sigma_inv && d_f && N > 0 && m >= 1 && sigma > 0, "arguments");
  PPBO_REQUIRE(ctx, N % (m + 1) == 0, "N must be n_q*(m+1)");
  hipStream_t s = (hipStream_t)stream;
  const int mblk = m + 1, n_q = N / mblk;
  double* v = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_SCRATCH, ((size_t)2 * N + n_q + 8) * sizeof(double));
  if (!v) return (int)hipErrorOutOfMemory;
  double* beta = v + N;
  double* tq = beta + N;
  double* sc = tq + n_q;
  if (int rc = ppbo_gemv_async(ctx, d_Sigma_inv, N, N, d_f, v, 0, 0, s)) return rc;
  laplace_kernel<<<(n_q + 3) / 4, 256, 0, s>>>(d_f, N, mblk, n_q, sigma, tq, beta, nullptr, nullptr);
  tlik_reduce_kernel<<<1, 1024, 0, s>>>(tq, n_q, m, sc + 2);
  grad_kernel<<<1, 1024, 0, s>>>(d_f, v, beta, N, nullptr, d_grad, sc);
  PPBO_LAUNCH_CHECK(ctx);
  if (h_T) {
    double h[3];
    PPBO_HIP_CHECK(ctx, hipMemcpyAsync(h, sc, sizeof(h), hipMemcpyDeviceToHost, s));
    PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
    *h_T = -0.5 * h[0] + h[2];
  }
  return 0;
}

int ppbo_fit_fmap(ppbo_ctx* ctx, const double* d_Sigma_inv, int N, int m, double sigma, const double* d_f_init,
                  const ppbo_fit_opts* opts, double* d_fMAP, ppbo_fit_stats* h_stats, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_Sigma_inv && d_f_init && d_fMAP, "null pointer");
  PPBO_REQUIRE(ctx, N > 0 && m >= 1 && sigma > 0, "sizes");
  PPBO_REQUIRE(ctx, N % (m + 1) == 0, "N must be n_q*(m+1)");
  const double gtol = (opts && opts->gtol > 0) ? opts->gtol : 1e-4;
  const int maxiter = (opts && opts->maxiter > 0) ? opts->maxiter : 200 * N;
  const int verbose = opts ? opts->verbose : 0;
  const bool judge_noise = opts && opts->judge_by_gradient_below_noise != 0;

  FitWork W{};
  W.ctx = ctx; W.s = (hipStream_t)stream; W.Sinv = d_Sigma_inv;
  W.N = N; W.m = m; W.mblk = m + 1; W.n_q = N / (m + 1); W.sigma = sigma;
  Vecs pt[2];
  if (int rc = carve_fit_work(ctx, W, pt, 2)) return rc;
  hipStream_t s = W.s;
  const size_t vbytes = (size_t)N * sizeof(double);
  int cur = 0;
  PPBO_HIP_CHECK(ctx, hipMemcpyAsync(pt[cur].f, d_f_init, vbytes, hipMemcpyDeviceToDevice, s));
  if (int rc = eval_point(W, pt[cur])) return rc;

  double radius = (opts && opts->initial_radius > 0.0) ? opts->initial_radius : 1.0;
  const double rmax = 1000.0, eta = 0.15, k_easy = 0.1, k_hard = 0.2;
  std::vector<double> host_p(N), host_z(N), host_lz(N), host_g(N);
  bool host_g_valid = false;
  double lam_lb_prev = 0.0;
  double prev_lam = 0.0, prev_radius = radius;
  bool prev_boundary = false, prev_failed = false;
  bool shrink = false, h_changed = true;
  double st_mindiag = 0, st_gmax = 0, st_gmin = 0, st_fro = 0, st_inf = 0;
  int it = 0, nchol = 0;
  bool h_at_zero = false;  // H buffer currently holds the unfactorized lam=0 matrix
  // Chord steps: after an interior Newton step that the model predicted well, the next iterate(s) reuse the
  // factor (and its inverse blocks) instead of refactoring -- near the optimum the Hessian barely moves, a chord
  // step costs two triangular applications instead of a Cholesky + inversion, and the step is still judged by
  // the same actual/predicted ratio (the model is then the quadratic with the OLD Hessian).
  int chord_left = 0, split = 0;

  if (!std::isfinite(pt[cur].gn2) || !std::isfinite(pt[cur].fSf))
    return ppbo_set_error(ctx, -3, "non-finite objective at the start vector");
  while (it < maxiter && std::sqrt(pt[cur].gn2) >= gtol) {
    ++it;
    Vecs& C = pt[cur];
    Vecs& T = pt[cur ^ 1];
    const double gnorm = std::sqrt(C.gn2);
    if (h_changed && chord_left == 0) {
      form_shifted_kernel<<<N, 256, 0, s>>>(W.Sinv, N, W.mblk, C.ld, C.lo, 0.0, W.H, W.rowstats);
      stats_reduce_kernel<<<1, 1024, 0, s>>>(W.rowstats, N, W.sc + 8);
      PPBO_LAUNCH_CHECK(ctx);
      PPBO_HIP_CHECK(ctx, hipMemcpyAsync(W.hsc, W.sc + 8, 5 * sizeof(double), hipMemcpyDeviceToHost, s));
      PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
      st_mindiag = W.hsc[0]; st_gmax = W.hsc[1]; st_gmin = W.hsc[2];
      st_fro = std::sqrt(W.hsc[3]); st_inf = W.hsc[4];
      h_changed = false;
      h_at_zero = true;
    }
    // ---- subproblem: min g'p + 1/2 p'Hp, |p| <= radius --------------------
    const double hn = std::fmin(st_fro, st_inf);
    double lb = std::fmax(0.0, std::fmax(-st_mindiag, gnorm / radius - std::fmin(st_gmax, hn)));
    double ub = std::fmax(0.0, gnorm / radius + std::fmin(-st_gmin, hn));
    if (shrink) lb = std::fmax(lb, lam_lb_prev);
    double lam = (lb == 0.0) ? 0.0 : std::fmax(std::sqrt(lb * ub), lb + 0.01 * (ub - lb));
    // Warm start (not in SciPy): the multiplier that hit the previous radius, rescaled for the new
    // radius, is a far better guess than the Gershgorin geometric mean (Sigma^-1 has 1e7-size
    // off-diagonals, so those bounds are loose).  lam = 0 is still tried first whenever it is not ruled out.
    double warm = 0.0;
    if (prev_lam > 0.0 && prev_boundary) warm = prev_lam * (prev_radius / radius);
    const bool had_guess = warm > 0.0;
    double best_pd = INFINITY;   // smallest shift of this subproblem whose factorization succeeded
    // lam = 0 (interior Newton step) is tried first unless it is ruled out by the Gershgorin bound or the
    // previous subproblem PROVED indefiniteness by a failed factorization (H changes little between iterates)
    if ((lb > 0.0 || prev_failed) && warm > lb && warm < ub) { lam = warm; warm = 0.0; }
    bool any_failed = false;
    bool boundary = true, have_step = false, used_hard = false;
    double pn = 0.0, gtp = 0.0, lam_used = 0.0, hard_pred = 0.0;
    host_g_valid = false;
    bool chord = (chord_left > 0);
    if (chord) lam = 0.0;
    bool chord_abort = false;
    for (int inner = 0; inner < 80; ++inner) {
      if (!chord) {
      if (!(h_at_zero && lam == 0.0)) {
        form_shifted_kernel<<<N, 256, 0, s>>>(W.Sinv, N, W.mblk, C.ld, C.lo, lam, W.H, nullptr);
        PPBO_LAUNCH_CHECK(ctx);
      }
      h_at_zero = false;
      ++nchol;
      if (int rc = ppbo_potrf_async(ctx, W.H, N, N, W.d_info, s, W.sc + 24)) return rc;
      int info = 0;
      PPBO_HIP_CHECK(ctx, hipMemcpyAsync(&info, W.d_info, sizeof(int), hipMemcpyDeviceToHost, s));
      PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
      if (verbose > 1) printf("    [inner %d] lam %.6e info %d lb %.3e ub %.3e radius %.3e\n", inner, lam, info, lb, ub, radius);
      if (info != 0) {
        any_failed = true;
        lb = std::fmax(lb, lam);
        // the failed pivot bounds the admissible shifts from below (as in SciPy's trust-exact): without it
        // the bracket only learns "more than lam" and closes in on the pole by geometric means
        if (ppbo_potrf_fail_bound_async(ctx, W.H, N, N, W.d_info, W.sc + 24, W.sc + 25, s) == 0) {
          PPBO_HIP_CHECK(ctx, hipMemcpyAsync(W.hsc, W.sc + 25, sizeof(double), hipMemcpyDeviceToHost, s));
          PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
          const double grow = W.hsc[0];
          if (std::isfinite(grow) && grow > 0.0) lb = std::fmax(lb, lam + grow);
          if (verbose > 1) printf("    [inner %d]   failed pivot: shift >= %.6e\n", inner, lam + grow);
        }
        if (lb == 0.0) lb = 1e-14 * std::fmax(st_inf, 1e-300);   // lam = 0 is ruled out from now on
        if (ub <= lb) ub = 2.0 * lb + 1e-12;
        if (warm > lb && warm < ub) { lam = warm; warm = 0.0; }
        else if (std::isfinite(best_pd)) lam = std::sqrt(lb * best_pd);    // between the failure and a known PD shift
        else if (had_guess) lam = 2.0 * lb;                                // right order of magnitude: double, do not
                                                                           // jump to the (very loose) Gershgorin mean
        else lam = std::fmax(std::sqrt(lb * ub), lb + 0.01 * (ub - lb));
        continue;
      }
      best_pd = std::fmin(best_pd, lam);
      // L^-1 is kept as two diagonal blocks + L21 (ppbo_apply_linv_async)
      if (int rc = ppbo_trtri_async(ctx, W.H, N, N, W.Linv, N, s, 1, &split)) return rc;
      }
      auto apply_linv = [&](const double* x, double* y, int trans) {
        return ppbo_apply_linv_async(ctx, W.Linv, N, W.H, N, N, split, x, y, trans, W.tmpv, s);
      };
      if (int rc = apply_linv(C.g, W.w, 0)) return rc;   // w = L^-1 g
      if (int rc = apply_linv(W.w, W.u, 1)) return rc;   // u = L^-T w
      step_kernel<<<1, 1024, 0, s>>>(W.u, C.g, C.f, N, W.p, T.f, W.sc + 16);           // p = -u, fn = f + p
      if (int rc = apply_linv(W.p, W.q, 0)) return rc;   // q = L^-1 p
      if (int rc = ppbo_dot_async(ctx, W.q, W.q, N, W.sc + 18, s)) return rc;
      PPBO_HIP_CHECK(ctx, hipMemcpyAsync(W.hsc, W.sc + 16, 3 * sizeof(double), hipMemcpyDeviceToHost, s));
      PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
      pn = std::sqrt(W.hsc[0]);
      gtp = W.hsc[1];
      const double qn2 = W.hsc[2];
      if (!std::isfinite(pn) || !std::isfinite(gtp) || !std::isfinite(qn2))
        return ppbo_set_error(ctx, -3, "non-finite trust-region step (|p|^2=%g g.p=%g |q|^2=%g) at iteration %d", W.hsc[0],
                              gtp, qn2, it);
      if (chord && !(pn <= radius)) { chord_abort = true; break; }   // stale factor, step too long: refactor
      have_step = true;
      lam_used = lam;
      if (verbose > 1) printf("    [inner %d]   |p| %.6e (radius %.3e)\n", inner, pn, radius);
      if (pn <= radius && lam == 0.0) { boundary = false; break; }
      if (std::fabs(pn - radius) <= k_easy * radius) break;
      double lam_new = lam + (pn * pn / qn2) * (pn - radius) / radius;
      if (pn < radius) {
        // Inside the region with lam > 0: the solution sits next to the pole -lambda_min(H) ("hard case",
        // Conn/Gould/Toint 7.3.1).  Two inverse iterations with the factor just computed give the eigenvector z
        // of the smallest eigenvalue s2 of H + lam I; p + tau z on the boundary is accepted when it changes the
        // model by less than k_hard, and lam - s2 is a tight lower bound on the admissible shifts.
        std::vector<double>& hp = host_p; std::vector<double>& hz = host_z; std::vector<double>& hl = host_lz;
        if (int rc = apply_linv(W.p, W.w, 0)) return rc;
        if (int rc = apply_linv(W.w, W.u, 1)) return rc;
        if (int rc = apply_linv(W.u, W.w, 0)) return rc;
        if (int rc = apply_linv(W.w, W.u, 1)) return rc;      // u ~ eigenvector
        if (int rc = ppbo_gemv_async(ctx, W.H, N, N, W.u, W.q, 1, 1, s)) return rc;         // q = L^T u
        PPBO_HIP_CHECK(ctx, hipMemcpyAsync(hp.data(), W.p, vbytes, hipMemcpyDeviceToHost, s));
        PPBO_HIP_CHECK(ctx, hipMemcpyAsync(hz.data(), W.u, vbytes, hipMemcpyDeviceToHost, s));
        PPBO_HIP_CHECK(ctx, hipMemcpyAsync(hl.data(), W.q, vbytes, hipMemcpyDeviceToHost, s));
        if (!host_g_valid) PPBO_HIP_CHECK(ctx, hipMemcpyAsync(host_g.data(), C.g, vbytes, hipMemcpyDeviceToHost, s));
        PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
        host_g_valid = true;
        double zn2 = 0.0, lz2 = 0.0, pz = 0.0, gz = 0.0;
        for (int i = 0; i < N; ++i) { zn2 += hz[i] * hz[i]; lz2 += hl[i] * hl[i]; pz += hp[i] * hz[i]; gz += host_g[i] * hz[i]; }
        if (zn2 > 0.0 && std::isfinite(zn2)) {
          const double zn = std::sqrt(zn2);
          const double s2 = lz2 / zn2;                       // z'(H + lam I)z for the unit vector
          pz /= zn; gz /= zn;
          const double cterm = pn * pn - radius * radius;    // < 0
          const double disc = std::sqrt(std::fmax(pz * pz - cterm, 0.0));
          const double ta = -pz + disc, tb = -pz - disc;
          const double tau = (std::fabs(ta) < std::fabs(tb)) ? ta : tb;
          const double quad = -gtp;                          // p'(H + lam I)p
          const double rel = tau * tau * s2 / (quad + lam * radius * radius);
          if (rel <= k_hard) {
            for (int i = 0; i < N; ++i) hp[i] += tau * hz[i] / zn;
            PPBO_HIP_CHECK(ctx, hipMemcpyAsync(W.p, hp.data(), vbytes, hipMemcpyHostToDevice, s));
            restep_kernel<<<1, 1024, 0, s>>>(W.p, C.g, C.f, N, T.f, W.sc + 16);
            PPBO_HIP_CHECK(ctx, hipMemcpyAsync(W.hsc, W.sc + 16, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
            PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
            const double gtp_new = W.hsc[1];
            // model decrease of the modified step: -(g'p' + 1/2 p'Hp'), p'Hp' = p'(H+lam I)p' - lam |p'|^2
            hard_pred = -0.5 * gtp_new + 0.5 * tau * gz - 0.5 * tau * tau * s2 + 0.5 * lam * radius * radius;
            pn = std::sqrt(W.hsc[0]);
            gtp = gtp_new;
            used_hard = true;
            break;
          }
          lb = std::fmax(lb, lam - s2);
        }
        ub = lam;
      } else {
        lb = lam;
      }
      if (!(lb < lam_new && lam_new < ub)) {
        // the secular-equation Newton step left the bracket; if lam = 0 is still possible (no failed
        // factorization there) and the step is inside the region, the interior Newton step is next
        if (lb == 0.0 && pn < radius) lam_new = 0.0;
        else lam_new = std::fmax(std::sqrt(lb * ub), lb + 0.01 * (ub - lb));
      }
      if (lam_new < 0.0) lam_new = 0.0;
      if (lam_new == lam) break;
      lam = lam_new;
    }
    if (chord_abort) { chord_left = 0; --it; continue; }
    lam_lb_prev = lb;
    if (!have_step) break;
    prev_lam = lam_used;
    prev_radius = radius;
    prev_boundary = boundary;
    // indefiniteness is sticky: H changes little between iterates, so once a factorization at (or near) lam = 0
    // has failed the warm multiplier goes first until an interior Newton step is accepted again
    if (any_failed) prev_failed = true;
    else if (!boundary) prev_failed = false;
    // (H + lam I) p = -g  =>  p'Hp = -g'p - lam |p|^2
    const double pred = used_hard ? hard_pred : (-0.5 * gtp + 0.5 * lam_used * pn * pn);
    if (!(pred > 0.0)) break;
    if (int rc = eval_point(W, T, W.p, C.v)) return rc;
    // phi(f) - phi(f+p) = -1/2 p'(Sf + S(f+p)) + (Tlik(f+p) - Tlik(f)), from small quantities
    const double actual = -0.5 * T.pv + (T.Tlik - C.Tlik);
    double rho = actual / pred;
    // Next to the optimum the predicted decrease drops below the rounding noise of the objective difference
    // (~1e-13 at phi ~ 10: the likelihood is a sum of N terms), rho is then a random number and a perfectly good
    // Newton step would be rejected and the radius quartered until it cuts the step (seen when the fit is
    // started from the whitened pre-phase's result with a gtol below that phase's floor): there the step is
    // judged by what it does to the gradient instead.
    // Only as the whitened search's finisher (opts->judge_by_gradient_below_noise): called on its own, this routine
    // follows SciPy's acceptance and radius rules to the letter.
    if (judge_noise && pred < 1e-11 * std::fmax(1.0, std::fabs(phi_of(C))) && T.gn2 < C.gn2) rho = 1.0;
    const double old_radius = radius;
    if (!(rho >= 0.25)) { if (!chord) radius *= 0.25; }   // a poor chord step blames the stale Hessian, not the radius
    else if (rho > 0.75 && boundary) radius = std::fmin(2.0 * radius, rmax);
    shrink = radius < old_radius;
    if (verbose)
      printf("[ppbo_fit] it %d lam %.3e |p| %.3e rho %.3f radius %.3e phi %.12e |g| %.3e nchol %d\n", it, lam_used,
             pn, rho, radius, phi_of(C), gnorm, nchol);
    if (rho > eta) {
      cur ^= 1;
      h_changed = true;
    }
    // enter only in the contracting regime (the gradient fell by 5x on a full Newton step)
    if (chord) chord_left = (rho > eta && rho >= 0.5) ? chord_left - 1 : 0;
    else chord_left = (!boundary && lam_used == 0.0 && rho > 0.75 && std::sqrt(T.gn2) < 0.2 * gnorm) ? 2 : 0;
    if (radius < 1e-14) break;
  }
  PPBO_HIP_CHECK(ctx, hipMemcpyAsync(d_fMAP, pt[cur].f, vbytes, hipMemcpyDeviceToDevice, s));
  PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
  if (h_stats) {
    h_stats->lbfgs_iterations = 0; h_stats->lbfgs_evals = 0; h_stats->lbfgs_status = -1;
    h_stats->iterations = it;
    h_stats->n_cholesky = nchol;
    h_stats->gradnorm = std::sqrt(pt[cur].gn2);
    h_stats->converged = (h_stats->gradnorm < gtol) ? 1 : 0;
    h_stats->T = -phi_of(pt[cur]);
  }
  return 0;
}

}  // extern "C"

namespace {

// How far the host may run ahead of the device with its slot enqueues: the host needs ~25 us per slot, the device
// ~45 us, so two slots of lead keep the device fed; whatever is queued beyond the slot that ends the search is a
// handful of gated no-op launches (the first version enqueued 4, 8, 16, ... slots per read-back: up to 8 dead slots
// = 48 launches and three stream synchronisations per fit).
constexpr int WH_AHEAD = 3;

struct WhitenedExtras {
  const double* d_Linv = nullptr;    // when given: z0 = L^-1 f_init by one triangular product (else L^T (Sigma^-1 f_init))
  bool start_is_z = false;           // d_f_init IS z0
  const int* d_factor_info = nullptr; // the device info word of the factorization that produced d_L (ahead on the same stream)
  bool sync_at_end = true;           // false: d_fMAP is only enqueued (the caller synchronises later)
  hipEvent_t join_event = nullptr;   // when given: Sigma^-1 (and L^-1) are being produced on another stream; the search's
  int gf_from = 0;                   // stream waits for this event before slot gf_from, the first that may read them
};

int whitened_search(ppbo_ctx* ctx, const double* d_L, int ldl, const double* d_Sigma_inv, int N, int m, double sigma,
                    const double* d_f_init, const ppbo_fit_opts* opts, double* d_fMAP, ppbo_fit_stats* h_stats,
                    hipStream_t s, const WhitenedExtras& ex) {
  const double gtol = (opts && opts->gtol > 0) ? opts->gtol : 1e-4;
  const int verbose = opts ? opts->verbose : 0;
  const int max_evals = (opts && opts->lbfgs_max_evals > 0) ? opts->lbfgs_max_evals : 4000;
  const int mblk = m + 1, n_q = N / mblk;
  // workspace: state | basis [LB_NB][N] | z zt d u v beta ft rowsq | tq
  const size_t st_doubles = (sizeof(WhState) + 7) / 8 + 8;
  const size_t nvec = (size_t)LB_NB + 8;
  const size_t dot_rows = (size_t)(N + 15) / 16 + 1;     // rows of partial inner products (PpboDotsOut): one per workgroup of the u launch
  double* base = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_LBFGS, (st_doubles + nvec * N + n_q + 64 + dot_rows * PPBO_DOTS_STRIDE) * sizeof(double));
  if (!base) return (int)hipErrorOutOfMemory;
  WhState* st = (WhState*)base;
  double* basis = base + st_doubles;
  double* z = basis + (size_t)LB_NB * N;
  double *zt = z + N, *dd = zt + N, *u = dd + N, *v = u + N, *beta = v + N, *ft = beta + N, *rowsq = ft + N;
  double* tq = rowsq + N;
  double* dot_part = tq + n_q + 64;
  PpboDotsOut dots;
  dots.zt = zt; dots.z = z; dots.gcur = basis + (size_t)(2 * LB_H) * N; dots.d = dd; dots.v = v; dots.beta = beta;
  dots.basis = basis; dots.nb = LB_NB; dots.partial = dot_part;
  // host-mapped progress word + head copy (the ctx's result record block: doubles [4, 4 + 16) hold the head, [24] the word)
  PpboHostRecord hr;
  if (int rc = ppbo_host_record(ctx, &hr)) return rc;
  static_assert(sizeof(WhHead) <= 16 * sizeof(double), "head copy fits its slot of the host-mapped block");
  WhProgress prog{reinterpret_cast<unsigned long long*>(hr.d_rec + 24), reinterpret_cast<int*>(hr.d_rec + 4)};
  volatile unsigned long long* h_word = reinterpret_cast<volatile unsigned long long*>(const_cast<double*>(hr.h_rec) + 24);
  const WhHead* h_head = reinterpret_cast<const WhHead*>(const_cast<double*>(hr.h_rec) + 4);
  *h_word = 0;                          // nothing of this ctx is in flight that could write it (one search per ctx at a time)
  row_sqnorm_kernel<<<(N + 3) / 4, 256, 0, s>>>(d_L, N, ldl, rowsq, base, st_doubles + (size_t)LB_NB * N);
  lbfgs_init_kernel<<<1, LB_T, 0, s>>>(st, rowsq, N, gtol, max_evals, 1, ex.gf_from, ex.d_factor_info, prog,
                                       ex.start_is_z ? d_f_init : nullptr, zt);
  PPBO_LAUNCH_CHECK(ctx);
  if (!ex.start_is_z) {
    if (ex.d_Linv) {
      if (int rc = ppbo_gemv_async(ctx, ex.d_Linv, N, N, d_f_init, zt, 0, 1, s)) return rc;     // z0 = L^-1 f_init
    } else {
      // z0 = L^-1 f_init = L^T (Sigma^-1 f_init)
      if (int rc = ppbo_gemv_async(ctx, d_Sigma_inv, N, N, d_f_init, v, 0, 0, s)) return rc;
      if (int rc = ppbo_gemv_async(ctx, d_L, N, ldl, v, zt, 1, 1, s)) return rc;
    }
  }
  PpboGate run; run.skip_if_nonzero = &st->status;
  PpboGate run_gf = run; run_gf.skip_if_zero = &st->need_gf;
  long long* dbg = nullptr;
  if (verbose > 1) {
    dbg = (long long*)ppbo_workspace(ctx, ppbo_ctx::WS_SEARCH_SMALL, 16 * 16 * sizeof(long long) + 512);
    if (dbg) (void)hipMemsetAsync(dbg, 0, 16 * 16 * sizeof(long long), s);
  }
  // beta and L^T beta in one launch through a row-major copy of L^T.  The transposition is paid once per fit and every
  // workgroup recomputes beta: a gain while launches dominate (C2, N = 512: 2.28 -> 1.97 ms; C4, N = 1024: 3.87 -> 2.96),
  // a loss once the matrix passes do (C3, N = 2048: 1.12 -> 1.19 ms; C5 warm, N = 4096: 0.44 -> 0.58)
  const bool fused = (N % 2) == 0 && N <= 1536;
  double* U = nullptr;
  const size_t blds = (size_t)2 * N * sizeof(double);
  if (fused) {
    U = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_LBFGS_U, (size_t)N * N * sizeof(double));
    if (!U) return (int)hipErrorOutOfMemory;
    const int nt32 = (N + 31) / 32;
    transpose_lower_kernel<<<dim3(nt32, nt32), 256, 0, s>>>(d_L, N, ldl, U);
    if (blds > 48 * 1024) ppbo_lds_limit(ctx, (const void*)beta_lt_kernel, 144 * 1024);
  }
  auto enqueue_slot = [&]() -> int {
    bool v_done = false;                // v = Sigma^-1 f rode along with the u launch
    int n_parts = 0;                    // > 0: the u launch left the judgement's inner products as that many rows of partial sums
    if (int rc = ppbo_gemv_async(ctx, d_L, N, ldl, zt, ft, 0, 1, s, run)) return rc;                  // f = L zt
    if (fused) {
      beta_lt_kernel<<<(N + 3) / 4, 256, blds, s>>>(&st->status, ft, N, mblk, n_q, sigma, U, beta, tq, u, &st->need_gf,
                                                    d_Sigma_inv, v);
      v_done = true;              // (the judgement's inner products stay with the step kernel here: at these sizes its one
                                  // workgroup forms them in ~2 us, and rows of partial sums from this launch cost as much)
    } else {
      // u = L^T beta(f): beta rebuilt inside the product's first pass where the star size allows (m = 31: yes)
      const int rcb = ppbo_gemvT_beta_async(ctx, d_L, N, ldl, ft, mblk, sigma, u, beta, tq, s, run, d_Sigma_inv, N, v,
                                            run_gf, dots, &n_parts);
      if (rcb > 1) return rcb;
      v_done = rcb == 0;
      if (rcb == 1) {
        laplace_kernel<<<(n_q + 3) / 4, 256, 0, s>>>(ft, N, mblk, n_q, sigma, tq, beta, nullptr, nullptr);
        if (int rc = ppbo_gemv_async(ctx, d_L, N, ldl, beta, u, 1, 1, s, run)) return rc;             // u = L^T beta
      }
    }
    if (!v_done)
      if (int rc = ppbo_gemv_async(ctx, d_Sigma_inv, N, N, ft, v, 0, 0, s, run_gf)) return rc;        // v = Sigma^-1 f
    lbfgs_step_kernel<<<1, LB_T, 0, s>>>(st, N, m, n_q, z, zt, dd, u, v, beta, tq, basis, dbg, prog, dot_part, n_parts);
    return 0;
  };
  int enq = 0, status = 0, evals = 0;
  bool stalled = false, joined = false;
  PpboSpinWait spin;
  spin.limit_s = 1e-3 * ctx->poll_limit_ms;
  for (;;) {
    const unsigned long long w = *h_word;
    status = (int)(w >> 32);
    evals = (int)(w & 0xffffffffu);
    if (status != 0) break;
    if (enq - evals < WH_AHEAD && enq < max_evals + WH_AHEAD) {
      if (ex.join_event && !joined && enq >= ex.gf_from) {
        PPBO_HIP_CHECK(ctx, hipStreamWaitEvent(s, ex.join_event, 0));
        joined = true;
      }
      if (int rc = enqueue_slot()) return rc;
      ++enq;
      continue;
    }
    if (!spin.idle(w)) continue;
    // no progress word for seconds (work queued in front of this call, ranks sharing the GPU, a profiler serialising
    // the stream ...): let the runtime wait for what is enqueued.  Afterwards every enqueued slot has run, so either the
    // search has ended, or it has advanced -- then it just needs more slots -- or the slots really do nothing.
    PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
    const unsigned long long w2 = *h_word;
    if ((int)(w2 >> 32) != 0 || (int)(w2 & 0xffffffffu) > evals) { spin.reset(); continue; }
    stalled = true;
    break;
  }
  PPBO_LAUNCH_CHECK(ctx);
  if (ex.join_event && !joined) PPBO_HIP_CHECK(ctx, hipStreamWaitEvent(s, ex.join_event, 0));   // the finisher, alpha, the posterior need them
  // (gated dead slots may still be queued behind the one that ended the search; they read the search's state in the
  // ctx's workspaces.  Every way out of here waits for the stream -- the standalone entry synchronises, ppbo_gp_fit's one
  // host wait is on a launch behind them -- so nothing of this search is in flight when the entry returns.)
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  WhHead head;
  if (stalled) {
    // a full synchronisation brought no new evaluation: read the state the ordinary way (the host-mapped word may be
    // the thing that is broken)
    PPBO_HIP_CHECK(ctx, hipMemcpy(&head, st, sizeof(WhHead), hipMemcpyDeviceToHost));
    if (head.status == 0)
      return ppbo_set_error(ctx, (int)hipErrorUnknown, "the whitened search made no progress (%d slots enqueued, %d evaluations)",
                            enq, head.evals);
  } else {
    std::memcpy(&head, h_head, sizeof(WhHead));
  }
  if (head.status == 6) {
    (void)hipStreamSynchronize(s);
    return ppbo_set_error(ctx, PPBO_ERR_NOT_PD, "the factor handed to the whitened search does not exist: its matrix is not positive definite");
  }
  if (verbose)
    printf("[ppbo_fit whitened] evals %d iters %d phi %.12e |grad_z| %.3e |grad_f| %.3e status %d (slots enqueued %d)\n",
           head.evals, head.iters, head.phi, std::sqrt(head.gz2), head.gf2 >= 0 ? std::sqrt(head.gf2) : -1.0, head.status, enq);
  if (dbg) {
    long long h[256];
    (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
    for (int e = 3; e < 10; ++e) {
      const long long* t = h + 16 * e;
      if (!t[1]) continue;
      if (!t[8]) continue;            // a backtracking slot ends before the later stamps
      printf("  [judgement at evaluation %d] pass1 %.2f | block sum %.2f | judge %.2f | gram %.2f | stop %.2f | two-loop %.2f | pass2 %.2f us\n",
             e, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01, (t[5] - t[4]) * 0.01, (t[6] - t[5]) * 0.01,
             (t[7] - t[6]) * 0.01, (t[8] - t[7]) * 0.01);
    }
  }
  const int lb_status = head.status, lb_iters = head.iters, lb_evals = head.evals;
  if (lb_status == 4) {
    // the start vector has no finite objective in whitened form: leave everything to the trust region
    if (ex.start_is_z) {
      if (int rc = ppbo_gemv_async(ctx, d_L, N, ldl, d_f_init, d_fMAP, 0, 1, s)) return rc;             // f = L z0
    } else {
      PPBO_HIP_CHECK(ctx, hipMemcpyAsync(d_fMAP, d_f_init, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
  } else {
    if (int rc = ppbo_gemv_async(ctx, d_L, N, ldl, z, d_fMAP, 0, 1, s)) return rc;                      // f = L z
  }
  if (lb_status == 1 && head.gf2 >= 0.0 && head.gf2 < gtol * gtol) {
    // ended on the reference's own rule |grad_f T| < gtol, evaluated at the accepted point: nothing is left for the
    // finisher to do (it used to re-evaluate the point and synchronise twice to find that out: ~0.1 ms)
    if (h_stats) {
      h_stats->iterations = 0; h_stats->n_cholesky = 0; h_stats->converged = 1;
      h_stats->T = -head.phi; h_stats->gradnorm = std::sqrt(head.gf2);
      h_stats->lbfgs_iterations = lb_iters; h_stats->lbfgs_evals = lb_evals; h_stats->lbfgs_status = lb_status;
    }
    if (ex.sync_at_end) PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
    return 0;
  }
  // finisher: the exact trust-region Newton from there
  ppbo_fit_stats tr{};
  ppbo_fit_opts fin{};
  if (opts) fin = *opts;
  fin.judge_by_gradient_below_noise = 1;
  const int rc = ppbo_fit_fmap(ctx, d_Sigma_inv, N, m, sigma, d_fMAP, &fin, d_fMAP, &tr, (void*)s);
  if (h_stats) {
    *h_stats = tr;
    h_stats->lbfgs_iterations = lb_iters;
    h_stats->lbfgs_evals = lb_evals;
    h_stats->lbfgs_status = lb_status;
  }
  return rc;
}

// Lambda_MAP, alpha, B = Sigma^-1 - Lambda = L_B L_B^T, R = L_B^-1, G = R Lambda (and P = R^T R) -- everything enqueued,
// the factorization's info word left on the device (d_info)
// the two info words of a fit into the ctx's host-mapped result record (as doubles), flag last: the host polls the flag
// instead of paying a device-to-host copy and a stream synchronisation for eight bytes
__global__ void publish_info_kernel(const int* __restrict__ info, int n, double* __restrict__ rec,
                                    unsigned long long* __restrict__ flag, unsigned long long epoch) {
  rec[0] = (double)info[0];
  rec[1] = n > 1 ? (double)info[1] : 0.0;
  __threadfence_system();
  __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int posterior_async(ppbo_ctx* ctx, const double* d_Sigma_inv, const double* d_fMAP, int N, int m, double sigma,
                    double* d_alpha, double* d_lam_diag, double* d_lam_off, double* d_G, double* d_P, int* d_info,
                    hipStream_t s) {
  const int mblk = m + 1, n_q = N / mblk;
  const size_t nn = (size_t)N * N;
  double* H = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_LINALG, 2 * nn * sizeof(double));
  if (!H) return (int)hipErrorOutOfMemory;
  double* R = H + nn;
  if (int rc = ppbo_gemv_async(ctx, d_Sigma_inv, N, N, d_fMAP, d_alpha, 0, 0, s)) return rc;
  laplace_kernel<<<(n_q + 3) / 4, 256, 0, s>>>(d_fMAP, N, mblk, n_q, sigma, nullptr, nullptr, d_lam_diag, d_lam_off);
  form_shifted_kernel<<<N, 256, 0, s>>>(d_Sigma_inv, N, mblk, d_lam_diag, d_lam_off, 0.0, H, nullptr);
  PPBO_LAUNCH_CHECK(ctx);
  if (int rc = ppbo_potrf_async(ctx, H, N, N, d_info, s)) return rc;
  if (int rc2 = ppbo_trtri_async(ctx, H, N, N, R, N, s, 0, nullptr, d_P ? 1 : 0)) return rc2;   // R^T R reads above the blocks, G = R Lambda does not
  g_build_kernel<<<dim3((N + 255) / 256, N), 256, 0, s>>>(R, N, mblk, d_lam_diag, d_lam_off, d_G);
  PPBO_LAUNCH_CHECK(ctx);
  if (d_P) {
    GemmArgs g{};  // P = R^T R
    g.A = R; g.lda = N; g.B = R; g.ldb = N; g.C = d_P; g.ldc = N;
    g.M = N; g.N = N; g.K = N; g.alpha = 1.0; g.beta = 0.0; g.klo_mode = 1; g.tri_block = 1;
    if (int rc3 = ppbo_gemm_launch(ctx, g, 1, 0, s)) return rc3;
  }
  return 0;
}

}  // namespace

extern "C" {

int ppbo_fit_fmap_whitened(ppbo_ctx* ctx, const double* d_L, int ldl, const double* d_Sigma_inv, int N, int m,
                           double sigma, const double* d_f_init, const ppbo_fit_opts* opts, double* d_fMAP,
                           ppbo_fit_stats* h_stats, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_L && d_Sigma_inv && d_f_init && d_fMAP, "null pointer");
  PPBO_REQUIRE(ctx, N > 0 && ldl >= N && m >= 1 && sigma > 0, "sizes");
  PPBO_REQUIRE(ctx, N % (m + 1) == 0, "N must be n_q*(m+1)");
  return whitened_search(ctx, d_L, ldl, d_Sigma_inv, N, m, sigma, d_f_init, opts, d_fMAP, h_stats, (hipStream_t)stream,
                         WhitenedExtras());
}

int ppbo_gp_fit(ppbo_ctx* ctx, int kernel_id, const double* d_X, int N, int D, const double theta[3], double shrink,
                int m, const double* d_f_init, const ppbo_fit_opts* opts, double* d_Sigma, double* d_Sigma_inv,
                double* d_L, double* d_Linv, double* d_fMAP, double* d_alpha, double* d_lam_diag, double* d_lam_off,
                double* d_G, ppbo_fit_stats* h_stats, int* h_info, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_X && theta && d_f_init && d_Sigma_inv && d_L && d_fMAP, "null pointer");
  PPBO_REQUIRE(ctx, (d_alpha && d_lam_diag && d_lam_off && d_G) || (!d_alpha && !d_lam_diag && !d_lam_off && !d_G),
               "the posterior outputs (alpha, lam_diag, lam_off, G) come together or not at all");
  PPBO_REQUIRE(ctx, N > 0 && D > 0 && m >= 1 && theta[0] > 0 && N % (m + 1) == 0, "sizes (N must be n_q*(m+1))");
  PPBO_REQUIRE(ctx, d_L != d_Sigma && d_L != d_Sigma_inv && d_Linv != d_Sigma_inv && d_Linv != d_L, "outputs must not alias");
  hipStream_t s = (hipStream_t)stream;
  if (h_info) *h_info = 0;
  const size_t nn = (size_t)N * N;
  double* W = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_LINALG, 2 * nn * sizeof(double));
  int* d_info = (int*)ppbo_workspace(ctx, ppbo_ctx::WS_SMALL, 4096);
  if (!W || !d_info) return (int)hipErrorOutOfMemory;
  double* Li = d_Linv ? d_Linv : W + nn;     // the posterior reuses this half for R -- after the GEMM below has read it
  // Sigma (kept for the caller) and the matrix the factorization overwrites: the Gram kernel runs twice (8.6 us at
  // N = 2048) instead of once plus a 33 MB device-to-device copy (12 us)
  // With a whitened start at N >= 1024 the |grad_f T| rule of the search is armed from evaluation fit_gf_from on -- in
  // the two-stream form because Sigma^-1 is not there before, in the one-stream form (PPBO_FIT_OVERLAP=0) so that both
  // give the same answer bit for bit: the result depends on the inputs, not on how the launches were laid out.
  const bool gf_deferred = opts && opts->start_is_whitened != 0 && N >= 1024;
  const bool overlap = ctx->fit_overlap && gf_deferred;
  hipStream_t s2 = s;
  // an error return after the fork must not leave the second stream working on buffers the caller may free
  struct SideGuard {
    hipStream_t side = nullptr;
    ~SideGuard() { if (side) (void)hipStreamSynchronize(side); }
  } side_guard;
  if (overlap) {
    if (!ctx->side_stream) {
      PPBO_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
      PPBO_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
      PPBO_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
    s2 = ctx->side_stream;
    side_guard.side = s2;
    PPBO_HIP_CHECK(ctx, hipEventRecord(ctx->ev_fork, s));            // X (and whatever else the caller enqueued) is there
    PPBO_HIP_CHECK(ctx, hipStreamWaitEvent(s2, ctx->ev_fork, 0));
  }
  if (d_Sigma)                                                        // the caller's copy: off the critical path
    if (int rc = ppbo_gram(ctx, kernel_id, d_X, N, D, theta, shrink, d_Sigma, (void*)s2)) return rc;
  if (int rc = ppbo_gram(ctx, kernel_id, d_X, N, D, theta, shrink, d_L, stream)) return rc;
  if (int rc = ppbo_potrf_async(ctx, d_L, N, N, d_info, s)) return rc;
  if (overlap) {
    PPBO_HIP_CHECK(ctx, hipEventRecord(ctx->ev_fork, s));            // the factor is there
    PPBO_HIP_CHECK(ctx, hipStreamWaitEvent(s2, ctx->ev_fork, 0));
  }
  if (int rc = ppbo_trtri_async(ctx, d_L, N, N, Li, N, s2)) return rc;
  // Sigma^-1 = L^-T L^-1: the lower triangle on the matrix cores, the upper one mirrored (bitwise symmetric).
  // With a whitened start the search needs only L until its |grad_f| rule is armed, so the triangular inverse and this
  // product run on the ctx's second stream beside the first evaluations; the search's stream joins them before slot
  // fit_gf_from, the first evaluation that may apply that rule (a fixed number: the result does not depend on timing).
  // Round 4 history: with the product on 64 x 64 tiles (1024 long-lived workgroups, 157 us) the search's small launches
  // starved beside it (profiles/r04_fit_side_stream.txt: no gain); on 32 x 32 tiles (90 us) the two streams do share the
  // chip: fit 2.49 -> 2.38 ms at N = 2048 (an LDS occupancy cap on the side GEMMs, 1 to 4 workgroups per CU, made no
  // difference either way).
  if (int rc = ppbo_syrk_inverse_async(ctx, Li, N, d_Sigma_inv, s2)) return rc;
  WhitenedExtras ex;
  if (overlap) {
    PPBO_HIP_CHECK(ctx, hipEventRecord(ctx->ev_join, s2));
    ex.join_event = ctx->ev_join;
  }
  if (gf_deferred) ex.gf_from = ctx->fit_gf_from;
  ex.d_factor_info = d_info;
  ex.d_Linv = Li;
  ex.start_is_z = opts && opts->start_is_whitened != 0;
  ex.sync_at_end = false;
  ppbo_fit_stats stt{};
  int rc = whitened_search(ctx, d_L, N, d_Sigma_inv, N, m, theta[0], d_f_init, opts, d_fMAP, &stt, s, ex);
  if (h_stats) *h_stats = stt;
  if (rc) {
    (void)hipStreamSynchronize(s);
    // whatever the search or its finisher made of it: a Sigma that is not positive definite is THE error of this call
    int info0 = 0;
    if (hipMemcpy(&info0, d_info, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess && info0 != 0) {
      if (h_info) *h_info = 1;
      return ppbo_set_error(ctx, PPBO_ERR_NOT_PD, "Sigma is not positive definite (leading minor %d)", info0);
    }
    return rc;
  }
  if (d_G)
    if (int rc2 = posterior_async(ctx, d_Sigma_inv, d_fMAP, N, m, theta[0], d_alpha, d_lam_diag, d_lam_off, d_G, nullptr,
                                  d_info + 1, s)) {
      (void)hipStreamSynchronize(s);
      return rc2;
    }
  // the call's ONE host wait: the last launch of the stream publishes the two info words through the host-mapped record
  PpboHostRecord hr;
  if (int rc3 = ppbo_host_record(ctx, &hr)) return rc3;
  publish_info_kernel<<<1, 1, 0, s>>>(d_info, d_G ? 2 : 1, hr.d_rec, hr.d_flag, hr.epoch);
  PPBO_LAUNCH_CHECK(ctx);
  if (int rc3 = ppbo_host_record_wait(ctx, hr, s)) return rc3;
  const int h2[2] = {(int)hr.h_rec[0], (int)hr.h_rec[1]};
  side_guard.side = nullptr;                      // s has waited for the second stream's event: everything is done
  if (h2[0] != 0) {
    if (h_info) *h_info = 1;
    return ppbo_set_error(ctx, PPBO_ERR_NOT_PD, "Sigma is not positive definite (leading minor %d)", h2[0]);
  }
  if (d_G && h2[1] != 0) {
    if (h_info) *h_info = 2;
    return ppbo_set_error(ctx, PPBO_ERR_NOT_PD, "Sigma^-1 - Lambda_MAP is not positive definite (leading minor %d)", h2[1]);
  }
  return 0;
}

int ppbo_posterior(ppbo_ctx* ctx, const double* d_Sigma_inv, const double* d_fMAP, int N, int m, double sigma,
                   double* d_alpha, double* d_lam_diag, double* d_lam_off, double* d_G, double* d_P, int* h_info,
                   void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_Sigma_inv && d_fMAP && d_alpha && d_lam_diag && d_lam_off && d_G, "null pointer");
  PPBO_REQUIRE(ctx, N > 0 && m >= 1 && sigma > 0 && N % (m + 1) == 0, "sizes");
  hipStream_t s = (hipStream_t)stream;
  if (h_info) *h_info = 0;
  int* d_info = (int*)ppbo_workspace(ctx, ppbo_ctx::WS_SMALL, 4096);
  if (!d_info) return (int)hipErrorOutOfMemory;
  // everything is enqueued first, the factorization's info word is looked at last: one host wait per call (the
  // triangular inverse of a failed factor is wasted work, but a failure is the rare case)
  if (int rc = posterior_async(ctx, d_Sigma_inv, d_fMAP, N, m, sigma, d_alpha, d_lam_diag, d_lam_off, d_G, d_P, d_info, s))
    return rc;
  int info = 0;
  PPBO_HIP_CHECK(ctx, hipMemcpyAsync(&info, d_info, sizeof(int), hipMemcpyDeviceToHost, s));
  PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
  if (h_info) *h_info = info;
  if (info != 0) return ppbo_set_error(ctx, PPBO_ERR_NOT_PD, "matrix is not positive definite (leading minor %d)", info);
  return 0;
}

}  // extern "C"
