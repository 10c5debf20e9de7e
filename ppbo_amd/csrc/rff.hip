// K7/K8/K9: random Fourier features of the SE kernel.
//   reference: random_fourier_sampler.py:45-58 (phiVec/update_phi_X), :166,170 (phi(x)^T omega),
//              :106-122 (S, S_grad, S_hessian -- the Hessian is diagonal).
// Phi is [F, N] row-major (feature-major, as the reference stores phi_X).
//   rff_project: 64x64 output tiles, both operand panels staged transposed in LDS, 4x4
//                micro-tiles per lane, epilogue sqrt(2 sf^2/F) cos(. + b_f); HBM-write bound
//                (8 F N bytes), full 512-byte row-segment stores.
//   rff_score  : one lane = two candidates in registers, W/b/omega rows broadcast from LDS,
//                features split across blockIdx.y into partial slabs; Phi(Xc) never exists.
//   rff_terms  : f = Phi^T omega (split GEMV), per-query likelihood weights (one wavefront per
//                query), then one wavefront per feature row for S_grad / diag(S_hessian).
#include <chrono>
#include "linalg.h"
#include "rffmath.h"
#include "score.h"

namespace {

constexpr int TS = 64;
constexpr double INV_SQRT_4PI = 0.28209479177387814347;

[[maybe_unused]] __device__ __forceinline__ void stage_T(const double* __restrict__ X, int n, int D, int r0, double* __restrict__ dstT) {
  for (int e = threadIdx.x; e < TS * D; e += blockDim.x) {
    const int r = e / D, d = e - r * D;
    const int gr = r0 + r;
    dstT[d * TS + r] = (gr < n) ? X[(size_t)gr * D + d] : 0.0;
  }
}

__device__ __forceinline__ double rff_cos(double x, const RffPoly& P) {
  if (!(fabs(x) < RFF_COS_FAST_RANGE)) return P.c[0] * rff_cos_slow(x);
  return rff_cos_fast(x, P);
}

// Phi strip [64 features x NT*64 points] per workgroup.  The phase w_f.x_n + b_f comes off the fp64 matrix cores
// (depth DP, compile time; b_f is the accumulator's initial value), the epilogue is the cosine above.
// Eight wavefronts: wave = (16 feature rows, 32 points of every 64-point tile).
// What the time is made of at C3 (F = 4096, N = 2048, 67 MB; tools/dev/rff_dev.hip drops one phase at a time):
// the write-only floor of the chip is 10.3 us, the arithmetic alone 14.7 us (fp64 MFMA and fp64 VALU do not
// overlap), and -- the expensive surprise -- every store INSTRUCTION costs issue time that does not overlap
// either: with one 8-byte store per element the kernel took 20.7 us even when all stores hit one cached
// megabyte.  Hence WIDE: lane pairs trade one value (DPP) so that each lane owns two adjacent columns of one
// row and stores 16 bytes (half the store instructions; 20.7 -> 16.9 us), and NT = 4: one round of resident
// workgroups whose operand staging is paid once.
template <int DP, int NT, bool WIDE>
__global__ __launch_bounds__(512) void rff_project_kernel(const double* __restrict__ X, int N, int D,
                                                          const double* __restrict__ W, int F,
                                                          const double* __restrict__ b, RffPoly P,
                                                          double* __restrict__ Phi) {
  constexpr int LD = DP + 2, Q = DP / 4;
  __shared__ __attribute__((aligned(16))) double Wa[TS * LD];
  __shared__ __attribute__((aligned(16))) double Xb[NT * TS * LD];
  const int f0 = blockIdx.y * TS, n0 = blockIdx.x * (TS * NT);
  {
    // 4 lanes per row; rows 0-63 are the feature panel, the rest the point strip
    const int part = threadIdx.x & 3;
#pragma unroll
    for (int p = 0; p < (NT + 2) / 2; ++p) {
      const int row = p * 128 + (threadIdx.x >> 2);
      if (row >= (NT + 1) * TS) break;
      const bool isw = row < TS;
      const double* src = isw ? W : X;
      double* dst = isw ? Wa : Xb;
      const int r = isw ? row : row - TS;
      const int g = (isw ? f0 : n0) + r, lim = isw ? F : N;
#pragma unroll
      for (int k = 0; k < Q; ++k) {
        const int d = part * Q + k;
        dst[r * LD + d] = (d < D && g < lim) ? src[(size_t)g * D + d] : 0.0;
      }
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, w = wv & 3, jh = wv >> 2;
  const int lr = lane & 15, lk = lane >> 4;
  double af[Q], bv[4];
#pragma unroll
  for (int kk = 0; kk < Q; ++kk) af[kk] = Wa[(w * 16 + lr) * LD + kk * 4 + lk];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = f0 + w * 16 + lk + 4 * r;
    bv[r] = (f < F) ? b[f] : 0.0;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 4 * t + 2 * jh + jj;
      double4_t acc = double4_t{bv[0], bv[1], bv[2], bv[3]};
#pragma unroll
      for (int kk = 0; kk < Q; ++kk)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], Xb[(j * 16 + lr) * LD + kk * 4 + lk], acc, 0, 0, 0);
      double cv[4];
      bool big = false;
#pragma unroll
      for (int r = 0; r < 4; ++r) big |= !(fabs(acc[r]) < RFF_COS_FAST_RANGE);
#pragma unroll
      for (int r = 0; r < 4; ++r) cv[r] = rff_cos_fast(acc[r], P);      // four independent chains, no branches
      if (__builtin_amdgcn_ballot_w64(big)) {                            // never taken for RFF phases
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (!(fabs(acc[r]) < RFF_COS_FAST_RANGE)) cv[r] = P.c[0] * rff_cos_slow(acc[r]);
      }
      if (WIDE) {
        // even lanes keep rows lk + {0, 8}, odd lanes rows lk + {4, 12}; each stores its two adjacent columns
        const bool odd = lane & 1;
        const int nb = n0 + j * 16 + (lr & ~1);
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
          const double got = lane_xor1(odd ? cv[2 * rp] : cv[2 * rp + 1]);
          const double x0 = odd ? got : cv[2 * rp], x1 = odd ? cv[2 * rp + 1] : got;
          const int f = f0 + w * 16 + lk + 4 * (2 * rp + (odd ? 1 : 0));
          if (f < F && nb < N) store_through2(Phi + (size_t)f * N + nb, x0, x1);      // N is even here
        }
      } else {
        const int n = n0 + j * 16 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = f0 + w * 16 + lk + 4 * r;
          if (f < F && n < N) store_through(Phi + (size_t)f * N + n, cv[r]);
        }
      }
    }
  }
}

constexpr int RS_THREADS = 256;
constexpr int RS_RF = 32;  // feature rows staged per step

template <int DP>
__global__ __launch_bounds__(RS_THREADS) void rff_score_kernel(const double* __restrict__ Xc, int M, int D,
                                                               const double* __restrict__ W, int F,
                                                               const double* __restrict__ b,
                                                               const double* __restrict__ omega, RffPoly P,
                                                               int f_per_split, double* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) double ws[RS_RF * DP];
  __shared__ double s_b[RS_RF], s_om[RS_RF];
  const int c0 = (blockIdx.x * RS_THREADS + threadIdx.x) * 2;
  double xa[DP], xb[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) {
    xa[d] = (d < D && c0 < M) ? Xc[(size_t)c0 * D + d] : 0.0;
    xb[d] = (d < D && c0 + 1 < M) ? Xc[(size_t)(c0 + 1) * D + d] : 0.0;
  }
  const int f_beg = blockIdx.y * f_per_split;
  int f_end = f_beg + f_per_split;
  if (f_end > F) f_end = F;
  double a0 = 0.0, a1 = 0.0;
  for (int r0 = f_beg; r0 < f_end; r0 += RS_RF) {
    __syncthreads();
    for (int e = threadIdx.x; e < RS_RF * DP; e += RS_THREADS) {
      const int r = e / DP, d = e - r * DP;
      const int f = r0 + r;
      ws[e] = (f < f_end && d < D) ? W[(size_t)f * D + d] : 0.0;
    }
    if (threadIdx.x < RS_RF) {
      const int f = r0 + threadIdx.x;
      s_b[threadIdx.x] = (f < f_end) ? b[f] : 0.0;
      s_om[threadIdx.x] = (f < f_end) ? omega[f] : 0.0;   // zero weight kills padded rows
    }
    __syncthreads();
    const int rmax = (f_end - r0 < RS_RF) ? (f_end - r0) : RS_RF;
    for (int r = 0; r < rmax; ++r) {
      const double* __restrict__ wr = ws + r * DP;
      double s0 = s_b[r], s1 = s_b[r];
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        const double w = wr[d];
        s0 += w * xa[d];
        s1 += w * xb[d];
      }
      const double om = s_om[r];
      double v0 = rff_cos_fast(s0, P), v1 = rff_cos_fast(s1, P);       // amplitude included
      if (__builtin_amdgcn_ballot_w64(!(fabs(s0) < RFF_COS_FAST_RANGE) || !(fabs(s1) < RFF_COS_FAST_RANGE))) {
        if (!(fabs(s0) < RFF_COS_FAST_RANGE)) v0 = P.c[0] * rff_cos_slow(s0);
        if (!(fabs(s1) < RFF_COS_FAST_RANGE)) v1 = P.c[0] * rff_cos_slow(s1);
      }
      a0 += om * v0;
      a1 += om * v1;
    }
  }
  if (c0 < M) part[(size_t)blockIdx.y * M + c0] = a0;
  if (c0 + 1 < M) part[(size_t)blockIdx.y * M + c0 + 1] = a1;
}


// K8 on the matrix cores (default; PPBO_RFF_SCORE_MFMA=0 selects rff_score_kernel above): phases of a 16-feature x
// 16-candidate tile from one MFMA chain (b_f = the accumulator's initial value), cosine and omega-weighted row sum on
// the VALU.  Wavefront = CGV groups of 16 candidates whose B fragments stay in registers; 32 feature rows staged per
// step in LDS and read as A fragments by all four wavefronts.  Output layout of v_mfma_f64_16x16x4: lane (lr, lk) holds
// features lk + 4r of the tile for candidate lr, so the weighted feature sum is 4 FMAs per lane and two xor-shuffles
// at the very end.  Measured at C3 (M = 65536, F = 4096, D = 20; tools/rff_score_time.py): VALU kernel 0.48 ms,
// this one 0.38 ms with 4 groups per wave (0.40 with 2 groups at 4 waves / SIMD).  It is NOT the halving a free
// matrix pipe would give: the fp64 MFMA chain (268 M phases x 20 MAC at 16 MAC / cycle / SIMD = 0.14 ms) and the
// cosine (0.19 ms of VALU issue) add up rather than overlap -- fp64 MFMA and fp64 VALU share the CU's DP datapath.
template <int DP, int CGV, int MINW>
__global__ __launch_bounds__(256, MINW) void rff_score_mfma_kernel(const double* __restrict__ Xc, int M, int D,
                                                                const double* __restrict__ W, int F,
                                                                const double* __restrict__ b,
                                                                const double* __restrict__ omega, RffPoly P,
                                                                int f_per_split, double* __restrict__ part) {
  constexpr int Q = DP / 4, LD = DP + 2, CG = CGV, RJ = 32;
  __shared__ __attribute__((aligned(16))) double ws[RJ * LD];
  __shared__ double s_b[RJ], s_om[RJ];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
  const int cw = blockIdx.x * (64 * CG) + wv * (16 * CG);
  double xb[CG][Q], a[CG];
#pragma unroll
  for (int g = 0; g < CG; ++g) {
    const int c = cw + 16 * g + lr;
#pragma unroll
    for (int kk = 0; kk < Q; ++kk) {
      const int d = kk * 4 + lk;
      xb[g][kk] = (d < D && c < M) ? Xc[(size_t)c * D + d] : 0.0;
    }
    a[g] = 0.0;
  }
  const int f_beg = blockIdx.y * f_per_split;
  int f_end = f_beg + f_per_split;
  if (f_end > F) f_end = F;
  for (int r0 = f_beg; r0 < f_end; r0 += RJ) {
    __syncthreads();
    for (int e = threadIdx.x; e < RJ * DP; e += 256) {
      const int r = e / DP, d = e - r * DP;
      const int f = r0 + r;
      ws[r * LD + d] = (f < f_end && d < D) ? W[(size_t)f * D + d] : 0.0;
    }
    if (threadIdx.x < RJ) {
      const int f = r0 + threadIdx.x;
      s_b[threadIdx.x] = (f < f_end) ? b[f] : 0.0;
      s_om[threadIdx.x] = (f < f_end) ? omega[f] : 0.0;   // zero weight kills padded rows
    }
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < RJ / 16; ++t) {
      if (r0 + 16 * t >= f_end) break;
      double af[Q], br[4], omr[4];
#pragma unroll
      for (int kk = 0; kk < Q; ++kk) af[kk] = ws[(16 * t + lr) * LD + kk * 4 + lk];
#pragma unroll
      for (int r = 0; r < 4; ++r) { br[r] = s_b[16 * t + lk + 4 * r]; omr[r] = s_om[16 * t + lk + 4 * r]; }
#pragma unroll
      for (int g = 0; g < CG; ++g) {
        double4_t acc = double4_t{br[0], br[1], br[2], br[3]};
#pragma unroll
        for (int kk = 0; kk < Q; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], xb[g][kk], acc, 0, 0, 0);
        double v[4];
        bool big = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) big |= !(fabs(acc[r]) < RFF_COS_FAST_RANGE);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rff_cos_fast(acc[r], P);
        if (__builtin_amdgcn_ballot_w64(big)) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (!(fabs(acc[r]) < RFF_COS_FAST_RANGE)) v[r] = P.c[0] * rff_cos_slow(acc[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) a[g] = fma(omr[r], v[r], a[g]);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < CG; ++g) {
    double t = a[g];
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    const int c = cw + 16 * g + lr;
    if (lk == 0 && c < M) part[(size_t)blockIdx.y * M + c] = t;
  }
}

// partial[split][n] = sum_{f in split} Phi[f][n] omega[f]
__global__ __launch_bounds__(256) void phiT_omega_kernel(const double* __restrict__ Phi, int F, int N,
                                                         const double* __restrict__ omega, int f_per_split,
                                                         double* __restrict__ partial) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int f0 = blockIdx.y * f_per_split;
  int f1 = f0 + f_per_split;
  if (f1 > F) f1 = F;
  double s = 0.0;
#pragma unroll 8
  for (int f = f0; f < f1; ++f) s += Phi[(size_t)f * N + n] * omega[f];
  partial[(size_t)blockIdx.y * N + n] = s;
}

__global__ __launch_bounds__(256) void sum_parts_kernel(const double* __restrict__ partial, int n_split, int N,
                                                        double* __restrict__ y) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  double s = 0.0;
#pragma unroll 8
  for (int k = 0; k < n_split; ++k) s += partial[(size_t)k * N + j];
  y[j] = s;
}

// per query: tq = sum_j Phi(Delta_j/sqrt2); a_j = phi2(Delta_j)/(sigma m); h_j = -1/2 Delta_j phi2(Delta_j)/(m sigma^2)
__global__ __launch_bounds__(256) void rff_weights_kernel(const double* __restrict__ f, int N, int mblk, int n_q,
                                                          double sigma, double* __restrict__ tq,
                                                          double* __restrict__ a, double* __restrict__ h) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= n_q) return;
  const int m = mblk - 1, i = q * mblk;
  const double f0 = f[i];
  double sphi = 0.0;
  for (int r = 1 + lane; r <= m; r += 64) {
    const int j = i + r;
    const double delta = (f[j] - f0) / sigma;
    sphi += 0.5 * erfc(-0.5 * delta);
    const double p2 = INV_SQRT_4PI * exp(-0.25 * (delta * delta));
    a[j] = p2 / (sigma * (double)m);
    h[j] = -0.5 * delta * p2 / ((double)m * sigma * sigma);
  }
  sphi = wave_sum(sphi);
  if (lane == 0) { tq[q] = sphi; a[i] = 0.0; h[i] = 0.0; }
}

// one wavefront per feature row: grad_f = -omega_f - sum_n dPhi a_n ; hdiag_f = -1 - sum_n dPhi^2 h_n
__global__ __launch_bounds__(256) void rff_rows_kernel(const double* __restrict__ Phi, int F, int N, int mblk,
                                                       const double* __restrict__ omega,
                                                       const double* __restrict__ a, const double* __restrict__ h,
                                                       double* __restrict__ grad, double* __restrict__ hdiag) {
  const int lane = threadIdx.x & 63;
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= F) return;
  const double* row = Phi + (size_t)f * N;
  double sg = 0.0, sh = 0.0;
  for (int n = lane; n < N; n += 64) {
    const int o = (n / mblk) * mblk;
    const double dphi = row[n] - row[o];
    sg += dphi * a[n];
    sh += dphi * dphi * h[n];
  }
  sg = wave_sum(sg);
  sh = wave_sum(sh);
  if (lane == 0) {
    if (grad) grad[f] = -omega[f] - sg;
    if (hdiag) hdiag[f] = -1.0 - sh;
  }
}

// out[0] = -1/2 omega.omega - (1/m) sum tq
__global__ __launch_bounds__(1024) void rff_S_kernel(const double* __restrict__ omega, int F,
                                                     const double* __restrict__ tq, int n_q, int m,
                                                     double* __restrict__ out) {
  __shared__ double sh[2][16];
  double a = 0.0, t = 0.0;
  for (int i = threadIdx.x; i < F; i += 1024) a += omega[i] * omega[i];
  for (int i = threadIdx.x; i < n_q; i += 1024) t += tq[i];
  a = wave_sum(a);
  t = wave_sum(t);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = t; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0.0, st = 0.0;
    for (int w = 0; w < 16; ++w) { sa += sh[0][w]; st += sh[1][w]; }
    *out = -0.5 * sa - st / (double)m;
  }
}

// ---- update_omega_MAP as a device-resident loop ---------------------------------------------------------------
// The trust region's state lives on the device and one 1024-thread workgroup takes its decisions (om_step_kernel), so
// an iteration is FOUR launches and no read-back: f-partials at the trial point, weights (summing the partials on the
// way), the row pass (gradient, Hessian diagonal), and the step kernel, which first judges the trial it finds
// evaluated (S, |gradient|, rho, radius, acceptance) and then forms the next one.  The host enqueues slots a few
// ahead of the device and watches a host-mapped progress word, exactly as the whitened f_MAP search does (fit.hip).
// Before: 9 enqueues and a stream synchronisation per iteration (381 of them at C2).
struct OmHead {
  int status;        // 0 running | 1 |gradient| < gtol | 2 maxiter | 3 radius < 1e-14
  int it;            // steps taken
  int cur;           // which of the two (omega, gradient, Hessian diagonal) sets is the accepted point
  int pending;       // 0 nothing to judge | 1 a trial step | 2 the start point (accepted as it is)
  int maxiter, pad_;
  double S, radius, gn, nrm, pred, gtol;
};
struct OmProgress {
  unsigned long long* word;   // device view of the host-mapped progress word: status << 32 | it
  int* head;                  // device view of the host-mapped copy of OmHead (written when status != 0)
};

__global__ void om_init_kernel(OmHead* st, int maxiter, double gtol) {
  st->status = 0; st->it = 0; st->cur = 1; st->pending = 2; st->maxiter = maxiter; st->pad_ = 0;
  st->S = 0.0; st->radius = 1.0; st->gn = INFINITY; st->nrm = 0.0; st->pred = 0.0; st->gtol = gtol;
}

// partial[split][n] = sum_{f in split} Phi[f][n] omega_trial[f]
__global__ __launch_bounds__(256) void om_phiT_kernel(const OmHead* __restrict__ st, const double* __restrict__ Phi,
                                                      int F, int N, const double* __restrict__ buf, int f_per_split,
                                                      double* __restrict__ partial) {
  if (st->status != 0) return;
  const double* omega = buf + (size_t)(1 - st->cur) * 3 * F;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int f0 = blockIdx.y * f_per_split;
  int f1 = f0 + f_per_split;
  if (f1 > F) f1 = F;
  double s = 0.0;
#pragma unroll 8
  for (int f = f0; f < f1; ++f) s += Phi[(size_t)f * N + n] * omega[f];      // eight loads in flight per lane
  partial[(size_t)blockIdx.y * N + n] = s;
}

// rff_weights_kernel with f summed from the partials on the way (same order as sum_parts_kernel)
template <int NS>
__global__ __launch_bounds__(256) void om_weights_kernel(const OmHead* __restrict__ st, const double* __restrict__ partial,
                                                         int N, int mblk, int n_q, double sigma,
                                                         double* __restrict__ tq, double* __restrict__ a,
                                                         double* __restrict__ h) {
  if (st->status != 0) return;
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= n_q) return;
  const int m = mblk - 1, i = q * mblk;
  auto fsum = [&](int j) {               // all NS loads in flight (a runtime trip count made them a dependent chain)
    double v[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) v[k] = partial[(size_t)k * N + j];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NS; ++k) s += v[k];
    return s;
  };
  const double f0 = fsum(i);
  double sphi = 0.0;
  for (int r = 1 + lane; r <= m; r += 64) {
    const int j = i + r;
    const double delta = (fsum(j) - f0) / sigma;
    sphi += 0.5 * erfc(-0.5 * delta);
    const double p2 = INV_SQRT_4PI * exp(-0.25 * (delta * delta));
    a[j] = p2 / (sigma * (double)m);
    h[j] = -0.5 * delta * p2 / ((double)m * sigma * sigma);
  }
  sphi = wave_sum(sphi);
  if (lane == 0) { tq[q] = sphi; a[i] = 0.0; h[i] = 0.0; }
}

// rff_rows_kernel on the trial set
__global__ __launch_bounds__(256) void om_rows_kernel(const OmHead* __restrict__ st, const double* __restrict__ Phi,
                                                      int F, int N, int mblk, double* __restrict__ buf,
                                                      const double* __restrict__ a, const double* __restrict__ h) {
  if (st->status != 0) return;
  double* set = buf + (size_t)(1 - st->cur) * 3 * F;
  const int lane = threadIdx.x & 63;
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= F) return;
  const double* row = Phi + (size_t)f * N;
  double sg = 0.0, sh = 0.0;
  for (int n = lane; n < N; n += 64) {
    const int o = (n / mblk) * mblk;
    const double dphi = row[n] - row[o];
    sg += dphi * a[n];
    sh += dphi * dphi * h[n];
  }
  sg = wave_sum(sg);
  sh = wave_sum(sh);
  if (lane == 0) {
    set[F + f] = -set[f] - sg;
    set[2 * F + f] = -1.0 - sh;
  }
}

// Judge the evaluated trial, then form the next one: one trust-region step of update_omega_MAP
// (random_fourier_sampler.py:124-132; acceptance and radius rules are SciPy trust-exact's).  The Hessian of S is
// diagonal, so the subproblem trust-exact solves -- min -g.s + 1/2 s.(C s), |s| <= radius, C = diag(max(-h, 1e-12)) --
// has the closed form s_i = g_i / (c_i + lam): lam = 0 when the Newton step fits, otherwise the root of
// |s(lam)| = radius, found by the More-Sorensen Newton iteration on 1/|s| - 1/radius (monotone from lam = 0 since
// every c_i + lam > 0).  trial = omega + s; nrm = |Newton step| (>= radius <=> the step is on the boundary),
// pred = g.s - 1/2 s.(C s), gn = |g| at the accepted point (what h_gradnorm reports).
template <int PER>
__global__ __launch_bounds__(1024) void om_step_kernel(OmHead* __restrict__ st, double* __restrict__ buf, int F,
                                                       const double* __restrict__ tq, int n_q, int m,
                                                       double* __restrict__ omega_out, OmProgress prog) {
  // PER > 0: F <= 1024 PER and both sets sit in registers after ONE round of loads (every later pass over the
  // vectors -- |gradient|, each root-finding round, the trial point -- would otherwise be a round trip to L2 of its
  // own: 15 us per call at F = 1000).  PER = 0: any F, from memory.
  __shared__ double sh[2][16];
  __shared__ double s_r[2];
  __shared__ OmHead hs;
  constexpr int HW = (int)(sizeof(OmHead) / 4);
  constexpr int PR = PER > 0 ? PER : 1;
  const int t = threadIdx.x;
  double r_om[2][PR], r_g[2][PR], r_h[2][PR];
  if (t < HW) reinterpret_cast<int*>(&hs)[t] = reinterpret_cast<const int*>(st)[t];
  double q = 0.0;                                      // this thread's share of sum tq (same round of loads as the head)
  for (int i = t; i < n_q; i += 1024) q += tq[i];
  if constexpr (PER > 0) {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int f = t + 1024 * k;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const double* set = buf + (size_t)c * 3 * F;
        r_om[c][k] = f < F ? set[f] : 0.0;
        r_g[c][k] = f < F ? set[F + f] : 0.0;
        r_h[c][k] = f < F ? set[2 * F + f] : -1.0;
      }
    }
  }
  __syncthreads();
  if (hs.status != 0) return;
  // -> s_r[0], s_r[1] on every thread.  DPP sums: sixteen wavefronts reducing through __shfl_xor (ds_bpermute) queue
  // on the CU's one LDS pipe, ~0.5 us per sum (common.h)
  auto block_sum2 = [&](double a, double b) {
    a = wave_sum_dpp(a); b = wave_sum_dpp(b);
    if ((t & 63) == 0) { sh[0][t >> 6] = a; sh[1][t >> 6] = b; }
    __syncthreads();
    if (t < 2) {
      double x = 0.0;
#pragma unroll
      for (int w = 0; w < 16; ++w) x += sh[t][w];
      s_r[t] = x;
    }
    __syncthreads();
  };
  if (hs.pending) {
    // ---- the trial (or the start point) has been evaluated: S = -1/2 |omega|^2 - (1/m) sum tq
    const int tr = 1 - hs.cur;
    const double* set = buf + (size_t)tr * 3 * F;
    double a = 0.0;
    if constexpr (PER > 0) {
#pragma unroll
      for (int k = 0; k < PER; ++k) { const double o = tr ? r_om[1][k] : r_om[0][k]; a += o * o; }
    } else {
      for (int i = t; i < F; i += 1024) a += set[i] * set[i];
    }
    block_sum2(a, q);
    if (t == 0) {
      const double Sn = -0.5 * s_r[0] - s_r[1] / (double)m;
      if (hs.pending == 2) { hs.cur ^= 1; hs.S = Sn; }
      else {
        const double rho = (hs.pred > 0.0) ? (Sn - hs.S) / hs.pred : -1.0;
        if (rho < 0.25) hs.radius *= 0.25;
        else if (rho > 0.75 && hs.nrm >= hs.radius) hs.radius = (2.0 * hs.radius < 1000.0) ? 2.0 * hs.radius : 1000.0;
        if (rho > 0.15) { hs.cur ^= 1; hs.S = Sn; }
        if (hs.radius < 1e-14) hs.status = 3;
      }
      hs.pending = 0;
    }
    __syncthreads();
  }
  const int cur = hs.cur;
  const double* om = buf + (size_t)cur * 3 * F;
  const double* g = om + F;
  const double* h = om + 2 * F;
  // the accepted set: omega, gradient, c = max(-h, 1e-12)
  double a_om[PR], a_g[PR], a_c[PR];
  if constexpr (PER > 0) {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      a_om[k] = cur ? r_om[1][k] : r_om[0][k];
      a_g[k] = cur ? r_g[1][k] : r_g[0][k];
      a_c[k] = fmax(-(cur ? r_h[1][k] : r_h[0][k]), 1e-12);
    }
  }
  // body(f, omega_f, g_f, c_f) over this thread's entries (padding entries of the register path: g = 0, c = 1)
  auto each = [&](auto&& body) {
    if constexpr (PER > 0) {
#pragma unroll
      for (int k = 0; k < PER; ++k) body(t + 1024 * k, a_om[k], a_g[k], a_c[k]);
    } else {
      for (int f = t; f < F; f += 1024) body(f, om[f], g[f], fmax(-h[f], 1e-12));
    }
  };
  {
    // ---- |gradient| and the Newton step length at the accepted point (also when the search ends here: the reported
    // |gradient| belongs to the point that is returned)
    double n2 = 0.0, g2 = 0.0;
    each([&](int, double, double gf, double c) { const double sf = gf / c; n2 += sf * sf; g2 += gf * gf; });
    block_sum2(n2, g2);
    const double nrm = sqrt(s_r[0]), gn = sqrt(s_r[1]);
    __syncthreads();
    if (t == 0) {
      hs.gn = gn; hs.nrm = nrm;
      if (hs.status == 0) {
        if (gn < hs.gtol) hs.status = 1;
        else if (hs.it >= hs.maxiter) hs.status = 2;
      }
    }
    __syncthreads();
    if (hs.status == 0) {
      const double radius = hs.radius;
      double lam = 0.0;
      if (nrm > radius) {
        for (int it = 0; it < 40; ++it) {
          double a = 0.0, b = 0.0;                       // |s|^2 and s.(C + lam)^-1 s
          each([&](int, double, double gf, double c) { const double d = c + lam, sf = gf / d; a += sf * sf; b += sf * sf / d; });
          block_sum2(a, b);
          const double sa = s_r[0], sb = s_r[1];
          const double sn = sqrt(sa);
          __syncthreads();                               // everyone has read s_r before the next round overwrites it
          if (fabs(sn - radius) <= 1e-12 * radius || !(sb > 0.0)) break;
          lam += (sa / sb) * ((sn - radius) / radius);
          if (lam < 0.0) lam = 0.0;
        }
      }
      double* trial = buf + (size_t)(1 - cur) * 3 * F;
      double pred = 0.0;
      each([&](int f, double of, double gf, double c) {
        const double sf = gf / (c + lam);
        if (f < F) trial[f] = of + sf;
        pred += gf * sf - 0.5 * sf * (c * sf);
      });
      block_sum2(pred, 0.0);
      if (t == 0) { hs.pred = s_r[0]; hs.it += 1; hs.pending = 1; }
      __syncthreads();
    }
  }
  if (hs.status != 0)
    each([&](int f, double of, double, double) { if (f < F) omega_out[f] = of; });
  __syncthreads();
  if (t < HW) reinterpret_cast<int*>(st)[t] = reinterpret_cast<const int*>(&hs)[t];
  if (hs.status != 0) {
    __threadfence();
    if (t < HW && prog.head) prog.head[t] = reinterpret_cast<const int*>(&hs)[t];
    __threadfence_system();
    __syncthreads();
  }
  if (t == 0 && prog.word)
    __hip_atomic_store(prog.word, ((unsigned long long)(unsigned)hs.status << 32) | (unsigned)hs.it, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// S (optional, into *d_S_out[0]), grad S and diag(S_hessian) at omega, enqueued only
int rff_terms_async(ppbo_ctx* ctx, const double* d_Phi, int F, int N, int m, double sigma, const double* d_omega,
                    bool want_S, double* d_grad, double* d_hdiag, double** d_S_out, hipStream_t s) {
  const int mblk = m + 1, n_q = N / mblk;
  const int n_split = 32;
  const int f_per_split = (F + n_split - 1) / n_split;
  double* ws = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_VEC, ((size_t)(n_split + 3) * N + n_q + 8) * sizeof(double));
  if (!ws) return (int)hipErrorOutOfMemory;
  double* part = ws;
  double* f = ws + (size_t)n_split * N;
  double* a = f + N;
  double* h = a + N;
  double* tq = h + N;
  double* sc = tq + n_q;
  phiT_omega_kernel<<<dim3((N + 255) / 256, n_split), 256, 0, s>>>(d_Phi, F, N, d_omega, f_per_split, part);
  sum_parts_kernel<<<(N + 255) / 256, 256, 0, s>>>(part, n_split, N, f);
  rff_weights_kernel<<<(n_q + 3) / 4, 256, 0, s>>>(f, N, mblk, n_q, sigma, tq, a, h);
  if (d_grad || d_hdiag)
    rff_rows_kernel<<<(F + 3) / 4, 256, 0, s>>>(d_Phi, F, N, mblk, d_omega, a, h, d_grad, d_hdiag);
  if (want_S) rff_S_kernel<<<1, 1024, 0, s>>>(d_omega, F, tq, n_q, m, sc);
  PPBO_LAUNCH_CHECK(ctx);
  if (d_S_out) *d_S_out = sc;
  return 0;
}

}  // namespace

extern "C" {

int ppbo_rff_project(ppbo_ctx* ctx, const double* d_X, int N, int D, const double* d_W, int F,
                     const double* d_b, double sigma_f, double* d_Phi, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_X && d_W && d_b && d_Phi, "null pointer");
  PPBO_REQUIRE(ctx, N > 0 && D > 0 && D <= 256 && F > 0, "sizes");
  const double scale = std::sqrt(2.0 * sigma_f * sigma_f / (double)F);
  PPBO_REQUIRE(ctx, D <= 64, "D<=64");
  // strip length NT: one round of resident workgroups whenever the problem is large enough for it
  const long long tiles = (long long)((N + TS - 1) / TS) * ((F + TS - 1) / TS);
  int nt = ctx->rff_nt > 0 ? ctx->rff_nt : (tiles >= 2048 ? 4 : (tiles >= 1024 ? 2 : 1));
  if (nt != 1 && nt != 2 && nt != 4) nt = 1;
  const RffPoly P = make_rff_poly(scale);
  const bool wide = (N % 2 == 0) && ((reinterpret_cast<uintptr_t>(d_Phi) & 15) == 0);   // 16-byte row-pair stores
  dim3 grid((N + TS * nt - 1) / (TS * nt), (F + TS - 1) / TS);
  hipStream_t s = (hipStream_t)stream;
  PpboProfScope pf(ctx, ppbo_ctx::PF_RFF_PROJECT, s);
#define RP_GO(DPV, NTV)                                                                                            \
  do {                                                                                                             \
    if (wide) rff_project_kernel<DPV, NTV, true><<<grid, 512, 0, s>>>(d_X, N, D, d_W, F, d_b, P, d_Phi);           \
    else rff_project_kernel<DPV, NTV, false><<<grid, 512, 0, s>>>(d_X, N, D, d_W, F, d_b, P, d_Phi);               \
  } while (0)
#define RP_LAUNCH(DPV)            \
  do {                            \
    if (nt == 4) RP_GO(DPV, 4);   \
    else if (nt == 2) RP_GO(DPV, 2); \
    else RP_GO(DPV, 1);           \
  } while (0)
  if (D <= 4) RP_LAUNCH(4);
  else if (D <= 8) RP_LAUNCH(8);
  else if (D <= 12) RP_LAUNCH(12);
  else if (D <= 16) RP_LAUNCH(16);
  else if (D <= 20) RP_LAUNCH(20);
  else if (D <= 24) RP_LAUNCH(24);
  else if (D <= 32) RP_LAUNCH(32);
  else {   // deep panels: one 64-point tile per workgroup (the strip would not leave room for 4 workgroups per CU)
    grid = dim3((N + TS - 1) / TS, (F + TS - 1) / TS);
    if (D <= 48) RP_GO(48, 1);
    else RP_GO(64, 1);
  }
#undef RP_LAUNCH
#undef RP_GO
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

int ppbo_rff_score(ppbo_ctx* ctx, const double* d_Xc, int64_t M, int D, const double* d_W, int F,
                   const double* d_b, double sigma_f, const double* d_omega, double* d_score,
                   double* h_best_val, int64_t* h_best_idx, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_Xc && d_W && d_b && d_omega, "null pointer");
  PPBO_REQUIRE(ctx, M > 0 && D > 0 && D <= 64 && F > 0, "sizes (D<=64)");
  hipStream_t s = (hipStream_t)stream;
  const double scale = std::sqrt(2.0 * sigma_f * sigma_f / (double)F);
  const int64_t chunk_cap = 65536;
  const int64_t n_chunks = (M + chunk_cap - 1) / chunk_cap;
  const int Mc_max = (int)(M < chunk_cap ? M : chunk_cap);
  const int cblocks = (Mc_max + RS_THREADS * 2 - 1) / (RS_THREADS * 2);
  int n_split = (2048 + cblocks - 1) / cblocks;
  if (n_split > (F + RS_RF - 1) / RS_RF) n_split = (F + RS_RF - 1) / RS_RF;
  if (n_split > 64) n_split = 64;
  if (n_split < 1) n_split = 1;
  int f_per_split = (F + n_split - 1) / n_split;
  f_per_split = ((f_per_split + RS_RF - 1) / RS_RF) * RS_RF;
  n_split = (F + f_per_split - 1) / f_per_split;
  double* part = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_PART, (size_t)n_split * Mc_max * sizeof(double));
  if (!part) return (int)hipErrorOutOfMemory;
  const int sblocks_max = score_blocks(Mc_max);
  Best* bests = (Best*)ppbo_workspace(ctx, ppbo_ctx::WS_SMALL, (size_t)(sblocks_max + n_chunks) * sizeof(Best));
  if (!bests) return (int)hipErrorOutOfMemory;
  Best* chunk_best = bests + sblocks_max;
  for (int64_t ch = 0; ch < n_chunks; ++ch) {
    const int64_t c_beg = ch * chunk_cap;
    const int Mc = (int)((M - c_beg) < chunk_cap ? (M - c_beg) : chunk_cap);
    const double* xc = d_Xc + (size_t)c_beg * D;
    dim3 grid((Mc + RS_THREADS * 2 - 1) / (RS_THREADS * 2), n_split);
    PpboProfScope pf(ctx, ppbo_ctx::PF_RFF_SCORE, s);
    if (ctx->rff_score_mfma) {
      dim3 gm((Mc + 255) / 256, n_split);
#define RM_LAUNCH(DP) \
  rff_score_mfma_kernel<DP, 4, 2><<<gm, 256, 0, s>>>(xc, Mc, D, d_W, F, d_b, d_omega, make_rff_poly(scale), f_per_split, part)
      if (D <= 4) RM_LAUNCH(4);
      else if (D <= 8) RM_LAUNCH(8);
      else if (D <= 12) RM_LAUNCH(12);
      else if (D <= 16) RM_LAUNCH(16);
      else if (D <= 20) RM_LAUNCH(20);
      else if (D <= 24) RM_LAUNCH(24);
      else if (D <= 32) RM_LAUNCH(32);
      else if (D <= 48) RM_LAUNCH(48);
      else RM_LAUNCH(64);
#undef RM_LAUNCH
    } else {
#define RS_LAUNCH(DP) \
  rff_score_kernel<DP><<<grid, RS_THREADS, 0, s>>>(xc, Mc, D, d_W, F, d_b, d_omega, make_rff_poly(scale), f_per_split, part)
    if (D <= 4) RS_LAUNCH(4);
    else if (D <= 6) RS_LAUNCH(6);
    else if (D <= 8) RS_LAUNCH(8);
    else if (D <= 10) RS_LAUNCH(10);
    else if (D <= 12) RS_LAUNCH(12);
    else if (D <= 16) RS_LAUNCH(16);
    else if (D <= 20) RS_LAUNCH(20);
    else if (D <= 24) RS_LAUNCH(24);
    else if (D <= 32) RS_LAUNCH(32);
    else if (D <= 48) RS_LAUNCH(48);
    else RS_LAUNCH(64);
#undef RS_LAUNCH
    }
    PPBO_LAUNCH_CHECK(ctx);
    const int sblocks = score_blocks(Mc);
    score_kernel<<<sblocks, SC_THREADS, 0, s>>>(part, n_split, nullptr, nullptr, 0, Mc, 0.0, PPBO_SCORE_MEAN, 0.0,
                                         (long long)c_beg, nullptr, nullptr, d_score ? d_score + c_beg : nullptr,
                                         bests);
    argmax_final_kernel<<<1, 256, 0, s>>>(bests, sblocks, chunk_best + ch);
    PPBO_LAUNCH_CHECK(ctx);
  }
  return merge_chunk_bests(ctx, chunk_best, (int)n_chunks, h_best_val, h_best_idx, s);
}

int ppbo_rff_terms(ppbo_ctx* ctx, const double* d_Phi, int F, int N, int m, double sigma,
                   const double* d_omega, double* h_S, double* d_grad, double* d_hdiag, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_Phi && d_omega, "null pointer");
  PPBO_REQUIRE(ctx, F > 0 && N > 0 && m >= 1 && sigma > 0 && N % (m + 1) == 0, "sizes");
  hipStream_t s = (hipStream_t)stream;
  double* sc = nullptr;
  if (int rc = rff_terms_async(ctx, d_Phi, F, N, m, sigma, d_omega, h_S != nullptr, d_grad, d_hdiag, &sc, s)) return rc;
  if (h_S) {
    PPBO_HIP_CHECK(ctx, hipMemcpyAsync(h_S, sc, sizeof(double), hipMemcpyDeviceToHost, s));
    PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
  }
  return 0;
}

int ppbo_rff_omega_map(ppbo_ctx* ctx, const double* d_Phi, int F, int N, int m, double sigma, double* d_omega,
                       int maxiter, double gtol, double* h_S, double* h_gradnorm, int* h_iterations, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_Phi && d_omega, "null pointer");
  PPBO_REQUIRE(ctx, F > 0 && N > 0 && m >= 1 && sigma > 0 && N % (m + 1) == 0 && maxiter >= 0 && gtol >= 0, "sizes");
  hipStream_t s = (hipStream_t)stream;
  const int mblk = m + 1, n_q = N / mblk;
  constexpr int OM_SPLIT = 32;
  const int n_split = OM_SPLIT;
  const int f_per_split = (F + n_split - 1) / n_split;
  // two (omega, gradient, Hessian diagonal) sets -- the accepted point and the trial -- and the state
  const size_t st_doubles = (sizeof(OmHead) + 7) / 8;
  double* buf = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_SEARCH_SMALL, ((size_t)6 * F + st_doubles + 8) * sizeof(double));
  double* ws = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_VEC, ((size_t)(n_split + 2) * N + n_q + 8) * sizeof(double));
  if (!buf || !ws) return ppbo_set_error(ctx, (int)hipErrorOutOfMemory, "omega_MAP workspace");
  OmHead* st = reinterpret_cast<OmHead*>(buf + (size_t)6 * F);
  double* part = ws;
  double* a = ws + (size_t)n_split * N;
  double* h = a + N;
  double* tq = h + N;
  // host-mapped progress word + head copy (the ctx's result record block, laid out as the whitened search uses it)
  PpboHostRecord hr;
  if (int rc = ppbo_host_record(ctx, &hr)) return rc;
  static_assert(sizeof(OmHead) <= 16 * sizeof(double), "head copy fits its slot of the host-mapped block");
  OmProgress prog{reinterpret_cast<unsigned long long*>(hr.d_rec + 24), reinterpret_cast<int*>(hr.d_rec + 4)};
  volatile unsigned long long* h_word = reinterpret_cast<volatile unsigned long long*>(const_cast<double*>(hr.h_rec) + 24);
  const OmHead* h_head = reinterpret_cast<const OmHead*>(const_cast<double*>(hr.h_rec) + 4);
  *h_word = 0;                          // nothing of this ctx is in flight that could write it (one search per ctx at a time)
  om_init_kernel<<<1, 1, 0, s>>>(st, maxiter, gtol);
  PPBO_HIP_CHECK(ctx, hipMemcpyAsync(buf, d_omega, (size_t)F * sizeof(double), hipMemcpyDeviceToDevice, s));   // set 0 = the start point (cur = 1)
  auto enqueue_slot = [&]() {
    om_phiT_kernel<<<dim3((N + 255) / 256, n_split), 256, 0, s>>>(st, d_Phi, F, N, buf, f_per_split, part);
    om_weights_kernel<OM_SPLIT><<<(n_q + 3) / 4, 256, 0, s>>>(st, part, N, mblk, n_q, sigma, tq, a, h);
    om_rows_kernel<<<(F + 3) / 4, 256, 0, s>>>(st, d_Phi, F, N, mblk, buf, a, h);
    if (F <= 1024) om_step_kernel<1><<<1, 1024, 0, s>>>(st, buf, F, tq, n_q, m, d_omega, prog);
    else if (F <= 2048) om_step_kernel<2><<<1, 1024, 0, s>>>(st, buf, F, tq, n_q, m, d_omega, prog);
    else if (F <= 4096) om_step_kernel<4><<<1, 1024, 0, s>>>(st, buf, F, tq, n_q, m, d_omega, prog);
    else om_step_kernel<0><<<1, 1024, 0, s>>>(st, buf, F, tq, n_q, m, d_omega, prog);
  };
  constexpr int OM_AHEAD = 4;
  int enq = 0, status = 0, done = 0;
  bool stalled = false;
  PpboSpinWait spin;
  spin.limit_s = 1e-3 * ctx->poll_limit_ms;
  for (;;) {
    const unsigned long long w = *h_word;
    status = (int)(w >> 32);
    if (status != 0) break;
    done = (int)(w & 0xffffffffu);                // slot k's step kernel reports it = k + 1 (slot 0 evaluates the start point)
    if (enq - done < OM_AHEAD && enq < maxiter + 2) {
      enqueue_slot();
      ++enq;
      continue;
    }
    if (!spin.idle(w)) continue;
    // no progress word for seconds: let the runtime wait for what is enqueued; if that advanced the search it simply
    // needs more slots (a slow or shared device is not an error)
    PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
    const unsigned long long w2 = *h_word;
    if ((int)(w2 >> 32) != 0 || (int)(w2 & 0xffffffffu) > done) { spin.reset(); continue; }
    stalled = true;
    break;
  }
  PPBO_LAUNCH_CHECK(ctx);
  // gated dead slots still queued behind the one that ended the search read `st` and the vectors in the ctx's workspaces:
  // nothing of this search may be in flight when the entry returns (the next call may come in on another stream, or
  // with another F, which moves `st`)
  if (enq > done) PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  OmHead head;
  if (stalled) {
    PPBO_HIP_CHECK(ctx, hipMemcpy(&head, st, sizeof(OmHead), hipMemcpyDeviceToHost));
    if (head.status == 0)
      return ppbo_set_error(ctx, (int)hipErrorUnknown, "the omega_MAP search made no progress (%d slots enqueued)", enq);
  } else {
    std::memcpy(&head, h_head, sizeof(OmHead));
  }
  if (h_S) *h_S = head.S;
  if (h_gradnorm) *h_gradnorm = head.gn;
  if (h_iterations) *h_iterations = head.it;
  return 0;
}

}  // extern "C"
