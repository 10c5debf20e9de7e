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
  for (int f = f0; f < f1; ++f) s += Phi[(size_t)f * N + n] * omega[f];
  partial[(size_t)blockIdx.y * N + n] = s;
}

__global__ __launch_bounds__(256) void sum_parts_kernel(const double* __restrict__ partial, int n_split, int N,
                                                        double* __restrict__ y) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  double s = 0.0;
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

// One trust-region step of update_omega_MAP (random_fourier_sampler.py:124-132).  The Hessian of S is diagonal, so the
// subproblem trust-exact solves -- min -g.s + 1/2 s.(C s), |s| <= radius, C = diag(max(-h, 1e-12)) -- has the closed
// form s_i = g_i / (c_i + lam): lam = 0 when the Newton step fits, otherwise the root of |s(lam)| = radius, found by
// the More-Sorensen Newton iteration on 1/|s| - 1/radius (monotone from lam = 0 since every c_i + lam > 0).
// trial = omega + s;  out[0] = |Newton step| (>= radius <=> the step is on the boundary), out[1] = predicted gain
// g.s - 1/2 s.(C s), out[2] = |g|.
__global__ __launch_bounds__(1024) void rff_newton_step_kernel(const double* __restrict__ omega, const double* __restrict__ g,
                                                               const double* __restrict__ h, int F, double radius,
                                                               double* __restrict__ trial, double* __restrict__ out) {
  __shared__ double sh[2][16];
  __shared__ double s_a, s_b;
  const int t = threadIdx.x;
  auto block_sum2 = [&](double a, double b) {        // -> s_a, s_b on every thread
    a = wave_sum(a);
    b = wave_sum(b);
    if ((t & 63) == 0) { sh[0][t >> 6] = a; sh[1][t >> 6] = b; }
    __syncthreads();
    if (t == 0) {
      double x = 0.0, y = 0.0;
      for (int w = 0; w < 16; ++w) { x += sh[0][w]; y += sh[1][w]; }
      s_a = x; s_b = y;
    }
    __syncthreads();
  };
  double n2 = 0.0, g2 = 0.0;
  for (int f = t; f < F; f += 1024) {
    const double c = fmax(-h[f], 1e-12), st = g[f] / c;
    n2 += st * st;
    g2 += g[f] * g[f];
  }
  block_sum2(n2, g2);
  const double nrm = sqrt(s_a);
  if (t == 0) { out[0] = nrm; out[2] = sqrt(s_b); }
  double lam = 0.0;
  if (nrm > radius) {
    for (int it = 0; it < 40; ++it) {
      double a = 0.0, b = 0.0;                       // |s|^2 and s.(C + lam)^-1 s
      for (int f = t; f < F; f += 1024) {
        const double d = fmax(-h[f], 1e-12) + lam, st = g[f] / d;
        a += st * st;
        b += st * st / d;
      }
      __syncthreads();                               // everyone has read s_a / s_b of the previous round
      block_sum2(a, b);
      const double sn = sqrt(s_a);
      if (fabs(sn - radius) <= 1e-12 * radius || !(s_b > 0.0)) break;
      lam += (s_a / s_b) * ((sn - radius) / radius);
      if (lam < 0.0) lam = 0.0;
    }
  }
  double pred = 0.0;
  for (int f = t; f < F; f += 1024) {
    const double c = fmax(-h[f], 1e-12), st = g[f] / (c + lam);
    trial[f] = omega[f] + st;
    pred += g[f] * st - 0.5 * st * (c * st);
  }
  __syncthreads();
  block_sum2(pred, 0.0);
  if (t == 0) out[1] = s_a;
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
  // omega / gradient / Hessian diagonal of the accepted point and of the trial point, 4 scalars, pinned read-back
  double* buf = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_SEARCH_SMALL, ((size_t)6 * F + 8) * sizeof(double));
  double* host = (double*)ppbo_pinned(ctx, (64 + 8) * sizeof(double));   // out: 5 of its 8 doubles are used
  if (!buf || !host) return ppbo_set_error(ctx, (int)hipErrorOutOfMemory, "omega_MAP workspace");
  host += 64;                                    // the first 64 doubles of the pinned block belong to the fit
  double *om = buf, *g = buf + F, *h = buf + 2 * F, *omt = buf + 3 * F, *gt = buf + 4 * F, *ht = buf + 5 * F;
  double* out = buf + 6 * F;                     // [0] |unscaled step|, [1] predicted gain, [2] |g|, [3] S(trial)
  PPBO_HIP_CHECK(ctx, hipMemcpyAsync(om, d_omega, (size_t)F * sizeof(double), hipMemcpyDeviceToDevice, s));
  double* sc = nullptr;
  if (int rc = rff_terms_async(ctx, d_Phi, F, N, m, sigma, om, true, g, h, &sc, s)) return rc;
  PPBO_HIP_CHECK(ctx, hipMemcpyAsync(host, sc, sizeof(double), hipMemcpyDeviceToHost, s));
  PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
  double S = host[0], radius = 1.0, gn = INFINITY;
  bool gn_current = false;                       // gn belongs to the point in `om` (else: to the one before the last acceptance)
  int it = 0;
  for (; it < maxiter; ++it) {
    if (gn_current && gn < gtol) break;          // the accepted point is already stationary: no trial evaluation is paid for
    // the step from the accepted point and, speculatively, the terms at the trial point (S, gradient, |gradient|^2,
    // Hessian diagonal): ONE read-back per iteration
    rff_newton_step_kernel<<<1, 1024, 0, s>>>(om, g, h, F, radius, omt, out);
    if (int rc = rff_terms_async(ctx, d_Phi, F, N, m, sigma, omt, true, gt, ht, &sc, s)) return rc;
    PPBO_HIP_CHECK(ctx, hipMemcpyAsync(out + 3, sc, sizeof(double), hipMemcpyDeviceToDevice, s));
    if (int rc = ppbo_dot_async(ctx, gt, gt, F, out + 4, s)) return rc;
    PPBO_HIP_CHECK(ctx, hipMemcpyAsync(host, out, 5 * sizeof(double), hipMemcpyDeviceToHost, s));
    PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
    const double nrm = host[0], pred = host[1], Sn = host[3];
    gn = host[2];
    gn_current = true;
    if (gn < gtol) break;                        // the accepted point is stationary: the trial is not used
    const double rho = (pred > 0.0) ? (Sn - S) / pred : -1.0;
    if (rho < 0.25) radius *= 0.25;
    else if (rho > 0.75 && nrm >= radius) radius = (2.0 * radius < 1000.0) ? 2.0 * radius : 1000.0;
    if (rho > 0.15) {
      std::swap(om, omt); std::swap(g, gt); std::swap(h, ht);
      S = Sn;
      gn = std::sqrt(host[4]);                   // |gradient| AT the accepted point (what h_gradnorm reports)
    }
    if (radius < 1e-14) { ++it; break; }
  }
  PPBO_HIP_CHECK(ctx, hipMemcpyAsync(d_omega, om, (size_t)F * sizeof(double), hipMemcpyDeviceToDevice, s));
  PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
  if (h_S) *h_S = S;
  if (h_gradnorm) *h_gradnorm = gn;
  if (h_iterations) *h_iterations = it;
  return 0;
}

}  // extern "C"
