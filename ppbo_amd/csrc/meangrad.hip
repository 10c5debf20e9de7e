// f-2: posterior mean and its analytic gradient at M points, the device half of the batched multi-start
// refinement that replaces mu_star's differential evolution (gp_model.py:415-437).
//   mu(x) = sum_i alpha_i k(x, x_i),   d mu / d x_d = sum_i alpha_i dk/dx_d
//   SE      dk/dx_d = -(x_d - x_i,d) / l^2 * k                               (kernels.py:19-25)
//   RQ      dk/dx_d = -(x_d - x_i,d) / l^2 * k / (1 + r^2 / (4 l^2))         (kernels.py:27-34, alpha = 2)
//   camphor dk/dx_d = -(2 pi / l^2) sin(2 pi (x_d - x_i,d)) * k  (d != 2),   -(x_2 - x_i,2) / (l + 0.05)^2 * k
//                                                                            (kernels.py:36-53)
// One 256-thread workgroup per point: lanes stride over the N design rows with the point held in
// registers, accumulate mu and D gradient components, then a shuffle + LDS reduction.  M is small here
// (a few hundred ascent iterates), so the work per launch is M * N * D * ~6 flops -- microseconds.
#include "common.h"
#include "rffmath.h"

namespace {

template <int KID, int DP>
__global__ __launch_bounds__(256) void mean_grad_kernel(const double* __restrict__ X, int N, int D, KernParams p,
                                                        const double* __restrict__ alpha,
                                                        const double* __restrict__ Xc, double* __restrict__ mu,
                                                        double* __restrict__ grad) {
  __shared__ double red[4][DP + 1];
  const int c = blockIdx.x;
  double xc[DP], g[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) {
    xc[d] = (d < D) ? Xc[(size_t)c * D + d] : 0.0;
    g[d] = 0.0;
  }
  double m = 0.0;
  for (int i = threadIdx.x; i < N; i += 256) {
    const double* __restrict__ xi = X + (size_t)i * D;
    double dx[DP], s = 0.0;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      dx[d] = (d < D) ? xc[d] - xi[d] : 0.0;
      s += kern_term<KID>(dx[d], d, p);
    }
    const double w = alpha[i] * kern_finish<KID>(s, p);
    m += w;
    if (KID == PPBO_KERNEL_CAMPHOR) {
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        if (d == 2) g[d] -= 2.0 * p.c1 * dx[d] * w;
        else if (d < 6) g[d] -= p.c0 * 3.14159265358979323846 * sinpi(2.0 * dx[d]) * w;
      }
    } else {
      const double coef = (KID == PPBO_KERNEL_SE) ? -2.0 * p.c0 * w : -4.0 * p.c0 * w / (1.0 + p.c0 * s);
#pragma unroll
      for (int d = 0; d < DP; ++d) g[d] += coef * dx[d];
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  m = wave_sum(m);
#pragma unroll
  for (int d = 0; d < DP; ++d) g[d] = wave_sum(g[d]);
  if (lane == 0) {
    red[wave][DP] = m;
#pragma unroll
    for (int d = 0; d < DP; ++d) red[wave][d] = g[d];
  }
  __syncthreads();
  if (threadIdx.x <= DP) {
    const int d = threadIdx.x;
    const double v = (red[0][d] + red[1][d]) + (red[2][d] + red[3][d]);
    if (d == DP) mu[c] = v;
    else if (d < D) grad[(size_t)c * D + d] = v;
  }
}

// ---------------------------------------------------------------------------------------------------
// Device-resident maximiser of the posterior mean (ppbo_mean_search): what mu_star's differential evolution
// (gp_model.py:415-437) is replaced by, without a host round trip per iterate.
//   1. group_max_kernel: the M scored candidates are cut into T <= 4096 groups of consecutive rows (T such that the
//      survivors and their coordinates fit one workgroup's LDS: 2622 at D = 6, 914 at D = 20); each group's best row
//      survives (the candidates are i.i.d. uniform, so this is a thinning, not a loss of coverage; the best candidate
//      always survives).
//   2. select_starts_kernel (one workgroup): greedy choice of the K best survivors that are pairwise more than
//      `sep` apart -- argmax, then strike everything within sep of the winner -- the rule the host loop applied.
//   3. mean_ascent_kernel: one workgroup per start runs the WHOLE projected Barzilai-Borwein ascent: evaluations
//      of mu and its gradient by all 256 threads (as mean_grad_kernel), the D-vector bookkeeping by wavefront 0 with
//      lane = coordinate (D <= 64), monotone safeguard, per-start stopping rule.

// The candidates of the trials of ONE mu_star call (ppbo_mean_search_multi), never materialised: trial t sees the
// resident uniform pool through its own rotation frac(pool + shift_t) (what ppbo_shift_points writes out for the
// one-trial entry: the same expression, the same bits), and trial 0 also the E extra points (the design, the previous
// x*).  Every trial has Mt = M + E slots; slots >= M of the later trials are absent (score -inf).
struct TrialCands {
  const double* pool = nullptr; long long M = 0; const double* shifts = nullptr;
  const double* extra = nullptr;   // E_rows rows (the design points: the model's own X, or caller-given points) ...
  const double* xprev = nullptr;   // ... followed by one more point (the previous x*) when given
  int E_rows = 0, E = 0, D = 0;    // E = E_rows + (xprev ? 1 : 0)
};
__device__ __forceinline__ double trial_coord(const TrialCands& c, int trial, long long i, int d) {
  if (i < c.M) { const double v = c.pool[(size_t)i * c.D + d] + c.shifts[(size_t)trial * c.D + d]; return v - floor(v); }
  const long long e = i - c.M;
  return e < c.E_rows ? c.extra[(size_t)e * c.D + d] : c.xprev[d];
}

// Screening pass of a mu_star trial: the posterior mean of every candidate, good enough to RANK them (the starts of the
// ascents are picked from it; every value that is reported comes from the fp64 ascent).  Kernel values in fp32 from
// direct differences -- v_exp_f32 instead of an 18-instruction fp64 exponential, fp32 FMAs at twice the fp64 rate --
// accumulated in fp64 (alpha has both signs and |mu| << sum |alpha_i k_i|).  Relative error of mu ~1e-6: two candidates
// closer than that may swap ranks.  grid (candidate blocks, row splits, trials); part[trial][split][Mt].
constexpr int SCR_T = 256, SCR_CPT = 4, SCR_RJ = 64;
typedef float float2_t __attribute__((ext_vector_type(2)));
// two candidates per packed operation (v_pk_add_f32 / v_pk_fma_f32: both halves at the price of one fp32 instruction)
template <int KID>
__device__ __forceinline__ float2_t screen_term2(float2_t dx, int d, float c0, float c1) {
  if (KID == PPBO_KERNEL_CAMPHOR) {
    if (d == 2) return c1 * dx * dx;
    float2_t sn;
    sn.x = sinpif(fabsf(dx.x)); sn.y = sinpif(fabsf(dx.y));
    return c0 * sn * sn;
  }
  return dx * dx;
}
template <int KID>
__device__ __forceinline__ float screen_finish(float s, float sf2, float c0) {
  if (KID == PPBO_KERNEL_RQ) { const float t = 1.0f + s * c0; return sf2 * __builtin_amdgcn_rcpf(t * t); }
  const float e = (KID == PPBO_KERNEL_SE) ? -c0 * s : -s;
  return sf2 * __builtin_amdgcn_exp2f(fmaxf(e * 1.44269504088896340736f, -126.0f));
}
template <int KID, int DP>
__global__ __launch_bounds__(SCR_T) void mean_screen_kernel(const double* __restrict__ X, int N, int D, KernParams p,
                                                            const double* __restrict__ alpha, TrialCands tc,
                                                            int rows_per_split, int n_split, double* __restrict__ part,
                                                            int extra_trial) {
  static_assert(SCR_CPT % 2 == 0, "candidates in packed pairs");
  constexpr int NP = SCR_CPT / 2;
  __shared__ __attribute__((aligned(16))) float xs[SCR_RJ][DP];
  __shared__ double sa[SCR_RJ];
  const int trial = blockIdx.z;
  const long long Mt = tc.M + tc.E;
  const long long c0 = ((long long)blockIdx.x * SCR_T + threadIdx.x) * SCR_CPT;
  float2_t xc[NP][DP];
  double mu[SCR_CPT];
#pragma unroll
  for (int q = 0; q < SCR_CPT; ++q) {
    mu[q] = 0.0;
    const bool there = c0 + q < tc.M || (trial == extra_trial && c0 + q < Mt);   // (the launch's trial that carries the extra points)
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      const float v = (d < D && there) ? (float)trial_coord(tc, trial, c0 + q, d) : 0.0f;
      if (q & 1) xc[q >> 1][d].y = v; else xc[q >> 1][d].x = v;
    }
  }
  const float c0f = (float)p.c0, c1f = (float)p.c1, sf2f = (float)p.sf2;
  const int j_beg = blockIdx.y * rows_per_split;
  int j_end = j_beg + rows_per_split;
  if (j_end > N) j_end = N;
  for (int row0 = j_beg; row0 < j_end; row0 += SCR_RJ) {
    __syncthreads();
    for (int e = threadIdx.x; e < SCR_RJ * DP; e += SCR_T) {
      const int r = e / DP, d = e - r * DP, j = row0 + r;
      xs[r][d] = (j < j_end && d < D) ? (float)X[(size_t)j * D + d] : 0.0f;
    }
    if (threadIdx.x < SCR_RJ) sa[threadIdx.x] = (row0 + (int)threadIdx.x < j_end) ? alpha[row0 + threadIdx.x] : 0.0;
    __syncthreads();
    const int rmax = (j_end - row0 < SCR_RJ) ? (j_end - row0) : SCR_RJ;
    for (int r = 0; r < rmax; ++r) {
      float2_t sv[NP];
#pragma unroll
      for (int q = 0; q < NP; ++q) sv[q] = float2_t{0.0f, 0.0f};
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        const float x = xs[r][d];
        const float2_t xx = {x, x};
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          if (KID == PPBO_KERNEL_CAMPHOR) sv[q] += screen_term2<KID>(xx - xc[q][d], d, c0f, c1f);
          else { const float2_t dx = xx - xc[q][d]; sv[q] = __builtin_elementwise_fma(dx, dx, sv[q]); }
        }
      }
      const double a = sa[r];
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        mu[2 * q] = fma(a, (double)screen_finish<KID>(sv[q].x, sf2f, c0f), mu[2 * q]);
        mu[2 * q + 1] = fma(a, (double)screen_finish<KID>(sv[q].y, sf2f, c0f), mu[2 * q + 1]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < SCR_CPT; ++q)
    if (c0 + q < Mt) part[((size_t)trial * n_split + blockIdx.y) * Mt + c0 + q] = mu[q];
}

// mu[trial][c] = sum over the row splits (fixed order); absent slots -inf
__global__ __launch_bounds__(256) void screen_sum_kernel(const double* __restrict__ part, int n_split, long long Mt,
                                                         long long M, int extra_trial, double* __restrict__ mu) {
  const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
  const int trial = blockIdx.y;
  if (c >= Mt) return;
  double s = 0.0;
  if (c >= M && trial != extra_trial) s = -INFINITY;
  else
    for (int k = 0; k < n_split; ++k) s += part[((size_t)trial * n_split + k) * Mt + c];
  mu[(size_t)trial * Mt + c] = s;
}

__global__ __launch_bounds__(256) void fill_kernel(double* __restrict__ p, long long n, double v) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

// candidates of trial t as rows (the fp64 screening path scores them with ppbo_predict)
__global__ __launch_bounds__(256) void trial_rows_kernel(TrialCands tc, int trial, double* __restrict__ out) {
  const long long n = (tc.M + (trial == 0 ? tc.E : 0)) * tc.D;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  out[e] = trial_coord(tc, trial, e / tc.D, (int)(e % tc.D));
}

// blockIdx.y = trial: mu, gval, gidx are per-trial arrays of M (resp. T) entries
__global__ __launch_bounds__(256) void group_max_kernel(const double* __restrict__ mu, int64_t M, int G, int T,
                                                        double* __restrict__ gval, int* __restrict__ gidx) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  mu += (size_t)blockIdx.y * M; gval += (size_t)blockIdx.y * T; gidx += (size_t)blockIdx.y * T;
  const int64_t lo = (int64_t)t * G, hi = (lo + G < M) ? lo + G : M;
  double best = -INFINITY;
  int64_t bi = lo;
  for (int64_t i = lo; i < hi; ++i) {
    const double v = mu[i];
    if (v > best) { best = v; bi = i; }         // NaN never wins
  }
  gval[t] = best;
  gidx[t] = (int)bi;
}

// (score, index) argmax over a wavefront by DPP (no LDS crossbar): larger score wins, ties go to the smaller index.
struct SelRec { double v; int i; };
__device__ __forceinline__ SelRec sel_merge(SelRec a, SelRec b) {
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
template <int CTRL>
__device__ __forceinline__ SelRec sel_dpp_step(SelRec a) {
  SelRec o;
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(a.v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(a.v), CTRL, 0xF, 0xF, true);
  o.i = __builtin_amdgcn_mov_dpp(a.i, CTRL, 0xF, 0xF, true);
  o.v = __hiloint2double(hi, lo);
  return sel_merge(a, o);
}
__device__ __forceinline__ SelRec wave_sel(SelRec a) {
  a = sel_dpp_step<0xB1>(a);
  a = sel_dpp_step<0x4E>(a);
  a = sel_dpp_step<0x141>(a);
  a = sel_dpp_step<0x140>(a);
  SelRec r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r[k].v = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a.v), 16 * k),
                              __builtin_amdgcn_readlane(__double2loint(a.v), 16 * k));
    r[k].i = __builtin_amdgcn_readlane(a.i, 16 * k);
  }
  return sel_merge(sel_merge(r[0], r[1]), sel_merge(r[2], r[3]));
}

// One workgroup.  The T survivors' scores AND coordinates live in LDS (the host sizes T for it): a pick is an argmax
// (DPP per wavefront, every wavefront merges the 16 wave records itself) and a strike pass over LDS -- no global
// round trip inside the K-step loop (the first form re-read gidx and the coordinates from memory in every step:
// 13 us per pick, 0.44 of the 0.68 ms of a mu_star trial).
// blockIdx.x = trial (ppbo_mean_search_multi; the one-trial entries launch one workgroup): every per-trial array is
// offset by it, and with tc.pool the candidate coordinates are formed on the fly (TrialCands) instead of read from `cand`.
__global__ __launch_bounds__(1024) void select_starts_kernel(const double* __restrict__ gval,
                                                             const int* __restrict__ gidx, int T,
                                                             const double* __restrict__ cand, int D, int K, double sep2,
                                                             double* __restrict__ starts, int* __restrict__ count,
                                                             TrialCands tc) {
  extern __shared__ double sv[];          // [T] survivor scores (struck: -inf) | [T][D] coordinates | 16 wave records
  double* xc = sv + T;
  __shared__ double wv[16];
  __shared__ int wi[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int trial = blockIdx.x;
  gval += (size_t)trial * T; gidx += (size_t)trial * T; starts += (size_t)trial * K * D; count += trial;
  for (int t = tid; t < T; t += 1024) sv[t] = gval[t];
  for (int e = tid; e < T * D; e += 1024) {
    const int t = e / D, d = e - t * D;
    xc[e] = tc.pool ? trial_coord(tc, trial, gidx[t], d) : cand[(size_t)gidx[t] * D + d];
  }
  __syncthreads();
  int k = 0;
  for (; k < K; ++k) {
    SelRec best{-INFINITY, 0x7fffffff};
    for (int t = tid; t < T; t += 1024) {
      const double v = sv[t];
      if (v > best.v) { best.v = v; best.i = t; }       // ascending t per thread: first index wins
    }
    best = wave_sel(best);
    if (lane == 0) { wv[wave] = best.v; wi[wave] = best.i; }
    __syncthreads();
    SelRec mine{-INFINITY, 0x7fffffff};
    if (lane < 16) { mine.v = wv[lane]; mine.i = wi[lane]; }
    const SelRec top = wave_sel(mine);                  // the same in every wavefront
    if (!(top.v > -INFINITY)) break;
    const int w = top.i;
    const double* pw = xc + (size_t)w * D;
    if (tid < D) starts[(size_t)k * D + tid] = pw[tid];
    for (int t = tid; t < T; t += 1024) {
      if (!(sv[t] > -INFINITY)) continue;
      const double* pt = xc + (size_t)t * D;
      double d2 = 0.0;
      for (int d = 0; d < D; ++d) { const double dx = pt[d] - pw[d]; d2 += dx * dx; }
      if (d2 <= sep2) sv[t] = -INFINITY;        // strikes the winner itself too
    }
    __syncthreads();                             // sv and the wave records are rewritten by the next pick
  }
  if (tid == 0) *count = k;
}

// survivors of the thinning: as many as fit one workgroup's LDS next to their coordinates (<= 4096)
static inline int select_capacity(int D) {
  const int cap = (144 * 1024) / (8 + 8 * D);
  return cap > 4096 ? 4096 : (cap < 64 ? 64 : cap);
}

// mu and its gradient at the point held in LDS (sx), partial sums of this thread's rows reduced into red[wave][.]
// TR: X is given TRANSPOSED ([D][N]): the threads of a wavefront then read consecutive addresses per coordinate.  With
// row-major X every lane reads its own row (a 8 D-byte segment each, 64 cache lines per load instruction) and the
// ascent is bound by the L1's transaction rate, not by arithmetic or latency.
template <int KID, int DP, int NT = 256, bool TR = false>
__device__ __forceinline__ void eval_mean_grad(const double* __restrict__ X, int N, int D, const KernParams& p,
                                               const double* __restrict__ alpha, const double* sx,
                                               double (*red)[DP + 1]) {
  double xc[DP], g[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) { xc[d] = sx[d]; g[d] = 0.0; }
  double m = 0.0;
  for (int i = threadIdx.x; i < N; i += NT) {
    const double* __restrict__ xi = TR ? X + i : X + (size_t)i * D;
    const size_t xs = TR ? (size_t)N : 1;
    double dx[DP], s = 0.0;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      dx[d] = (d < D) ? xc[d] - xi[d * xs] : 0.0;
      s += kern_term<KID>(dx[d], d, p);
    }
    const double w = alpha[i] * kern_finish<KID>(s, p);
    m += w;
    if (KID == PPBO_KERNEL_CAMPHOR) {
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        if (d == 2) g[d] -= 2.0 * p.c1 * dx[d] * w;
        else if (d < 6) g[d] -= p.c0 * 3.14159265358979323846 * sinpi(2.0 * dx[d]) * w;
      }
    } else {
      const double coef = (KID == PPBO_KERNEL_SE) ? -2.0 * p.c0 * w : -4.0 * p.c0 * w / (1.0 + p.c0 * s);
#pragma unroll
      for (int d = 0; d < DP; ++d) g[d] += coef * dx[d];
    }
  }
  // DPP sums over rows of 16 lanes (DP + 1 reductions through ds_bpermute would queue on the LDS pipe, and finishing
  // each of them across the four rows costs eight v_readlane more): one record per row, 4 per wavefront
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  m = row16_sum_dpp(m);
#pragma unroll
  for (int d = 0; d < DP; ++d) g[d] = row16_sum_dpp(g[d]);
  if ((lane & 15) == 0) {
    double* r = red[4 * wave + (lane >> 4)];
    r[DP] = m;
#pragma unroll
    for (int d = 0; d < DP; ++d) r[d] = g[d];
  }
}

// the objective of an ascent: fills red[wave][0..DP) with the gradient's and red[wave][DP] with the value's partial sums
template <int KID, int DP, int NT>
struct MeanEval {
  const double* X; int N, D; KernParams p; const double* alpha;
  __device__ __forceinline__ void operator()(const double* sx, double (*red)[DP + 1]) const {
    eval_mean_grad<KID, DP, NT, true>(X, N, D, p, alpha, sx, red);       // X: the transposed design [D][N]
  }
};

// one posterior sample of the utility in weight space (random_fourier_sampler.py:45-53,166):
//   f(x) = a sum_f omega_f cos(w_f.x + b_f),   grad f = -a sum_f omega_f sin(w_f.x + b_f) w_f,   a = sqrt(2 sf^2 / F)
template <int DP, int NT>
struct RffEval {
  const double* W; int F, D; const double* b; const double* omega; double amp; RffPoly P;   // P: unit amplitude
  __device__ __forceinline__ void operator()(const double* sx, double (*red)[DP + 1]) const {
    double xc[DP], g[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) { xc[d] = sx[d]; g[d] = 0.0; }
    double m = 0.0;
    for (int f = threadIdx.x; f < F; f += NT) {
      const double* __restrict__ wf = W + f;                 // W: the transposed basis [D][F] (coalesced per coordinate)
      double wv[DP], ph = b[f];
#pragma unroll
      for (int d = 0; d < DP; ++d) { wv[d] = (d < D) ? wf[(size_t)d * F] : 0.0; ph = fma(wv[d], xc[d], ph); }
      // cos and sin = cos(. - pi/2) by the branch-free polynomial of the RFF kernels (2 x 20 instructions against a
      // library sincos with its own range reduction); phases beyond its range take the library path
      double sn, cs;
      if (fabs(ph) < 0.5 * RFF_COS_FAST_RANGE) {
        cs = rff_cos_fast(ph, P);
        sn = rff_cos_fast(ph - 1.57079632679489661923, P);
      } else {
        sincos(ph, &sn, &cs);
      }
      const double om = amp * omega[f];
      m = fma(om, cs, m);
      const double c = -om * sn;
#pragma unroll
      for (int d = 0; d < DP; ++d) g[d] = fma(c, wv[d], g[d]);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    m = row16_sum_dpp(m);
#pragma unroll
    for (int d = 0; d < DP; ++d) g[d] = row16_sum_dpp(g[d]);
    if ((lane & 15) == 0) {
      double* r = red[4 * wave + (lane >> 4)];
      r[DP] = m;
#pragma unroll
      for (int d = 0; d < DP; ++d) r[d] = g[d];
    }
  }
};

// fixed-order sum of the NW records of column c
template <int NW, int DP>
__device__ __forceinline__ double red_col(const double (*red)[DP + 1], int c) {
  double t = red[0][c];
#pragma unroll
  for (int w = 1; w < NW; ++w) t += red[w][c];
  return t;
}

template <int DP, class EVAL, int NT>
__global__ __launch_bounds__(NT) void bb_ascent_kernel(EVAL ev, int D, const double* __restrict__ starts,
                                                        const int* __restrict__ count, int iters, double tol,
                                                        double* __restrict__ x_out, double* __restrict__ mu_out,
                                                        int* __restrict__ it_out, int per_trial) {
  static_assert(DP <= 64, "lane = coordinate");
  constexpr int NW = NT / 16;              // one record per row of 16 lanes
  __shared__ double red[NW][DP + 1];
  __shared__ double sx[DP];
  __shared__ int done;
  const int c = blockIdx.x, tid = threadIdx.x;
  // count: starts that exist -- one number for the launch (per_trial = 0) or one per block of per_trial starts (the
  // trials of ppbo_mean_search_multi)
  if (count && (per_trial > 0 ? (c % per_trial) >= count[c / per_trial] : c >= *count)) {
    if (tid == 0) { mu_out[c] = -INFINITY; if (it_out) it_out[c] = 0; }
    return;
  }
  const bool w0 = tid < 64;
  const int d = tid;                       // coordinate of this lane (wavefront 0 only)
  const bool live = w0 && d < D;
  double x = 0.0, g = 0.0, mu = 0.0, step = 0.0, xn = 0.0;
  if (tid < DP) {
    x = live ? fmin(fmax(starts[(size_t)c * D + d], 0.0), 1.0) : 0.0;
    sx[tid] = x;
  }
  if (tid == 0) done = 0;
  __syncthreads();
  ev(sx, red);
  __syncthreads();
  if (w0) {
    const int dd = d < DP ? d : DP - 1;
    g = live ? red_col<NW, DP>(red, dd) : 0.0;
    mu = red_col<NW, DP>(red, DP);
    const double gn = sqrt(wave_sum_dpp(g * g));
    step = 0.02 / fmax(gn, 1e-300);        // first move: 0.02 in the unit box
  }
  int it = 0;
  for (; it < iters; ++it) {
    if (w0) {
      const double pg = ((x <= 0.0 && g < 0.0) || (x >= 1.0 && g > 0.0)) ? 0.0 : g;
      const double pn = sqrt(wave_sum_dpp(pg * pg));
      if (!(pn * step >= tol)) { if (tid == 0) done = 1; }
      else {
        xn = fmin(fmax(x + step * pg, 0.0), 1.0);
        if (tid < DP) sx[tid] = live ? xn : 0.0;
      }
    }
    __syncthreads();
    if (done) break;
    ev(sx, red);
    __syncthreads();
    if (w0) {
      const int dd = d < DP ? d : DP - 1;
      const double gnew = live ? red_col<NW, DP>(red, dd) : 0.0;
      const double mun = red_col<NW, DP>(red, DP);
      const bool ok = mun >= mu;
      const double sv = live ? xn - x : 0.0, yv = gnew - g;
      const double curv = -wave_sum_dpp(sv * yv);          // > 0 where mu is locally concave along the move
      const double ss = wave_sum_dpp(sv * sv);
      step = ok ? (curv > 0.0 ? ss / fmax(curv, 1e-300) : 2.0 * step) : 0.25 * step;
      if (ok) { x = xn; g = gnew; mu = mun; }
    }
    // the next write to sx / red happens after every wave has passed the barrier above
  }
  if (live) x_out[(size_t)c * D + d] = x;
  if (tid == 0) { mu_out[c] = mu; if (it_out) it_out[c] = it; }
}

// out = frac(in + shift): a Cranley-Patterson rotation of a resident uniform candidate pool (keeps it uniform)
__global__ __launch_bounds__(256) void shift_points_kernel(const double* __restrict__ in, int64_t n, int D,
                                                           const double* __restrict__ shift,
                                                           double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double v = in[i] + shift[i % D];
  out[i] = v - floor(v);
}

// out[d][r] = in[r][d]  (R x D row-major -> D x R): the ascent kernels read their operand transposed
__global__ __launch_bounds__(256) void transpose_rows_kernel(const double* __restrict__ in, int R, int D,
                                                             double* __restrict__ out) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  for (int d = 0; d < D; ++d) out[(size_t)d * R + r] = in[(size_t)r * D + d];
}

template <int KID>
int launch_mean_ascent(ppbo_ctx* ctx, const ppbo_model* m, const KernParams& p, const double* starts, const int* count, int K,
                       int iters, double tol, double* x_out, double* mu_out, int* it_out, hipStream_t s, int per_trial = 0) {
  double* Xt = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_TRANSPOSE, (size_t)m->N * m->D * sizeof(double));
  if (!Xt) return (int)hipErrorOutOfMemory;
  transpose_rows_kernel<<<(m->N + 255) / 256, 256, 0, s>>>(m->d_X, m->N, m->D, Xt);
  // threads per start: one wavefront per SIMD (256 threads) leaves every dependent fp64 instruction's latency exposed;
  // more wavefronts hide it, as far as the registers of the dimension bucket allow (and the design has rows for them)
#define MA_LAUNCH(DP, NT)                                                                                             \
  do {                                                                                                                \
    MeanEval<KID, DP, NT> ev{Xt, m->N, m->D, p, m->d_alpha};                                                        \
    bb_ascent_kernel<DP, MeanEval<KID, DP, NT>, NT><<<K, NT, 0, s>>>(ev, m->D, starts, count, iters, tol, x_out, mu_out, it_out, per_trial); \
  } while (0)
  const bool tall = m->N >= 1024;
  if (KID == PPBO_KERNEL_CAMPHOR || m->D <= 8) { if (tall) MA_LAUNCH(8, 1024); else MA_LAUNCH(8, 256); }
  else if (m->D <= 24) { if (tall) MA_LAUNCH(24, 512); else MA_LAUNCH(24, 256); }
  else MA_LAUNCH(64, 256);
#undef MA_LAUNCH
  return 0;
}

int launch_rff_ascent(ppbo_ctx* ctx, const double* W_rows, int F, int D, const double* b, const double* omega, double amp,
                      const double* starts, const int* count, int K, int iters, double tol, double* x_out,
                      double* v_out, hipStream_t s) {
  double* W = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_TRANSPOSE, (size_t)F * D * sizeof(double));
  if (!W) return (int)hipErrorOutOfMemory;
  transpose_rows_kernel<<<(F + 255) / 256, 256, 0, s>>>(W_rows, F, D, W);
#define RA_LAUNCH(DP, NT)                                                                                      \
  do {                                                                                                         \
    RffEval<DP, NT> ev{W, F, D, b, omega, amp, make_rff_poly(1.0)};                                            \
    bb_ascent_kernel<DP, RffEval<DP, NT>, NT><<<K, NT, 0, s>>>(ev, D, starts, count, iters, tol, x_out, v_out, nullptr, 0); \
  } while (0)
  const bool wide = F >= 1024;
  if (D <= 8) { if (wide) RA_LAUNCH(8, 1024); else RA_LAUNCH(8, 256); }
  else if (D <= 24) { if (wide) RA_LAUNCH(24, 512); else RA_LAUNCH(24, 256); }
  else RA_LAUNCH(64, 256);
#undef RA_LAUNCH
  return 0;
}

template <int KID>
void launch_mean_grad(const ppbo_model* m, const KernParams& p, const double* d_Xc, int M, double* d_mu,
                      double* d_grad, hipStream_t s) {
  if (KID == PPBO_KERNEL_CAMPHOR || m->D <= 8)
    mean_grad_kernel<KID, 8><<<M, 256, 0, s>>>(m->d_X, m->N, m->D, p, m->d_alpha, d_Xc, d_mu, d_grad);
  else if (m->D <= 24)
    mean_grad_kernel<KID, 24><<<M, 256, 0, s>>>(m->d_X, m->N, m->D, p, m->d_alpha, d_Xc, d_mu, d_grad);
  else
    mean_grad_kernel<KID, 64><<<M, 256, 0, s>>>(m->d_X, m->N, m->D, p, m->d_alpha, d_Xc, d_mu, d_grad);
}

}  // namespace

extern "C" int ppbo_mean_grad(ppbo_ctx* ctx, const ppbo_model* m, const double* d_Xc, int64_t M, double* d_mu,
                              double* d_grad, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, m != nullptr && m->d_X && m->d_alpha, "model X/alpha");
  PPBO_REQUIRE(ctx, m->N > 0 && m->D > 0 && m->D <= 64, "model sizes (D<=64)");
  PPBO_REQUIRE(ctx, m->kernel_id >= 0 && m->kernel_id <= 2, "kernel_id");
  PPBO_REQUIRE(ctx, m->kernel_id != PPBO_KERNEL_CAMPHOR || m->D == 6, "camphor kernel needs D == 6");
  PPBO_REQUIRE(ctx, d_Xc && d_mu && d_grad && M >= 0 && M < (1 << 30), "points / outputs");
  if (M == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const KernParams p = make_kern_params(m->kernel_id, m->theta);
  switch (m->kernel_id) {
    case PPBO_KERNEL_SE: launch_mean_grad<PPBO_KERNEL_SE>(m, p, d_Xc, (int)M, d_mu, d_grad, s); break;
    case PPBO_KERNEL_RQ: launch_mean_grad<PPBO_KERNEL_RQ>(m, p, d_Xc, (int)M, d_mu, d_grad, s); break;
    default: launch_mean_grad<PPBO_KERNEL_CAMPHOR>(m, p, d_Xc, (int)M, d_mu, d_grad, s); break;
  }
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}


extern "C" int ppbo_shift_points(ppbo_ctx* ctx, const double* d_in, int64_t M, int D, const double* h_shift,
                                 double* d_out, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_in && d_out && h_shift && M > 0 && D > 0 && D <= 64, "arguments (D <= 64)");
  hipStream_t s = (hipStream_t)stream;
  double* dsh = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_SEARCH_SMALL, 64 * sizeof(double));
  if (!dsh) return (int)hipErrorOutOfMemory;
  PPBO_HIP_CHECK(ctx, hipMemcpyAsync(dsh, h_shift, (size_t)D * sizeof(double), hipMemcpyHostToDevice, s));
  const int64_t n = M * D;
  shift_points_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(d_in, n, D, dsh, d_out);
  PPBO_LAUNCH_CHECK(ctx);
  // h_shift is pageable host memory: the copy above is only asynchronous with respect to the DEVICE, the host
  // buffer has been consumed when hipMemcpyAsync returns
  return 0;
}

extern "C" int ppbo_mean_ascent(ppbo_ctx* ctx, const ppbo_model* m, const double* d_starts, int K, int iters,
                                double tol, double* d_x, double* d_mu, int* d_iters, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, m != nullptr && m->d_X && m->d_alpha, "model X/alpha");
  PPBO_REQUIRE(ctx, m->N > 0 && m->D > 0 && m->D <= 64, "model sizes (D<=64)");
  PPBO_REQUIRE(ctx, m->kernel_id >= 0 && m->kernel_id <= 2, "kernel_id");
  PPBO_REQUIRE(ctx, m->kernel_id != PPBO_KERNEL_CAMPHOR || m->D == 6, "camphor kernel needs D == 6");
  PPBO_REQUIRE(ctx, d_starts && d_x && d_mu && K > 0 && K <= 65536 && iters >= 0 && tol >= 0, "starts / outputs");
  hipStream_t s = (hipStream_t)stream;
  const KernParams p = make_kern_params(m->kernel_id, m->theta);
  int rc = 0;
  switch (m->kernel_id) {
    case PPBO_KERNEL_SE: rc = launch_mean_ascent<PPBO_KERNEL_SE>(ctx, m, p, d_starts, nullptr, K, iters, tol, d_x, d_mu, d_iters, s); break;
    case PPBO_KERNEL_RQ: rc = launch_mean_ascent<PPBO_KERNEL_RQ>(ctx, m, p, d_starts, nullptr, K, iters, tol, d_x, d_mu, d_iters, s); break;
    default: rc = launch_mean_ascent<PPBO_KERNEL_CAMPHOR>(ctx, m, p, d_starts, nullptr, K, iters, tol, d_x, d_mu, d_iters, s); break;
  }
  if (rc) return rc;
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

extern "C" int ppbo_mean_search(ppbo_ctx* ctx, const ppbo_model* m, const double* d_cand, int64_t M, int K, double sep,
                                int iters, double tol, double* d_x, double* d_mu, int* h_found, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, m != nullptr && m->d_X && m->d_alpha, "model X/alpha");
  PPBO_REQUIRE(ctx, m->N > 0 && m->D > 0 && m->D <= 64, "model sizes (D<=64)");
  PPBO_REQUIRE(ctx, m->kernel_id >= 0 && m->kernel_id <= 2, "kernel_id");
  PPBO_REQUIRE(ctx, m->kernel_id != PPBO_KERNEL_CAMPHOR || m->D == 6, "camphor kernel needs D == 6");
  PPBO_REQUIRE(ctx, d_cand && d_x && d_mu && M > 0 && M < ((int64_t)1 << 31), "candidates / outputs");
  PPBO_REQUIRE(ctx, K > 0 && K <= 1024 && sep >= 0 && iters >= 0 && tol >= 0, "K (<= 1024) / sep / iters / tol");
  hipStream_t s = (hipStream_t)stream;
  const int D = m->D;
  const int T_MAX = select_capacity(D);
  const int G = (int)((M + T_MAX - 1) / T_MAX);
  const int T = (int)((M + G - 1) / G);
  // workspace: mu[M] | gval[T] | starts[K*D] | gidx[T] (int) | count (int)
  const size_t nd = (size_t)M + T + (size_t)K * D;
  double* mu = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_SEARCH, nd * sizeof(double) + ((size_t)T + 16) * sizeof(int));
  if (!mu) return (int)hipErrorOutOfMemory;
  double* gval = mu + M;
  double* starts = gval + T;
  int* gidx = (int*)(starts + (size_t)K * D);
  int* count = gidx + T;
  ppbo_model mean_only = *m;
  mean_only.d_G = nullptr;
  if (int rc = ppbo_predict(ctx, &mean_only, d_cand, M, PPBO_SCORE_MEAN, 0.0, mu, nullptr, nullptr, nullptr, nullptr, stream))
    return rc;
  group_max_kernel<<<(T + 255) / 256, 256, 0, s>>>(mu, M, G, T, gval, gidx);
  {
    const size_t sel_lds = (size_t)T * (1 + D) * sizeof(double);
    if (sel_lds > 64 * 1024) ppbo_lds_limit(ctx, (const void*)select_starts_kernel, 150 * 1024);
    select_starts_kernel<<<1, 1024, sel_lds, s>>>(gval, gidx, T, d_cand, D, K, sep * sep, starts, count, TrialCands{});
  }
  const KernParams p = make_kern_params(m->kernel_id, m->theta);
  switch (m->kernel_id) {
    case PPBO_KERNEL_SE: if (int rc = launch_mean_ascent<PPBO_KERNEL_SE>(ctx, m, p, starts, count, K, iters, tol, d_x, d_mu, nullptr, s)) return rc; break;
    case PPBO_KERNEL_RQ: if (int rc = launch_mean_ascent<PPBO_KERNEL_RQ>(ctx, m, p, starts, count, K, iters, tol, d_x, d_mu, nullptr, s)) return rc; break;
    default: if (int rc = launch_mean_ascent<PPBO_KERNEL_CAMPHOR>(ctx, m, p, starts, count, K, iters, tol, d_x, d_mu, nullptr, s)) return rc; break;
  }
  PPBO_LAUNCH_CHECK(ctx);
  if (h_found) {
    PPBO_HIP_CHECK(ctx, hipMemcpyAsync(h_found, count, sizeof(int), hipMemcpyDeviceToHost, s));
    PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
  }
  return 0;
}


template <int KID>
void launch_screen(const ppbo_model* m, const KernParams& p, const TrialCands& tc, int nb, int rows_per_split, int n_split,
                   double* part, int extra_trial, hipStream_t s) {
  const long long Mt = tc.M + tc.E;
  const dim3 grid((unsigned)((Mt + SCR_T * SCR_CPT - 1) / (SCR_T * SCR_CPT)), n_split, nb);
#define SCR_LAUNCH(DP) mean_screen_kernel<KID, DP><<<grid, SCR_T, 0, s>>>(m->d_X, m->N, m->D, p, m->d_alpha, tc, rows_per_split, n_split, part, extra_trial)
  if (KID == PPBO_KERNEL_CAMPHOR) SCR_LAUNCH(6);
  else if (m->D <= 4) SCR_LAUNCH(4);
  else if (m->D <= 8) SCR_LAUNCH(8);
  else if (m->D <= 12) SCR_LAUNCH(12);
  else if (m->D <= 16) SCR_LAUNCH(16);
  else if (m->D <= 20) SCR_LAUNCH(20);
  else if (m->D <= 24) SCR_LAUNCH(24);
  else if (m->D <= 32) SCR_LAUNCH(32);
  else if (m->D <= 48) SCR_LAUNCH(48);
  else SCR_LAUNCH(64);
#undef SCR_LAUNCH
}

extern "C" int ppbo_mean_search_multi(ppbo_ctx* ctx, const ppbo_model* m, const double* d_pool, int64_t M,
                                      const double* h_shifts, int T, const double* d_extra, int E_rows,
                                      const double* h_xprev, int K, double sep, int iters, double tol, int screen_fp32,
                                      double* d_x, double* d_mu, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, m != nullptr && m->d_X && m->d_alpha, "model X/alpha");
  PPBO_REQUIRE(ctx, m->N > 0 && m->D > 0 && m->D <= 64, "model sizes (D<=64)");
  PPBO_REQUIRE(ctx, m->kernel_id >= 0 && m->kernel_id <= 2, "kernel_id");
  PPBO_REQUIRE(ctx, m->kernel_id != PPBO_KERNEL_CAMPHOR || m->D == 6, "camphor kernel needs D == 6");
  PPBO_REQUIRE(ctx, d_pool && h_shifts && d_x && d_mu && M > 0 && E_rows >= 0 && M + E_rows + 1 < ((int64_t)1 << 31),
               "pool / shifts / extra points / outputs");
  if (E_rows > 0 && !d_extra) {       // NULL with a row count: the model's own design points
    PPBO_REQUIRE(ctx, E_rows == m->N, "d_extra = NULL stands for the model's N design points: E_rows must be N");
    d_extra = m->d_X;
  }
  const int E = E_rows + (h_xprev ? 1 : 0);
  PPBO_REQUIRE(ctx, T >= 1 && T <= 64 && K > 0 && K <= 1024 && sep >= 0 && iters >= 0 && tol >= 0, "T (<= 64) / K (<= 1024) / sep / iters / tol");
  hipStream_t s = (hipStream_t)stream;
  const int D = m->D;
  const long long Mt = M + E;
  const int T_MAX = select_capacity(D);
  const int G = (int)((Mt + T_MAX - 1) / T_MAX);
  const int Tg = (int)((Mt + G - 1) / G);
  // workspace: shifts[T][D] + xprev[D] | mu[T][Mt] | gval[T][Tg] | starts[T][K][D] | gidx[T][Tg] (int) | counts[T] (int)
  const size_t nd = (size_t)(T + 1) * D + (size_t)T * Mt + (size_t)T * Tg + (size_t)T * K * D;
  double* base = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_SEARCH, nd * sizeof(double) + ((size_t)T * Tg + T + 16) * sizeof(int));
  if (!base) return (int)hipErrorOutOfMemory;
  double* shifts = base;
  double* xprev = shifts + (size_t)T * D;
  double* mu = xprev + D;
  double* gval = mu + (size_t)T * Mt;
  double* starts = gval + (size_t)T * Tg;
  int* gidx = (int*)(starts + (size_t)T * K * D);
  int* counts = gidx + (size_t)T * Tg;
  // the shifts and the previous x* leave the host through pinned upload slots of the ctx (ppbo_upload_async): the call
  // has consumed h_shifts / h_xprev when it returns and never blocks on the stream
  if (int rc = ppbo_upload_async(ctx, shifts, h_shifts, (size_t)T * D * sizeof(double), s)) return rc;
  if (h_xprev)
    if (int rc = ppbo_upload_async(ctx, xprev, h_xprev, (size_t)D * sizeof(double), s)) return rc;
  TrialCands tc;
  tc.pool = d_pool; tc.M = M; tc.shifts = shifts; tc.extra = d_extra; tc.xprev = h_xprev ? xprev : nullptr;
  tc.E_rows = E_rows; tc.E = E; tc.D = D;
  const KernParams p = make_kern_params(m->kernel_id, m->theta);
  if (screen_fp32) {
    // the trials in batches of <= 8 (bounds the partial sums: 8 x n_split x Mt doubles)
    const int blocks_x = (int)((Mt + SCR_T * SCR_CPT - 1) / (SCR_T * SCR_CPT));
    for (int t0 = 0; t0 < T; t0 += 8) {
      const int nb = (T - t0 < 8) ? (T - t0) : 8;
      int n_split = (2048 + blocks_x * nb - 1) / (blocks_x * nb);
      if (n_split > 16) n_split = 16;
      if (n_split > (m->N + SCR_RJ - 1) / SCR_RJ) n_split = (m->N + SCR_RJ - 1) / SCR_RJ;
      if (n_split < 1) n_split = 1;
      const int rows_per_split = (m->N + n_split - 1) / n_split;
      n_split = (m->N + rows_per_split - 1) / rows_per_split;
      double* part = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_PART, (size_t)nb * n_split * Mt * sizeof(double));
      if (!part) return (int)hipErrorOutOfMemory;
      TrialCands tb = tc;
      tb.shifts = shifts + (size_t)t0 * D;
      const int extra_trial = (t0 == 0 && E > 0) ? 0 : -1;     // only the job's trial 0 sees the extra points
      switch (m->kernel_id) {
        case PPBO_KERNEL_SE: launch_screen<PPBO_KERNEL_SE>(m, p, tb, nb, rows_per_split, n_split, part, extra_trial, s); break;
        case PPBO_KERNEL_RQ: launch_screen<PPBO_KERNEL_RQ>(m, p, tb, nb, rows_per_split, n_split, part, extra_trial, s); break;
        default: launch_screen<PPBO_KERNEL_CAMPHOR>(m, p, tb, nb, rows_per_split, n_split, part, extra_trial, s); break;
      }
      screen_sum_kernel<<<dim3((unsigned)((Mt + 255) / 256), nb), 256, 0, s>>>(part, n_split, Mt, M, extra_trial, mu + (size_t)t0 * Mt);
      PPBO_LAUNCH_CHECK(ctx);
    }
  } else {
    // fp64 screening: every trial's candidates written out as rows and scored by ppbo_predict (mean only), exactly
    // what the one-trial entry does with the rows ppbo_shift_points leaves
    double* rows = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_SEARCH_ROWS, (size_t)Mt * D * sizeof(double));
    if (!rows) return (int)hipErrorOutOfMemory;
    ppbo_model mean_only = *m;
    mean_only.d_G = nullptr;
    for (int t = 0; t < T; ++t) {
      const long long nt = M + (t == 0 ? E : 0);
      trial_rows_kernel<<<(unsigned)((nt * D + 255) / 256), 256, 0, s>>>(tc, t, rows);
      PPBO_LAUNCH_CHECK(ctx);
      if (int rc = ppbo_predict(ctx, &mean_only, rows, nt, PPBO_SCORE_MEAN, 0.0, mu + (size_t)t * Mt, nullptr, nullptr, nullptr,
                                nullptr, stream))
        return rc;
      if (nt < Mt) {
        // absent slots: -inf (0xFFF0000000000000 is not a byte pattern: a tiny fill kernel)
        fill_kernel<<<(unsigned)((Mt - nt + 255) / 256), 256, 0, s>>>(mu + (size_t)t * Mt + nt, Mt - nt, -INFINITY);
      }
    }
  }
  group_max_kernel<<<dim3((Tg + 255) / 256, T), 256, 0, s>>>(mu, Mt, G, Tg, gval, gidx);
  {
    const size_t sel_lds = (size_t)Tg * (1 + D) * sizeof(double);
    if (sel_lds > 64 * 1024) ppbo_lds_limit(ctx, (const void*)select_starts_kernel, 150 * 1024);
    select_starts_kernel<<<T, 1024, sel_lds, s>>>(gval, gidx, Tg, nullptr, D, K, sep * sep, starts, counts, tc);
  }
  PPBO_LAUNCH_CHECK(ctx);
  switch (m->kernel_id) {
    case PPBO_KERNEL_SE: if (int rc = launch_mean_ascent<PPBO_KERNEL_SE>(ctx, m, p, starts, counts, T * K, iters, tol, d_x, d_mu, nullptr, s, K)) return rc; break;
    case PPBO_KERNEL_RQ: if (int rc = launch_mean_ascent<PPBO_KERNEL_RQ>(ctx, m, p, starts, counts, T * K, iters, tol, d_x, d_mu, nullptr, s, K)) return rc; break;
    default: if (int rc = launch_mean_ascent<PPBO_KERNEL_CAMPHOR>(ctx, m, p, starts, counts, T * K, iters, tol, d_x, d_mu, nullptr, s, K)) return rc; break;
  }
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

extern "C" int ppbo_rff_search(ppbo_ctx* ctx, const double* d_cand, int64_t M, int D, const double* d_W, int F,
                               const double* d_b, double sigma_f, const double* d_omega, int K, double sep, int iters,
                               double tol, double* d_x, double* d_val, int* h_found, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_cand && d_W && d_b && d_omega && d_x && d_val, "null pointer");
  PPBO_REQUIRE(ctx, M > 0 && M < ((int64_t)1 << 31) && D > 0 && D <= 64 && F > 0, "sizes (D <= 64)");
  PPBO_REQUIRE(ctx, K > 0 && K <= 1024 && sep >= 0 && iters >= 0 && tol >= 0, "K (<= 1024) / sep / iters / tol");
  hipStream_t s = (hipStream_t)stream;
  const int T_MAX = select_capacity(D);
  const int G = (int)((M + T_MAX - 1) / T_MAX);
  const int T = (int)((M + G - 1) / G);
  const size_t nd = (size_t)M + T + (size_t)K * D;
  double* sc = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_SEARCH, nd * sizeof(double) + ((size_t)T + 16) * sizeof(int));
  if (!sc) return (int)hipErrorOutOfMemory;
  double* gval = sc + M;
  double* starts = gval + T;
  int* gidx = (int*)(starts + (size_t)K * D);
  int* count = gidx + T;
  if (int rc = ppbo_rff_score(ctx, d_cand, M, D, d_W, F, d_b, sigma_f, d_omega, sc, nullptr, nullptr, stream)) return rc;
  group_max_kernel<<<(T + 255) / 256, 256, 0, s>>>(sc, M, G, T, gval, gidx);
  {
    const size_t sel_lds = (size_t)T * (1 + D) * sizeof(double);
    if (sel_lds > 64 * 1024) ppbo_lds_limit(ctx, (const void*)select_starts_kernel, 150 * 1024);
    select_starts_kernel<<<1, 1024, sel_lds, s>>>(gval, gidx, T, d_cand, D, K, sep * sep, starts, count, TrialCands{});
  }
  if (int rc = launch_rff_ascent(ctx, d_W, F, D, d_b, d_omega, std::sqrt(2.0 * sigma_f * sigma_f / (double)F), starts, count, K,
                                 iters, tol, d_x, d_val, s))
    return rc;
  PPBO_LAUNCH_CHECK(ctx);
  if (h_found) {
    PPBO_HIP_CHECK(ctx, hipMemcpyAsync(h_found, count, sizeof(int), hipMemcpyDeviceToHost, s));
    PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
  }
  return 0;
}
