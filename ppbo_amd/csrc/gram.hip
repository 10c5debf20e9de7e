// K1 / K2: Gram matrix (symmetric, shrinkage fused) and raw cross-covariance.
//   reference: kernels.py:3-53 (dist, SE/RQ/camphor), gp_model.py:147-155,
//              misc.py:71-88 (regularize_covariance == closed-form shrink).
// Layout: X[N,D] row-major fp64 in HBM.  One 256-thread workgroup produces one 64x64 output tile;
// gram only visits tiles on/above the diagonal and writes the mirror image itself, every HBM store
// completing whole 128-byte lines.  SE/RQ tiles run on the matrix cores (gram_mfma_kernel), the
// camphor kernel on the vector ALUs (gram_kernel).  HBM-write bound: algorithmic bytes = 8 N^2 + 8 N D;
// the write-only floor of the chip at these sizes is measured by tools/store_floor.hip.
#include "common.h"

namespace {

constexpr int TS = 64;   // tile side
constexpr int TP = TS + 2;  // padded LDS row (transpose staging)

// staged "dimensions" per row: the camphor kernel is evaluated in feature form (below), 12 values per row
template <int KID>
__device__ __forceinline__ int staged_dims(int D) { return KID == PPBO_KERNEL_CAMPHOR ? 12 : D; }

template <int KID>
__device__ __forceinline__ void tile_eval(const double* __restrict__ XaT, const double* __restrict__ XbT,
                                          int D, int ty, int tx, const KernParams& p, double v[4][4]) {
  double s[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) s[a][b] = 0.0;
  if (KID == PPBO_KERNEL_CAMPHOR) {
    // sin^2(pi (a - b)) = (1 - cos 2pi a cos 2pi b - sin 2pi a sin 2pi b) / 2: kernels.py:36-53's exponent
    //   c0 sum_k sin^2(pi |a_k - b_k|) + c1 (a_2 - b_2)^2 = (c0 / 2) (5 - phi(a).phi(b)) + c1 (a_2 - b_2)^2
    // with phi = (cos 2pi x_k, sin 2pi x_k) over the periodic coordinates, staged once per row (stage_panel): ten FMAs per
    // pair instead of five sinpi evaluations.  The dot product is the same chain of the same products for (a, b) and
    // (b, a): the matrix stays bitwise symmetric.
    for (int f = 0; f < 10; ++f) {
      double xa[4], xb[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) xa[a] = XaT[f * TS + ty * 4 + a];
#pragma unroll
      for (int b = 0; b < 4; ++b) xb[b] = XbT[f * TS + tx * 4 + b];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) s[a][b] = fma(xa[a], xb[b], s[a][b]);
    }
    const double h = 0.5 * p.c0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const double dd = XaT[10 * TS + ty * 4 + a] - XbT[10 * TS + tx * 4 + b];
        v[a][b] = kern_finish<KID>(fma(p.c1 * dd, dd, h * (5.0 - s[a][b])), p);
      }
    return;
  }
  for (int d = 0; d < D; ++d) {
    double xa[4], xb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) xa[a] = XaT[d * TS + ty * 4 + a];
#pragma unroll
    for (int b = 0; b < 4; ++b) xb[b] = XbT[d * TS + tx * 4 + b];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) s[a][b] += kern_term<KID>(xa[a] - xb[b], d, p);
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) v[a][b] = kern_finish<KID>(s[a][b], p);
}

template <int KID>
__device__ __forceinline__ void stage_panel(const double* __restrict__ X, int n, int D, int r0, double* __restrict__ dstT) {
  if (KID == PPBO_KERNEL_CAMPHOR) {
    // dstT[k][r], dstT[5 + k][r] = cos, sin (2 pi x_d) for the periodic coordinates d = 0, 1, 3, 4, 5; dstT[10][r] = x_2
    for (int e = threadIdx.x; e < TS * 6; e += blockDim.x) {
      const int r = e / 6, k = e - r * 6;
      const int gr = r0 + r;
      if (k < 5) {
        const double x = (gr < n) ? X[(size_t)gr * D + (k < 2 ? k : k + 1)] : 0.0;
        double sn, cs;
        sincospi(2.0 * x, &sn, &cs);
        dstT[k * TS + r] = cs;
        dstT[(5 + k) * TS + r] = sn;
      } else {
        dstT[10 * TS + r] = (gr < n) ? X[(size_t)gr * D + 2] : 0.0;
        dstT[11 * TS + r] = 0.0;
      }
    }
    return;
  }
  // dstT[d][r] = X[r0 + r][d], zero beyond n
  for (int e = threadIdx.x; e < TS * D; e += blockDim.x) {
    const int r = e / D, d = e - r * D;
    const int gr = r0 + r;
    dstT[d * TS + r] = (gr < n) ? X[(size_t)gr * D + d] : 0.0;
  }
}

template <int KID>
__global__ __launch_bounds__(256) void gram_kernel(const double* __restrict__ X, int N, int D, KernParams p,
                                                    double shrink, double* __restrict__ Sigma, int nt) {
  extern __shared__ double smem[];
  double* XaT = smem;                // [D][64]
  const int Ds = staged_dims<KID>(D);
  double* XbT = smem + (size_t)Ds * TS;  // [Ds][64]
  double* Tt = XbT + (size_t)Ds * TS;    // [64][TP] transpose staging

  // linear block id -> (bi <= bj) over the upper triangle, row by row
  const int t = blockIdx.x;
  const double q = 2.0 * nt + 1.0;
  int bi = (int)floor((q - sqrt(q * q - 8.0 * (double)t)) * 0.5);
  // guard against rounding at row boundaries
  while (bi > 0 && t < bi * nt - bi * (bi - 1) / 2) --bi;
  while (t >= (bi + 1) * nt - (bi + 1) * bi / 2) ++bi;
  const int bj = bi + (t - (bi * nt - bi * (bi - 1) / 2));

  const int i0 = bi * TS, j0 = bj * TS;
  stage_panel<KID>(X, N, D, i0, XaT);
  stage_panel<KID>(X, N, D, j0, XbT);
  __syncthreads();

  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double v[4][4];
  tile_eval<KID>(XaT, XbT, D, ty, tx, p, v);

  const double one_minus = 1.0 - shrink;
  const double diagv = one_minus * p.sf2 + shrink * p.sf2;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int gi = i0 + ty * 4 + a;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int gj = j0 + tx * 4 + b;
      v[a][b] = (gi == gj) ? diagv : one_minus * v[a][b];
    }
  }
  const bool vec_ok = ((N & 1) == 0);
  // direct tile (rows i, cols j)
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int gi = i0 + ty * 4 + a;
    if (gi >= N) continue;
    const int gj = j0 + tx * 4;
    double* dst = Sigma + (size_t)gi * N + gj;
    if (vec_ok && gj + 3 < N) {
      store_through2(dst, v[a][0], v[a][1]);
      store_through2(dst + 2, v[a][2], v[a][3]);
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (gj + b < N) dst[b] = v[a][b];
    }
  }
  if (bi == bj) return;
  // mirrored tile through LDS: Tt[col][row]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) Tt[(tx * 4 + b) * TP + ty * 4 + a] = v[a][b];
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int r = ty * 4 + a;       // row of the mirrored tile = column index j
    const int gi = j0 + r;
    if (gi >= N) continue;
    const int gj = i0 + tx * 4;
    const double* src = Tt + r * TP + tx * 4;
    double* dst = Sigma + (size_t)gi * N + gj;
    if (vec_ok && gj + 3 < N) {
      store_through2(dst, src[0], src[1]);
      store_through2(dst + 2, src[2], src[3]);
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (gj + b < N) dst[b] = src[b];
    }
  }
}

template <int KID>
__device__ __forceinline__ double gram_finish(double r2, const KernParams& p, double scale) {
  if (KID == PPBO_KERNEL_SE) return scale * exp_nonpos(-p.c0 * r2);
  const double t = 1.0 + r2 * p.c0;
  return scale / (t * t);
}

// SE / RQ Gram tile with the pairwise dot products on the fp64 matrix cores and the reference's
// expansion  r_ij^2 = (|x_i|^2 + |x_j|^2) - 2 x_i.x_j  (kernels.py:7-10), clipped at 0.  The MFMA chain
// multiplies x_i by (-2 x_j): products and accumulation order are the same for (i,j) and (j,i), and
// the norm sum commutes, so the matrix is bitwise symmetric by construction.
// DP = D padded to a multiple of 4 (compile time).  One 256-thread workgroup owns one 64x64 tile on or
// above the diagonal; wave w owns its rows 16w..16w+15.  After the single barrier that publishes the
// two operand panels the waves are independent: per 16x16 sub-tile a wave runs DP/4 MFMAs, finishes
// its four entries per lane, stores them straight from registers (16 lanes = one 128-byte line), and
// -- off the diagonal -- transposes the sub-tile through a private 2 KB LDS strip and stores the
// mirror image as 16-byte pieces (8 lanes = one line).  Stores therefore start after the first
// quarter of the wave's arithmetic and overlap the rest.  HBM-write bound: 8 N^2 + 8 N D bytes.
template <int KID, int DP>
__global__ __launch_bounds__(512) void gram_mfma_kernel(const double* __restrict__ X, int N, int D, KernParams p,
                                                         double shrink, double* __restrict__ Sigma, int nt,
                                                         int order) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int LD = DP + 2;       // 2*odd: fragment reads and panel writes are bank-conflict free
  constexpr int LT = 18;           // transpose strip row pitch (conflict-free b64 writes, 16-byte aligned rows)
  double* Xa = smem;               // [64][LD]
  double* Xb = Xa + TS * LD;       // [64][LD]  (-2 x_j)
  double* na = Xb + TS * LD;       // [64] |x_i|^2
  double* nb = na + TS;            // [64] |x_j|^2
  double* Tw = nb + TS;            // [8 waves][16][LT]

  // off-diagonal tiles first (row by row over the strict upper triangle), the nt diagonal tiles last: the
  // diagonal tiles have no mirror image to write, and the last-dispatched workgroups are the ones that land as
  // a third tile on an already busy CU (528 tiles on 256 CUs at N = 2048)
  const int n1 = nt - 1, n_off = n1 * (n1 + 1) / 2;
  int bi, bj;
  if ((int)blockIdx.x >= n_off) {
    bi = bj = blockIdx.x - n_off;
  } else {
    const int t = blockIdx.x;
    const double q = 2.0 * n1 + 1.0;
    bi = (int)floor((q - sqrt(q * q - 8.0 * (double)t)) * 0.5);
    while (bi > 0 && t < bi * n1 - bi * (bi - 1) / 2) --bi;
    while (t >= (bi + 1) * n1 - (bi + 1) * bi / 2) ++bi;
    bj = bi + (t - (bi * n1 - bi * (bi - 1) / 2)) + 1;
    if (order == 1) {
      // diagonal-major: the same decomposition read as (distance k = bi + 1 from the diagonal, position r along
      // it).  Row-major order makes co-resident workgroups share ONE 512-byte column window for their mirror
      // stores (same bi, consecutive bj): with a power-of-two row pitch those all fall on the same memory
      // channels.  Along a diagonal both the direct and the mirror windows move with every workgroup.
      const int k = bi + 1, r = bj - bi - 1;
      bi = r;
      bj = r + k;
    }
  }
  const int i0 = bi * TS, j0 = bj * TS;

  // 4 lanes per row, DP/4 elements each; threads 0-255 stage the row panel, 256-511 the column panel
  // (scaled by -2); squared norms by two xor-shuffles
  {
    constexpr int Q = DP / 4;
    const int half = threadIdx.x >> 8, r = (threadIdx.x & 255) >> 2, part = threadIdx.x & 3;
    const int g0 = half ? j0 : i0;
    double* Xp = half ? Xb : Xa;
    const double sc = half ? -2.0 : 1.0;
    double xv[Q];
    double sn = 0.0;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      const int d = part * Q + k;
      xv[k] = (d < D && g0 + r < N) ? X[(size_t)(g0 + r) * D + d] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      sn += xv[k] * xv[k];
      Xp[r * LD + part * Q + k] = sc * xv[k];
    }
    sn += __shfl_xor(sn, 1, 64); sn += __shfl_xor(sn, 2, 64);
    if (part == 0) (half ? nb : na)[r] = sn;
  }
  __syncthreads();
  // wave = (row strip w of 16 rows, column half jh of 32 columns): eight wavefronts keep four per SIMD in
  // flight at two tiles per CU, and a wave's first stores leave after half as much arithmetic
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, w = wv & 3, jh = wv >> 2;
  const int lr = lane & 15, lk = lane >> 4;
  double af[DP / 4], nai[4];
#pragma unroll
  for (int kk = 0; kk < DP / 4; ++kk) af[kk] = Xa[(w * 16 + lr) * LD + kk * 4 + lk];
#pragma unroll
  for (int r = 0; r < 4; ++r) nai[r] = na[w * 16 + lk + 4 * r];

  const double one_minus = 1.0 - shrink;
  const double scale = one_minus * p.sf2;
  const double diagv = one_minus * p.sf2 + shrink * p.sf2;
  const bool vec_ok = ((N & 1) == 0);
  const bool full = (i0 + TS <= N) && (j0 + TS <= N);
  double* Ts = Tw + wv * 16 * LT;
  // direct piece: rows lk+4r of the wave's strip, column lr of sub-tile j
  double* ddst = Sigma + (size_t)(i0 + w * 16 + lk) * N + j0 + lr;
  // mirror piece: lane = (row c = lane>>3 (+8), column pair 2*(lane&7)) of the transposed sub-tile
  const int mc = lane >> 3, mp = (lane & 7) * 2;
  double* mdst = Sigma + (size_t)(j0 + mc) * N + i0 + w * 16 + mp;

#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int j = 2 * jh + jj;
    double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < DP / 4; ++kk)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], Xb[(j * 16 + lr) * LD + kk * 4 + lk], acc, 0, 0, 0);
    const double nbj = nb[j * 16 + lr];
    double v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = gram_finish<KID>(fmax(acc[r] + (nai[r] + nbj), 0.0), p, scale);
    if (bi == bj && j == w) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (lk + 4 * r == lr) v[r] = diagv;
    }
    if (full) {
#pragma unroll
      for (int r = 0; r < 4; ++r) store_through(ddst + (size_t)(4 * r) * N + j * 16, v[r]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (i0 + w * 16 + lk + 4 * r < N && j0 + j * 16 + lr < N) ddst[(size_t)(4 * r) * N + j * 16] = v[r];
    }
    if (bi == bj) continue;
    // transpose through the wave's private strip; LDS traffic of one wave is ordered, the fences
    // only keep the compiler from moving the accesses across each other
#pragma unroll
    for (int r = 0; r < 4; ++r) Ts[lr * LT + lk + 4 * r] = v[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = mc + 8 * h;
      const double2 m = *reinterpret_cast<const double2*>(Ts + c * LT + mp);
      double* dst = mdst + (size_t)(j * 16 + 8 * h) * N;
      const int gi = j0 + j * 16 + c, gj = i0 + w * 16 + mp;
      if (full && vec_ok) store_through2(dst, m.x, m.y);
      else if (gi < N) {
        if (gj < N) dst[0] = m.x;
        if (gj + 1 < N) dst[1] = m.y;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <int KID>
__global__ __launch_bounds__(256) void crosscov_kernel(const double* __restrict__ X1, int n1,
                                                        const double* __restrict__ X2, int n2, int D,
                                                        KernParams p, double* __restrict__ K, int ldk) {
  extern __shared__ double smem[];
  double* XaT = smem;
  double* XbT = smem + (size_t)staged_dims<KID>(D) * TS;
  const int i0 = blockIdx.y * TS, j0 = blockIdx.x * TS;
  stage_panel<KID>(X1, n1, D, i0, XaT);
  stage_panel<KID>(X2, n2, D, j0, XbT);
  __syncthreads();
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double v[4][4];
  tile_eval<KID>(XaT, XbT, D, ty, tx, p, v);
  const bool vec_ok = ((ldk & 1) == 0) && ((reinterpret_cast<uintptr_t>(K) & 15) == 0);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int gi = i0 + ty * 4 + a;
    if (gi >= n1) continue;
    const int gj = j0 + tx * 4;
    double* dst = K + (size_t)gi * ldk + gj;
    if (vec_ok && gj + 3 < n2) {
      *reinterpret_cast<double2*>(dst) = make_double2(v[a][0], v[a][1]);
      *reinterpret_cast<double2*>(dst + 2) = make_double2(v[a][2], v[a][3]);
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (gj + b < n2) dst[b] = v[a][b];
    }
  }
}


// Measurement probe (ppbo_store_floor): a WRITE-ONLY pass over an N x N fp64 matrix, 32 x 128 tiles, 16-byte
// write-through stores -- the shape tools/store_floor.hip found fastest at HBM-sized N.  It is the ceiling any
// Gram kernel can reach at that N on this chip; bench.py launches it next to ppbo_gram instead of quoting constants.
__global__ __launch_bounds__(256) void store_floor_kernel(double* __restrict__ S, int N, double v) {
  constexpr int TR = 32, TC = 128, LPR = TC / 2, RPP = 256 / LPR;
  const int ntc = (N + TC - 1) / TC;
  const int bi = blockIdx.x / ntc, bj = blockIdx.x % ntc;
  const int c2 = (threadIdx.x % LPR) * 2, r0 = threadIdx.x / LPR;
#pragma unroll
  for (int a = 0; a < TR / RPP; ++a) {
    const int r = bi * TR + a * RPP + r0, c = bj * TC + c2;
    if (r < N && c + 1 < N) store_through2(S + (size_t)r * N + c, v, v + (double)r);
    else if (r < N && c < N) S[(size_t)r * N + c] = v;
  }
}


// a-3 as a stand-alone operator (misc.py:71-88 applied to a matrix that did not come out of ppbo_gram):
//   diagonal entries < 0 -> jitter; K <- (1 - s) K + s (tr K / n) I.  The SVD round trip of misc.py:79-80 is the
// identity (oracle/ppbo_oracle.py regularize_covariance, tests/test_oracle_golden.py) and is not executed.
// Pass 1 (one workgroup): clamp the diagonal in place and leave mu = tr(K)/n in mu_out.
__global__ __launch_bounds__(1024) void regcov_trace_kernel(double* __restrict__ K, int N, int ldk, int pos_diag,
                                                            double jitter, double* __restrict__ mu_out) {
  __shared__ double sh[16];
  double t = 0.0;
  for (int i = threadIdx.x; i < N; i += 1024) {
    double d = K[(size_t)i * ldk + i];
    if (pos_diag && d < 0.0) { d = jitter; K[(size_t)i * ldk + i] = d; }
    t += d;
  }
  t = wave_sum(t);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0;
    for (int w = 0; w < 16; ++w) a += sh[w];
    *mu_out = a / (double)N;
  }
}

// Pass 2: one thread per pair of columns (16-byte accesses when the row is aligned), rows on blockIdx.y.
__global__ __launch_bounds__(256) void regcov_shrink_kernel(double* __restrict__ K, int N, int ldk, double shrink,
                                                            const double* __restrict__ mu) {
  const int i = blockIdx.y;
  const double add = shrink * (*mu), om = 1.0 - shrink;
  double* row = K + (size_t)i * ldk;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < N; j += gridDim.x * 256) {
    const double v = om * row[j];
    row[j] = (j == i) ? v + add : v;
  }
}

}  // namespace

extern "C" {

int ppbo_gram(ppbo_ctx* ctx, int kernel_id, const double* d_X, int N, int D, const double h_theta[3],
              double shrink, double* d_Sigma, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_X && d_Sigma && h_theta, "null pointer");
  PPBO_REQUIRE(ctx, N > 0 && D > 0 && D <= 64, "N>0, 0<D<=64");
  PPBO_REQUIRE(ctx, kernel_id >= 0 && kernel_id <= 2, "kernel_id");
  PPBO_REQUIRE(ctx, kernel_id != PPBO_KERNEL_CAMPHOR || D == 6, "camphor kernel needs D == 6");
  const KernParams p = make_kern_params(kernel_id, h_theta);
  const int nt = (N + TS - 1) / TS;
  const int nblk = nt * (nt + 1) / 2;
  hipStream_t s = (hipStream_t)stream;
  ppbo_lds_limit(ctx, (const void*)gram_kernel<PPBO_KERNEL_CAMPHOR>, 112 * 1024);
  PpboProfScope pf(ctx, ppbo_ctx::PF_GRAM, s);
  // tile order (PPBO_GRAM_VARIANT: 0 row-major over the upper triangle, 1 diagonal-major, -1 = by size)
  const int order = ctx->gram_variant >= 0 ? ctx->gram_variant : (N >= 4096 ? 1 : 0);
  if (kernel_id != PPBO_KERNEL_CAMPHOR) {
#define GM_LAUNCH(DPV)                                                                                       \
  do {                                                                                                       \
    constexpr int LDv = DPV + 2;                                                                             \
    constexpr int body = 2 * TS * LDv + 2 * TS + 8 * 16 * 18;                                                \
    const size_t lds = (size_t)body * sizeof(double);                                                        \
    if (kernel_id == PPBO_KERNEL_SE) {                                                                       \
      if (lds > 64 * 1024) ppbo_lds_limit(ctx, (const void*)gram_mfma_kernel<PPBO_KERNEL_SE, DPV>, (int)lds); \
      gram_mfma_kernel<PPBO_KERNEL_SE, DPV><<<nblk, 512, lds, s>>>(d_X, N, D, p, shrink, d_Sigma, nt, order);       \
    } else {                                                                                                 \
      if (lds > 64 * 1024) ppbo_lds_limit(ctx, (const void*)gram_mfma_kernel<PPBO_KERNEL_RQ, DPV>, (int)lds); \
      gram_mfma_kernel<PPBO_KERNEL_RQ, DPV><<<nblk, 512, lds, s>>>(d_X, N, D, p, shrink, d_Sigma, nt, order);       \
    }                                                                                                        \
  } while (0)
    if (D <= 4) GM_LAUNCH(4);
    else if (D <= 8) GM_LAUNCH(8);
    else if (D <= 12) GM_LAUNCH(12);
    else if (D <= 16) GM_LAUNCH(16);
    else if (D <= 20) GM_LAUNCH(20);
    else if (D <= 24) GM_LAUNCH(24);
    else if (D <= 32) GM_LAUNCH(32);
    else if (D <= 48) GM_LAUNCH(48);
    else GM_LAUNCH(64);
#undef GM_LAUNCH
  } else {
    const size_t lds = ((size_t)2 * 12 * TS + (size_t)TS * TP) * sizeof(double);     // 12 staged features per row
    gram_kernel<PPBO_KERNEL_CAMPHOR><<<nblk, 256, lds, s>>>(d_X, N, D, p, shrink, d_Sigma, nt);
  }
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

int ppbo_store_floor(ppbo_ctx* ctx, double* d_S, int N, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_S && N > 0 && (N % 2) == 0, "matrix (even N)");
  const int nt = ((N + 31) / 32) * ((N + 127) / 128);
  store_floor_kernel<<<nt, 256, 0, (hipStream_t)stream>>>(d_S, N, 1.0);
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

int ppbo_cross_cov(ppbo_ctx* ctx, int kernel_id, const double* d_X1, int n1, const double* d_X2, int n2,
                   int D, const double h_theta[3], double* d_K, int ldk, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_X1 && d_X2 && d_K && h_theta, "null pointer");
  PPBO_REQUIRE(ctx, n1 > 0 && n2 > 0 && D > 0 && D <= 64 && ldk >= n2, "sizes (D<=64)");
  PPBO_REQUIRE(ctx, kernel_id >= 0 && kernel_id <= 2, "kernel_id");
  PPBO_REQUIRE(ctx, kernel_id != PPBO_KERNEL_CAMPHOR || D == 6, "camphor kernel needs D == 6");
  const KernParams p = make_kern_params(kernel_id, h_theta);
  dim3 grid((n2 + TS - 1) / TS, (n1 + TS - 1) / TS);
  const size_t lds = (size_t)2 * (kernel_id == PPBO_KERNEL_CAMPHOR ? 12 : D) * TS * sizeof(double);
  hipStream_t s = (hipStream_t)stream;
  if (lds > 64 * 1024) {   // D up to 64 needs up to 64 KB + padding
    ppbo_lds_limit(ctx, (const void*)crosscov_kernel<PPBO_KERNEL_SE>, 112 * 1024);
    ppbo_lds_limit(ctx, (const void*)crosscov_kernel<PPBO_KERNEL_RQ>, 112 * 1024);
  }
  switch (kernel_id) {
    case PPBO_KERNEL_SE: crosscov_kernel<PPBO_KERNEL_SE><<<grid, 256, lds, s>>>(d_X1, n1, d_X2, n2, D, p, d_K, ldk); break;
    case PPBO_KERNEL_RQ: crosscov_kernel<PPBO_KERNEL_RQ><<<grid, 256, lds, s>>>(d_X1, n1, d_X2, n2, D, p, d_K, ldk); break;
    default: crosscov_kernel<PPBO_KERNEL_CAMPHOR><<<grid, 256, lds, s>>>(d_X1, n1, d_X2, n2, D, p, d_K, ldk); break;
  }
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

int ppbo_regularize_covariance(ppbo_ctx* ctx, double* d_K, int N, int ldk, double reg_level, int pos_diag, double jitter,
                               void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_K && N > 0 && ldk >= N, "matrix");
  PPBO_REQUIRE(ctx, reg_level >= 0.0 && reg_level <= 1.0, "reg_level must be in [0, 1] (sklearn shrunk_covariance)");
  hipStream_t s = (hipStream_t)stream;
  double* mu = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_VEC, sizeof(double));
  if (!mu) return (int)hipErrorOutOfMemory;
  regcov_trace_kernel<<<1, 1024, 0, s>>>(d_K, N, ldk, pos_diag, jitter, mu);
  const int bx = (N + 255) / 256 < 8 ? (N + 255) / 256 : 8;
  regcov_shrink_kernel<<<dim3(bx, N), 256, 0, s>>>(d_K, N, ldk, reg_level, mu);
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

}  // extern "C"
