// K1 / K2: Gram matrix (symmetric, shrinkage fused) and raw cross-covariance.
//   reference: kernels.py:3-53 (dist, SE/RQ/camphor), gp_model.py:147-155,
//              misc.py:71-88 (regularize_covariance == closed-form shrink).
// Layout: X[N,D] row-major fp64 in HBM.  One 256-thread workgroup produces one
// 64x64 output tile; the two 64-row operand panels are staged in LDS transposed
// ([D][64]) so a lane's four columns are one 32-byte LDS read.  gram only visits
// tiles on/above the diagonal and mirrors them through an LDS transpose, so
// every HBM store is a full 512-byte row segment.  HBM-write bound:
// algorithmic bytes = 8 N^2 + 8 N D.
#include "common.h"

namespace {

constexpr int TS = 64;   // tile side
constexpr int TP = TS + 2;  // padded LDS row (transpose staging)

template <int KID>
__device__ __forceinline__ void tile_eval(const double* __restrict__ XaT, const double* __restrict__ XbT,
                                          int D, int ty, int tx, const KernParams& p, double v[4][4]) {
  double s[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) s[a][b] = 0.0;
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

__device__ __forceinline__ void stage_panel(const double* __restrict__ X, int n, int D, int r0, double* __restrict__ dstT) {
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
  double* XbT = smem + (size_t)D * TS;  // [D][64]
  double* Tt = XbT + (size_t)D * TS;    // [64][TP] transpose staging

  // linear block id -> (bi <= bj) over the upper triangle, row by row
  const int t = blockIdx.x;
  const double q = 2.0 * nt + 1.0;
  int bi = (int)floor((q - sqrt(q * q - 8.0 * (double)t)) * 0.5);
  // guard against rounding at row boundaries
  while (bi > 0 && t < bi * nt - bi * (bi - 1) / 2) --bi;
  while (t >= (bi + 1) * nt - (bi + 1) * bi / 2) ++bi;
  const int bj = bi + (t - (bi * nt - bi * (bi - 1) / 2));

  const int i0 = bi * TS, j0 = bj * TS;
  stage_panel(X, N, D, i0, XaT);
  stage_panel(X, N, D, j0, XbT);
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
      *reinterpret_cast<double2*>(dst) = make_double2(v[a][0], v[a][1]);
      *reinterpret_cast<double2*>(dst + 2) = make_double2(v[a][2], v[a][3]);
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
      *reinterpret_cast<double2*>(dst) = make_double2(src[0], src[1]);
      *reinterpret_cast<double2*>(dst + 2) = make_double2(src[2], src[3]);
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (gj + b < N) dst[b] = src[b];
    }
  }
}

// SE / RQ Gram tile with the inner products on the fp64 matrix cores:
//   r^2 = |x_i|^2 + |x_j|^2 - 2 x_i.x_j  (the reference's own formula, kernels.py:7-10, clipped at 0),
// x_i.x_j by v_mfma_f64_16x16x4_f64 over the zero-padded dimension, so the VALU only does the
// norm combine + exp.  Wave w owns rows 16w..16w+15 of the 64x64 tile (4 MFMA column tiles).
// The finished tile is staged in LDS (aliasing the dead operand panels) so that both the tile
// and its mirror leave as 512-byte row segments.
template <int KID>
__global__ __launch_bounds__(256) void gram_mfma_kernel(const double* __restrict__ X, int N, int D, KernParams p,
                                                         double shrink, double* __restrict__ Sigma, int nt) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int Dp = (D + 3) & ~3, LD = Dp + 2;
  const int tile_doubles = TS * TP;
  const int pan_doubles = 2 * TS * LD;
  const int body = tile_doubles > pan_doubles ? tile_doubles : pan_doubles;
  double* Xa = smem;              // [64][LD]
  double* Xb = smem + TS * LD;    // [64][LD]
  double* Tt = smem;              // [64][TP] (aliases the panels once the MFMAs are done)
  double* na = smem + body;       // [64]
  double* nb = na + TS;           // [64]

  const int t = blockIdx.x;
  const double q = 2.0 * nt + 1.0;
  int bi = (int)floor((q - sqrt(q * q - 8.0 * (double)t)) * 0.5);
  while (bi > 0 && t < bi * nt - bi * (bi - 1) / 2) --bi;
  while (t >= (bi + 1) * nt - (bi + 1) * bi / 2) ++bi;
  const int bj = bi + (t - (bi * nt - bi * (bi - 1) / 2));
  const int i0 = bi * TS, j0 = bj * TS;

  // coalesced panel loads (64 rows of X are one contiguous 64*D block), zero padding
  for (int e = threadIdx.x; e < TS * Dp; e += 256) {
    const int r = e / Dp, d = e - r * Dp;
    Xa[r * LD + d] = (d < D && i0 + r < N) ? X[(size_t)(i0 + r) * D + d] : 0.0;
    Xb[r * LD + d] = (d < D && j0 + r < N) ? X[(size_t)(j0 + r) * D + d] : 0.0;
  }
  __syncthreads();
  if (threadIdx.x < 2 * TS) {
    const double* row = (threadIdx.x < TS ? Xa : Xb) + (threadIdx.x & (TS - 1)) * LD;
    double s = 0.0;
    for (int d = 0; d < Dp; ++d) s += row[d] * row[d];
    na[threadIdx.x] = s;          // na | nb are contiguous
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  double4_t acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = double4_t{0.0, 0.0, 0.0, 0.0};
  for (int kk = 0; kk < Dp; kk += 4) {
    const double a = Xa[(w * 16 + lr) * LD + kk + lk];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double b = Xb[(j * 16 + lr) * LD + kk + lk];
      acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    }
  }
  __syncthreads();   // panels are dead from here; norms are visible
  const double one_minus = 1.0 - shrink;
  const double diagv = one_minus * p.sf2 + shrink * p.sf2;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = j * 16 + lr;
    const double nj = nb[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = w * 16 + lk + 4 * r;
      double r2 = na[row] + nj - 2.0 * acc[j][r];
      r2 = r2 > 0.0 ? r2 : 0.0;
      const double k = kern_finish<KID>(r2, p);
      Tt[row * TP + col] = ((i0 + row) == (j0 + col)) ? diagv : one_minus * k;
    }
  }
  __syncthreads();
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const bool vec_ok = ((N & 1) == 0);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int r = ty * 4 + a;
    const int gi = i0 + r;
    if (gi >= N) continue;
    const int gj = j0 + tx * 4;
    const double* src = Tt + r * TP + tx * 4;
    double* dst = Sigma + (size_t)gi * N + gj;
    if (vec_ok && gj + 3 < N) {
      *reinterpret_cast<double2*>(dst) = *reinterpret_cast<const double2*>(src);
      *reinterpret_cast<double2*>(dst + 2) = *reinterpret_cast<const double2*>(src + 2);
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (gj + b < N) dst[b] = src[b];
    }
  }
  if (bi == bj) return;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int r = ty * 4 + a;       // row of the mirrored tile = column of Tt
    const int gi = j0 + r;
    if (gi >= N) continue;
    const int gj = i0 + tx * 4;
    double v[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) v[b] = Tt[(tx * 4 + b) * TP + r];
    double* dst = Sigma + (size_t)gi * N + gj;
    if (vec_ok && gj + 3 < N) {
      *reinterpret_cast<double2*>(dst) = make_double2(v[0], v[1]);
      *reinterpret_cast<double2*>(dst + 2) = make_double2(v[2], v[3]);
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (gj + b < N) dst[b] = v[b];
    }
  }
}

template <int KID>
__global__ __launch_bounds__(256) void crosscov_kernel(const double* __restrict__ X1, int n1,
                                                        const double* __restrict__ X2, int n2, int D,
                                                        KernParams p, double* __restrict__ K, int ldk) {
  extern __shared__ double smem[];
  double* XaT = smem;
  double* XbT = smem + (size_t)D * TS;
  const int i0 = blockIdx.y * TS, j0 = blockIdx.x * TS;
  stage_panel(X1, n1, D, i0, XaT);
  stage_panel(X2, n2, D, j0, XbT);
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

}  // namespace

extern "C" {

int ppbo_gram(ppbo_ctx* ctx, int kernel_id, const double* d_X, int N, int D, const double h_theta[3],
              double shrink, double* d_Sigma, void* stream) {
  PPBO_REQUIRE(ctx, ctx != nullptr, "ctx");
  PPBO_REQUIRE(ctx, d_X && d_Sigma && h_theta, "null pointer");
  PPBO_REQUIRE(ctx, N > 0 && D > 0 && D <= 64, "N>0, 0<D<=64");
  PPBO_REQUIRE(ctx, kernel_id >= 0 && kernel_id <= 2, "kernel_id");
  PPBO_REQUIRE(ctx, kernel_id != PPBO_KERNEL_CAMPHOR || D == 6, "camphor kernel needs D == 6");
  const KernParams p = make_kern_params(kernel_id, h_theta);
  const int nt = (N + TS - 1) / TS;
  const int nblk = nt * (nt + 1) / 2;
  hipStream_t s = (hipStream_t)stream;
  static bool attr_done = false;
  if (!attr_done) {   // D up to 64 needs more than the default 64 KB of dynamic LDS
    const int cap = 112 * 1024;
    (void)hipFuncSetAttribute((const void*)gram_mfma_kernel<PPBO_KERNEL_SE>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute((const void*)gram_mfma_kernel<PPBO_KERNEL_RQ>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute((const void*)gram_kernel<PPBO_KERNEL_CAMPHOR>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute((const void*)crosscov_kernel<PPBO_KERNEL_SE>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute((const void*)crosscov_kernel<PPBO_KERNEL_RQ>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    attr_done = true;
  }
  PpboProfScope pf(ctx, ppbo_ctx::PF_GRAM, s);
  if (kernel_id != PPBO_KERNEL_CAMPHOR) {
    const int Dp = (D + 3) & ~3, LD = Dp + 2;
    const int body = (TS * TP > 2 * TS * LD) ? TS * TP : 2 * TS * LD;
    const size_t lds = (size_t)(body + 2 * TS) * sizeof(double);
    if (kernel_id == PPBO_KERNEL_SE) gram_mfma_kernel<PPBO_KERNEL_SE><<<nblk, 256, lds, s>>>(d_X, N, D, p, shrink, d_Sigma, nt);
    else gram_mfma_kernel<PPBO_KERNEL_RQ><<<nblk, 256, lds, s>>>(d_X, N, D, p, shrink, d_Sigma, nt);
  } else {
    const size_t lds = ((size_t)2 * D * TS + (size_t)TS * TP) * sizeof(double);
    gram_kernel<PPBO_KERNEL_CAMPHOR><<<nblk, 256, lds, s>>>(d_X, N, D, p, shrink, d_Sigma, nt);
  }
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

int ppbo_cross_cov(ppbo_ctx* ctx, int kernel_id, const double* d_X1, int n1, const double* d_X2, int n2,
                   int D, const double h_theta[3], double* d_K, int ldk, void* stream) {
  PPBO_REQUIRE(ctx, ctx != nullptr, "ctx");
  PPBO_REQUIRE(ctx, d_X1 && d_X2 && d_K && h_theta, "null pointer");
  PPBO_REQUIRE(ctx, n1 > 0 && n2 > 0 && D > 0 && D <= 64 && ldk >= n2, "sizes (D<=64)");
  PPBO_REQUIRE(ctx, kernel_id >= 0 && kernel_id <= 2, "kernel_id");
  PPBO_REQUIRE(ctx, kernel_id != PPBO_KERNEL_CAMPHOR || D == 6, "camphor kernel needs D == 6");
  const KernParams p = make_kern_params(kernel_id, h_theta);
  dim3 grid((n2 + TS - 1) / TS, (n1 + TS - 1) / TS);
  const size_t lds = (size_t)2 * D * TS * sizeof(double);
  hipStream_t s = (hipStream_t)stream;
  switch (kernel_id) {
    case PPBO_KERNEL_SE: crosscov_kernel<PPBO_KERNEL_SE><<<grid, 256, lds, s>>>(d_X1, n1, d_X2, n2, D, p, d_K, ldk); break;
    case PPBO_KERNEL_RQ: crosscov_kernel<PPBO_KERNEL_RQ><<<grid, 256, lds, s>>>(d_X1, n1, d_X2, n2, D, p, d_K, ldk); break;
    default: crosscov_kernel<PPBO_KERNEL_CAMPHOR><<<grid, 256, lds, s>>>(d_X1, n1, d_X2, n2, D, p, d_K, ldk); break;
  }
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

}  // extern "C"
