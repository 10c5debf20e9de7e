// a-9: signed log-determinant of I + Sigma*Lambda by LU with partial pivoting.
//   reference: gp_model.py:301-310 -- scipy.linalg.lu(matrix) then slogdet of P, L and U summed as
//   sign*logdet.  P and L contribute 0 (logdet 0), so the value is  sign(prod diag U) * sum log|diag U|,
//   which depends on the pivot sequence; LAPACK's rule is reproduced: pivot = first row of maximal
//   |entry| in the column at and below the diagonal.
// Right-looking blocked LU, NB = 16:
//   getrf_panel   one 1024-thread workgroup: per column an argmax reduction (first index wins ties),
//                 the row swap inside the panel, scaling and the rank-1 update of the panel columns.
//                 Up to 2048 rows below the diagonal the panel lives in REGISTERS for the whole factorization
//                 (getrf_panel_reg_kernel: a thread owns one or two rows of 16 entries; the pivot row and the row it
//                 displaces travel through LDS) -- the memory-resident form (getrf_panel_kernel, kept for taller
//                 panels) pays ~5 dependent global round trips per column, 20 us x N columns = 40 of the 57 ms of
//                 one evidence at N = 2048
//   laswp         the panel's row swaps applied to all columns outside it (one lane per column)
//   trsm_unit     U12 = L11^-1 A12, one lane per column, L11 broadcast from LDS
//   dgemm         A22 -= L21 U12 on the fp64 MFMA engine
// The matrix M = I + Sigma*Lambda itself is formed from the star-graph Lambda by column operations.
#include "linalg.h"

namespace {

constexpr int LNB = 16;

struct PivRec {
  double val;
  int idx;
};
__device__ __forceinline__ PivRec piv_merge(PivRec a, PivRec b) {
  // larger |value| wins; ties -> smaller row index (idamax: first maximal element)
  if (b.idx < 0) return a;
  if (a.idx < 0) return b;
  if (b.val > a.val || (b.val == a.val && b.idx < a.idx)) return b;
  return a;
}

__global__ __launch_bounds__(1024) void getrf_panel_kernel(double* __restrict__ A, int lda, int N, int k0, int nb,
                                                           int* __restrict__ ipiv, int* __restrict__ info) {
  __shared__ PivRec sh[16];
  __shared__ double prow[LNB];
  __shared__ int s_p;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int j = 0; j < nb; ++j) {
    const int col = k0 + j;
    PivRec best{0.0, -1};
    for (int r = col + t; r < N; r += 1024) {
      const double v = fabs(A[(size_t)r * lda + col]);
      if (best.idx < 0 || v > best.val) { best.val = v; best.idx = r; }   // ascending r: strict > keeps the first
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      PivRec other;
      other.val = __shfl_xor(best.val, o, 64);
      other.idx = __shfl_xor(best.idx, o, 64);
      best = piv_merge(best, other);
    }
    if (lane == 0) sh[w] = best;
    __syncthreads();
    if (t == 0) {
      PivRec b = sh[0];
      for (int k = 1; k < 16; ++k) b = piv_merge(b, sh[k]);
      s_p = b.idx;
      ipiv[col] = b.idx;
      if (b.val == 0.0 && *info == 0) *info = col + 1;
    }
    __syncthreads();
    const int p = s_p;
    if (p != col && t < nb) {
      double* ra = A + (size_t)col * lda + k0 + t;
      double* rb = A + (size_t)p * lda + k0 + t;
      const double x = *ra;
      *ra = *rb;
      *rb = x;
    }
    __syncthreads();
    if (t < nb) prow[t] = A[(size_t)col * lda + k0 + t];
    __syncthreads();
    const double piv = prow[j];
    if (piv != 0.0) {
      for (int r = col + 1 + t; r < N; r += 1024) {
        double* row = A + (size_t)r * lda + k0;
        const double l = row[j] / piv;
        row[j] = l;
        for (int c = j + 1; c < nb; ++c) row[c] -= l * prow[c];
      }
    }
    __syncthreads();
  }
}

// (value, index) argmax over a wavefront without the LDS crossbar: four DPP steps inside each row of 16 lanes
// (xor 1, xor 2, row_half_mirror, row_mirror), then the four row winners through scalar registers.  piv_merge is
// a max under a total order (larger |value|, then smaller index), so any reduction order gives the same winner.
template <int CTRL>
__device__ __forceinline__ PivRec piv_dpp_step(PivRec a) {
  PivRec o;
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(a.val), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(a.val), CTRL, 0xF, 0xF, true);
  o.idx = __builtin_amdgcn_mov_dpp(a.idx, CTRL, 0xF, 0xF, true);
  o.val = __hiloint2double(hi, lo);
  return piv_merge(a, o);
}
__device__ __forceinline__ PivRec wave_piv(PivRec a) {
  a = piv_dpp_step<0xB1>(a);
  a = piv_dpp_step<0x4E>(a);
  a = piv_dpp_step<0x141>(a);
  a = piv_dpp_step<0x140>(a);
  PivRec r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r[k].val = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a.val), 16 * k),
                                __builtin_amdgcn_readlane(__double2loint(a.val), 16 * k));
    r[k].idx = __builtin_amdgcn_readlane(a.idx, 16 * k);
  }
  return piv_merge(piv_merge(r[0], r[1]), piv_merge(r[2], r[3]));
}

// One column step of the register-resident panel (J is a template parameter: every index into a[][] is static, or the
// panel would be demoted to scratch).
template <int RPT, int NT, int J>
__device__ __forceinline__ void lu_reg_column(double (&a)[RPT][LNB], int N, int k0, int* __restrict__ ipiv,
                                              int* __restrict__ info, PivRec* sh, double* rowc, double* prow) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int col = k0 + J;
  PivRec best{0.0, -1};
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = k0 + t + NT * i;
    if (r >= col && r < N) {
      const double v = fabs(a[i][J]);
      if (best.idx < 0 || v > best.val) { best.val = v; best.idx = r; }   // ascending r: strict > keeps the first
    }
  }
  best = wave_piv(best);
  if (lane == 0) sh[w] = best;
  __syncthreads();
  // every wavefront merges the 16 records itself (lane l < 16 takes record l): no serial merge by one thread, no
  // second barrier to publish the winner
  PivRec mine{0.0, -1};
  if (lane < NT / 64) mine = sh[lane];
  const PivRec win = wave_piv(mine);
  if (t == 0) {
    ipiv[col] = win.idx;
    if (win.val == 0.0 && *info == 0) *info = col + 1;
  }
  const int p = win.idx;
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = k0 + t + NT * i;
    if (r == col) {
#pragma unroll
      for (int c = 0; c < LNB; ++c) rowc[c] = a[i][c];
    }
    if (r == p) {
#pragma unroll
      for (int c = 0; c < LNB; ++c) prow[c] = a[i][c];
    }
  }
  __syncthreads();
  if (p != col) {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = k0 + t + NT * i;
      if (r == col) {
#pragma unroll
        for (int c = 0; c < LNB; ++c) a[i][c] = prow[c];
      } else if (r == p) {
#pragma unroll
        for (int c = 0; c < LNB; ++c) a[i][c] = rowc[c];
      }
    }
  }
  const double piv = prow[J];
  if (piv != 0.0) {
    // LAPACK's dgetf2 / dgetrf2 scale the column by the reciprocal of the pivot (for |pivot| >= sfmin), not by a
    // division per entry; one correctly rounded reciprocal per thread and column instead of RPT divisions
    const bool tiny = fabs(piv) < 2.2250738585072014e-308;
    const double rinv = 1.0 / piv;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = k0 + t + NT * i;
      if (r > col && r < N) {
        const double l = tiny ? a[i][J] / piv : a[i][J] * rinv;
        a[i][J] = l;
#pragma unroll
        for (int c = J + 1; c < LNB; ++c) a[i][c] -= l * prow[c];
      }
    }
  }
  // no barrier here: the next column rewrites sh only after every thread has passed the barrier above (all reads of
  // sh precede it), and rowc / prow only after the next column's first barrier (all reads of them precede that)
}

// The panel in registers: thread t owns rows k0 + t + NT i (i < RPT) of the panel's nb <= 16 columns; NT shrinks with
// the number of rows left (fewer wavefronts to issue and to meet at the two barriers of a column).
template <int RPT, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT / 256, NT / 256))) void getrf_panel_reg_kernel(
    double* __restrict__ A, int lda, int N, int k0, int nb, int* __restrict__ ipiv, int* __restrict__ info) {
  __shared__ PivRec sh[16];
  __shared__ double rowc[LNB], prow[LNB];    // the row at the diagonal before the swap; the pivot row
  const int t = threadIdx.x;
  double a[RPT][LNB];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = k0 + t + NT * i;
#pragma unroll
    for (int c = 0; c < LNB; ++c) a[i][c] = (r < N && c < nb) ? A[(size_t)r * lda + k0 + c] : 0.0;
  }
#define LU_COL(J) if (J < nb) lu_reg_column<RPT, NT, J>(a, N, k0, ipiv, info, sh, rowc, prow);     /* nb is uniform */
  LU_COL(0) LU_COL(1) LU_COL(2) LU_COL(3) LU_COL(4) LU_COL(5) LU_COL(6) LU_COL(7)
  LU_COL(8) LU_COL(9) LU_COL(10) LU_COL(11) LU_COL(12) LU_COL(13) LU_COL(14) LU_COL(15)
#undef LU_COL
  static_assert(LNB == 16, "LU_COL list");
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = k0 + t + NT * i;
    if (r < N) {
#pragma unroll
      for (int c = 0; c < LNB; ++c)
        if (c < nb) A[(size_t)r * lda + k0 + c] = a[i][c];
    }
  }
}

// apply the panel's swaps to the columns outside [k0, k0+nb)
__global__ __launch_bounds__(256) void laswp_kernel(double* __restrict__ A, int lda, int N, int k0, int nb,
                                                    const int* __restrict__ ipiv) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N - nb) return;
  if (c >= k0) c += nb;
  for (int j = 0; j < nb; ++j) {
    const int r = k0 + j, p = ipiv[r];
    if (p != r) {
      const double x = A[(size_t)r * lda + c];
      A[(size_t)r * lda + c] = A[(size_t)p * lda + c];
      A[(size_t)p * lda + c] = x;
    }
  }
}

// A12 <- L11^-1 A12 (unit lower L11), one lane per column right of the panel
__global__ __launch_bounds__(256) void trsm_unit_kernel(double* __restrict__ A, int lda, int N, int k0, int nb) {
  __shared__ double Ls[LNB * LNB];
  for (int e = threadIdx.x; e < nb * nb; e += blockDim.x) {
    const int r = e / nb, c = e - r * nb;
    Ls[r * LNB + c] = (c < r) ? A[(size_t)(k0 + r) * lda + k0 + c] : 0.0;
  }
  __syncthreads();
  const int c = k0 + nb + blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  double x[LNB];
#pragma unroll
  for (int r = 0; r < LNB; ++r) x[r] = (r < nb) ? A[(size_t)(k0 + r) * lda + c] : 0.0;
#pragma unroll
  for (int r = 1; r < LNB; ++r) {
    double v = x[r];
#pragma unroll
    for (int k = 0; k < r; ++k) v -= Ls[r * LNB + k] * x[k];
    x[r] = v;
  }
#pragma unroll
  for (int r = 0; r < LNB; ++r)
    if (r < nb) A[(size_t)(k0 + r) * lda + c] = x[r];
}

// out[0] = prod sign(u_ii), out[1] = sum log|u_ii|
__global__ __launch_bounds__(1024) void diag_slogdet_kernel(const double* __restrict__ A, int lda, int N,
                                                            double* __restrict__ out) {
  __shared__ double sh[2][16];
  double ld = 0.0;
  int neg = 0;
  for (int i = threadIdx.x; i < N; i += 1024) {
    const double u = A[(size_t)i * lda + i];
    ld += log(fabs(u));
    neg ^= (u < 0.0) ? 1 : 0;
  }
  double negd = (double)neg;
  ld = wave_sum(ld);
  negd = wave_sum(negd);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = ld; sh[1][threadIdx.x >> 6] = negd; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < 16; ++w) { a += sh[0][w]; b += sh[1][w]; }
    out[0] = (((long long)b) & 1) ? -1.0 : 1.0;
    out[1] = a;
  }
}

// M = I + Sigma * Lambda for the star-structured Lambda; thread = (row i, star q)
__global__ __launch_bounds__(256) void ipsl_kernel(const double* __restrict__ S, int N, int mblk,
                                                   const double* __restrict__ lam_diag,
                                                   const double* __restrict__ lam_off, double* __restrict__ M) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int q0 = blockIdx.y * mblk;
  if (i >= N) return;
  const double* s = S + (size_t)i * N + q0;
  double* m = M + (size_t)i * N + q0;
  const double s0 = s[0];
  double acc = s0 * lam_diag[q0];
  for (int k = 1; k < mblk && q0 + k < N; ++k) {
    const double sk = s[k];
    const double lo = lam_off[q0 + k];
    m[k] = sk * lam_diag[q0 + k] + s0 * lo + ((i == q0 + k) ? 1.0 : 0.0);
    acc += sk * lo;
  }
  m[0] = acc + ((i == q0) ? 1.0 : 0.0);
}

}  // namespace

extern "C" {

int ppbo_lu_slogdet(ppbo_ctx* ctx, double* d_A, int N, int lda, double* h_u_sign, double* h_u_logdet, int* h_info,
                    void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_A && N > 0 && lda >= N, "matrix");
  hipStream_t s = (hipStream_t)stream;
  double* d_out = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_VEC, 4 * sizeof(double) + ((size_t)N + 8) * sizeof(int));
  if (!d_out) return (int)hipErrorOutOfMemory;
  int* d_ipiv = reinterpret_cast<int*>(d_out + 4);
  int* d_info = d_ipiv + N;
  PPBO_HIP_CHECK(ctx, hipMemsetAsync(d_info, 0, sizeof(int), s));
  for (int k0 = 0; k0 < N; k0 += LNB) {
    const int nb = (N - k0 < LNB) ? (N - k0) : LNB;
    if (N - k0 <= 256) getrf_panel_reg_kernel<1, 256><<<1, 256, 0, s>>>(d_A, lda, N, k0, nb, d_ipiv, d_info);
    else if (N - k0 <= 512) getrf_panel_reg_kernel<1, 512><<<1, 512, 0, s>>>(d_A, lda, N, k0, nb, d_ipiv, d_info);
    else if (N - k0 <= 1024) getrf_panel_reg_kernel<1, 1024><<<1, 1024, 0, s>>>(d_A, lda, N, k0, nb, d_ipiv, d_info);
    else if (N - k0 <= 2048) getrf_panel_reg_kernel<2, 1024><<<1, 1024, 0, s>>>(d_A, lda, N, k0, nb, d_ipiv, d_info);
    else getrf_panel_kernel<<<1, 1024, 0, s>>>(d_A, lda, N, k0, nb, d_ipiv, d_info);
    if (N - nb > 0) laswp_kernel<<<(N - nb + 255) / 256, 256, 0, s>>>(d_A, lda, N, k0, nb, d_ipiv);
    const int rest = N - k0 - nb;
    if (rest > 0) {
      trsm_unit_kernel<<<(rest + 255) / 256, 256, 0, s>>>(d_A, lda, N, k0, nb);
      GemmArgs g{};
      g.A = d_A + (size_t)(k0 + nb) * lda + k0; g.lda = lda;
      g.B = d_A + (size_t)k0 * lda + (k0 + nb); g.ldb = lda;
      g.C = d_A + (size_t)(k0 + nb) * lda + (k0 + nb); g.ldc = lda;
      g.M = rest; g.N = rest; g.K = nb; g.alpha = -1.0; g.beta = 1.0; g.tri_block = 1;
      if (int rc = ppbo_gemm_launch(ctx, g, 0, 0, s)) return rc;
    }
  }
  diag_slogdet_kernel<<<1, 1024, 0, s>>>(d_A, lda, N, d_out);
  PPBO_LAUNCH_CHECK(ctx);
  double h[2];
  int info = 0;
  PPBO_HIP_CHECK(ctx, hipMemcpyAsync(h, d_out, sizeof(h), hipMemcpyDeviceToHost, s));
  PPBO_HIP_CHECK(ctx, hipMemcpyAsync(&info, d_info, sizeof(int), hipMemcpyDeviceToHost, s));
  PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
  if (h_u_sign) *h_u_sign = h[0];
  if (h_u_logdet) *h_u_logdet = h[1];
  if (h_info) *h_info = info;
  return 0;
}

int ppbo_laplace_logdet(ppbo_ctx* ctx, const double* d_Sigma, const double* d_lam_diag, const double* d_lam_off,
                        int N, int m, double* h_u_sign, double* h_u_logdet, int* h_info, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_Sigma && d_lam_diag && d_lam_off, "null pointer");
  PPBO_REQUIRE(ctx, N > 0 && m >= 1 && N % (m + 1) == 0, "sizes");
  hipStream_t s = (hipStream_t)stream;
  const int mblk = m + 1, n_q = N / mblk;
  double* M = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_LINALG, (size_t)2 * N * N * sizeof(double));
  if (!M) return (int)hipErrorOutOfMemory;
  ipsl_kernel<<<dim3((N + 255) / 256, n_q), 256, 0, s>>>(d_Sigma, N, mblk, d_lam_diag, d_lam_off, M);
  PPBO_LAUNCH_CHECK(ctx);
  return ppbo_lu_slogdet(ctx, M, N, N, h_u_sign, h_u_logdet, h_info, stream);
}

}  // extern "C"
