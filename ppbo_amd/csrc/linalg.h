// Internal (non-ABI) dense linear-algebra entry points shared between .hip files.
#pragma once
#include "common.h"

struct GemmArgs {
  const double* A; int lda;
  const double* B; int ldb;
  double* C; int ldc;
  int M, N, K;
  double alpha, beta;
  int lower_only;   // skip output tiles strictly above the diagonal
  int khi_mode;     // 0: K   1: A block-lower-triangular (k < roundup(m0+128, tri_block); each wavefront stops at the end of the
                    //           block of ITS last row, as quadform_kernel does)   2: k < min(m0,n0)+128
  int klo_mode;     // 0: 0   1: k >= max(m0, n0)   2: k >= n0
  int tri_block;
  int nt_chunk;     // > 0 (128 x 128-tile configuration, no batch): tiles are walked in chunks of nt_chunk column tiles -- all
                    // row tiles of a chunk, longest K range first, before the next chunk -- so that the chunk's slice
                    // of B is re-read from the Infinity Cache instead of HBM once per row tile (the order of
                    // quadform_kernel); 0: row-tile-major as launched
  int force_cfg;                  // 0: tile configuration by size | 1: 128 x 128 | 2: 64 x 64 | 3: 32 x 32
  int batch;                      // grid.y; operand b lives at base + b*stride (elements); 0/1 = single
  long long strideA, strideB, strideC;
};

// C = alpha op(A) op(B) + beta C on the fp64 MFMA engine (see gemm.hip)
int ppbo_gemm_launch(ppbo_ctx* ctx, const GemmArgs& g, int transA, int transB, hipStream_t s);

// in-place lower Cholesky; d_info (device int) receives 0 or the failing 1-based column
// d_fail_pivot (optional, device double): receives the non-positive pivot of a failed factorization
// (only the default one-launch-per-step path writes it; callers pre-set it to NaN)
int ppbo_potrf_async(ppbo_ctx* ctx, double* d_A, int N, int lda, int* d_info, hipStream_t s,
                     double* d_fail_pivot = nullptr);
// after a failure: d_out[0] = (-pivot) / |v|^2, the amount by which the shift must at least grow
// (Conn/Gould/Toint 7.3; scipy.optimize._trustregion_exact.singular_leading_submatrix).  Returns 1 when
// the bound is not available for this size.
int ppbo_potrf_fail_bound_async(ppbo_ctx* ctx, const double* d_L, int N, int ldl, const int* d_info,
                                const double* d_fail_pivot, double* d_out, hipStream_t s);
// d_Linv (full N x N, upper part zeroed) = inverse of lower-triangular L
// skip_top: leave the last doubling level (the split [0, b) | [b, N), b the largest 64 * 2^k < N) unformed and
// report b in *split_out; the result is then applied with ppbo_apply_linv_async
// zero_upper = 0: the caller promises never to read above the 64 x 64 diagonal blocks (whose own upper triangles are
// written as zeros); the 8 N^2-byte memset of the result is skipped
int ppbo_trtri_async(ppbo_ctx* ctx, const double* d_L, int N, int ldl, double* d_Linv, int ldi, hipStream_t s,
                     int skip_top = 0, int* split_out = nullptr, int zero_upper = 1);
// d_Ainv (N x N, row pitch N) = Linv^T Linv: lower triangle on the matrix cores, upper triangle mirrored
int ppbo_syrk_inverse_async(ppbo_ctx* ctx, const double* d_Linv, int N, double* d_Ainv, hipStream_t s);
int ppbo_apply_linv_async(ppbo_ctx* ctx, const double* d_Linv, int ldi, const double* d_L, int ldl, int N, int split,
                          const double* d_x, double* d_y, int trans, double* d_tmp, hipStream_t s);
// Device-side gate of a launch: the kernel returns at once when *skip_if_nonzero != 0 or *skip_if_zero == 0
// (null pointers = no condition).  Lets the host enqueue a fixed sequence of launches whose tail turns into
// no-ops once a device-resident iteration has decided that it is finished (csrc/fit.hip, whitened L-BFGS).
struct PpboGate {
  const int* skip_if_nonzero = nullptr;
  const int* skip_if_zero = nullptr;
#ifdef __HIPCC__
  __device__ __forceinline__ bool closed() const {
    return (skip_if_nonzero && *skip_if_nonzero != 0) || (skip_if_zero && *skip_if_zero == 0);
  }
#endif
};

// Optional by-product of the launch that finishes u = L^T beta in the whitened f_MAP search: the inner products the
// judgement of the trial point needs, as one row of partial sums per workgroup (summed, in a fixed order, by the
// judgement's one workgroup).  Row layout (PPBO_DOTS_STRIDE doubles): [0] zt.zt  [1] |v - beta|^2  [2] s.y  [3] s.s
// [4] y.y  [5] gt.d  [6] gt.gt  [8 + l] gt.b_l  with gt = zt - u, s = zt - z, y = gt - gcur.
constexpr int PPBO_DOTS_STRIDE = 32;
struct PpboDotsOut {
  const double *zt = nullptr, *z = nullptr, *gcur = nullptr, *d = nullptr, *v = nullptr, *beta = nullptr, *basis = nullptr;
  int nb = 0;                 // basis vectors [nb][N], nb <= PPBO_DOTS_STRIDE - 8
  double* partial = nullptr;  // [n_part][PPBO_DOTS_STRIDE]; null: no by-product
};
// y = T x (trans=0) or y = T^T x (trans=1) for a lower-triangular (lower=1) or full N x N matrix
int ppbo_gemv_async(ppbo_ctx* ctx, const double* d_T, int N, int ldt, const double* d_x, double* d_y, int trans,
                    int lower, hipStream_t s, PpboGate gate = PpboGate());
// u = L^T beta(f) in two launches with beta[N] and tq[N / mblk] as by-products (one launch less than laplace_kernel +
// ppbo_gemv_async); returns 1 without enqueueing anything when a star (mblk rows) exceeds one wavefront.
// With d_R: the first launch also carries rv = R f (full N x N, by rows) behind its own gate -- a second product with
// the same f that would otherwise be a launch of its own
int ppbo_gemvT_beta_async(ppbo_ctx* ctx, const double* d_L, int N, int ldl, const double* d_f, int mblk, double sigma,
                          double* d_u, double* d_beta, double* d_tq, hipStream_t s, PpboGate gate = PpboGate(),
                          const double* d_R = nullptr, int ldr = 0, double* d_rv = nullptr,
                          PpboGate rider_gate = PpboGate(), PpboDotsOut dots = PpboDotsOut(), int* n_dot_parts = nullptr);
// out[0] = sum_i x_i y_i  (deterministic single-block reduction)
int ppbo_dot_async(ppbo_ctx* ctx, const double* d_x, const double* d_y, int N, double* d_out, hipStream_t s);
