// Generic fp64 MFMA GEMM on the gemm64 tile engine:  C = alpha op(A) op(B) + beta C.
// Used by the dense factorizations (SYRK/TRSM-as-GEMM updates, triangular
// inverse products) and exposed as ppbo_dgemm for tests.
#include "gemm_f64.h"
#include "linalg.h"

namespace {

using namespace gemm64;
using GCBig = Cfg<4, 2, 2, 4>;     // 128 x 128 tile, 8 wavefronts of 32 x 64 (4 waves/SIMD): large products
using GCBig16 = Cfg<4, 4, 2, 2>;  // the same tile as 16 wavefronts of 32 x 32 (8 waves/SIMD): the default for large products since
                                  // round 6 (Y = G K* of 512 lines at C3: 2.18 against 2.23 ms; same bits: the order of a K sum does
                                  // not depend on the tile shape); PPBO_GEMM_BIG16=0 selects GCBig
using GCSmall = Cfg<2, 2, 2, 2>;   // 64 x 64 tile, 4 wavefronts of 32 x 32: short-K panel updates, where the
                                   // grid must put >= 2 wavefronts on every SIMD to reach the 64-cycle MFMA rate
using GCTiny = Cfg<2, 2, 1, 1>;    // 32 x 32 tile, 4 wavefronts of 16 x 16: products with so few 64 x 64 tiles that a CU
                                   // would hold ONE workgroup, whose single stage of prefetch (one k-chunk = 1024 MFMA
                                   // cycles) cannot cover a memory round trip; four small workgroups per CU can

template <class GC, int ALAY, int BLAY>
__global__ __launch_bounds__(GC::NT, (GC::NT == 1024) ? 8 : (GC::NT == 512) ? 4 : 2) void dgemm_kernel(GemmArgs g) {
  constexpr bool PRELOAD = (GC::NT == 256);   // the small-tile configuration (latency-bound panel updates)
  constexpr int BM = GC::BM, BN = GC::BN;
  if (g.batch > 1) {
    g.A += (size_t)blockIdx.y * g.strideA;
    g.B += (size_t)blockIdx.y * g.strideB;
    g.C += (size_t)blockIdx.y * g.strideC;
  }
  const int ntn = (g.N + BN - 1) / BN;
  const int id = blockIdx.x;      // (an XCD-aware remap of the ids was measured on Y = G K*: 2.49 -> 3.45 ms)
  int mt = id / ntn, nt = id % ntn;
  if (g.nt_chunk > 0) {
    const int ntm = (g.M + BM - 1) / BM, per_chunk = ntm * g.nt_chunk;
    const int c = id / per_chunk, rem = id - c * per_chunk;
    mt = ntm - 1 - rem / g.nt_chunk;
    nt = c * g.nt_chunk + rem % g.nt_chunk;
    if (nt >= ntn) return;               // the ragged last chunk
  }
  const int m0 = mt * BM, n0 = nt * BN;
  if (g.lower_only && n0 > m0 + BM - 1) return;
  int kbeg = 0, kend = g.K;
  if (g.khi_mode == 1) {
    // tri_block > 1 (A = G of a posterior): rounded up to the chunk depth -- G carries explicit zeros right of a row's
    // last star, and a K range that is a whole number of chunks keeps the tile on the unguarded loop whatever the star
    // size (m = 25, the reference's default: 26-row stars).  tri_block == 1 (the triangular inverse's products on an
    // Linv whose strict upper part is unwritten): NO round-up -- the K range must end at the triangle's edge
    int e = ((m0 + BM + g.tri_block - 1) / g.tri_block) * g.tri_block;
    if (g.tri_block > 1) e = (e + BK - 1) & ~(BK - 1);
    kend = e < g.K ? e : g.K;
  } else if (g.khi_mode == 2) {
    const int e = (m0 < n0 ? m0 : n0) + BM;
    kend = e < g.K ? e : g.K;
  }
  if (g.klo_mode == 1) kbeg = (m0 > n0 ? m0 : n0);
  else if (g.klo_mode == 2) kbeg = n0;
  kbeg &= ~1;
  double4_t acc[GC::TM][GC::TN];
  // rank-k updates C -= A B^T (alpha = -beta) start from the accumulators holding -C: the read of C is
  // issued with the first operand tiles instead of as a second memory round trip after the main loop,
  // and alpha * (A B - C) = alpha * A B + beta * C
  const bool preload = PRELOAD && (g.beta != 0.0) && (g.alpha == -g.beta || g.alpha == g.beta);
  if (preload) {
    const double sgn = (g.alpha == g.beta) ? 1.0 : -1.0;
#pragma unroll
    for (int i = 0; i < GC::TM; ++i)
#pragma unroll
      for (int j = 0; j < GC::TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + acc_row<GC>(i, r), col = n0 + acc_col<GC>(j);
          acc[i][j][r] = (row < g.M && col < g.N) ? sgn * g.C[(size_t)row * g.ldc + col] : 0.0;
        }
  } else {
    zero_acc<GC>(acc);
  }
  int k_wave_end = 0x7fffffff;
  if (g.khi_mode == 1) {          // A is zero right of the block of a row: Y = G K* 2.55 -> 2.49 ms
    const int wrow_end = m0 + ((int)(threadIdx.x >> 6) / GC::WN + 1) * GC::TM * 16;
    k_wave_end = ((wrow_end + g.tri_block - 1) / g.tri_block) * g.tri_block;
  }
  mainloop<GC, ALAY, BLAY>(g.A, g.lda, g.B, g.ldb, g.M, g.N, m0, n0, kbeg, kend, acc, k_wave_end);
#pragma unroll
  for (int i = 0; i < GC::TM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + acc_row<GC>(i, r);
      if (row >= g.M) continue;
#pragma unroll
      for (int j = 0; j < GC::TN; ++j) {
        const int col = n0 + acc_col<GC>(j);
        if (col >= g.N) continue;
        double* c = g.C + (size_t)row * g.ldc + col;
        double v = g.alpha * acc[i][j][r];
        if (!preload && g.beta != 0.0) v += g.beta * (*c);
        store_through(c, v);
      }
    }
}

template <class GC>
int launch_cfg(ppbo_ctx* ctx, const GemmArgs& g, int transA, int transB, hipStream_t s) {
  const int ntm = (g.M + GC::BM - 1) / GC::BM, ntn = (g.N + GC::BN - 1) / GC::BN;
  const size_t lds = GC::LDS_DOUBLES * sizeof(double);
  if (lds > 64 * 1024) {
    ppbo_lds_limit(ctx, (const void*)dgemm_kernel<GC, KC, RC>, (int)lds);
    ppbo_lds_limit(ctx, (const void*)dgemm_kernel<GC, KC, KC>, (int)lds);
    ppbo_lds_limit(ctx, (const void*)dgemm_kernel<GC, RC, RC>, (int)lds);
    ppbo_lds_limit(ctx, (const void*)dgemm_kernel<GC, RC, KC>, (int)lds);
  }
  const bool chunked = g.nt_chunk > 0 && g.batch <= 1;
  const int ntn_grid = chunked ? ((ntn + g.nt_chunk - 1) / g.nt_chunk) * g.nt_chunk : ntn;
  GemmArgs gg = g;
  if (!chunked) gg.nt_chunk = 0;
  const dim3 grid(ntm * ntn_grid, g.batch > 1 ? g.batch : 1);
  if (!transA && !transB) dgemm_kernel<GC, KC, RC><<<grid, GC::NT, lds, s>>>(gg);
  else if (!transA && transB) dgemm_kernel<GC, KC, KC><<<grid, GC::NT, lds, s>>>(gg);
  else if (transA && !transB) dgemm_kernel<GC, RC, RC><<<grid, GC::NT, lds, s>>>(gg);
  else dgemm_kernel<GC, RC, KC><<<grid, GC::NT, lds, s>>>(gg);
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

}  // namespace

int ppbo_gemm_launch(ppbo_ctx* ctx, const GemmArgs& g, int transA, int transB, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0) return 0;
  if (g.force_cfg == 1) return launch_cfg<GCBig>(ctx, g, transA, transB, s);
  if (g.force_cfg == 2) return launch_cfg<GCSmall>(ctx, g, transA, transB, s);
  if (g.force_cfg == 3) return launch_cfg<GCTiny>(ctx, g, transA, transB, s);
  // big tiles only when they still give every CU several workgroups
  const long long big_tiles = (long long)((g.M + 127) / 128) * ((g.N + 127) / 128) * (g.batch > 1 ? g.batch : 1) /
                              (g.lower_only ? 2 : 1);
  if (big_tiles >= 1024 && g.K >= 256)
    return ctx->gemm_big16 ? launch_cfg<GCBig16>(ctx, g, transA, transB, s) : launch_cfg<GCBig>(ctx, g, transA, transB, s);
  const long long small_tiles = (long long)((g.M + 63) / 64) * ((g.N + 63) / 64) * (g.batch > 1 ? g.batch : 1) /
                                (g.lower_only ? 2 : 1);
  if (small_tiles <= 384 && g.K >= 128) return launch_cfg<GCTiny>(ctx, g, transA, transB, s);
  return launch_cfg<GCSmall>(ctx, g, transA, transB, s);
}

extern "C" int ppbo_dgemm(ppbo_ctx* ctx, int transA, int transB, int M, int N, int K, double alpha,
                          const double* d_A, int lda, const double* d_B, int ldb, double beta, double* d_C,
                          int ldc, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_A && d_B && d_C, "null pointer");
  PPBO_REQUIRE(ctx, M >= 0 && N >= 0 && K >= 0, "sizes");
  GemmArgs g{};
  g.A = d_A; g.lda = lda; g.B = d_B; g.ldb = ldb; g.C = d_C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta;
  g.tri_block = 1;
  return ppbo_gemm_launch(ctx, g, transA, transB, (hipStream_t)stream);
}
