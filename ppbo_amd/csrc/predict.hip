// K2+K3+K4: batched candidate scoring (posterior mean, variance, score, argmax).
//   reference: gp_model.py:441-461 (mu_Sigma_pred / mu_pred), called per candidate by
//   mu_star's DE (:415-437) and per 70-point line by EI/varmax (acquisition.py:72-81,170-178).
//
// Posterior state (see ppbo_posterior): alpha = Sigma^-1 f_MAP, Lambda_MAP in star form,
// G = R Lambda with R = chol(Sigma^-1 - Lambda)^-1 (block lower triangular).  For a candidate x
//   mu(x)  = k*' alpha
//   var(x) = sigma_f^2 - k*' A k*,  A = Sigma^-1 - Sigma^-1 P Sigma^-1  (gp_model.py:449)
//          = sigma_f^2 + k*' Lambda k* + |G k*|^2                        (Woodbury, same operator)
// Pass 1 (kstar_kernel, VALU): one lane = two candidates held in registers, X rows staged in LDS
//   and read as broadcasts (SE / RQ: the reference's expansion with the candidate pre-scaled by -2, one FMA
//   per dimension; camphor: differences); writes K*[N, Mc] (j-major) once and reduces mu and k*'Lambda k*
//   in registers.
// Pass 2 (quadform_kernel, fp64 MFMA): Y = G K* on 128x128 tiles (16 wavefronts of 32x32), K range cut
//   at the block-triangular edge per wavefront, epilogue = column sums of Y^2 into per-row-tile slabs.
//   Workgroups are ordered candidate-tile-fastest in chunks of 128 tiles (PPBO_QF_ORDER, default 514):
//   all resident workgroups stream the same G row panel out of L2 while their K* chunk sits in the
//   Infinity Cache.  PPBO_QF_VARIANT (default 4: 16 wavefronts of 32x32) selects the measured tile shapes, see DESIGN.md.
// Pass 3 (score_kernel): slab sums -> var, score, per-block argmax; (argmax_final_kernel) -> 1 value.
#include "gemm_f64.h"
#include "linalg.h"
#include "score.h"

namespace {

using namespace gemm64;

constexpr int KS_THREADS = 256;
constexpr int KS_CPT = 2;  // candidates per thread (even); 4 halves the LDS reads per pair but was measured slower (fewer waves, 32-byte store pieces)

constexpr int KS_RJ = 32;   // X rows staged in LDS per step

// DP = padded dimension (compile time): X rows are zero-padded to DP in LDS and the
// candidate registers likewise, so the inner product loop is branch-free and fully
// unrolled; every lane reads the same LDS address (broadcast), one ds_read_b128 per 2 dims.
// KS_CPT candidates per lane share each of those reads: with the one-FMA-per-dimension form the kernel is
// bound by the LDS pipe (a broadcast b128 read still returns 1 KB per wavefront), not by the vector ALUs.
// fp32 kernel evaluation (F32 = true): the "fp32 tolerance" variant BASELINE config 5 names.  K* entries are
// evaluated in single precision from direct differences (the expansion form loses 4e-5 of mu in fp32),
// then widened; every accumulation (mu, k*'Lambda k*, the MFMA contraction) stays fp64.
template <int KID>
__device__ __forceinline__ float kern_term32(float dx, int d, float c0, float c1) {
  if (KID == PPBO_KERNEL_CAMPHOR) {
    if (d == 2) return c1 * dx * dx;
    const float sn = sinpif(fabsf(dx));
    return c0 * sn * sn;
  }
  return dx * dx;
}
template <int KID>
__device__ __forceinline__ float kern_finish32(float s, float sf2, float c0) {
  if (KID == PPBO_KERNEL_SE) return sf2 * expf(-c0 * s);
  if (KID == PPBO_KERNEL_RQ) { const float t = 1.0f + s * c0; return sf2 / (t * t); }
  return sf2 * expf(-s);
}

template <int KID, int DP, bool F32 = false>
__global__ __launch_bounds__(KS_THREADS) void kstar_kernel(
    const double* __restrict__ X, int N, int D, KernParams p, const double* __restrict__ alpha,
    const double* __restrict__ lam_diag, const double* __restrict__ lam_off, int mblk,
    const double* __restrict__ Xc, int M, double* __restrict__ Kt, int ldk, double* __restrict__ mu_part,
    double* __restrict__ t_part, int q_per_split, int n_q) {
  __shared__ __attribute__((aligned(16))) double xs[KS_RJ * DP];
  __shared__ double s_alpha[KS_RJ], s_ld[KS_RJ], s_lo[KS_RJ], s_nx[KS_RJ];
  const int c0 = (blockIdx.x * KS_THREADS + threadIdx.x) * KS_CPT;
  // The camphor kernel in FEATURE form (fp64 path; DP = 12): sin^2(pi (a - b)) = (1 - cos 2pi a cos 2pi b - sin 2pi a sin 2pi b) / 2
  // turns kernels.py:36-53's exponent  c0 sum_k sin^2(pi |a_k - b_k|) + c1 (a_2 - b_2)^2  into
  //   5 c0 / 2 + sum_{f < 10} phi_f(a) psi_f(b) + c1 (a_2 - b_2)^2,  phi = (cos 2pi a_k, sin 2pi a_k)_k,  psi = -(c0 / 2) phi(b):
  // ten FMAs per pair instead of five sinpi evaluations (~20 instructions each: K* was 1.88 ms of C5's 16.6 ms step, six
  // times the SE kernel's cost per pair).  The features of a design row are formed once when it is staged (per 512
  // candidates), those of a candidate once per thread.  Terms of size c0 cancel to the true exponent with an absolute
  // error of ~1e-15 (the kernel value moves by <= 1e-14 relative; Sigma / K* parity is asserted at 1e-12).
  constexpr bool CAMF = (KID == PPBO_KERNEL_CAMPHOR) && !F32 && DP == 12;   // (the only fp64 camphor bucket launched)
  // SE / RQ use the reference's own expansion r^2 = (|x|^2 + |c|^2) - 2 x.c (kernels.py:7-10) with the
  // candidate pre-scaled by -2: one FMA per dimension and pair instead of a subtract and an FMA.  Its
  // rounding is the rounding every entry of Sigma already carries (posterior mean / variance move by
  // <= 3e-13 / 1e-13 sigma_f^2 on the fixtures against direct differences).  The camphor kernel needs the
  // differences themselves.
  constexpr bool EXPAND = (KID != PPBO_KERNEL_CAMPHOR) && !F32;
  double xc[KS_CPT][DP], nc[KS_CPT], mu[KS_CPT], tl[KS_CPT], ko[KS_CPT];
  float xcf[KS_CPT][DP];
  const float c0f = (float)p.c0, c1f = (float)p.c1, sf2f = (float)p.sf2;
#pragma unroll
  for (int q = 0; q < KS_CPT; ++q) {
    nc[q] = 0.0; mu[q] = 0.0; tl[q] = 0.0; ko[q] = 0.0;
    if (CAMF) {
      // psi: -(c0 / 2) (cos, sin)(2 pi c_k) for the periodic coordinates k = 0, 1, 3, 4, 5; then c_2 itself
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int d = k < 2 ? k : k + 1;
        const double v = (c0 + q < M) ? Xc[(size_t)(c0 + q) * D + d] : 0.0;
        double sn, cs;
        sincospi(2.0 * v, &sn, &cs);
        xc[q][k] = -0.5 * p.c0 * cs;
        xc[q][5 + k] = -0.5 * p.c0 * sn;
      }
      xc[q][10] = (c0 + q < M) ? Xc[(size_t)(c0 + q) * D + 2] : 0.0;
      xc[q][11] = 0.0;
      continue;
    }
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      double v = (d < D && c0 + q < M) ? Xc[(size_t)(c0 + q) * D + d] : 0.0;
      if (F32) xcf[q][d] = (float)v;
      if (EXPAND) { nc[q] = fma(v, v, nc[q]); v *= -2.0; }
      xc[q][d] = v;
    }
  }
  const int j_beg = blockIdx.y * q_per_split * mblk;
  int j_end = j_beg + q_per_split * mblk;
  if (j_end > N) j_end = N;
  const bool vec = (Kt != nullptr) && ((ldk & 1) == 0) && (c0 + KS_CPT - 1 < M);
  const bool has_lam = (lam_diag != nullptr);
  int rb = 0;  // row index inside the current star block (splits start on a block edge)
  for (int row0 = j_beg; row0 < j_end; row0 += KS_RJ) {
    __syncthreads();
    if (CAMF) {                            // phi of the staged rows: (cos, sin)(2 pi x_k) for k = 0, 1, 3, 4, 5; then x_2
      for (int e = threadIdx.x; e < KS_RJ * 6; e += KS_THREADS) {
        const int r = e / 6, k = e - r * 6;
        const int j = row0 + r;
        if (k < 5) {
          const double v = (j < j_end) ? X[(size_t)j * D + (k < 2 ? k : k + 1)] : 0.0;
          double sn, cs;
          sincospi(2.0 * v, &sn, &cs);
          xs[r * DP + k] = cs;
          xs[r * DP + 5 + k] = sn;
        } else {
          xs[r * DP + 10] = (j < j_end) ? X[(size_t)j * D + 2] : 0.0;
          xs[r * DP + 11] = 0.0;
        }
      }
    } else
    for (int e = threadIdx.x; e < KS_RJ * DP; e += KS_THREADS) {
      const int r = e / DP, d = e - r * DP;
      const int j = row0 + r;
      xs[e] = (j < j_end && d < D) ? X[(size_t)j * D + d] : 0.0;
    }
    if (threadIdx.x < KS_RJ) {
      const int j = row0 + threadIdx.x;
      const bool ok = j < j_end;
      s_alpha[threadIdx.x] = ok ? alpha[j] : 0.0;
      s_ld[threadIdx.x] = (ok && has_lam) ? lam_diag[j] : 0.0;
      s_lo[threadIdx.x] = (ok && has_lam) ? lam_off[j] : 0.0;
      if (EXPAND) {
        double nx = 0.0;
        if (ok)
          for (int d = 0; d < D; ++d) { const double v = X[(size_t)j * D + d]; nx = fma(v, v, nx); }
        s_nx[threadIdx.x] = nx;
      }
    }
    __syncthreads();
    const int rmax = (j_end - row0 < KS_RJ) ? (j_end - row0) : KS_RJ;
    for (int r = 0; r < rmax; ++r) {
      const double* __restrict__ xr = xs + r * DP;
      double sv[KS_CPT];
      double kv[KS_CPT];
#pragma unroll
      for (int q = 0; q < KS_CPT; ++q) sv[q] = 0.0;
      if (F32) {
        float sf[KS_CPT];
#pragma unroll
        for (int q = 0; q < KS_CPT; ++q) sf[q] = 0.0f;
#pragma unroll
        for (int d = 0; d < DP; ++d) {
          const float x = (float)xr[d];
#pragma unroll
          for (int q = 0; q < KS_CPT; ++q) sf[q] += kern_term32<KID>(x - xcf[q][d], d, c0f, c1f);
        }
#pragma unroll
        for (int q = 0; q < KS_CPT; ++q) kv[q] = (double)kern_finish32<KID>(sf[q], sf2f, c0f);
      } else if (EXPAND) {
#pragma unroll
        for (int d = 0; d < DP; ++d) {
          const double x = xr[d];
#pragma unroll
          for (int q = 0; q < KS_CPT; ++q) sv[q] = fma(x, xc[q][d], sv[q]);
        }
        const double nx = s_nx[r];
#pragma unroll
        for (int q = 0; q < KS_CPT; ++q) sv[q] = fmax(sv[q] + (nx + nc[q]), 0.0);
      } else if (CAMF) {
#pragma unroll
        for (int d = 0; d < 10; ++d) {
          const double x = xr[d];
#pragma unroll
          for (int q = 0; q < KS_CPT; ++q) sv[q] = fma(x, xc[q][d], sv[q]);
        }
        const double x2 = xr[10], base = 2.5 * p.c0;
#pragma unroll
        for (int q = 0; q < KS_CPT; ++q) {
          const double dd = x2 - xc[q][10];
          sv[q] = fma(p.c1 * dd, dd, sv[q] + base);
        }
      } else {
#pragma unroll
        for (int d = 0; d < DP; ++d) {
          const double x = xr[d];
#pragma unroll
          for (int q = 0; q < KS_CPT; ++q) sv[q] += kern_term<KID>(x - xc[q][d], d, p);
        }
      }
      if (!F32) {
#pragma unroll
        for (int q = 0; q < KS_CPT; ++q) kv[q] = kern_finish<KID>(sv[q], p);
      }
      if (Kt) {
        double* dst = Kt + (size_t)(row0 + r) * ldk + c0;
        if (vec) {      // 1 GB streamed out once, read back by the next kernel
#pragma unroll
          for (int q = 0; q < KS_CPT; q += 2) store_through2(dst + q, kv[q], kv[q + 1]);
        } else {
#pragma unroll
          for (int q = 0; q < KS_CPT; ++q)
            if (c0 + q < M) dst[q] = kv[q];
        }
      }
      const double a = s_alpha[r];
#pragma unroll
      for (int q = 0; q < KS_CPT; ++q) mu[q] += a * kv[q];
      if (has_lam) {
        const double ld = s_ld[r];
        if (rb == 0) {
#pragma unroll
          for (int q = 0; q < KS_CPT; ++q) { ko[q] = kv[q]; tl[q] += ld * kv[q] * kv[q]; }
        } else {
          const double lo2 = 2.0 * s_lo[r];
#pragma unroll
          for (int q = 0; q < KS_CPT; ++q) tl[q] += kv[q] * (ld * kv[q] + lo2 * ko[q]);
        }
      }
      if (++rb == mblk) rb = 0;
    }
  }
#pragma unroll
  for (int q = 0; q < KS_CPT; ++q) {
    if (c0 + q < M) {
      mu_part[(size_t)blockIdx.y * M + c0 + q] = mu[q];
      if (t_part) t_part[(size_t)blockIdx.y * M + c0 + q] = tl[q];
    }
  }
}

// Y = G K* ; slab[mt][c] = sum over the BM rows of tile mt of Y[i,c]^2
template <class C, int MINW, bool ALWAYS_FAST>
__global__ __launch_bounds__(C::NT, MINW) void quadform_kernel(const double* __restrict__ G, int N,
                                                               const double* __restrict__ Kt, int ldk, int M,
                                                               int tri_block, double* __restrict__ slab, int ntm,
                                                               int ntn, int swizzle, int n_rows) {
  int id = blockIdx.x;
  if (swizzle & 1) {                   // XCD-aware: consecutive logical ids share an XCD / L2 (workgroup b runs on XCD b % 8)
    const int per = gridDim.x >> 3, rem = gridDim.x & 7;
    const int xcd = id & 7, pos = id >> 3;            // a bijection for any grid size: XCDs [0, rem) hold per + 1 ids
    id = pos + (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per);
  }
  int mt, nt;
  if (swizzle & 2) {                   // candidate tile fastest: co-resident workgroups share one G row panel
    const int ch = swizzle >> 2;       // optional: candidate tiles in chunks of ch (K* chunk stays in the Infinity Cache)
    if (ch > 0 && ch < ntn) {
      const int per_chunk = ntm * ch, full = ntn / ch;
      if (id < full * per_chunk) {
        const int c = id / per_chunk, r = id % per_chunk;
        mt = ntm - 1 - (r / ch); nt = c * ch + (r % ch);
      } else {                         // the last, narrower chunk
        const int cw = ntn - full * ch, r = id - full * per_chunk;
        mt = ntm - 1 - (r / cw); nt = full * ch + (r % cw);
      }
    } else {
      mt = ntm - 1 - (id / ntn); nt = id % ntn;
    }
  } else {                             // row tile fastest, heaviest (largest K range) first: share one K* panel
    mt = ntm - 1 - (id % ntm); nt = id / ntm;
  }
  const int m0 = mt * C::BM, n0 = nt * C::BN;
  // G is zero right of the last star that reaches into this row tile; the K range ends there, rounded UP to the chunk
  // depth (g_build_kernel writes the zeros of every row explicitly, so the extra columns multiply zeros): every K
  // range is then a whole number of chunks whatever the star size -- m = 25, the reference's default, has 26-row stars
  const int e = (((m0 + C::BM + tri_block - 1) / tri_block) * tri_block + BK - 1) & ~(BK - 1);
  const int kend = e < N ? e : N;
  double4_t acc[C::TM][C::TN];
  zero_acc<C>(acc);
  // rows of this wavefront end at m0 + (wm+1)*TM*16: G is zero beyond the end of their last star block
  const int wrow_beg = m0 + ((int)(threadIdx.x >> 6) / C::WN) * C::TM * 16;
  const int wrow_end = wrow_beg + C::TM * 16;
  // (a wavefront whose rows all lie in the zero frame below the real matrix multiplies nothing)
  const int wk = wrow_beg >= n_rows ? 0 : ((wrow_end + tri_block - 1) / tri_block) * tri_block;
  mainloop<C, KC, RC, ALWAYS_FAST>(G, N, Kt, ldk, N, M, m0, n0, 0, kend, acc, wk < kend ? wk : kend);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / C::WN, wn = wave % C::WN;
  double* red = lds_dyn;  // [WM][BN]; mainloop ended with a barrier
#pragma unroll
  for (int j = 0; j < C::TN; ++j) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += acc[i][j][r] * acc[i][j][r];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lane < 16) red[wm * C::BN + (wn * C::TN + j) * 16 + lane] = s;
  }
  __syncthreads();
  if (threadIdx.x < C::BN) {
    const int c = n0 + threadIdx.x;
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < C::WM; ++w) t += red[w * C::BN + threadIdx.x];
    if (c < M) slab[(size_t)mt * M + c] = t;
  }
}

// tile shapes tried on MI355X: variant id -> (WM, WN, TM, TN, min waves/SIMD)
using QF0 = Cfg<2, 2, 4, 4>;   // 4 waves of 64x64, 2 WG/CU -> 2 waves/SIMD
using QF1 = Cfg<2, 4, 4, 2>;   // 8 waves of 64x32, 2 WG/CU -> 4 waves/SIMD
using QF2 = Cfg<4, 2, 2, 4>;   // 8 waves of 32x64
using QF3 = Cfg<4, 4, 2, 2>;   // 16 waves of 32x32, 1 WG/CU -> 4 waves/SIMD
using QF4 = Cfg<4, 4, 2, 2>;   // same tile, 2 WG/CU -> 8 waves/SIMD (<= 64 VGPRs)
using QF5 = Cfg<8, 2, 2, 4>;   // 256 x 128 tile, 16 waves of 32x64, 1 WG/CU: K* re-read once per 256 G rows

// G [rows >= ntm BM][N] with row stride N, Kt [N][ldk]: `N` is the K extent AND G's row stride.  The lean (unguarded)
// loop runs when every tile is in bounds and every K range a whole number of chunks: N % 16 == 0, the row tiles
// backed by memory (g_rows >= ntm BM) and the candidate tiles too (ldk >= ntn BN; what the columns beyond Mc hold does
// not matter, each output column depends on its own K* column only and the slab is written for c < Mc).
// predict_passes pads its operands so that this always holds.
template <class C, int MINW>
int launch_quadform(ppbo_ctx* ctx, const double* G, int N, int g_rows, int n_rows, const double* Kt, int ldk, int Mc,
                    int mblk, double* slab, hipStream_t s) {
  const size_t lds = C::LDS_DOUBLES * sizeof(double);
  ppbo_lds_limit(ctx, (const void*)quadform_kernel<C, MINW, true>, (int)lds);
  ppbo_lds_limit(ctx, (const void*)quadform_kernel<C, MINW, false>, (int)lds);
  const int ntm = (g_rows + C::BM - 1) / C::BM, ntn = (Mc + C::BN - 1) / C::BN;
  const int grid = ntm * ntn;
  // PPBO_QF_ORDER; default: candidate-tile fastest in chunks of 128 tiles (514; measured best at N >= 2048:
  // profiles/r04_quadform_traffic_vs_order.txt), of 256 tiles up to N = 1024 (1026: the same 268 MB of K* per chunk as 128
  // tiles at N = 2048; round 6, interleaved on one box at C4: 0.899-0.903 against 0.891-0.894 of the MFMA peak)
  const int order = ctx->qf_order >= 0 ? ctx->qf_order : (N <= 1024 ? 1026 : 514);
  const int swz = (grid >= 64 ? (order & 1) : 0) | (order & ~1);
  // every tile in bounds, 16-byte aligned, and every K range a multiple of 16?
  const bool fast = (g_rows % C::BM == 0) && (ldk >= ntn * C::BN) && (N % BK == 0) && (ldk % 2 == 0) &&
                    ((reinterpret_cast<uintptr_t>(G) & 15) == 0) && ((reinterpret_cast<uintptr_t>(Kt) & 15) == 0);
  if (fast) quadform_kernel<C, MINW, true><<<grid, C::NT, lds, s>>>(G, N, Kt, ldk, Mc, mblk, slab, ntm, ntn, swz, n_rows);
  else quadform_kernel<C, MINW, false><<<grid, C::NT, lds, s>>>(G, N, Kt, ldk, Mc, mblk, slab, ntm, ntn, swz, n_rows);
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

// ctx->qf_variant (PPBO_QF_VARIANT); default 4 since round 6: 16 wavefronts of 32x32, two workgroups per CU = 8 waves/SIMD.
// Rounds 1-5 shipped variant 2 (8 wavefronts of 32x64, 4 waves/SIMD: the winner of round 1's sweep, 58.1 against 54.7 TF, taken
// BEFORE the lean main loop); re-measured interleaved on one box in round 6: variant 4 3.74-3.75 ms against 3.78-3.79 at C3 (0.948
// against 0.938 of the MFMA peak), 0.979 against 0.996 ms at C4 (0.921 / 0.906), level at C5 -- with no vector instruction left in
// the loop to pay for, twice the wavefronts hide more of the LDS and barrier latency
int quadform_variant(const ppbo_ctx* ctx) { return ctx->qf_variant; }

int dispatch_quadform(ppbo_ctx* ctx, const double* G, int N, int g_rows, int n_rows, const double* Kt, int ldk, int Mc,
                      int mblk, double* slab, hipStream_t s) {
  switch (quadform_variant(ctx)) {
    case 1: return launch_quadform<QF1, 4>(ctx, G, N, g_rows, n_rows, Kt, ldk, Mc, mblk, slab, s);
    case 2: return launch_quadform<QF2, 4>(ctx, G, N, g_rows, n_rows, Kt, ldk, Mc, mblk, slab, s);
    case 3: return launch_quadform<QF3, 4>(ctx, G, N, g_rows, n_rows, Kt, ldk, Mc, mblk, slab, s);
    case 4: return launch_quadform<QF4, 8>(ctx, G, N, g_rows, n_rows, Kt, ldk, Mc, mblk, slab, s);
    case 5: return launch_quadform<QF5, 4>(ctx, G, N, g_rows, n_rows, Kt, ldk, Mc, mblk, slab, s);
    default: return launch_quadform<QF0, 2>(ctx, G, N, g_rows, n_rows, Kt, ldk, Mc, mblk, slab, s);
  }
}

// Gp [rows_p][cols_p] = G [N][N] framed with zeros: the operand of the quadratic form when N is not a multiple of its
// row tile / chunk depth (every tile of the launch is then in bounds and takes the lean loop; 3 N^2 x 8 bytes moved
// once per call -- 20 us at N = 2080 against the ~1 ms the guarded loop costs per 65536 candidates)
__global__ __launch_bounds__(256) void pad_square_kernel(const double* __restrict__ G, int N, double* __restrict__ Gp,
                                                         int rows_p, int cols_p) {
  const int i = blockIdx.y;
  const int j = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (j >= cols_p) return;
  double2 v = {0.0, 0.0};
  if (i < N) {
    if (j < N) v.x = G[(size_t)i * N + j];
    if (j + 1 < N) v.y = G[(size_t)i * N + j + 1];
  }
  *reinterpret_cast<double2*>(Gp + (size_t)i * cols_p + j) = v;
}

// G of the model as the block-triangular products want it: rows up to a whole 128-row tile, columns up to the chunk
// depth, framed with zeros -- the model's own array when it already has that shape.  *Nk_out = its row stride = K extent.
// worth = false (few columns on the other side of the product: the copy would cost more than the guarded loop does):
// the model's own array, whatever its shape.
const double* padded_G(ppbo_ctx* ctx, const ppbo_model* model, int row_tile, bool worth, int* Nk_out, int* g_rows_out,
                       hipStream_t s) {
  const int N = model->N;
  const int Nk = (N + BK - 1) & ~(BK - 1);
  const int g_rows = ((N + row_tile - 1) / row_tile) * row_tile;
  *Nk_out = N; *g_rows_out = N;
  if (!worth || (g_rows == N && Nk == N && (reinterpret_cast<uintptr_t>(model->d_G) & 15) == 0)) return model->d_G;
  *Nk_out = Nk; *g_rows_out = g_rows;
  double* Gp = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_GPAD, (size_t)g_rows * Nk * sizeof(double));
  if (!Gp) return nullptr;
  pad_square_kernel<<<dim3((Nk / 2 + 255) / 256, g_rows), 256, 0, s>>>(model->d_G, N, Gp, g_rows, Nk);
  return Gp;
}

// Z = Lambda K*  (star-graph rows), Kt/Z are [N, M] with row stride ld
__global__ void lam_apply_kernel(const double* __restrict__ Kt, int ld, int N, int M, int mblk,
                                 const double* __restrict__ lam_diag, const double* __restrict__ lam_off,
                                 double* __restrict__ Z) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int q = blockIdx.y;
  if (c >= M) return;
  const int i = q * mblk;
  const double ko = Kt[(size_t)i * ld + c];
  double zo = lam_diag[i] * ko;
  for (int r = 1; r < mblk && i + r < N; ++r) {
    const int j = i + r;
    const double kj = Kt[(size_t)j * ld + c];
    const double lo = lam_off[j];
    Z[(size_t)j * ld + c] = lam_diag[j] * kj + lo * ko;
    zo += lo * kj;
  }
  Z[(size_t)i * ld + c] = zo;
}

// prior block of one line: cov_b[g][h] = (1-s) k(x_g, x_h), diagonal (1-s) sf2 + s sf2  (gp_model.py:447)
template <int KID>
__global__ __launch_bounds__(256) void line_prior_kernel(const double* __restrict__ grid, int G, int D,
                                                         KernParams p, double shrink, double* __restrict__ cov) {
  const double* xg = grid + (size_t)blockIdx.x * G * D;
  double* c = cov + (size_t)blockIdx.x * G * G;
  if (KID == PPBO_KERNEL_CAMPHOR) {
    // feature form of the camphor exponent (see kstar_kernel): (cos, sin)(2 pi x_k) of the line's points once, then ten
    // FMAs per pair; the same chain of the same products for (g, h) and (h, g)
    __shared__ double ph[128 * 11];        // G <= 128 (line_acq_impl)
    for (int e = threadIdx.x; e < G * 6; e += blockDim.x) {
      const int g = e / 6, k = e - g * 6;
      if (k < 5) {
        double sn, cs;
        sincospi(2.0 * xg[g * D + (k < 2 ? k : k + 1)], &sn, &cs);
        ph[g * 11 + k] = cs;
        ph[g * 11 + 5 + k] = sn;
      } else {
        ph[g * 11 + 10] = xg[g * D + 2];
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < G * G; e += blockDim.x) {
      const int g = e / G, h = e - g * G;
      double dot = 0.0;
#pragma unroll
      for (int f = 0; f < 10; ++f) dot = fma(ph[g * 11 + f], ph[h * 11 + f], dot);
      const double dd = ph[g * 11 + 10] - ph[h * 11 + 10];
      const double s = fma(p.c1 * dd, dd, 0.5 * p.c0 * (5.0 - dot));
      c[e] = (g == h) ? ((1.0 - shrink) * p.sf2 + shrink * p.sf2) : (1.0 - shrink) * kern_finish<KID>(s, p);
    }
    return;
  }
  for (int e = threadIdx.x; e < G * G; e += blockDim.x) {
    const int g = e / G, h = e - g * G;
    double s = 0.0;
    for (int d = 0; d < D; ++d) s += kern_term<KID>(xg[g * D + d] - xg[h * D + d], d, p);
    c[e] = (g == h) ? ((1.0 - shrink) * p.sf2 + shrink * p.sf2) : (1.0 - shrink) * kern_finish<KID>(s, p);
  }
}

// The data term of a line's G x G predictive covariance, C_b = K*_b' Lambda K*_b + Y_b' Y_b (Y = G K*; gp_model.py:447-450
// through the Woodbury form of DESIGN 2.4), for one line per workgroup and one slice of the N rows per blockIdx.y.
// Replaces lam_apply_kernel + two batched 128 x 128-tile GEMMs whose 70 x 70 outputs used 30 % of every tile
// (1.61 ms of the 4.6 ms of 512 lines x 70 points at N = 2048): the rows are streamed once, 32 at a time, through
// LDS (next chunk in registers while the current one is multiplied), Lambda is applied while a chunk is staged --
//   K*' Lambda K* = sum_j k_j b_j' + (its transpose) / 2,  b_j = lam_jj k_j + 2 lam_off,j k_obs(j)  (pseudo rows),
//                                                          b_obs = lam_obs k_obs,
// the ONE-SIDED form: a row's b needs only its own row and its star's observation row; the consumer symmetrises --
// and both products accumulate in the same NT16 x NT16 grid of 16 x 16 MFMA tiles.  out[blockIdx.y][b][G*G].
// tile t of the NT16 x NT16 grid in the order the wavefronts share them out: SYM walks the lower triangle (mt >= nt)
// only, row by row
template <int NT16, bool SYM>
struct LcTiles {
  static constexpr int N = SYM ? NT16 * (NT16 + 1) / 2 : NT16 * NT16;
  static constexpr int mt(int t) {
    if (!SYM) return t / NT16;
    int m = 0;
    while ((m + 1) * (m + 2) / 2 <= t) ++m;
    return m;
  }
  static constexpr int nt(int t) { return SYM ? t - mt(t) * (mt(t) + 1) / 2 : t % NT16; }
};

// the MFMAs of one k-step over the tiles W, W + 4, ... (compile-time recursion: the fragment indices are constants)
template <int NT16, bool SYM, int W, int I>
__device__ __forceinline__ void lc_mfma_k(const double (&fk)[NT16], const double (&fb)[NT16],
                                          double4_t (&acc)[(LcTiles<NT16, SYM>::N + 3) / 4]) {
  using T = LcTiles<NT16, SYM>;
  constexpr int tile = W + 4 * I;
  if constexpr (tile < T::N) {
    acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(fk[T::mt(tile)], fb[T::nt(tile)], acc[I], 0, 0, 0);
    lc_mfma_k<NT16, SYM, W, I + 1>(fk, fb, acc);
  }
}
template <int NT16, bool SYM, int W, int I>
__device__ __forceinline__ void lc_mfma_y(const double (&fy)[NT16], double4_t (&acc)[(LcTiles<NT16, SYM>::N + 3) / 4]) {
  using T = LcTiles<NT16, SYM>;
  constexpr int tile = W + 4 * I;
  if constexpr (tile < T::N) {
    acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(fy[T::mt(tile)], fy[T::nt(tile)], acc[I], 0, 0, 0);
    lc_mfma_y<NT16, SYM, W, I + 1>(fy, acc);
  }
}

// one 32-row chunk of line_cov_kernel for wavefront W: its tiles are W, W + 4, ... of the tile list
template <int NT16, bool SYM, int W, int LDC>
__device__ __forceinline__ void line_cov_mm(const double* Ks, const double* Bs, const double* Ys, int lk, int lr,
                                            double4_t (&accK)[(LcTiles<NT16, SYM>::N + 3) / 4],
                                            double4_t (&accY)[(LcTiles<NT16, SYM>::N + 3) / 4], int ksteps) {
#pragma unroll 1
  for (int kk = 0; kk < ksteps; ++kk) {     // not unrolled: 3 NT16 fragments per step are live, not 24 NT16
    const int ko = (4 * kk + lk) * LDC + lr;
    // this k-step's fragments of all NT16 column blocks, once: every tile of the wavefront draws on them
    double fk[NT16], fb[NT16], fy[NT16];
#pragma unroll
    for (int c = 0; c < NT16; ++c) { fk[c] = Ks[ko + 16 * c]; fb[c] = Bs[ko + 16 * c]; fy[c] = Ys[ko + 16 * c]; }
    lc_mfma_k<NT16, SYM, W, 0>(fk, fb, accK);
    lc_mfma_y<NT16, SYM, W, 0>(fy, accY);
  }
}

constexpr int LC_MAXROWS = 512;      // rows of one (line, slice) workgroup: their Lambda entries sit in LDS

// SYM (the host picks it for stars of up to 32 rows, with row slices that hold whole stars): a chunk is `cs` rows = as many
// WHOLE stars as fit 32 rows (one at m = 31 and at the reference's default m = 25, two at m = 15, ...; the rest of the 32
// staged rows are zeros and k-steps beyond them are not run), so Lambda K* is formed EXACTLY -- every observation row gets
// lam_obs k_obs + sum_j lam_off,j k_j from the staged chunk -- both products are symmetric and only the tiles of the
// lower triangle (15 of 25 at G = 70) are computed.  Not SYM: cs = 32 rows of anything, the one-sided form.
template <int NT16, bool SYM>
__global__ __launch_bounds__(256, (NT16 <= 5) ? 2 : 1) void line_cov_kernel(const double* __restrict__ Kt, const double* __restrict__ Y, int ld,
                                                          int N, int G, int mblk, const double* __restrict__ lam_diag,
                                                          const double* __restrict__ lam_off, int rows_per_split,
                                                          double* __restrict__ out, long long out_stride, int cs) {
  using TL = LcTiles<NT16, SYM>;
  constexpr int CH = 32, LDC = 16 * NT16 + 8, NTILE = TL::N, TPW = (NTILE + 3) / 4;
  constexpr int NQ = (CH * 16 * NT16 + 255) / 256;
  __shared__ __attribute__((aligned(16))) double Ks[CH * LDC], Bs[CH * LDC], Ys[CH * LDC];
  __shared__ double s_ld[LC_MAXROWS], s_lo[LC_MAXROWS];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 15, lk = lane >> 4;
  const int col0 = blockIdx.x * G;
  const int r_beg = blockIdx.y * rows_per_split;
  int r_end = r_beg + rows_per_split;
  if (r_end > N) r_end = N;
  // two accumulator sets (K*'B and Y'Y): consecutive MFMAs never share an accumulator (a dependent fp64 MFMA issues
  // every ~138 cycles, an independent one every 64)
  double4_t accK[TPW], accY[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) { accK[i] = double4_t{0.0, 0.0, 0.0, 0.0}; accY[i] = double4_t{0.0, 0.0, 0.0, 0.0}; }
  for (int e = t; e < CH * LDC; e += 256) { Ks[e] = 0.0; Bs[e] = 0.0; Ys[e] = 0.0; }    // the padding columns stay zero
  for (int i = t; i < r_end - r_beg; i += 256) { s_ld[i] = lam_diag[r_beg + i]; s_lo[i] = lam_off[r_beg + i]; }
  // element q of this thread inside a chunk: (row rq, column gq), fixed for the whole walk (no division per chunk)
  int rq[NQ], gq[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int e = t + 256 * q;
    rq[q] = e / G;
    gq[q] = e - rq[q] * G;
  }
  // A chunk is FETCHED by unconditional loads only (addresses clamped into the matrix, values masked when they are
  // staged): the first version computed lam * k inside per-element guards, which made every one of the NQ elements its
  // own memory round trip -- 9 dependent round trips per 32-row chunk, 1.0 ms for a pass that is 0.35 ms of MFMA work.
  double rk[NQ], ro[NQ], ry[NQ];
  auto fetch = [&](int row0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      int j = row0 + rq[q];
      j = j < N ? j : N - 1;
      const int q0 = (j / mblk) * mblk;
      const size_t c = (size_t)col0 + gq[q];
      rk[q] = Kt[(size_t)j * ld + c];
      ro[q] = Kt[(size_t)q0 * ld + c];           // the star's observation row (L2: the same row for the whole star)
      ry[q] = Y[(size_t)j * ld + c];
    }
  };
  fetch(r_beg);
  const int ksteps = (cs + 3) >> 2;
  for (int row0 = r_beg; row0 < r_end; row0 += cs) {
    __syncthreads();                         // the previous chunk's operands are done with (and, first time, s_ld / s_lo are there)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int r = rq[q], j = row0 + r;
      if (r < CH) {
        const bool on = r < cs && j < r_end;
        const int jl = on ? j - r_beg : 0;
        const double kv = on ? rk[q] : 0.0;
        // Lambda row: lam_jj k_j + c lam_off,j k_obs (lam_off is 0 on observation rows); c = 2: the one-sided form,
        // c = 1 (SYM): the pseudo-observation rows of the exact product, the observation row follows below
        Ks[r * LDC + gq[q]] = kv;
        Bs[r * LDC + gq[q]] = on ? s_ld[jl] * kv + (SYM ? 1.0 : 2.0) * s_lo[jl] * ro[q] : 0.0;
        Ys[r * LDC + gq[q]] = on ? ry[q] : 0.0;
      }
    }
    __syncthreads();
    if (SYM) {
      // the chunk is whole stars, each led by its observation row o: B[o][g] += sum_r lam_off,o+r K[o+r][g] (m terms per
      // column; rows past the slice's end were staged as zeros and are not touched)
      if (t < 16 * NT16) {
        const int jl0 = row0 - r_beg;
        for (int o = 0; o < cs && row0 + o < r_end; o += mblk) {
          double sg = 0.0;
#pragma unroll 8
          for (int r = 1; r < mblk; ++r) sg += s_lo[jl0 + o + r] * Ks[(o + r) * LDC + t];
          Bs[o * LDC + t] += sg;
        }
      }
      __syncthreads();
    }
    if (row0 + cs < r_end) fetch(row0 + cs);
    // the tile list of a wavefront is a compile-time list (W + 4 i): a run-time tile index would turn the fragment
    // arrays into dynamically indexed registers (measured: 1.84 ms instead of 0.99)
    switch (wave) {
      case 0: line_cov_mm<NT16, SYM, 0, LDC>(Ks, Bs, Ys, lk, lr, accK, accY, ksteps); break;
      case 1: line_cov_mm<NT16, SYM, 1, LDC>(Ks, Bs, Ys, lk, lr, accK, accY, ksteps); break;
      case 2: line_cov_mm<NT16, SYM, 2, LDC>(Ks, Bs, Ys, lk, lr, accK, accY, ksteps); break;
      default: line_cov_mm<NT16, SYM, 3, LDC>(Ks, Bs, Ys, lk, lr, accK, accY, ksteps); break;
    }
  }
  double* o = out + (size_t)blockIdx.y * out_stride + (size_t)blockIdx.x * G * G;
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int tile = wave + 4 * i;
    if (tile >= NTILE) continue;
    const int mt = TL::mt(tile), nt = TL::nt(tile);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int g = 16 * mt + lk + 4 * r, h = 16 * nt + lr;
      if (g < G && h < G) o[g * G + h] = accK[i][r] + accY[i][r];
    }
  }
}

// Monte-Carlo part of EI / varmax on a line (acquisition.py:72-81, :170-178): f = mu + L z for S draws, max over the
// line, statistics of the maxima.  grid = (lines, draw splits); one workgroup = one line x one slice of the draws.
//   1. Cholesky of the G x G posterior covariance in LDS (every split of a line repeats it: 70 short steps, and the
//      splits run side by side);
//   2. F = Z L^T on the fp64 matrix cores: a wavefront takes 16 draws at a time (A fragments straight from the
//      shared z[S][G], which stays in L2), one 16 x 16 tile of F per 16 grid points with mu as the accumulator's
//      initial value and the contraction cut at the triangle's edge; the maximum over the line is an elementwise
//      max over the tiles and one 16-lane reduction;
//   3. partial sums (sum max(f - mustar, 0), sum f, sum f^2) per (line, split); mc_finish_kernel combines them.
// Round 1-2's form (lane = draw, a scalar loop over the triangle with z re-read from memory) took 1.3 ms for 1200
// draws of a 70-point line whatever the batch size; this one ~0.1 ms.
constexpr int MC_MAXG16 = 128;
__global__ __launch_bounds__(256) void line_mc_kernel(const double* __restrict__ mu, const double* __restrict__ cov,
                                                      int G, const double* __restrict__ z, int S, double mustar,
                                                      double jitter, int draws_per_split, double* __restrict__ part,
                                                      const double* __restrict__ cov_parts = nullptr, int n_parts = 0,
                                                      long long part_stride = 0, int parts_lower_only = 0) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int G16 = (G + 15) & ~15, ld = G16 + 2;
  double* Lm = sm;                           // [G16][ld], rows / columns >= G zero
  double* mus = sm + (size_t)G16 * ld;       // [G16], -inf beyond G: a padded grid point never is the maximum
  double* red = mus + G16;                   // [3][4]
  const double* c = cov + (size_t)blockIdx.x * G * G;
  const double* m = mu + (size_t)blockIdx.x * G;
  // cov_parts: the data term K*' Lambda K* + Y'Y as line_cov_kernel leaves it -- n_parts row-split slabs per line, the
  // Lambda part in its one-sided form -- added here in slab order
  const double* cp = cov_parts ? cov_parts + (size_t)blockIdx.x * G * G : nullptr;
  for (int e = threadIdx.x; e < G16 * ld; e += blockDim.x) {
    const int g = e / ld, h = e - g * ld;
    double v = 0.0;
    if (g < G && h < G) {
      double a = c[g * G + h], b = c[h * G + g];
      // parts_lower_only: the slabs are exactly symmetric and hold the 16 x 16 tiles of the lower triangle only
      const bool up = parts_lower_only && (g >> 4) < (h >> 4);
      const int ia = up ? h * G + g : g * G + h, ib = (parts_lower_only && !up && (g >> 4) > (h >> 4)) ? g * G + h : h * G + g;
      for (int k = 0; k < n_parts; ++k) { a += cp[(size_t)k * part_stride + ia]; b += cp[(size_t)k * part_stride + ib]; }
      // symmetrise (the contributions are symmetric only up to rounding; line_cov_kernel's Lambda term only after this)
      v = 0.5 * (a + b) + ((g == h) ? jitter : 0.0);
    }
    Lm[e] = v;
  }
  for (int g = threadIdx.x; g < G16; g += blockDim.x) mus[g] = (g < G) ? m[g] : -INFINITY;
  __syncthreads();
  // right-looking Cholesky in LDS, two barriers per column: scale the column below the diagonal, then the 16 x 16
  // thread grid sweeps the trailing lower triangle (no integer division per element)
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  for (int j = 0; j < G; ++j) {
    const double d = Lm[j * ld + j];
    const double piv = (d > 0.0) ? sqrt(d) : 0.0;     // semi-definite: drop the direction
    for (int i = j + 1 + threadIdx.x; i < G; i += blockDim.x) Lm[i * ld + j] = (piv > 0.0) ? Lm[i * ld + j] / piv : 0.0;
    __syncthreads();
    if (threadIdx.x == 0) Lm[j * ld + j] = piv;        // nobody reads the diagonal entry below
    for (int a = j + 1 + ty; a < G; a += 16) {
      const double la = Lm[a * ld + j];
      for (int b = j + 1 + tx; b <= a; b += 16) Lm[a * ld + b] -= la * Lm[b * ld + j];
    }
    __syncthreads();
  }
  // the strict upper triangle still holds the symmetric input: the contraction below stops at the diagonal TILE, so
  // inside the diagonal tiles it must read zeros there
  for (int e = threadIdx.x; e < G * G; e += blockDim.x) {
    const int g = e / G, h = e - g * G;
    if (h > g) Lm[g * ld + h] = 0.0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
  const int s_beg = blockIdx.y * draws_per_split;
  int s_end = s_beg + draws_per_split;
  if (s_end > S) s_end = S;
  const int nj = G16 >> 4;
  double sum_ei = 0.0, sum_f = 0.0, sum_f2 = 0.0;
  for (int s0 = s_beg + 16 * wave; s0 < s_end; s0 += 64) {
    // A fragments of the 16 draws s0 .. s0+15: lane (lr, lk) holds z[s0 + lr][4 kk + lk]
    double az[MC_MAXG16 / 4];
    const int sr = s0 + lr;
    const double* zr = z + (size_t)(sr < s_end ? sr : s_beg) * G;
#pragma unroll
    for (int kk = 0; kk < MC_MAXG16 / 4; ++kk) {
      const int k = 4 * kk + lk;
      az[kk] = (kk < (G16 >> 2) && k < G) ? zr[k] : 0.0;
    }
    double fmx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int j = 0; j < MC_MAXG16 / 16; ++j) {                        // unrolled: the triangle's edge is a constant per j
      if (j >= nj) break;
      const double mg = mus[16 * j + lr];
      double4_t acc = double4_t{mg, mg, mg, mg};
      const double* lrow = Lm + (size_t)(16 * j + lr) * ld + lk;      // B operand: L^T[k][g] = L[g][k]
#pragma unroll
      for (int kk = 0; kk < 4 * (j + 1); ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(az[kk], lrow[4 * kk], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) fmx[r] = fmax(fmx[r], acc[r]);
    }
    // acc[r] belongs to draw s0 + lk + 4 r and grid point 16 j + lr: maximum over the 16 lanes that share lk
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double v = fmx[r];
      v = fmax(v, __shfl_xor(v, 1, 64));
      v = fmax(v, __shfl_xor(v, 2, 64));
      v = fmax(v, __shfl_xor(v, 4, 64));
      v = fmax(v, __shfl_xor(v, 8, 64));
      if (lr == 0 && s0 + lk + 4 * r < s_end) {
        sum_ei += fmax(v - mustar, 0.0);
        sum_f += v;
        sum_f2 += v * v;
      }
    }
  }
  sum_ei = wave_sum(sum_ei);
  sum_f = wave_sum(sum_f);
  sum_f2 = wave_sum(sum_f2);
  if (lane == 0) { red[wave] = sum_ei; red[4 + wave] = sum_f; red[8 + wave] = sum_f2; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const double* r = red + 4 * threadIdx.x;
    part[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 3 + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
  }
}

// EI and varmax of every line from the per-split partial sums (fixed order: deterministic)
__global__ __launch_bounds__(256) void mc_finish_kernel(const double* __restrict__ part, int B, int nsplit, int S,
                                                        double* __restrict__ ei, double* __restrict__ varmax) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double e = 0.0, f1 = 0.0, f2 = 0.0;
  for (int k = 0; k < nsplit; ++k) {
    const double* p = part + ((size_t)b * nsplit + k) * 3;
    e += p[0]; f1 += p[1]; f2 += p[2];
  }
  if (ei) ei[b] = e / S;
  const double mean = f1 / S;
  if (varmax) varmax[b] = f2 / S - mean * mean;
}

// Standard normal draws for the Monte-Carlo acquisitions, generated where they are used (4800 x 70 draws cost the host
// 3 ms -- as much as the whole search they feed).  Counter-based: Philox-4x32-10 (Salmon et al. 2011) keyed by the
// seed, counter = output pair index; two 53-bit uniforms per counter -> one Box-Muller pair.  The stream is a pure
// function of (seed, index): reproducible, independent of launch geometry.
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
  const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
  const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__global__ __launch_bounds__(256) void randn_kernel(unsigned long long seed, double* __restrict__ out, long long n) {
  const long long pair = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * pair >= n) return;
  unsigned c[4] = {(unsigned)pair, (unsigned)((unsigned long long)pair >> 32), 0u, 0u};
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const unsigned long long a = ((unsigned long long)c[1] << 32) | c[0], b = ((unsigned long long)c[3] << 32) | c[2];
  const double u1 = ((double)(a >> 11) + 0.5) * 0x1.0p-53;       // (0, 1): the logarithm is finite
  const double u2 = ((double)(b >> 11) + 0.5) * 0x1.0p-53;
  const double rad = sqrt(-2.0 * log(u1));
  double sn, cs;
  sincospi(2.0 * u2, &sn, &cs);
  out[2 * pair] = rad * cs;
  if (2 * pair + 1 < n) out[2 * pair + 1] = rad * sn;
}

template <int KID>
int launch_kstar(const ppbo_model* m, const KernParams& p, const double* d_Xc, int M, double* Kt, int ldk,
                 double* mu_part, double* t_part, int q_per_split, int n_split, bool with_lam, hipStream_t s) {
  dim3 grid((M + KS_THREADS * KS_CPT - 1) / (KS_THREADS * KS_CPT), n_split);
  const int mblk = m->m + 1, n_q = (m->N + mblk - 1) / mblk;
  const double* ld = with_lam ? m->d_lam_diag : nullptr;
  const double* lo = with_lam ? m->d_lam_off : nullptr;
#define KS_LAUNCH32(DP)                                                                                              \
  kstar_kernel<KID, DP, true><<<grid, KS_THREADS, 0, s>>>(m->d_X, m->N, m->D, p, m->d_alpha, ld, lo, mblk, d_Xc, M, \
                                                          Kt, ldk, mu_part, with_lam ? t_part : nullptr,            \
                                                          q_per_split, n_q)
  if (m->kstar_fp32) {       // the fp32-tolerance report: the BASELINE shapes get their own bucket, the rest a generic one
    if (KID == PPBO_KERNEL_CAMPHOR || m->D <= 6) KS_LAUNCH32(6);
    else if (m->D <= 20) KS_LAUNCH32(20);
    else KS_LAUNCH32(64);
    return 0;
  }
#undef KS_LAUNCH32
#define KS_LAUNCH(DP)                                                                                          \
  kstar_kernel<KID, DP><<<grid, KS_THREADS, 0, s>>>(m->d_X, m->N, m->D, p, m->d_alpha, ld, lo, mblk, d_Xc, M, \
                                                    Kt, ldk, mu_part, with_lam ? t_part : nullptr,            \
                                                    q_per_split, n_q)
  if (KID == PPBO_KERNEL_CAMPHOR) KS_LAUNCH(12);      // the feature form: 12 staged values per row
  else if (m->D <= 4) KS_LAUNCH(4);
  else if (m->D <= 6) KS_LAUNCH(6);
  else if (m->D <= 8) KS_LAUNCH(8);
  else if (m->D <= 10) KS_LAUNCH(10);
  else if (m->D <= 12) KS_LAUNCH(12);
  else if (m->D <= 16) KS_LAUNCH(16);
  else if (m->D <= 20) KS_LAUNCH(20);
  else if (m->D <= 24) KS_LAUNCH(24);
  else if (m->D <= 32) KS_LAUNCH(32);
  else if (m->D <= 48) KS_LAUNCH(48);
  else KS_LAUNCH(64);
#undef KS_LAUNCH
  return 0;
}

int dispatch_kstar(const ppbo_model* m, const double* d_Xc, int M, double* Kt, int ldk, double* mu_part,
                   double* t_part, int q_per_split, int n_split, bool with_lam, hipStream_t s) {
  const KernParams p = make_kern_params(m->kernel_id, m->theta);
  switch (m->kernel_id) {
    case PPBO_KERNEL_SE: return launch_kstar<PPBO_KERNEL_SE>(m, p, d_Xc, M, Kt, ldk, mu_part, t_part, q_per_split, n_split, with_lam, s);
    case PPBO_KERNEL_RQ: return launch_kstar<PPBO_KERNEL_RQ>(m, p, d_Xc, M, Kt, ldk, mu_part, t_part, q_per_split, n_split, with_lam, s);
    default: return launch_kstar<PPBO_KERNEL_CAMPHOR>(m, p, d_Xc, M, Kt, ldk, mu_part, t_part, q_per_split, n_split, with_lam, s);
  }
}

int check_model(ppbo_ctx* ctx, const ppbo_model* m) {
  PPBO_REQUIRE(ctx, m != nullptr, "model");
  PPBO_REQUIRE(ctx, m->d_X && m->d_alpha, "model X/alpha");
  PPBO_REQUIRE(ctx, m->N > 0 && m->D > 0 && m->D <= 64 && m->m >= 1, "model sizes (D<=64)");
  PPBO_REQUIRE(ctx, m->N % (m->m + 1) == 0, "N must be n_q*(m+1) (feedback_processing.py:110-130)");
  PPBO_REQUIRE(ctx, m->kernel_id >= 0 && m->kernel_id <= 2, "kernel_id");
  PPBO_REQUIRE(ctx, m->kernel_id != PPBO_KERNEL_CAMPHOR || m->D == 6, "camphor kernel needs D == 6");
  return 0;
}

int pick_split(int M, int n_q) {
  // enough (block, split) pairs to give every SIMD several wavefronts
  const int blocks = (M + KS_THREADS * KS_CPT - 1) / (KS_THREADS * KS_CPT);
  // ~1024 workgroups: from 16384 candidates on a workgroup then covers 64+ rows (round 5 sweep at N = 2048, K* launch:
  // 16384 candidates 0.112 -> 0.099 ms against a target of 2048; 8192 and 65536 candidates unchanged at 0.064 / 0.305)
  int want = (1024 + blocks - 1) / blocks;
  if (want > n_q) want = n_q;
  if (want > 64) want = 64;
  if (want < 1) want = 1;
  return want;
}

}  // namespace

extern "C" {

// the scoring passes of ppbo_predict / ppbo_predict_record: per 65536-candidate chunk kstar -> quadform -> score ->
// one-workgroup argmax.  Leaves the per-chunk bests in *chunk_best_out (device); with d_record and ONE chunk the
// argmax launch writes the record (and raises the publication flag) itself.
static int predict_passes(ppbo_ctx* ctx, const ppbo_model* model, const double* d_Xc, int64_t M, int score_kind,
                          double mustar, double* d_mu, double* d_var, double* d_score, bool want_best,
                          double* d_record, int64_t record_offset, Best** chunk_best_out, int* n_chunks_out,
                          hipStream_t s, unsigned long long* publish = nullptr, unsigned long long epoch = 0) {
  if (int rc = check_model(ctx, model)) return rc;
  PPBO_REQUIRE(ctx, d_Xc != nullptr && M > 0, "candidates");
  PPBO_REQUIRE(ctx, score_kind >= 0 && score_kind <= 2, "score_kind");
  const bool want_var = (model->d_G != nullptr);
  PPBO_REQUIRE(ctx, want_var || (d_var == nullptr && score_kind == PPBO_SCORE_MEAN),
               "variance / EI scores need model->d_G");
  PPBO_REQUIRE(ctx, !want_var || (model->d_lam_diag && model->d_lam_off), "model Lambda");
  const int N = model->N, mblk = model->m + 1, n_q = N / mblk;
  if (want_var && ppbo_fused_eligible(ctx, model)) {
    // Models of up to 1024 rows: ONE launch forms K* in LDS, contracts it with G on the matrix cores and scores
    // (fused.hip) -- no K* in HBM, no slab pass, no candidate chunks -- then the one-workgroup argmax.  The choice
    // depends on the model only: a shard of a sharded search scores a candidate exactly as the unsharded search does.
    int ldgt = N;
    const double* Gt = ppbo_fused_transposed_G(ctx, model, &ldgt, s);
    if (!Gt) return (int)hipErrorOutOfMemory;
    PPBO_LAUNCH_CHECK(ctx);
    const long long nblk = (M + 31) / 32;
    PPBO_REQUIRE(ctx, nblk < ((long long)1 << 31), "candidate count");
    Best* bests = (Best*)ppbo_workspace(ctx, ppbo_ctx::WS_SMALL, (size_t)(nblk + 1) * sizeof(Best));
    if (!bests) return (int)hipErrorOutOfMemory;
    Best* chunk_best = bests + nblk;
    {
      PpboProfScope pf(ctx, ppbo_ctx::PF_FUSED, s);
      if (int rc = ppbo_fused_score(ctx, model, Gt, ldgt, d_Xc, (long long)M, score_kind, mustar, 0, d_mu, d_var, d_score,
                                    want_best ? bests : nullptr, s))
        return rc;
    }
    if (want_best) {
      PpboProfScope pfs(ctx, ppbo_ctx::PF_SCORE, s);
      argmax_final_kernel<<<1, 256, 0, s>>>(bests, (int)nblk, chunk_best, d_record, (long long)record_offset, publish, epoch);
      PPBO_LAUNCH_CHECK(ctx);
    }
    *chunk_best_out = chunk_best;
    *n_chunks_out = 1;
    return 0;
  }
  const int64_t chunk_cap = 65536;
  const int64_t n_chunks = (M + chunk_cap - 1) / chunk_cap;
  const int qf_bm = (quadform_variant(ctx) == 5) ? 256 : 128;
  const int ntm = (N + qf_bm - 1) / qf_bm;   // slabs = row tiles of the shipped quadform shape
  // The quadratic form's operands are PADDED so that every tile of its launch takes the unguarded main loop whatever
  // N, the star size and the candidate count are (the reference's default m = 25 gives N = 26 n_q: the guarded loop
  // cost 23 % at N = 2080, profiles/r05_ragged_shapes.txt): K extent Nk = N rounded up to the chunk depth (K* rows
  // [N, Nk) zeroed), G's rows up to a whole row tile (copied into a zero frame when N itself is not one), K*'s row
  // stride up to a whole candidate tile (the surplus columns feed only their own, unread, outputs).
  int Nk = N, g_rows = N;
  const double* Gq = nullptr;
  if (want_var) {
    Gq = padded_G(ctx, model, qf_bm, M >= 2048, &Nk, &g_rows, s);
    if (!Gq) return (int)hipErrorOutOfMemory;
    PPBO_LAUNCH_CHECK(ctx);
  }

  // workspaces sized for the largest chunk
  const int Mc_max = (int)(M < chunk_cap ? M : chunk_cap);
  const int ldk = (Mc_max + 127) & ~127;
  const int n_split = pick_split(Mc_max, n_q);
  const int q_per_split = (n_q + n_split - 1) / n_split;
  const int n_split_eff = (n_q + q_per_split - 1) / q_per_split;
  double* Kt = nullptr;
  if (want_var) {
    Kt = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_KSTAR, (size_t)Nk * ldk * sizeof(double));
    if (!Kt) return (int)hipErrorOutOfMemory;
    if (Nk != N) PPBO_HIP_CHECK(ctx, hipMemsetAsync(Kt + (size_t)N * ldk, 0, (size_t)(Nk - N) * ldk * sizeof(double), s));
  }
  const size_t part_doubles = (size_t)(2 * n_split_eff + ntm) * Mc_max;
  double* part = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_PART, part_doubles * sizeof(double));
  if (!part) return (int)hipErrorOutOfMemory;
  const int sblocks_max = score_blocks(Mc_max);
  Best* bests = (Best*)ppbo_workspace(ctx, ppbo_ctx::WS_SMALL, (size_t)(sblocks_max + n_chunks) * sizeof(Best));
  if (!bests) return (int)hipErrorOutOfMemory;
  Best* chunk_best = bests + sblocks_max;
  for (int64_t ch = 0; ch < n_chunks; ++ch) {
    const int64_t c_beg = ch * chunk_cap;
    const int Mc = (int)((M - c_beg) < chunk_cap ? (M - c_beg) : chunk_cap);
    const double* xc = d_Xc + (size_t)c_beg * model->D;
    double* mu_part = part;
    double* t_part = part + (size_t)n_split_eff * Mc;
    double* slab = part + (size_t)2 * n_split_eff * Mc;
    {
      PpboProfScope pf(ctx, ppbo_ctx::PF_KSTAR, s);
      dispatch_kstar(model, xc, Mc, Kt, ldk, mu_part, t_part, q_per_split, n_split_eff, want_var, s);
    }
    PPBO_LAUNCH_CHECK(ctx);
    if (want_var) {
      PpboProfScope pf(ctx, ppbo_ctx::PF_QUADFORM, s);
      if (int rc = dispatch_quadform(ctx, Gq, Nk, g_rows, N, Kt, ldk, Mc, mblk, slab, s)) return rc;
    }
    const int sblocks = score_blocks(Mc);
    PpboProfScope pfs(ctx, ppbo_ctx::PF_SCORE, s);
    const bool one = (n_chunks == 1);
    score_kernel<<<sblocks, SC_THREADS, 0, s>>>(mu_part, n_split_eff, t_part, want_var ? slab : nullptr, ntm, Mc,
                                                model->theta[2] * model->theta[2], score_kind, mustar,
                                                (long long)c_beg, d_mu ? d_mu + c_beg : nullptr,
                                                d_var ? d_var + c_beg : nullptr, d_score ? d_score + c_beg : nullptr,
                                                want_best ? bests : nullptr);
    PPBO_LAUNCH_CHECK(ctx);
    if (want_best) {
      argmax_final_kernel<<<1, 256, 0, s>>>(bests, sblocks, chunk_best + ch, one ? d_record : nullptr,
                                            (long long)record_offset, one ? publish : nullptr, epoch);
      PPBO_LAUNCH_CHECK(ctx);
    }
  }
  if (d_record && n_chunks > 1) {
    best_record_kernel<<<1, 64, 0, s>>>(chunk_best, (int)n_chunks, (long long)record_offset, d_record, publish, epoch);
    PPBO_LAUNCH_CHECK(ctx);
  }
  *chunk_best_out = chunk_best;
  *n_chunks_out = (int)n_chunks;
  return 0;
}

int ppbo_predict(ppbo_ctx* ctx, const ppbo_model* model, const double* d_Xc, int64_t M, int score_kind,
                 double mustar, double* d_mu, double* d_var, double* d_score, double* h_best_val,
                 int64_t* h_best_idx, void* stream) {
  PPBO_ENTER(ctx);
  hipStream_t s = (hipStream_t)stream;
  Best* chunk_best = nullptr;
  int n_chunks = 0;
  const bool want_best = h_best_val || h_best_idx;
  if (!want_best)
    return predict_passes(ctx, model, d_Xc, M, score_kind, mustar, d_mu, d_var, d_score, false, nullptr, 0,
                          &chunk_best, &n_chunks, s);
  // the best comes back through the ctx's host-mapped record (written by the last score workgroup, flag polled by the
  // host) instead of a device-to-host copy + stream synchronisation
  PpboHostRecord hr;
  if (int rc = ppbo_host_record(ctx, &hr)) return rc;
  if (int rc = predict_passes(ctx, model, d_Xc, M, score_kind, mustar, d_mu, d_var, d_score, true, hr.d_rec, 0,
                              &chunk_best, &n_chunks, s, hr.d_flag, hr.epoch))
    return rc;
  if (int rc = ppbo_host_record_wait(ctx, hr, s)) return rc;
  if (h_best_val) *h_best_val = hr.h_rec[1] < 0.0 ? 0.0 : hr.h_rec[0];
  if (h_best_idx) *h_best_idx = (int64_t)hr.h_rec[1];
  return 0;
}

int ppbo_predict_record(ppbo_ctx* ctx, const ppbo_model* model, const double* d_Xc, int64_t M, int score_kind,
                        double mustar, int64_t index_offset, double* d_record, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_record != nullptr, "d_record");
  Best* chunk_best = nullptr;
  int n_chunks = 0;
  return predict_passes(ctx, model, d_Xc, M, score_kind, mustar, nullptr, nullptr, nullptr, true, d_record,
                        index_offset, &chunk_best, &n_chunks, (hipStream_t)stream);
}

}  // extern "C"

// internal (dist.hip): ppbo_predict_record whose record lives in host-mapped memory, the flag raised after it
int ppbo_predict_record_publish(ppbo_ctx* ctx, const ppbo_model* model, const double* d_Xc, int64_t M, int score_kind,
                                double mustar, int64_t index_offset, double* d_record, unsigned long long* d_flag,
                                unsigned long long epoch, hipStream_t s) {
  Best* chunk_best = nullptr;
  int n_chunks = 0;
  return predict_passes(ctx, model, d_Xc, M, score_kind, mustar, nullptr, nullptr, nullptr, true, d_record,
                        index_offset, &chunk_best, &n_chunks, s, d_flag, epoch);
}

extern "C" {

int ppbo_predict_cov(ppbo_ctx* ctx, const ppbo_model* model, const double* d_Xc, int M, double shrink,
                     double* d_mu, double* d_cov, void* stream) {
  PPBO_ENTER(ctx);
  if (int rc = check_model(ctx, model)) return rc;
  PPBO_REQUIRE(ctx, d_Xc && d_cov && M > 0 && M <= 16384, "candidates (M <= 16384 for a full covariance)");
  PPBO_REQUIRE(ctx, model->d_G && model->d_lam_diag && model->d_lam_off, "model G/Lambda");
  hipStream_t s = (hipStream_t)stream;
  const int N = model->N, mblk = model->m + 1, n_q = N / mblk;
  const int ld = (M + 1) & ~1;
  double* ws = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_KSTAR, (size_t)3 * N * ld * sizeof(double));
  if (!ws) return (int)hipErrorOutOfMemory;
  double* Kt = ws;
  double* Z = ws + (size_t)N * ld;
  double* Y = ws + (size_t)2 * N * ld;
  const int n_split = pick_split(M, n_q);
  const int q_per_split = (n_q + n_split - 1) / n_split;
  const int n_split_eff = (n_q + q_per_split - 1) / q_per_split;
  double* part = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_PART, (size_t)n_split_eff * M * sizeof(double));
  if (!part) return (int)hipErrorOutOfMemory;
  dispatch_kstar(model, d_Xc, M, Kt, ld, part, nullptr, q_per_split, n_split_eff, false, s);
  PPBO_LAUNCH_CHECK(ctx);
  if (d_mu) {
    score_kernel<<<score_blocks(M), SC_THREADS, 0, s>>>(part, n_split_eff, nullptr, nullptr, 0, M, 0.0, PPBO_SCORE_MEAN,
                                                        0.0, 0, d_mu, nullptr, nullptr, nullptr);
    PPBO_LAUNCH_CHECK(ctx);
  }
  // prior block with the reference's shrinkage (gp_model.py:447)
  if (int rc = ppbo_gram(ctx, model->kernel_id, d_Xc, M, model->D, model->theta, shrink, d_cov, stream)) return rc;
  // + K*' Lambda K*
  lam_apply_kernel<<<dim3((M + 127) / 128, n_q), 128, 0, s>>>(Kt, ld, N, M, mblk, model->d_lam_diag,
                                                              model->d_lam_off, Z);
  PPBO_LAUNCH_CHECK(ctx);
  GemmArgs g{};
  g.A = Kt; g.lda = ld; g.B = Z; g.ldb = ld; g.C = d_cov; g.ldc = M;
  g.M = M; g.N = M; g.K = N; g.alpha = 1.0; g.beta = 1.0; g.tri_block = 1;
  if (int rc = ppbo_gemm_launch(ctx, g, 1, 0, s)) return rc;
  // + (G K*)' (G K*)
  GemmArgs y{};
  y.A = model->d_G; y.lda = N; y.B = Kt; y.ldb = ld; y.C = Y; y.ldc = ld;
  y.M = N; y.N = M; y.K = N; y.alpha = 1.0; y.beta = 0.0; y.khi_mode = 1; y.tri_block = mblk;
  if (int rc = ppbo_gemm_launch(ctx, y, 0, 0, s)) return rc;
  GemmArgs c{};
  c.A = Y; c.lda = ld; c.B = Y; c.ldb = ld; c.C = d_cov; c.ldc = M;
  c.M = M; c.N = M; c.K = N; c.alpha = 1.0; c.beta = 1.0; c.tri_block = 1;
  return ppbo_gemm_launch(ctx, c, 1, 0, s);
}

}  // extern "C"

namespace {

// grid[b][g][:] = alpha_(b)[g] * xi[b][:] + x[b][:]: the G points of line b (what FeedbackProcessing.xi_grid returns with
// is_scaled = True, src/feedback_processing.py:57-107, for abscissae alpha that the host has drawn); per_line = 0: one
// shared abscissa vector alpha[G] (common random numbers across lines), 1: alpha[B][G]
__global__ __launch_bounds__(256) void line_grid_kernel(const double* __restrict__ xi, const double* __restrict__ x,
                                                        const double* __restrict__ alpha, int per_line, int B, int G,
                                                        int D, double* __restrict__ grid) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)B * G * D) return;
  const int d = (int)(e % D);
  const long long bg = e / D;
  const int g = (int)(bg % G), b = (int)(bg / G);
  const double a = alpha[per_line ? (size_t)b * G + g : g];
  grid[e] = a * xi[(size_t)b * D + d] + x[(size_t)b * D + d];
}

// EI / varmax of B lines whose grid points are either given (d_grid) or formed on the device from (xi, x, alpha)
int line_acq_impl(ppbo_ctx* ctx, const ppbo_model* model, const double* d_grid, const double* d_xi, const double* d_x,
                  const double* d_alpha, int alpha_per_line, int B, int G, double shrink, const double* d_z, int S,
                  double mustar, double jitter, double* d_ei, double* d_varmax, hipStream_t s) {
  if (int rc = check_model(ctx, model)) return rc;
  PPBO_REQUIRE(ctx, (d_grid || (d_xi && d_x && d_alpha)) && d_z && B > 0 && G > 0 && G <= 128 && S > 0,
               "line arguments (G <= 128)");
  PPBO_REQUIRE(ctx, model->d_G && model->d_lam_diag && model->d_lam_off, "model G/Lambda");
  const int N = model->N, mblk = model->m + 1, n_q = N / mblk, D = model->D;
  const KernParams p = make_kern_params(model->kernel_id, model->theta);
  const int Bc_max = (B < 512) ? B : 512;
  // operands of Y = G K* padded to whole tiles / chunks, as in predict_passes: every tile of that product then takes
  // the unguarded loop whatever N, the star size and B G are (m = 25: 4.16 -> 3.6 ms for 512 lines at N = 2080)
  int Nk = N, g_rows = N;
  const double* Gq = padded_G(ctx, model, 128, (long long)B * G >= 2048, &Nk, &g_rows, s);
  if (!Gq) return (int)hipErrorOutOfMemory;
  PPBO_LAUNCH_CHECK(ctx);
  const int ld = ((Bc_max * G) + 127) & ~127;
  double* ws = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_KSTAR, (size_t)(Nk + g_rows) * ld * sizeof(double));
  if (!ws) return (int)hipErrorOutOfMemory;
  double* Kt = ws;
  double* Y = ws + (size_t)Nk * ld;
  if (Nk != N) PPBO_HIP_CHECK(ctx, hipMemsetAsync(Kt + (size_t)N * ld, 0, (size_t)(Nk - N) * ld * sizeof(double), s));
  double* gridws = nullptr;
  if (!d_grid) {
    gridws = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_SEARCH, (size_t)Bc_max * G * D * sizeof(double));
    if (!gridws) return (int)hipErrorOutOfMemory;
  }
  const int n_split = pick_split(Bc_max * G, n_q);
  const int q_per_split = (n_q + n_split - 1) / n_split;
  const int n_split_eff = (n_q + q_per_split - 1) / q_per_split;
  // row slices of the covariance's data term: enough (line, slice) workgroups for every CU to hold several
  // a slice walks its rows in chunks of cov_cs: whole stars (as many as fit the 32 staged rows) where a star has at most
  // 32 rows -- the symmetric form of line_cov_kernel --, else 32 rows of anything
  const bool sym = mblk <= 32;
  const int cov_cs = sym ? (32 / mblk) * mblk : 32;
  int cov_splits = (1024 + Bc_max - 1) / Bc_max;
  if (cov_splits < (N + LC_MAXROWS - 1) / LC_MAXROWS) cov_splits = (N + LC_MAXROWS - 1) / LC_MAXROWS;   // <= LC_MAXROWS rows per slice
  if (cov_splits > (N + cov_cs - 1) / cov_cs) cov_splits = (N + cov_cs - 1) / cov_cs;
  if (cov_splits < 1) cov_splits = 1;
  int cov_rows = ((((N + cov_splits - 1) / cov_splits) + cov_cs - 1) / cov_cs) * cov_cs;
  if (cov_rows > LC_MAXROWS) cov_rows = (LC_MAXROWS / cov_cs) * cov_cs;
  cov_splits = (N + cov_rows - 1) / cov_rows;
  double* part = (double*)ppbo_workspace(
      ctx, ppbo_ctx::WS_PART,
      ((size_t)(n_split_eff + 1) * Bc_max * G + (size_t)(1 + cov_splits) * Bc_max * G * G) * sizeof(double));
  if (!part) return (int)hipErrorOutOfMemory;
  double* mu = part + (size_t)n_split_eff * Bc_max * G;
  double* cov = mu + (size_t)Bc_max * G;
  double* cov_parts = cov + (size_t)Bc_max * G * G;
  const int G16 = (G + 15) & ~15;
  const size_t mc_lds = ((size_t)G16 * (G16 + 2) + G16 + 16) * sizeof(double);   // G = 128: 134 KB of the CU's 160 KB
  if (mc_lds > 64 * 1024) ppbo_lds_limit(ctx, (const void*)line_mc_kernel, (128 * 130 + 128 + 16) * (int)sizeof(double));
  // draws of a line spread over several workgroups when the batch alone does not fill the chip (>= 64 draws each)
  int nsplit = (2 * 256 + Bc_max - 1) / Bc_max;
  if (nsplit > (S + 63) / 64) nsplit = (S + 63) / 64;
  if (nsplit < 1) nsplit = 1;
  const int draws_per_split = (((S + nsplit - 1) / nsplit) + 15) & ~15;
  nsplit = (S + draws_per_split - 1) / draws_per_split;
  double* mc_part = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_SMALL, (size_t)Bc_max * nsplit * 3 * sizeof(double));
  if (!mc_part) return (int)hipErrorOutOfMemory;
  for (int b0 = 0; b0 < B; b0 += Bc_max) {
    const int Bc = (B - b0 < Bc_max) ? (B - b0) : Bc_max;
    const int M = Bc * G;
    const double* xg;
    if (d_grid) {
      xg = d_grid + (size_t)b0 * G * D;
    } else {
      const long long ne = (long long)Bc * G * D;
      line_grid_kernel<<<(unsigned)((ne + 255) / 256), 256, 0, s>>>(d_xi + (size_t)b0 * D, d_x + (size_t)b0 * D,
                                                                    alpha_per_line ? d_alpha + (size_t)b0 * G : d_alpha,
                                                                    alpha_per_line, Bc, G, D, gridws);
      xg = gridws;
    }
    {
      PpboProfScope pf(ctx, ppbo_ctx::PF_LINE_KSTAR, s);
      dispatch_kstar(model, xg, M, Kt, ld, part, nullptr, q_per_split, n_split_eff, false, s);
    }
    score_kernel<<<score_blocks(M), SC_THREADS, 0, s>>>(part, n_split_eff, nullptr, nullptr, 0, M, 0.0, PPBO_SCORE_MEAN,
                                                        0.0, 0, mu, nullptr, nullptr, nullptr);
    switch (model->kernel_id) {
      case PPBO_KERNEL_SE: line_prior_kernel<PPBO_KERNEL_SE><<<Bc, 256, 0, s>>>(xg, G, D, p, shrink, cov); break;
      case PPBO_KERNEL_RQ: line_prior_kernel<PPBO_KERNEL_RQ><<<Bc, 256, 0, s>>>(xg, G, D, p, shrink, cov); break;
      default: line_prior_kernel<PPBO_KERNEL_CAMPHOR><<<Bc, 256, 0, s>>>(xg, G, D, p, shrink, cov); break;
    }
    PPBO_LAUNCH_CHECK(ctx);
    GemmArgs y{};  // Y = G K*
    y.A = Gq; y.lda = Nk; y.B = Kt; y.ldb = ld; y.C = Y; y.ldc = ld;
    y.M = g_rows; y.N = (M + 127) & ~127; y.K = Nk; y.alpha = 1.0; y.beta = 0.0; y.khi_mode = 1; y.tri_block = mblk;
    // (rows [N, g_rows) of Y come out as zeros, columns [M, y.N) as whatever the K* padding holds: neither is read)
    // column tiles in chunks through all row tiles, heaviest first (a chunk's slice of K* -- 134 MB at 64 tiles -- stays
    // in the Infinity Cache); equal chunks of at most 72 tiles: a ragged last chunk costs 5 % (512 lines: 280 tiles in
    // chunks of 70: 2.33 ms; 64 + ragged 24: 2.47; 56: 2.35; 96: 2.35; 128: 4.85; one chunk: 4.2)
    const int y_tiles = (M + 127) / 128, y_chunks = (y_tiles + 71) / 72;
    y.nt_chunk = ctx->line_y_chunk > 0 ? ctx->line_y_chunk : (y_tiles + y_chunks - 1) / y_chunks;
    {
      PpboProfScope pf(ctx, ppbo_ctx::PF_LINE_Y, s);
      if (int rc = ppbo_gemm_launch(ctx, y, 0, 0, s)) return rc;
    }
    // data term of every line's covariance: K*' Lambda K* + Y'Y, one workgroup per (line, row slice)
    {
      PpboProfScope pf(ctx, ppbo_ctx::PF_LINE_COV, s);
      const dim3 cg(Bc, cov_splits);
      const long long pst = (long long)Bc_max * G * G;
      // sym: every chunk is whole stars: the symmetric form (lower-triangle tiles only)
#define LC_LAUNCH(NT)                                                                                                   \
  do {                                                                                                                  \
    if (sym) line_cov_kernel<NT, true><<<cg, 256, 0, s>>>(Kt, Y, ld, N, G, mblk, model->d_lam_diag, model->d_lam_off,   \
                                                          cov_rows, cov_parts, pst, cov_cs);                            \
    else line_cov_kernel<NT, false><<<cg, 256, 0, s>>>(Kt, Y, ld, N, G, mblk, model->d_lam_diag, model->d_lam_off,      \
                                                       cov_rows, cov_parts, pst, cov_cs);                               \
  } while (0)
      switch ((G + 15) / 16) {
        case 1: LC_LAUNCH(1); break;
        case 2: LC_LAUNCH(2); break;
        case 3: LC_LAUNCH(3); break;
        case 4: LC_LAUNCH(4); break;
        case 5: LC_LAUNCH(5); break;
        case 6: LC_LAUNCH(6); break;
        case 7: LC_LAUNCH(7); break;
        default: LC_LAUNCH(8); break;
      }
#undef LC_LAUNCH
    }
    {
      PpboProfScope pf(ctx, ppbo_ctx::PF_LINE_MC, s);
      line_mc_kernel<<<dim3(Bc, nsplit), 256, mc_lds, s>>>(mu, cov, G, d_z, S, mustar, jitter, draws_per_split, mc_part,
                                                           cov_parts, cov_splits, (long long)Bc_max * G * G, sym ? 1 : 0);
    }
    mc_finish_kernel<<<(Bc + 255) / 256, 256, 0, s>>>(mc_part, Bc, nsplit, S, d_ei ? d_ei + b0 : nullptr,
                                                      d_varmax ? d_varmax + b0 : nullptr);
    PPBO_LAUNCH_CHECK(ctx);
  }
  return 0;
}

}  // namespace

extern "C" {

int ppbo_line_acq(ppbo_ctx* ctx, const ppbo_model* model, const double* d_grid, int B, int G, double shrink,
                  const double* d_z, int S, double mustar, double jitter, double* d_ei, double* d_varmax,
                  void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_grid != nullptr, "d_grid");
  return line_acq_impl(ctx, model, d_grid, nullptr, nullptr, nullptr, 0, B, G, shrink, d_z, S, mustar, jitter, d_ei,
                       d_varmax, (hipStream_t)stream);
}

int ppbo_line_acq_xi(ppbo_ctx* ctx, const ppbo_model* model, const double* d_xi, const double* d_x,
                     const double* d_alpha, int alpha_per_line, int B, int G, double shrink, const double* d_z, int S,
                     double mustar, double jitter, double* d_ei, double* d_varmax, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_xi && d_x && d_alpha, "xi / x / alpha");
  return line_acq_impl(ctx, model, nullptr, d_xi, d_x, d_alpha, alpha_per_line != 0, B, G, shrink, d_z, S, mustar,
                       jitter, d_ei, d_varmax, (hipStream_t)stream);
}

int ppbo_randn(ppbo_ctx* ctx, uint64_t seed, double* d_out, int64_t n, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_out && n > 0, "output");
  const long long pairs = (n + 1) / 2;
  PPBO_REQUIRE(ctx, (pairs + 255) / 256 < ((long long)1 << 31), "n too large for one launch");
  randn_kernel<<<(unsigned)((pairs + 255) / 256), 256, 0, (hipStream_t)stream>>>((unsigned long long)seed, d_out, (long long)n);
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

}  // extern "C"
