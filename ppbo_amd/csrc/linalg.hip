// K6: hand-written fp64 dense kernels: blocked Cholesky, triangular inverse, GEMV, dot.
//   reference: misc.py:96-100 (pd_inverse = LAPACK posv with B = I), the Cholesky
//   factorizations inside SciPy's trust-exact (gp_model.py:382-384).
//
// potrf (right-looking, NB = 64) is latency-bound at these sizes (2.9 GFLOP at N = 2048), so the design
// minimises the dependent chain per panel step rather than flops:
//   default   potrf_step_kernel: ONE launch per step.  Its "panel" workgroups each factor the 64x64
//             diagonal block themselves (16-column slabs: one wavefront with lane = row and v_readlane
//             multipliers, MFMA updates between slabs) and solve their 64 rows below it with the block
//             inverses on the matrix cores; its "update" workgroups apply the PREVIOUS panel's rank-64
//             update to everything right of the current block column (one step of lookahead).
//   PPBO_POTRF_GEN=2   three launches per step: potf2_block, trsm_mfma, SYRK on the gemm64 engine
//             (the simple form; kept as the cross-check for the fused one).
// trtri: 64x64 diagonal inverses (lane = column), then log2(N/64) levels of batched MFMA GEMMs
//   X21 = -inv(L22) (L21 inv(L11)).
// A failing pivot writes its 1-based column to *d_info; later kernels see it and return at once.
#include "linalg.h"

namespace {

constexpr int NB = 64;
typedef unsigned v2u32_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double lane_bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double rsqrt_refined(double d) {
  double y = __builtin_amdgcn_rsq(d);             // v_rsq_f64, ~26 bits
  y = y * fma(-0.5 * d, y * y, 1.5);
  y = y * fma(-0.5 * d, y * y, 1.5);
  return y;
}

// ---- panel kernels, blocked by 16 columns so that the O(64^3) work runs on the matrix cores
constexpr int BLD = NB + 2;   // padded LDS row (doubles): conflict-free MFMA fragment reads

// 64x64 diagonal block, four wavefronts.  For each 16-column slab: wave 0 (lane = row) factors the slab
// with v_readlane-moved multipliers -- 120 rank-1 column updates instead of 2016 -- which also solves the
// rows below the slab's diagonal; then all waves apply the slab to the trailing columns of the block with
// v_mfma_f64_16x16x4_f64 (tile -= L_slab(rows) L_slab(cols)^T, operands and accumulators in LDS).
__global__ __launch_bounds__(256) void potf2_block_kernel(double* __restrict__ A, int lda, int k0, int kb,
                                                          int* __restrict__ info) {
  __shared__ __attribute__((aligned(16))) double Ab[NB * BLD];
  __shared__ int s_fail;
  if (*info != 0) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int e = t; e < NB * NB; e += 256) {
    const int r = e / NB, c = e - r * NB;
    Ab[r * BLD + c] = (r < kb && c < kb && c <= r) ? A[(size_t)(k0 + r) * lda + k0 + c] : ((r == c) ? 1.0 : 0.0);
  }
  if (t == 0) s_fail = 0;
  __syncthreads();
  const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int c0 = 16 * s;
    if (wave == 0) {
      double a[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = Ab[lane * BLD + c0 + j];
      int fail = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double d = lane_bcast(a[j], c0 + j);
        if (!(d > 0.0) && fail == 0 && c0 + j < kb) fail = c0 + j + 1;
        const double rs = rsqrt_refined(d);
        const double lij = (lane == c0 + j) ? d * rs : a[j] * rs;
        a[j] = lij;
#pragma unroll
        for (int k = j + 1; k < 16; ++k) a[k] -= lij * lane_bcast(lij, c0 + k);
      }
      if (lane >= c0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) Ab[lane * BLD + c0 + j] = a[j];
      }
      if (fail && lane == 0 && s_fail == 0) s_fail = fail;
    }
    __syncthreads();
    if (s_fail) break;
    // trailing tiles (rb >= cb > s) of the 4x4 tile grid, round-robin over the four waves
    int tile = 0;
#pragma unroll
    for (int cb = 1; cb < 4; ++cb) {
#pragma unroll
      for (int rb = 1; rb < 4; ++rb) {
        if (cb <= s || rb < cb) continue;
        if ((tile++ & 3) != wave) continue;
        double4_t acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = Ab[(16 * rb + lk + 4 * r) * BLD + 16 * cb + lr];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const double av = -Ab[(16 * rb + lr) * BLD + c0 + kk * 4 + lk];
          const double bv = Ab[(16 * cb + lr) * BLD + c0 + kk * 4 + lk];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Ab[(16 * rb + lk + 4 * r) * BLD + 16 * cb + lr] = acc[r];
      }
    }
    __syncthreads();
  }
  if (s_fail) {
    if (t == 0) *info = k0 + s_fail;
    return;
  }
  for (int e = t; e < NB * NB; e += 256) {
    const int r = e / NB, c = e - r * NB;
    if (r < kb && c <= r) A[(size_t)(k0 + r) * lda + k0 + c] = Ab[r * BLD + c];
  }
}

// rows i >= k0+64: X = B L_kk^-T by 16-column blocks on the matrix cores.  One wavefront = 16 rows.
//   X_c = (B_c - sum_{p<c} X_p L_cp^T) inv(L_cc)^T,  c = 0..3
// inv(L_cc) (four 16x16 triangles) is computed once per workgroup by wave 0 (lane = block*16 + column).
__global__ __launch_bounds__(256) void trsm_mfma_kernel(double* __restrict__ A, int lda, int N, int k0,
                                                        const int* __restrict__ info) {
  __shared__ __attribute__((aligned(16))) double Lk[NB * BLD];        // L_kk
  __shared__ __attribute__((aligned(16))) double Li[4 * 16 * 18];     // inv(L_cc)[j][k], row stride 18
  __shared__ __attribute__((aligned(16))) double Xs[4][16 * BLD];     // per wave: solved / staged row tile
  if (*info != 0) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  for (int e = t; e < NB * NB; e += 256) {
    const int r = e / NB, c = e - r * NB;
    Lk[r * BLD + c] = (c <= r) ? A[(size_t)(k0 + r) * lda + k0 + c] : 0.0;
  }
  __syncthreads();
  if (wave == 0) {
    const int b = lane >> 4, c = lane & 15;      // column c of inv(L_bb)
    double y[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      double v = (r == c) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < r; ++k) v -= Lk[(16 * b + r) * BLD + 16 * b + k] * y[k];
      y[r] = (r >= c) ? v / Lk[(16 * b + r) * BLD + 16 * b + r] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Li[(b * 16 + r) * 18 + c] = y[r];
  }
  __syncthreads();
  const int row0 = k0 + NB + (blockIdx.x * 4 + wave) * 16;
  if (row0 >= N) return;
  double* xs = Xs[wave];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double4_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gr = row0 + lk + 4 * r;
      acc[r] = (gr < N) ? A[(size_t)gr * lda + k0 + 16 * c + lr] : 0.0;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (p >= c) continue;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const double av = -xs[lr * BLD + 16 * p + kk * 4 + lk];
        const double bv = Lk[(16 * c + lr) * BLD + 16 * p + kk * 4 + lk];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
      }
    }
    // residual tile -> A-operand orientation through this wave's staging columns 16c..16c+15
#pragma unroll
    for (int r = 0; r < 4; ++r) xs[(lk + 4 * r) * BLD + 16 * c + lr] = acc[r];
    __builtin_amdgcn_wave_barrier();
    double4_t x = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const double av = xs[lr * BLD + 16 * c + kk * 4 + lk];
      const double bv = Li[(c * 16 + lr) * 18 + kk * 4 + lk];       // B[k][j] = inv(L_cc)[j][k]
      x = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, x, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xs[(lk + 4 * r) * BLD + 16 * c + lr] = x[r];
      const int gr = row0 + lk + 4 * r;
      if (gr < N) A[(size_t)gr * lda + k0 + 16 * c + lr] = x[r];
    }
    __builtin_amdgcn_wave_barrier();
  }
}


// inverse of the 16x16 lower-triangular diagonal block b of the factored panel block, by one level of
// recursion so that the longest dependent FMA chain is 28 + 8 + 8 instead of 120 (the last block's inverse
// sits on the panel's critical path).  Lanes 0-7 / 8-15: column (lane & 7) of the inverse of the leading /
// trailing 8x8 triangle; then lanes 0-7: their column of X21 = -inv(L22) (L21 inv(L11)).
// Rinv holds the reciprocal pivots left by the factorization.  // out points at the (0,0) element of the 16x16 result, ldo is its row pitch.
__device__ __forceinline__ void invert_diag16(const double* __restrict__ Ab, const double* __restrict__ Rinv,
                                              double* __restrict__ out, int ldo, int b, int lane) {
  if (lane >= 16) return;
  const int h = lane >> 3, cc = lane & 7, c0 = 16 * b, o = c0 + 8 * h;
  // Every LDS read of the first two phases up front: their addresses depend on nothing computed here, and one
  // wavefront pays a read's ~130 cycles in full whenever it has to wait for one (read row by row, behind the selects'
  // branches, the inverse took 1.9 us: more than the slab factor it has to hide under).  No select either: for the
  // rows above the column's own the substitution yields 0 by itself.
  // (L21 and inv(L22) in halves of four rows: with all of them in registers at once the kernel needed 324 VGPRs, and
  // two workgroups no longer fitted one CU -- the mode the large-N steps run in: potrf(4096) +3.6 %)
  double l[8][8], ri[8], la[4][8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    ri[r] = Rinv[o + r];
#pragma unroll
    for (int k = 0; k < r; ++k) l[r][k] = Ab[(o + r) * BLD + o + k];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int k = 0; k < 8; ++k) la[r][k] = Ab[(c0 + 8 + r) * BLD + c0 + k];
  double y[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    double v = (r == cc) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < r; ++k) v -= l[r][k] * y[k];
    y[r] = v * ri[r];
  }
  // rows 0-3 of L21 inv(L11) need nothing from the other lanes: formed here, so that the first half of L21 is dead
  // before the second is requested
  double tv[8];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) v += la[r][k] * y[k];
    tv[r] = v;
  }
  double lb[4][8];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int k = 0; k < 8; ++k) lb[r][k] = Ab[(c0 + 12 + r) * BLD + c0 + k];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    out[(8 * h + r) * ldo + 8 * h + cc] = y[r];
    if (h == 1) out[r * ldo + 8 + cc] = 0.0;               // the block above the diagonal
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (h == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) v += lb[r][k] * y[k];
      tv[4 + r] = v;
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      double i22[4][8];                                     // inv(L22), written by lanes 8-15 just now
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j <= 4 * half + r; ++j) i22[r][j] = out[(8 + 4 * half + r) * ldo + 8 + j];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j <= 4 * half + r; ++j) v -= i22[r][j] * tv[j];
        out[(8 + 4 * half + r) * ldo + cc] = v;
      }
    }
  }
}

// ---- one 16-column slab of the diagonal block, ONE wavefront, lane = row (see potrf_panel_part).
// A lone wavefront issues in order; what a lane broadcast costs it (tools/dev/bcast_bench.hip): two v_readlane and the
// fma that uses them 23.6 cycles, against 5.8 for an fma on registers -- the 136 broadcasts of a left-looking slab were
// two thirds of its time (2.3 us inside the panel kernel, 1.5-1.7 us in this form).  Hence:
//  * right-looking, and column j-1's multipliers L[c0 + k][j - 1], k > j, are read back from LDS, where the finished
//    column is stored anyway: one ds_read_b64 with a uniform address (a broadcast, no bank conflict) per multiplier,
//    all issued at the top of a stage and consumed under the pivot chain;
//  * only the pivot and the NEXT column's newest multiplier go through v_readlane, both taken from the unscaled column
//    in one round trip: L[c0 + j + 1][j] = v[lane c0 + j + 1] * rs, scaled by the same product the lane itself forms
//    (bit-identical to broadcasting a[j] afterwards);
//  * the pivot of column j (rsqrt + one third-order step + scale: one dependent chain) is interleaved BY HAND with the
//    rank-1 update column j-1 applies to the columns right of j; the scheduler fences keep that order (hipcc sinks
//    every update to its use otherwise, and the two chains run one after the other).
// Every element still receives its updates in column order, as in the left-looking form.
struct SlabState {
  double a[16];       // the slab's columns of this lane's row: final (scaled) left of the current column
  double ap;          // column j-1, scaled
  double piv[16];     // the pivots (wave-uniform): scanned for the first non-positive one only if the LAST is not positive
};

template <int K0, int K1>
__device__ __forceinline__ void slab_fill(SlabState& st, const double (&mk)[16]) {
#pragma unroll
  for (int k = K0; k < K1; ++k) st.a[k] -= st.ap * mk[k];
  __builtin_amdgcn_sched_barrier(0);
}

template <int J>
__device__ __forceinline__ void slab_column(SlabState& st, double* __restrict__ Ab, double* __restrict__ Rinv, int c0, int lane) {
  // the multipliers are on their way from LDS while the first half of the pivot chain runs (a wait for them would
  // hold the chain too: one wavefront, in order); the updates share the second half, THIRD at a time
  constexpr int NF = (J > 0) ? 15 - J : 0, PER = (NF + 2) / 3, F = J + 1;     // column j-1 updates columns F .. 15
  constexpr auto at = [](int q) { return (F + q * PER < 16) ? F + q * PER : 16; };
  const double v = st.a[J];
  const double d = lane_bcast(v, c0 + J);
  const double mraw = (J < 15) ? lane_bcast(v, c0 + J + 1) : 0.0;
  __builtin_amdgcn_sched_barrier(0);
  double mk[16];
#pragma unroll
  for (int k = F; k < 16; ++k) mk[k] = (J > 0) ? Ab[(c0 + k) * BLD + c0 + J - 1] : 0.0;
  // 1 / sqrt(d): v_rsq_f64 (~26 bits) and ONE third-order step, y0 (1 + e (1/2 + 3/8 e)) with e = 1 - d y0^2 -- the
  // remaining error is (5/16) e^3, far below an ulp -- five dependent-chain instructions instead of the seven of two
  // Newton steps (rsqrt_refined).  A non-positive pivot turns everything right of it into NaN (rsq of d < 0 is NaN,
  // of 0 inf and 0 * inf NaN): the pivots are scanned for the first one only if the last is not positive.
  const double y0 = __builtin_amdgcn_rsq(d);
  st.piv[J] = d;
  const double dy = d * y0;
  const double e = fma(-dy, y0, 1.0);
  __builtin_amdgcn_sched_barrier(0);
  slab_fill<at(0), at(1)>(st, mk);
  const double pq = fma(0.375, e, 0.5), ye = y0 * e;
  slab_fill<at(1), at(2)>(st, mk);
  const double y = fma(ye, pq, y0);
  slab_fill<at(2), 16>(st, mk);
  const double aj = v * y;
  if (J < 15) st.a[J + 1] -= aj * (mraw * y);
  st.a[J] = aj;
  st.ap = aj;
  // Final; the next stage reads its multipliers here.  Every lane stores, unmasked: rows above the diagonal receive
  // garbage, which nothing reads (the block's consumers touch its lower triangle only; the copy to diag_out zeroes
  // the upper one) -- the select and the exec mask were 5 of a column's ~40 instructions.  1 / l_jj: the same value
  // from every lane to one address.
  Ab[lane * BLD + c0 + J] = aj;
  Rinv[c0 + J] = y;
  if constexpr (J < 15) slab_column<J + 1>(st, Ab, Rinv, c0, lane);
}

// One launch per panel step, update part: C -= P P^T with P = block column k-1, on the tiles right of block
// column k.  Persistent: a workgroup walks the folded tile list; the next tile's operands and C values are
// requested before the current tile's MFMAs, so a tile costs its 64 MFMAs per wavefront, not a memory round
// trip.  rank / r0: this workgroup's slot in the round-robin (see tile_of) and its first round.
constexpr int PANEL_ROUNDS = 6;

__device__ __forceinline__ void potrf_update_part(double* __restrict__ A, int lda, int N, int k0, int nP, int ntS,
                                                  double* __restrict__ plds, int rank, int r0, int info_in) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  double* Pi = plds;
  double* Pj = plds + NB * BLD;
  const int kp = k0 - NB, cs = k0 + NB;
  const int nSW = gridDim.x - nP, total = (ntS + 1) * ((ntS + 1) / 2);
  // round r hands out nSW tiles to the update workgroups and, from round PANEL_ROUNDS on, nP more to the
  // panel workgroups, which join once their panel is done (a panel takes about as long as six tiles)
  auto tile_of = [&](int r) { return r * nSW + (r > PANEL_ROUNDS ? (r - PANEL_ROUNDS) * nP : 0) + rank; };
  double pi[16], pj[16];
  double4_t accN[4];
  int ti = 0, tj = 0;
  bool valid = false;
  auto fetch = [&](int idx) {
    const int c = idx % (ntS + 1), tr = idx / (ntS + 1);
    if (c <= tr) { ti = tr; tj = c; valid = true; }
    else { ti = ntS - 1 - tr; tj = c - tr - 1; valid = (ti != tr); }
    const int i0 = cs + ti * NB, j0 = cs + tj * NB;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = t + 256 * q, r = e >> 6, cc = e & 63;
      pi[q] = (valid && i0 + r < N) ? A[(size_t)(i0 + r) * lda + kp + cc] : 0.0;
      pj[q] = (valid && ti != tj && j0 + r < N) ? A[(size_t)(j0 + r) * lda + kp + cc] : 0.0;
    }
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gr = i0 + 16 * wave + lk + 4 * r, gc = j0 + 16 * jt + lr;
        accN[jt][r] = (valid && gr < N && gc < N) ? A[(size_t)gr * lda + gc] : 0.0;
      }
  };
  int rnd = r0;
  int cur = tile_of(rnd);
  if (cur < total) fetch(cur);
  if (info_in != 0) return;      // a failed factorization: checked only now, so that the flag's round trip does not
                                 // delay the first tile's loads (it used to cost every launch ~1 us up front)
  for (; cur < total;) {
    const int nxt = tile_of(rnd + 1);
    const bool v_cur = valid, diag_cur = (ti == tj);
    const int i0 = cs + ti * NB, j0 = cs + tj * NB;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = t + 256 * q;
      Pi[(e >> 6) * BLD + (e & 63)] = pi[q];
      if (!diag_cur) Pj[(e >> 6) * BLD + (e & 63)] = pj[q];
    }
    double4_t acc[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) acc[jt] = accN[jt];
    __syncthreads();
    if (nxt < total) fetch(nxt);
    if (v_cur) {
      const double* Pb = diag_cur ? Pi : Pj;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const double av = -Pi[(16 * wave + lr) * BLD + 4 * kk + lk];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
          acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Pb[(16 * jt + lr) * BLD + 4 * kk + lk], acc[jt], 0, 0, 0);
      }
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gr = i0 + 16 * wave + lk + 4 * r, gc = j0 + 16 * jt + lr;
          if (gr < N && gc < N) store_through(A + (size_t)gr * lda + gc, acc[jt][r]);
        }
    }
    __syncthreads();
    cur = nxt;
    ++rnd;
  }
}

// One launch per panel step, panel part (see potrf_step_kernel).  Returns false when the factorization has
// failed (here or earlier) and the workgroup has nothing more to do.
__device__ __forceinline__ bool potrf_panel_part(double* __restrict__ A, int lda, int N, int k0, int has_prev,
                                                 double* __restrict__ diag_out, int* __restrict__ info,
                                                 double* __restrict__ fail_pivot, double* __restrict__ plds,
                                                 int info_in) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  // Wave 0 only factors; waves 1-3 own 16 panel rows each (48 rows per workgroup) and fill the time wave 0
  // spends inside a slab (~2.2 us, waves 1-3 would otherwise idle at the barrier):
  //   during slab 0   panel k-1's update of their own rows
  //   during slab s   wave 3 inverts diagonal block s-1; waves 1-3 solve column block s-2 of their rows
  // so that after the last slab only two column blocks and one 16x16 inverse remain.
  double* Ab = plds;                    // [64][BLD]  A_kk -> L_kk
  double* Li = Ab + NB * BLD;           // [4][16][18] inverses of the 16x16 diagonal blocks
  double* Xs = Li + 4 * 16 * 18;        // [4 waves][16][BLD] solve staging; until slab 0 is done: L[k, k-1] as [64][BLD]
  double* Rinv = Xs + 4 * 16 * BLD;     // [64] 1 / l_jj
  int& s_fail = *reinterpret_cast<int*>(Rinv + NB);
  double& s_fpiv = Rinv[NB + 1];        // the non-positive pivot that stopped the factorization
  const int kb = (N - k0 < NB) ? (N - k0) : NB;
  // global reads, most urgent first (the counter retires them in order): diagonal block and L[k, k-1]
  // gate the factorization, the wave's own rows are not needed before slab 0 is under way.
  // Through two buffer descriptors (the diagonal block's rows; this wavefront's 16 panel rows), each ending with its
  // last valid row, so that rows past the block / past N read as zero without a branch, and with per-lane offsets
  // fixed for the kernel: a load is one add + one instruction.  (As 64 guarded global loads with 64-bit addresses
  // the requests alone took ~2 us of a step to ISSUE: 12 instructions and a branch each.)
  const int cbase = has_prev ? k0 - NB : k0, dcol = k0 - cbase;       // leftmost column read: block column k-1
  const int rowB = lda * 8;                                            // bytes per row (64 rows x lda x 8 < 4 GB)
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      A + (size_t)k0 * lda + cbase, 0, ((kb - 1) * lda + (N - cbase)) * 8, 0x00020000);
  const int row0 = __builtin_amdgcn_readfirstlane((wave == 0) ? N : k0 + NB + (blockIdx.x * 3 + wave - 1) * 16);
  const int nrow = (N - row0 < 16) ? (N - row0 < 0 ? 0 : N - row0) : 16;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      A + (size_t)(nrow ? row0 : k0) * lda + cbase, 0, nrow ? ((nrow - 1) * lda + (N - cbase)) * 8 : 0, 0x00020000);
  auto ldb = [](__amdgpu_buffer_rsrc_t r, int off) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
  };
  // the first step has no block column k-1: the same loads through descriptors of size zero (no request leaves the
  // CU, the code stays one straight line -- with a branch around them hipcc waited for the first 32 loads before it
  // issued the rest)
  const __amdgpu_buffer_rsrc_t rdp = __builtin_amdgcn_make_buffer_rsrc(
      A + (size_t)k0 * lda + cbase, 0, has_prev ? ((kb - 1) * lda + (N - cbase)) * 8 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwp = __builtin_amdgcn_make_buffer_rsrc(
      A + (size_t)(nrow ? row0 : k0) * lda + cbase, 0, (nrow && has_prev) ? ((nrow - 1) * lda + (N - cbase)) * 8 : 0, 0x00020000);
  // 16-byte requests where a lane's elements are neighbours (the block and L[k, k-1]: two columns per lane; the
  // wave's rows of block column k-1: see the k order below): 40 requests per lane instead of 64 -- a wavefront cannot
  // have more than 63 in flight, the 64th waited for the first to return
  auto ldb2 = [](__amdgpu_buffer_rsrc_t r, int off) {
    return __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  };
  double2_t a11[8], lp[8], afr[8];
  const int vd = (t >> 5) * rowB + (t & 31) * 16;                      // elements (t >> 5, cbase + 2 (t & 31) + {0, 1}) of the block's rows
#pragma unroll
  for (int q = 0; q < 8; ++q) a11[q] = ldb2(rd, vd + 8 * q * rowB + dcol * 8);
#pragma unroll
  for (int q = 0; q < 8; ++q) lp[q] = ldb2(rdp, vd + 8 * q * rowB);
  __builtin_amdgcn_sched_barrier(0);      // hipcc's scheduler would put the wave's own rows first
  double4_t rowv[4];
  // A operand of the wave's own update, k order permuted: MFMA step 2 j + h takes column 8 j + 2 lk + h from lane
  // (lr, lk), on both operands (the B side reads L[k, k-1] from LDS in the same order)
  const int va = lr * rowB + lk * 16;
#pragma unroll
  for (int j = 0; j < 8; ++j) afr[j] = ldb2(rwp, va + 64 * j);
  const int vr = lk * rowB + (dcol + lr) * 8;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) rowv[c][r] = ldb(rw, vr + 4 * r * rowB + 128 * c);
  if (info_in != 0) return false;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int r = (t >> 5) + 8 * q, c = (t & 31) * 2;
    // lower triangle; identity padding of a partial last block
    const double2_t v = {(c <= r) ? ((r >= kb && r == c) ? 1.0 : a11[q][0]) : 0.0,
                         (c + 1 <= r) ? ((r >= kb && r == c + 1) ? 1.0 : a11[q][1]) : 0.0};
    *reinterpret_cast<double2_t*>(Ab + r * BLD + c) = v;
    *reinterpret_cast<double2_t*>(Xs + r * BLD + c) = lp[q];
  }
  if (t == 0) s_fail = 0;
  __syncthreads();
  // panel k-1's update of the diagonal block, tile (rb, cb) -= L[k, k-1](rb) L[k, k-1](cb)^T with L[k, k-1] staged in Xs.
  // Two accumulators over the even / odd k steps (a dependent fp64 MFMA issues every ~138 cycles, an independent one
  // every 64).  Only block column 0 gates slab 0 and is done up front, one tile per wavefront; the other six tiles
  // are updates like any other and are applied by waves 1-3 while wave 0 factors: column 1 during slab 0 (before
  // the first trailing update reads it), columns 2 and 3 during slab 1 (Xs is free until the solves of slab 2).
  // (All ten tiles up front, three per wavefront, were 2.3 us of every step.)
  auto diag_tile = [&](int rb, int cb) {
    const int ra = (16 * rb + lr) * BLD + lk, ca = (16 * cb + lr) * BLD + lk, o = (16 * rb + lk) * BLD + 16 * cb + lr;
    double4_t d0, d1 = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) d0[r] = Ab[o + 4 * r * BLD];
#pragma unroll
    for (int kk = 0; kk < 16; kk += 2) {
      d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Xs[ra + 4 * kk], Xs[ca + 4 * kk], d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Xs[ra + 4 * kk + 4], Xs[ca + 4 * kk + 4], d1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Ab[o + 4 * r * BLD] = d0[r] + d1[r];
  };
  if (has_prev) {
    diag_tile(wave, 0);
    __syncthreads();
  }
  double* xs = Xs + wave * 16 * BLD;
  // column block c of this wave's rows: X_c = (R_c - sum_{p<c} X_p L_cp^T) inv(L_cc)^T
  auto solve_block = [&](int c) {
    double4_t acc = rowv[c], acc1 = double4_t{0.0, 0.0, 0.0, 0.0};      // two chains: a dependent fp64 MFMA waits ~138 cycles
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (p >= c) continue;
#pragma unroll
      for (int kk = 0; kk < 4; kk += 2) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xs[lr * BLD + 16 * p + kk * 4 + lk], Ab[(16 * c + lr) * BLD + 16 * p + kk * 4 + lk], acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xs[lr * BLD + 16 * p + kk * 4 + 4 + lk], Ab[(16 * c + lr) * BLD + 16 * p + kk * 4 + 4 + lk], acc1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) xs[(lk + 4 * r) * BLD + 16 * c + lr] = acc[r] + acc1[r];
    __builtin_amdgcn_wave_barrier();
    double4_t x = double4_t{0.0, 0.0, 0.0, 0.0}, x1 = x;
#pragma unroll
    for (int kk = 0; kk < 4; kk += 2) {
      x = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[lr * BLD + 16 * c + kk * 4 + lk], Li[(c * 16 + lr) * 18 + kk * 4 + lk], x, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[lr * BLD + 16 * c + kk * 4 + 4 + lk], Li[(c * 16 + lr) * 18 + kk * 4 + 4 + lk], x1, 0, 0, 0);
    }
    x += x1;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xs[(lk + 4 * r) * BLD + 16 * c + lr] = x[r];
      // (two halves by hand: __builtin_bit_cast of an ext-vector ELEMENT stored element 0 four times)
      const v2u32_t w = {(unsigned)__double2loint(x[r]), (unsigned)__double2hiint(x[r])};
      __builtin_amdgcn_raw_buffer_store_b64(w, rw, vr + 4 * r * rowB + 128 * c, 0, 0);   // rows >= N: dropped
    }
    __builtin_amdgcn_wave_barrier();
  };
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int c0 = 16 * s;
    if (wave == 0) {
      SlabState st;
#pragma unroll
      for (int j = 0; j < 16; ++j) st.a[j] = Ab[lane * BLD + c0 + j];
      st.ap = 0.0;
      slab_column<0>(st, Ab, Rinv, c0, lane);
      int fail = 0;
      double fpiv = 0.0;
      if (!(st.piv[15] > 0.0)) {
#pragma unroll
        for (int j = 15; j >= 0; --j)
          if (!(st.piv[j] > 0.0)) { fail = c0 + j + 1; fpiv = st.piv[j]; }
      }
      if (fail && lane == 0 && s_fail == 0) { s_fail = fail; s_fpiv = fpiv; }
    } else {
      if (s == 1 && has_prev) {
        if (wave == 1) { diag_tile(2, 2); diag_tile(3, 2); }
        if (wave == 2) diag_tile(3, 3);
      }
      if (s == 0 && has_prev) {
        diag_tile(wave, 1);
        // panel k-1's update of this wave's rows of block column k (L[k, k-1] is still staged in Xs)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          double2_t bv[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) bv[c] = *reinterpret_cast<const double2_t*>(Xs + (16 * c + lr) * BLD + 8 * j + 2 * lk);
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c = 0; c < 4; ++c)
              rowv[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(-afr[j][h], bv[c][h], rowv[c], 0, 0, 0);
        }
      }
      if (s >= 1 && wave == 3) invert_diag16(Ab, Rinv, Li + (s - 1) * 16 * 18, 18, s - 1, lane);
      if (s >= 2 && row0 < N) solve_block(s - 2);
    }
    __syncthreads();
    if (s_fail) break;
    int tile = 0;
#pragma unroll
    for (int cb = 1; cb < 4; ++cb) {
#pragma unroll
      for (int rb = 1; rb < 4; ++rb) {
        if (cb <= s || rb < cb) continue;
        if ((tile++ & 3) != wave) continue;
        double4_t acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = Ab[(16 * rb + lk + 4 * r) * BLD + 16 * cb + lr];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const double av = -Ab[(16 * rb + lr) * BLD + c0 + kk * 4 + lk];
          const double bv = Ab[(16 * cb + lr) * BLD + c0 + kk * 4 + lk];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Ab[(16 * rb + lk + 4 * r) * BLD + 16 * cb + lr] = acc[r];
      }
    }
    if (s < 3) __syncthreads();
  }
  if (!s_fail) {
    if (wave == 0) invert_diag16(Ab, Rinv, Li + 3 * 16 * 18, 18, 3, lane);
    else if (row0 < N) solve_block(2);
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    // also after a failure: the columns left of the failing one are final and the trust-region loop
    // turns them into a lower bound on the shift (potrf_fail_bound_kernel)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = t + 256 * q;
      diag_out[e] = ((e & 63) <= (e >> 6)) ? Ab[(e >> 6) * BLD + (e & 63)] : 0.0;     // the slabs leave garbage above the diagonal
    }
    if (s_fail && t == 0) {
      *info = k0 + s_fail;
      if (fail_pivot) *fail_pivot = s_fpiv;
    }
  }
  if (s_fail) return false;
  if (row0 < N) solve_block(3);
  return true;
}

// ---- ONE launch per panel step, with one step of lookahead.
// Launch k holds two independent kinds of workgroup:
//   panel part (the first nP workgroups, one per 48 rows below the panel): first applies panel k-1's
//     rank-64 update to block column k only (its own 48 rows and, redundantly, the diagonal block), then
//     factors the diagonal block ITSELF -- the factor never travels between workgroups, the redundant
//     flops are free on <= 42 of 256 CUs -- and solves its 48 rows against it.  Nobody may overwrite A_kk
//     while another workgroup can still be reading it, so workgroup 0 parks L_kk in a side buffer
//     (diag_out, one 64x64 slot per panel) that scatter_diag_kernel copies back once at the end;
//   update part (the remaining workgroups, persistent over the 64x64 tiles): panel k-1's trailing update
//     on block columns >= k+1, i.e. everything the panel part does not touch.
// Block column k therefore meets panel k-1's update one launch late, inside the panel part, and the
// bulk trailing update runs in the shadow of the latency-bound panel instead of after it: a step costs
// max(panel, update) and one launch instead of panel + update and two.  No workgroup reads what another
// one writes in the same launch: the panel part reads columns k-1 and k and writes column k (the diagonal
// factor to the side buffer), the update part reads column k-1 and read-modify-writes columns >= k+1.
// More than half of the CU's 160 KB: exactly one workgroup per CU, so a panel workgroup never shares its SIMDs
// with MFMA-saturated update workgroups (measured: sharing stretches the panel part from 23.5 to 30.5 us).
constexpr int STEP_LDS = 82 * 1024;
constexpr int STEP_LDS_SHARED = (NB * BLD + 4 * 16 * 18 + 4 * 16 * BLD + NB + 2) * (int)sizeof(double);   // what the panel part needs: two per CU

// Two builds of the same code.  CROWDED = the steps whose trailing update is so large that two workgroups share a CU
// (see ppbo_potrf_async): two wavefronts per SIMD means at most 256 VGPRs -- hipcc spills 67 of the panel part's
// values to fit, which costs the lone-workgroup steps 11 % (potrf(2048) 0.53 -> 0.59 ms) but buys the crowded ones
// their second workgroup (potrf(4096) 1.65 -> 1.53 ms); the other build takes the registers it wants (324).
template <bool CROWDED>
__global__ __launch_bounds__(256, CROWDED ? 2 : 1) void potrf_step_kernel(double* __restrict__ A, int lda, int N, int k0, int has_prev,
                                                         int nP, int ntS, double* __restrict__ diag_out,
                                                         int* __restrict__ info, double* __restrict__ fail_pivot) {
  extern __shared__ __attribute__((aligned(16))) double plds[];
  // consumed after the first loads are in flight (both parts).  The FIRST step of a factorization does not read the
  // word, it resets it (and the failed pivot): the thread that would report a failure -- workgroup 0, thread 0 -- does
  // so first, in program order; a launch of its own for two stores cost 4.4 us per factorization
  int info_in = 0;
  if (has_prev) info_in = *info;
  else if (blockIdx.x == 0 && threadIdx.x == 0) { *info = 0; if (fail_pivot) *fail_pivot = NAN; }
  const int nSW = gridDim.x - nP;
  if ((int)blockIdx.x >= nP) {
    potrf_update_part(A, lda, N, k0, nP, ntS, plds, blockIdx.x - nP, 0, info_in);
    return;
  }
  if (!potrf_panel_part(A, lda, N, k0, has_prev, diag_out, info, fail_pivot, plds, info_in)) return;
  // the panel is done; help with the trailing update (the tile lists of the rounds >= PANEL_ROUNDS)
  if (ntS == 0 || PANEL_ROUNDS * nSW >= (ntS + 1) * ((ntS + 1) / 2)) return;
  __syncthreads();
  potrf_update_part(A, lda, N, k0, nP, ntS, plds, nSW + blockIdx.x, PANEL_ROUNDS, 0);
}

// After a failed factorization of H + lam I at (0-based) column kf with pivot d <= 0, Conn/Gould/Toint's
// (and SciPy trust-exact's `singular_leading_submatrix`) vector v = (-L1^-T l, 1, 0..), L1 = L[:kf, :kf],
// l = L[kf, :kf], satisfies v'(H + lam I)v = d, hence every admissible shift is >= lam + (-d) / |v|^2.
// One workgroup: back substitution with L1^T in 64-column blocks from the bottom up; the right-hand side
// lives in LDS, each block's triangle is staged there too, the updates above it stream rows of L.
// out[0] = (-d) / |v|^2  (NaN when the pivot itself was not finite).
__global__ __launch_bounds__(512) void potrf_fail_bound_kernel(const double* __restrict__ L, int ldl,
                                                                const int* __restrict__ info,
                                                                const double* __restrict__ fail_pivot,
                                                                double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) double flds[];
  double* Tb = flds;                 // [64][65] the current diagonal block of L
  double* vb = Tb + 64 * 65;         // [64] solved entries of the block
  double* rhs = vb + 64;             // [kf] right-hand side / solution
  const int kf = *info - 1;
  const double d = *fail_pivot;
  const int t = threadIdx.x;
  if (kf < 0) { if (t == 0) out[0] = NAN; return; }
  for (int i = t; i < kf; i += 512) rhs[i] = -L[(size_t)kf * ldl + i];
  __syncthreads();
  double vn2 = 0.0;                  // wave 0's share of |v1|^2
  for (int b1 = kf; b1 > 0; b1 -= 64) {
    const int b0 = (b1 >= 64) ? b1 - 64 : 0, nb = b1 - b0;
    for (int e = t; e < nb * nb; e += 512) {
      const int r = e / nb, c = e - r * nb;
      Tb[r * 65 + c] = (c <= r) ? L[(size_t)(b0 + r) * ldl + b0 + c] : 0.0;
    }
    __syncthreads();
    if (t < 64) {
      // U x = r with U = Tb^T: lane i owns r_i; columns from the last to the first
      double ri = (t < nb) ? rhs[b0 + t] : 0.0;
      const double rv = (t < nb) ? 1.0 / Tb[t * 65 + t] : 0.0;     // one division per lane, not one per step
      for (int j = nb - 1; j >= 0; --j) {
        const double xj = __shfl(ri * rv, j, 64);
        if (t == j) ri = xj;
        else if (t < j) ri -= Tb[j * 65 + t] * xj;
      }
      vb[t] = (t < nb) ? ri : 0.0;
      if (t < nb) { rhs[b0 + t] = ri; vn2 += ri * ri; }
    }
    __syncthreads();
    // rows above the block: r_i -= sum_j L[b0+j][i] x_j.  One column per thread, all 64 rows requested at once
    // (x_j = 0 beyond nb): the kernel is a chain of memory round trips, so each one carries as much as it can
    for (int i = t; i < b0; i += 512) {
      double lv[64];
#pragma unroll
      for (int j = 0; j < 64; ++j) lv[j] = (j < nb) ? L[(size_t)(b0 + j) * ldl + i] : 0.0;
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 64; ++j) acc += lv[j] * vb[j];
      rhs[i] -= acc;
    }
    __syncthreads();
  }
  if (t < 64) {
    vn2 = wave_sum(vn2);
    if (t == 0) out[0] = (-d) / (1.0 + vn2);
  }
}

// diagonal factors parked by potrf_step_kernel -> lower triangles of A's diagonal blocks
__global__ __launch_bounds__(256) void scatter_diag_kernel(double* __restrict__ A, int lda, int N,
                                                           const double* __restrict__ diag, const int* __restrict__ info) {
  const int bad = *info;                       // 1-based failing column, 0 = success
  const int k0 = blockIdx.x * NB;
  if (bad != 0 && k0 >= bad) return;           // panels right of the failure were never factored
  const double* src = diag + (size_t)blockIdx.x * NB * NB;
  for (int e = threadIdx.x; e < NB * NB; e += 256) {
    const int r = e / NB, c = e - r * NB;
    if (c <= r && k0 + r < N) A[(size_t)(k0 + r) * lda + k0 + c] = src[e];
  }
}

// inverse of the 64x64 diagonal blocks of lower-triangular L, one 256-thread workgroup per block:
// the four 16x16 diagonal triangles are inverted by one wavefront each (invert_diag16), then two
// doubling levels X21 = -inv(L22) (L21 inv(L11)) on the matrix cores, operands in LDS.
// (The first version -- one wavefront, lane = column, 64 sequential rows -- took 64 us per launch.)
__global__ __launch_bounds__(256) void trtri_diag_kernel(const double* __restrict__ L, int ldl, int N,
                                                         double* __restrict__ Li, int ldi) {
  __shared__ __attribute__((aligned(16))) double Ls[NB * BLD];   // the block of L
  __shared__ __attribute__((aligned(16))) double Is[NB * BLD];   // its inverse
  __shared__ __attribute__((aligned(16))) double Ts[32 * BLD];   // L21 inv(L11) of the current level
  __shared__ double Rinv[NB];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int b0 = blockIdx.x * NB;
  const int kb = (N - b0 < NB) ? (N - b0) : NB;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int e = t + 256 * q, r = e >> 6, c = e & 63;
    const double v = (r < kb && c <= r) ? L[(size_t)(b0 + r) * ldl + b0 + c] : 0.0;
    Ls[r * BLD + c] = (r >= kb && r == c) ? 1.0 : v;            // identity padding of a partial last block
    Is[r * BLD + c] = 0.0;
    if (r == c) Rinv[r] = (r < kb) ? 1.0 / v : 1.0;
  }
  __syncthreads();
  invert_diag16(Ls, Rinv, Is + (16 * wave) * BLD + 16 * wave, BLD, wave, lane);
  __syncthreads();
  // 16 -> 32: waves 0 and 1 own the two 32x32 diagonal blocks
  if (wave < 2) {
    const int o = 32 * wave;
    double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)       // T = L21 inv(L11)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[(o + 16 + lr) * BLD + o + 4 * kk + lk],
                                                 Is[(o + 4 * kk + lk) * BLD + o + lr], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Ts[(16 * wave + lk + 4 * r) * BLD + lr] = acc[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double4_t x = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)       // X21 = -inv(L22) T
      x = __builtin_amdgcn_mfma_f64_16x16x4f64(-Is[(o + 16 + lr) * BLD + o + 16 + 4 * kk + lk],
                                               Ts[(16 * wave + 4 * kk + lk) * BLD + lr], x, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Is[(o + 16 + lk + 4 * r) * BLD + o + lr] = x[r];
  }
  __syncthreads();
  // 32 -> 64: one 16x16 tile of the 32x32 products per wave
  {
    const int tr = wave >> 1, tc = wave & 1;
    double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)       // T = L21 inv(L11), L21 = L[32:64, 0:32]
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[(32 + 16 * tr + lr) * BLD + 4 * kk + lk],
                                                 Is[(4 * kk + lk) * BLD + 16 * tc + lr], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Ts[(16 * tr + lk + 4 * r) * BLD + 16 * tc + lr] = acc[r];
    __syncthreads();
    double4_t x = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)       // X21 = -inv(L22) T, inv(L22) = Is[32:64, 32:64]
      x = __builtin_amdgcn_mfma_f64_16x16x4f64(-Is[(32 + 16 * tr + lr) * BLD + 32 + 4 * kk + lk],
                                               Ts[(4 * kk + lk) * BLD + 16 * tc + lr], x, 0, 0, 0);
    __syncthreads();                     // every wave has read inv(L11) / inv(L22) before the corner is written
#pragma unroll
    for (int r = 0; r < 4; ++r) Is[(32 + 16 * tr + lk + 4 * r) * BLD + 16 * tc + lr] = x[r];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int e = t + 256 * q, r = e >> 6, c = e & 63;
    if (r < kb && c < kb) Li[(size_t)(b0 + r) * ldi + b0 + c] = Is[r * BLD + c];
  }
}

// row i of y = T x by one wavefront (all 64 lanes call it)
__device__ __forceinline__ void gemv_row(const double* __restrict__ T, int rows, int cols, int ldt,
                                         const double* __restrict__ x, const double* __restrict__ y0,
                                         double* __restrict__ y, int lower, int i, int lane) {
  if (i >= rows) return;
  const int kend = (lower && i + 1 < cols) ? (i + 1) : cols;
  const double* row = T + (size_t)i * ldt;
  double s = 0.0;
  const bool vec = ((ldt & 1) == 0) && ((reinterpret_cast<uintptr_t>(T) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  if (vec) {
    // four independent 16-byte load pairs in flight per lane: a 16 KB row used to be 16 dependent round trips to L2
    // (9.3 us for the 2048 x 2048 triangle, set by its longest rows)
    const int kv = kend & ~1;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = lane * 2;
    for (; k + 384 < kv; k += 512) {
      const double2 t0 = *reinterpret_cast<const double2*>(row + k), x0 = *reinterpret_cast<const double2*>(x + k);
      const double2 t1 = *reinterpret_cast<const double2*>(row + k + 128), x1 = *reinterpret_cast<const double2*>(x + k + 128);
      const double2 t2 = *reinterpret_cast<const double2*>(row + k + 256), x2 = *reinterpret_cast<const double2*>(x + k + 256);
      const double2 t3 = *reinterpret_cast<const double2*>(row + k + 384), x3 = *reinterpret_cast<const double2*>(x + k + 384);
      s += t0.x * x0.x + t0.y * x0.y;
      s1 += t1.x * x1.x + t1.y * x1.y;
      s2 += t2.x * x2.x + t2.y * x2.y;
      s3 += t3.x * x3.x + t3.y * x3.y;
    }
    for (; k < kv; k += 128) {
      const double2 t = *reinterpret_cast<const double2*>(row + k);
      const double2 xv = *reinterpret_cast<const double2*>(x + k);
      s += t.x * xv.x + t.y * xv.y;
    }
    s = (s + s1) + (s2 + s3);
    if ((kend & 1) && lane == 0) s += row[kend - 1] * x[kend - 1];
  } else {
    for (int k = lane; k < kend; k += 64) s += row[k] * x[k];
  }
  s = wave_sum(s);
  if (lane == 0) y[i] = y0 ? y0[i] - s : s;
}

// y = T x, one wavefront per row
__global__ __launch_bounds__(256) void gemv_rows_kernel(const double* __restrict__ T, int rows, int cols, int ldt,
                                                        const double* __restrict__ x, const double* __restrict__ y0,
                                                        double* __restrict__ y, int lower, PpboGate gate) {
  if (gate.closed()) return;
  gemv_row(T, rows, cols, ldt, x, y0, y, lower, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

// y = T^T x in two deterministic passes.  partial[split][j] = sum_{i in split, i >= (lower ? j : 0)} T[i][j] x[i]
// with GT_ROWS rows per split: every lane keeps all its loads in flight (the first version looped over 64
// rows and was latency-bound at 0.8 TB/s); splits that lie entirely above the diagonal are skipped by both
// passes.
constexpr int GT_ROWS = 16;

__global__ __launch_bounds__(256) void gemvT_partial_kernel(const double* __restrict__ T, int rows, int cols, int ldt,
                                                            const double* __restrict__ x,
                                                            double* __restrict__ partial, int lower, PpboGate gate) {
  if (gate.closed()) return;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i0 = blockIdx.y * GT_ROWS;
  if (lower && i0 + GT_ROWS <= (int)blockIdx.x * 256) return;      // no row of this split reaches these columns
  if (j >= cols) return;
  double v[GT_ROWS];
#pragma unroll
  for (int r = 0; r < GT_ROWS; ++r) {
    const int i = i0 + r;
    v[r] = (i < rows && (!lower || i >= j)) ? T[(size_t)i * ldt + j] : 0.0;
  }
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < GT_ROWS; ++r) s += v[r] * ((i0 + r < rows) ? x[i0 + r] : 0.0);
  partial[(size_t)blockIdx.y * cols + j] = s;
}

// u = L^T beta(f), first pass, with beta REBUILT PER WORKGROUP from f instead of read from a vector that a launch of its
// own (laplace_kernel) would have to write first: one launch less per evaluation of the whitened f_MAP search.
// Requires mblk <= 64: a star's rows fit one wavefront -- lane r-1 holds pseudo-observation row r exactly as in laplace_kernel (src/gp_model.py:228-240: beta_j =
// -phi2(Delta_j) / (sigma m) on pseudo rows, beta_obs = sum_j phi2(Delta_j) / (sigma m)), same wave_sum, same bits.
// The workgroups of the first column block also publish beta (the judgement needs it for |grad_f|) and, one per star,
// the likelihood sum tq[q] = sum_j Phi(Delta_j / sqrt2) (:221-226).
__global__ __launch_bounds__(256) void gemvT_beta_partial_kernel(const double* __restrict__ T, int N, int ldt,
                                                                 const double* __restrict__ f, int mblk, double sigma,
                                                                 double* __restrict__ partial,
                                                                 double* __restrict__ beta_out, double* __restrict__ tq,
                                                                 PpboGate gate, int n_split, const double* __restrict__ R,
                                                                 int ldr, double* __restrict__ rv, PpboGate rider_gate) {
  if (gate.closed()) return;
  if ((int)blockIdx.y >= n_split) {
    // the rider: rv = R f by rows (the search's v = Sigma^-1 f, wanted only near the end: a launch of its own costs
    // its 4-5 us in EVERY slot, gated off or not)
    if (rider_gate.closed()) return;
    gemv_row(R, N, N, ldr, f, nullptr, rv, 0, (((int)blockIdx.y - n_split) * (int)gridDim.x + (int)blockIdx.x) * 4 + (threadIdx.x >> 6),
             threadIdx.x & 63);
    return;
  }
  __shared__ double sb[GT_ROWS];
  const int i0 = blockIdx.y * GT_ROWS;
  if (i0 + GT_ROWS <= (int)blockIdx.x * 256) return;      // no row of this split reaches these columns
  const int m = mblk - 1;
  {
    // the stars that reach into this split (one when the star size is a multiple of GT_ROWS, two for e.g. the
    // reference's default m = 25, more for tiny stars), one wavefront each.  A star is PUBLISHED (beta_out, tq) by the
    // one split that holds its observation row.
    const int i_last = (i0 + GT_ROWS < N ? i0 + GT_ROWS : N) - 1;
    const int qa = i0 / mblk, qb = i_last / mblk;
    const int lane = threadIdx.x & 63, r = lane + 1;
    const double bsc = sigma * (double)m;
    for (int q = qa + (int)(threadIdx.x >> 6); q <= qb; q += 4) {
      const int q0 = q * mblk;
      const bool pub = blockIdx.x == 0 && q0 >= i0;
      const double f0 = f[q0];
      double p2 = 0.0, ph = 0.0;
      if (r <= m) {
        const double delta = (f[q0 + r] - f0) / sigma;
        p2 = 0.28209479177387814347 * exp(-0.25 * (delta * delta));
        if (pub) ph = 0.5 * erfc(-0.5 * delta);
      }
      const double sp2 = wave_sum(p2);
      if (r <= m) {
        const int k = q0 + r - i0;
        if (k >= 0 && k < GT_ROWS) sb[k] = -p2 / bsc;
        if (pub) beta_out[q0 + r] = -p2 / bsc;
      }
      if (lane == 0 && q0 >= i0) { sb[q0 - i0] = sp2 / bsc; if (pub) beta_out[q0] = sp2 / bsc; }
      if (pub) {
        const double sphi = wave_sum(ph);
        if (lane == 0) tq[q] = sphi;
      }
    }
  }
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  double v[GT_ROWS];
#pragma unroll
  for (int r = 0; r < GT_ROWS; ++r) {
    const int i = i0 + r;
    v[r] = (i < N && i >= j) ? T[(size_t)i * ldt + j] : 0.0;
  }
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < GT_ROWS; ++r) s += v[r] * ((i0 + r < N) ? sb[r] : 0.0);
  partial[(size_t)blockIdx.y * N + j] = s;
}

// y[j] = sum over the splits that were written; 16 columns x 16 split groups per workgroup, fixed summation order
__global__ __launch_bounds__(256) void sum_slabs_kernel(const double* __restrict__ partial, int n_split, int N,
                                                        const double* __restrict__ y0, double* __restrict__ y,
                                                        int lower, PpboGate gate) {
  if (gate.closed()) return;
  __shared__ double sh[16][17];
  const int c = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + c;
  // first split whose partial kernel ran for this column's 256-column block
  const int first = lower ? ((j / 256) * 256) / GT_ROWS : 0;
  double s = 0.0;
  if (j < N) {
#pragma unroll 4
    for (int k = first + grp; k < n_split; k += 16) s += partial[(size_t)k * N + j];
  }
  sh[grp][c] = s;
  __syncthreads();
  if (grp == 0 && j < N) {
    double t = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += sh[g][c];
    y[j] = y0 ? y0[j] - t : t;
  }
}

// sum_slabs_kernel (lower = 1, no y0) that also leaves the workgroup's row of PpboDotsOut partial sums over its 16 columns
__global__ __launch_bounds__(256) void sum_slabs_dots_kernel(const double* __restrict__ partial, int n_split, int N,
                                                             double* __restrict__ y, PpboGate gate, PpboDotsOut dd) {
  if (gate.closed()) return;
  __shared__ double sh[16][17];
  __shared__ double su[16];
  const int c = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + c;
  const int first = ((j / 256) * 256) / GT_ROWS;
  double s = 0.0;
  if (j < N) {
#pragma unroll 4
    for (int k = first + grp; k < n_split; k += 16) s += partial[(size_t)k * N + j];
  }
  sh[grp][c] = s;
  __syncthreads();
  if (grp == 0) {
    double t = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += sh[g][c];
    if (j < N) y[j] = t;
    su[c] = t;
  }
  __syncthreads();
  // thread (grp, c): products number grp and 16 + grp of column c; the sum over the 16 columns by four xor steps
  const bool in = j < N;
  const double zi = in ? dd.zt[j] : 0.0;
  const double gt = in ? zi - su[c] : 0.0;
  double p0 = 0.0, p1 = 0.0;
  if (in) {
    if (grp < dd.nb) p0 = gt * dd.basis[(size_t)grp * N + j];
    const int k1 = 16 + grp;                        // 16 .. 31: the remaining basis vectors, then the seven scalars
    if (k1 < dd.nb) p1 = gt * dd.basis[(size_t)k1 * N + j];
    else {
      const int q = k1 - dd.nb;                     // 0 zz, 1 gf2, 2 sy, 3 ss, 4 yy, 5 gt.d, 6 gt.gt
      if (q == 0) p1 = zi * zi;
      else if (q == 1) { const double gf = dd.v[j] - dd.beta[j]; p1 = gf * gf; }
      else if (q == 5) p1 = gt * dd.d[j];
      else if (q == 6) p1 = gt * gt;
      else if (q >= 2 && q <= 4) {
        const double sv = zi - dd.z[j], yv = gt - dd.gcur[j];
        p1 = (q == 2) ? sv * yv : (q == 3 ? sv * sv : yv * yv);
      }
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) { p0 += __shfl_xor(p0, o, 64); p1 += __shfl_xor(p1, o, 64); }
  if (c == 0) {
    double* row = dd.partial + (size_t)blockIdx.x * PPBO_DOTS_STRIDE;
    if (grp < dd.nb) row[8 + grp] = p0;
    const int k1 = 16 + grp;
    if (k1 < dd.nb) row[8 + k1] = p1;
    else if (k1 - dd.nb < 7) row[k1 - dd.nb] = p1;
  }
}

__global__ __launch_bounds__(1024) void dot_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                   int N, double* __restrict__ out) {
  __shared__ double sh[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < N; i += 1024) s += x[i] * y[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += sh[w];
    *out = t;
  }
}

// borders of the appended inverses (n = n1 + k):
//   Ainv[0:n1, n1:] = -W, Ainv[n1:, 0:n1] = -W^T, Ainv[n1:, n1:] = BR
//   Linv[0:n1, n1:] = 0,  Linv[n1:, 0:n1] = -X,   Linv[n1:, n1:] = L22i
__global__ __launch_bounds__(256) void append_border_kernel(const double* __restrict__ W, const double* __restrict__ BR,
                                                            const double* __restrict__ X, const double* __restrict__ L22i,
                                                            int n1, int k, double* __restrict__ Ainv,
                                                            double* __restrict__ Linv, int ld) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int c = threadIdx.x & 63;
  if (i >= n1 + k || c >= k) return;
  if (i < n1) {
    const double w = -W[(size_t)i * k + c];
    Ainv[(size_t)i * ld + n1 + c] = w;
    Ainv[(size_t)(n1 + c) * ld + i] = w;
    Linv[(size_t)i * ld + n1 + c] = 0.0;
    Linv[(size_t)(n1 + c) * ld + i] = -X[(size_t)c * n1 + i];
  } else {
    Ainv[(size_t)i * ld + n1 + c] = BR[(size_t)(i - n1) * k + c];
    Linv[(size_t)i * ld + n1 + c] = L22i[(size_t)(i - n1) * k + c];
  }
}

// bordered Cholesky factor: L[n1:, 0:n1] = Y^T, L[n1:, n1:] = L22 (lower), L[0:n1, n1:] = 0
__global__ __launch_bounds__(256) void append_factor_kernel(const double* __restrict__ Y, const double* __restrict__ L22,
                                                            int n1, int k, double* __restrict__ L, int ld) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int c = threadIdx.x & 63;
  if (i >= n1 + k || c >= k) return;
  if (i < n1) {
    L[(size_t)(n1 + c) * ld + i] = Y[(size_t)i * k + c];
    L[(size_t)i * ld + n1 + c] = 0.0;
  } else {
    L[(size_t)i * ld + n1 + c] = (c <= i - n1) ? L22[(size_t)(i - n1) * k + c] : 0.0;
  }
}

__global__ void set_int_kernel(int* p, int v, double* fail_pivot) {
  *p = v;
  if (fail_pivot) *fail_pivot = NAN;
}

}  // namespace

// ctx->potrf_gen (PPBO_POTRF_GEN): 3 (default): potrf_step_kernel; 2: potf2_block + trsm_mfma + gemm64 SYRK
int ppbo_potrf_async(ppbo_ctx* ctx, double* d_A, int N, int lda, int* d_info, hipStream_t s, double* d_fail_pivot) {
  PpboProfScope pf(ctx, ppbo_ctx::PF_POTRF, s);
  if (ctx->potrf_gen < 3) set_int_kernel<<<1, 1, 0, s>>>(d_info, 0, d_fail_pivot);   // generation 3: the first step does it
  if (ctx->potrf_gen >= 3) {
    ppbo_lds_limit(ctx, (const void*)potrf_step_kernel<false>, STEP_LDS);
    ppbo_lds_limit(ctx, (const void*)potrf_step_kernel<true>, STEP_LDS);
    const int npanel = (N + NB - 1) / NB;
    double* diag = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_POTRF, (size_t)npanel * NB * NB * sizeof(double));
    if (!diag) return (int)hipErrorOutOfMemory;
    for (int k0 = 0; k0 < N; k0 += NB) {
      const int rest = N - k0 - NB;
      const int nP = rest > 0 ? (rest + 47) / 48 : 1;   // panel part: 3 row-owning wavefronts of 16 rows
      const int ntS = (k0 > 0 && rest > 0) ? (rest + NB - 1) / NB : 0;
      const int nS = ntS ? (ntS + 1) * ((ntS + 1) / 2) : 0;
      // One workgroup per CU (STEP_LDS) keeps the panel undisturbed; when the trailing update is so large that
      // it, not the panel, sets the step time (N >= ~3000), two workgroups share a CU instead: the panel slows
      // down by a third, the update nearly doubles its rate (a lone update workgroup is bound by the latency
      // of its one-tile-ahead prefetch).
      const bool crowded = nS > 900;
      const int slots = (crowded ? 512 : 256) - nP;
      const int nSW = nS < slots ? nS : slots;
      if (crowded)
        potrf_step_kernel<true><<<nP + nSW, 256, STEP_LDS_SHARED, s>>>(d_A, lda, N, k0, k0 > 0 ? 1 : 0, nP, ntS,
                                                                       diag + (size_t)(k0 / NB) * NB * NB, d_info, d_fail_pivot);
      else
        potrf_step_kernel<false><<<nP + nSW, 256, STEP_LDS, s>>>(d_A, lda, N, k0, k0 > 0 ? 1 : 0, nP, ntS,
                                                                 diag + (size_t)(k0 / NB) * NB * NB, d_info, d_fail_pivot);
    }
    scatter_diag_kernel<<<npanel, 256, 0, s>>>(d_A, lda, N, diag, d_info);
    PPBO_LAUNCH_CHECK(ctx);
    return 0;
  }
  for (int k0 = 0; k0 < N; k0 += NB) {
    const int kb = (N - k0 < NB) ? (N - k0) : NB;
    const int rest = N - k0 - NB;
    potf2_block_kernel<<<1, 256, 0, s>>>(d_A, lda, k0, kb, d_info);
    if (rest > 0) {
      trsm_mfma_kernel<<<(rest + 63) / 64, 256, 0, s>>>(d_A, lda, N, k0, d_info);
      GemmArgs g{};
      g.A = d_A + (size_t)(k0 + NB) * lda + k0; g.lda = lda;
      g.B = g.A; g.ldb = lda;
      g.C = d_A + (size_t)(k0 + NB) * lda + (k0 + NB); g.ldc = lda;
      g.M = rest; g.N = rest; g.K = NB; g.alpha = -1.0; g.beta = 1.0;
      g.lower_only = 1; g.tri_block = 1;
      if (int rc = ppbo_gemm_launch(ctx, g, 0, 1, s)) return rc;
    }
  }
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

int ppbo_potrf_fail_bound_async(ppbo_ctx* ctx, const double* d_L, int N, int ldl, const int* d_info,
                                const double* d_fail_pivot, double* d_out, hipStream_t s) {
  const size_t lds = ((size_t)64 * 65 + 64 + N) * sizeof(double);
  if (lds > 150 * 1024) return 1;     // leading block too large for the LDS-resident right-hand side: no bound
  ppbo_lds_limit(ctx, (const void*)potrf_fail_bound_kernel, 150 * 1024);
  potrf_fail_bound_kernel<<<1, 512, lds, s>>>(d_L, ldl, d_info, d_fail_pivot, d_out);
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

int ppbo_trtri_async(ppbo_ctx* ctx, const double* d_L, int N, int ldl, double* d_Linv, int ldi, hipStream_t s,
                     int skip_top, int* split_out, int zero_upper) {
  if (split_out) *split_out = 0;
  // The full result promises zeros above the diagonal.  With skip_top the caller only ever applies the two
  // diagonal blocks as lower-triangular operators (ppbo_apply_linv_async), nothing above the diagonal blocks
  // is read, and the 33 MB memset per trust-region trial (a blit with its own barriers) is skipped.
  if (!skip_top && zero_upper) PPBO_HIP_CHECK(ctx, hipMemsetAsync(d_Linv, 0, (size_t)N * ldi * sizeof(double), s));
  const int nblk = (N + NB - 1) / NB;
  trtri_diag_kernel<<<nblk, 256, 0, s>>>(d_L, ldl, N, d_Linv, ldi);
  PPBO_LAUNCH_CHECK(ctx);
  if (N <= NB) return 0;
  double* Tw = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_LINALG2, (size_t)N * N / 2 * sizeof(double) + 4096);
  if (!Tw) return (int)hipErrorOutOfMemory;
  for (int b = NB; b < N; b *= 2) {
    if (skip_top && 2 * b >= N) {       // the top split [0, b) | [b, N) stays implicit (ppbo_apply_linv_async)
      if (split_out) *split_out = b;
      break;
    }
    const int step = 2 * b;
    int nfull = 0;
    int ragged_r0 = -1, ragged_b2 = 0;
    for (int r0 = 0; r0 + b < N; r0 += step) {
      const int b2 = (N - r0 - b < b) ? (N - r0 - b) : b;
      if (b2 == b) ++nfull;
      else { ragged_r0 = r0; ragged_b2 = b2; }
    }
    for (int pass = 0; pass < 2; ++pass) {
      const bool rag = (pass == 1);
      if (rag && ragged_r0 < 0) continue;
      if (!rag && nfull == 0) continue;
      const int r0 = rag ? ragged_r0 : 0;
      const int b2 = rag ? ragged_b2 : b;
      double* T = Tw + (rag ? (size_t)nfull * b * b : 0);
      GemmArgs g1{};  // T = L21 inv(L11)
      g1.A = d_L + (size_t)(r0 + b) * ldl + r0; g1.lda = ldl;
      g1.B = d_Linv + (size_t)r0 * ldi + r0; g1.ldb = ldi;
      g1.C = T; g1.ldc = b;
      g1.M = b2; g1.N = b; g1.K = b; g1.alpha = 1.0; g1.beta = 0.0;
      g1.klo_mode = 2; g1.tri_block = 1;
      g1.batch = rag ? 1 : nfull;
      g1.strideA = (long long)step * ldl + step; g1.strideB = (long long)step * ldi + step;
      g1.strideC = (long long)b * b;
      if (int rc = ppbo_gemm_launch(ctx, g1, 0, 0, s)) return rc;
      GemmArgs g2{};  // X21 = -inv(L22) T
      g2.A = d_Linv + (size_t)(r0 + b) * ldi + (r0 + b); g2.lda = ldi;
      g2.B = T; g2.ldb = b;
      g2.C = d_Linv + (size_t)(r0 + b) * ldi + r0; g2.ldc = ldi;
      g2.M = b2; g2.N = b; g2.K = b2; g2.alpha = -1.0; g2.beta = 0.0;
      g2.khi_mode = 1; g2.tri_block = 1;
      g2.batch = rag ? 1 : nfull;
      g2.strideA = (long long)step * ldi + step; g2.strideB = (long long)b * b;
      g2.strideC = (long long)step * ldi + step;
      if (int rc = ppbo_gemm_launch(ctx, g2, 0, 0, s)) return rc;
    }
  }
  return 0;
}

// y = T x (trans = 0, T rows x cols) or y = T^T x (trans = 1); with y0: y = y0 - (that product)
int ppbo_gemv_rect_async(ppbo_ctx* ctx, const double* d_T, int rows, int cols, int ldt, const double* d_x,
                         const double* d_y0, double* d_y, int trans, int lower, hipStream_t s,
                         PpboGate gate = PpboGate()) {
  if (!trans) {
    gemv_rows_kernel<<<(rows + 3) / 4, 256, 0, s>>>(d_T, rows, cols, ldt, d_x, d_y0, d_y, lower, gate);
  } else {
    const int n_split = (rows + GT_ROWS - 1) / GT_ROWS;
    double* part = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_VEC, (size_t)n_split * cols * sizeof(double));
    if (!part) return (int)hipErrorOutOfMemory;
    gemvT_partial_kernel<<<dim3((cols + 255) / 256, n_split), 256, 0, s>>>(d_T, rows, cols, ldt, d_x, part, lower, gate);
    sum_slabs_kernel<<<(cols + 15) / 16, 256, 0, s>>>(part, n_split, cols, d_y0, d_y, lower, gate);
  }
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

// u = L^T beta(f) with beta and the per-query likelihood sums as by-products (see gemvT_beta_partial_kernel); returns 1
// (nothing enqueued) when the star size does not allow it: the caller then runs laplace_kernel + ppbo_gemv_async
int ppbo_gemvT_beta_async(ppbo_ctx* ctx, const double* d_L, int N, int ldl, const double* d_f, int mblk, double sigma,
                          double* d_u, double* d_beta, double* d_tq, hipStream_t s, PpboGate gate, const double* d_R,
                          int ldr, double* d_rv, PpboGate rider_gate, PpboDotsOut dots, int* n_dot_parts) {
  if (mblk > 64 || N % mblk != 0) return 1;      // a star's pseudo-observations fit one wavefront
  const int n_split = (N + GT_ROWS - 1) / GT_ROWS;
  double* part = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_VEC, (size_t)n_split * N * sizeof(double));
  if (!part) return (int)hipErrorOutOfMemory;
  const int gx = (N + 255) / 256;
  const int rider_y = d_R ? ((N + 3) / 4 + gx - 1) / gx : 0;      // extra workgroup rows: four rows of R per workgroup
  gemvT_beta_partial_kernel<<<dim3(gx, n_split + rider_y), 256, 0, s>>>(d_L, N, ldl, d_f, mblk, sigma, part, d_beta,
                                                                        d_tq, gate, n_split, d_R, ldr, d_rv, rider_gate);
  if (n_dot_parts) *n_dot_parts = 0;
  if (dots.partial && dots.nb >= 16 && dots.nb <= 25) {       // 32 products per column: nb basis vectors + 7 scalars
    sum_slabs_dots_kernel<<<(N + 15) / 16, 256, 0, s>>>(part, n_split, N, d_u, gate, dots);
    if (n_dot_parts) *n_dot_parts = (N + 15) / 16;
  } else {
    sum_slabs_kernel<<<(N + 15) / 16, 256, 0, s>>>(part, n_split, N, nullptr, d_u, 1, gate);
  }
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

int ppbo_gemv_async(ppbo_ctx* ctx, const double* d_T, int N, int ldt, const double* d_x, double* d_y, int trans,
                    int lower, hipStream_t s, PpboGate gate) {
  return ppbo_gemv_rect_async(ctx, d_T, N, N, ldt, d_x, nullptr, d_y, trans, lower, s, gate);
}

// y = L^-1 x / y = L^-T x with the inverse held as ppbo_trtri_async(skip_top = 1) leaves it: the two diagonal
// blocks W11 = inv(L11) ([0, split)) and W22 = inv(L22), the coupling applied through L21 itself:
//   L^-1 x = ( W11 x1 , W22 (x2 - L21 W11 x1) ),   L^-T x = ( W11^T (x1 - L21^T W22^T x2) , W22^T x2 ).
// Same bytes as one GEMV with the full inverse, two more launches, and the two largest GEMMs of the
// inversion (half its time) are never run.  d_tmp: N doubles of scratch.
int ppbo_apply_linv_async(ppbo_ctx* ctx, const double* d_Linv, int ldi, const double* d_L, int ldl, int N, int split,
                          const double* d_x, double* d_y, int trans, double* d_tmp, hipStream_t s) {
  if (split <= 0 || split >= N) return ppbo_gemv_async(ctx, d_Linv, N, ldi, d_x, d_y, trans, 1, s);
  const int n2 = N - split;
  const double* W22 = d_Linv + (size_t)split * ldi + split;
  const double* L21 = d_L + (size_t)split * ldl;
  if (!trans) {
    if (int rc = ppbo_gemv_rect_async(ctx, d_Linv, split, split, ldi, d_x, nullptr, d_y, 0, 1, s)) return rc;
    if (int rc = ppbo_gemv_rect_async(ctx, L21, n2, split, ldl, d_y, d_x + split, d_tmp, 0, 0, s)) return rc;
    return ppbo_gemv_rect_async(ctx, W22, n2, n2, ldi, d_tmp, nullptr, d_y + split, 0, 1, s);
  }
  if (int rc = ppbo_gemv_rect_async(ctx, W22, n2, n2, ldi, d_x + split, nullptr, d_y + split, 1, 1, s)) return rc;
  if (int rc = ppbo_gemv_rect_async(ctx, L21, n2, split, ldl, d_y + split, d_x, d_tmp, 1, 0, s)) return rc;
  return ppbo_gemv_rect_async(ctx, d_Linv, split, split, ldi, d_tmp, nullptr, d_y, 1, 1, s);
}

namespace {
// A[j][i] = A[i][j] for i > j: 32 x 32 tiles through LDS, one workgroup per tile of the lower triangle
__global__ __launch_bounds__(256) void mirror_lower_kernel(double* __restrict__ A, int N, int lda) {
  __shared__ double t[32][33];
  // linear index over the tiles (bi >= bj) of the lower triangle
  int bi = (int)((sqrt(8.0 * blockIdx.x + 1.0) - 1.0) * 0.5);
  while ((bi + 1) * (bi + 2) / 2 <= (int)blockIdx.x) ++bi;
  while (bi * (bi + 1) / 2 > (int)blockIdx.x) --bi;
  const int bj = blockIdx.x - bi * (bi + 1) / 2;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int i = bi * 32 + r, j = bj * 32 + tx;
    t[r][tx] = (i < N && j < N) ? A[(size_t)i * lda + j] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int j = bj * 32 + r, i = bi * 32 + tx;     // target A[j][i], strictly above the diagonal
    if (i < N && j < N && i > j) store_through(A + (size_t)j * lda + i, t[tx][r]);
  }
}
}  // namespace

// d_Ainv = Linv^T Linv for a lower-triangular Linv: only the tiles of the lower triangle are computed (the K range of a
// tile starts at its row), the upper triangle is their mirror image (exactly symmetric).  Up to N = 3072 the product runs on
// 32 x 32 tiles: it has few output tiles and long K ranges, so with 64 x 64 tiles (two wavefronts per SIMD at N = 2048)
// every chunk waits for its one-chunk-ahead prefetch -- 158 us for 3.1 GF; 32 x 32 tiles put eight wavefronts on a SIMD:
// 90 us.  Splitting K into windows summed by the mirror pass was measured too (profiles/r04_syrk_inverse_variants.txt):
// no gain on any tile size.
int ppbo_syrk_inverse_async(ppbo_ctx* ctx, const double* d_Linv, int N, double* d_Ainv, hipStream_t s) {
  GemmArgs g{};
  g.A = d_Linv; g.lda = N; g.B = d_Linv; g.ldb = N; g.C = d_Ainv; g.ldc = N;
  g.M = N; g.N = N; g.K = N; g.alpha = 1.0; g.beta = 0.0; g.klo_mode = 1; g.tri_block = 1; g.lower_only = 1;
  g.force_cfg = ctx->syrk_cfg > 0 ? ctx->syrk_cfg : (N <= 3072 ? 3 : 0);
  if (int rc = ppbo_gemm_launch(ctx, g, 1, 0, s)) return rc;
  const int nt = (N + 31) / 32;
  mirror_lower_kernel<<<nt * (nt + 1) / 2, 256, 0, s>>>(d_Ainv, N, N);
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

int ppbo_dot_async(ppbo_ctx* ctx, const double* d_x, const double* d_y, int N, double* d_out, hipStream_t s) {
  dot_kernel<<<1, 1024, 0, s>>>(d_x, d_y, N, d_out);
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

extern "C" {

int ppbo_potrf(ppbo_ctx* ctx, double* d_A, int N, int lda, int* h_info, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_A && N > 0 && lda >= N, "matrix");
  hipStream_t s = (hipStream_t)stream;
  int* d_info = (int*)ppbo_workspace(ctx, ppbo_ctx::WS_SMALL, 4096);
  if (!d_info) return (int)hipErrorOutOfMemory;
  if (int rc = ppbo_potrf_async(ctx, d_A, N, lda, d_info, s)) return rc;
  int info = 0;
  PPBO_HIP_CHECK(ctx, hipMemcpyAsync(&info, d_info, sizeof(int), hipMemcpyDeviceToHost, s));
  PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
  if (h_info) *h_info = info;
  if (info != 0) return ppbo_set_error(ctx, PPBO_ERR_NOT_PD, "matrix is not positive definite (leading minor %d)", info);
  return 0;
}

int ppbo_dgemv(ppbo_ctx* ctx, int trans, int lower, int N, const double* d_A, int lda, const double* d_x,
               double* d_y, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_A && d_x && d_y && N > 0 && lda >= N, "arguments");
  return ppbo_gemv_async(ctx, d_A, N, lda, d_x, d_y, trans, lower, (hipStream_t)stream);
}

int ppbo_pd_inverse_append(ppbo_ctx* ctx, const double* d_A, int N, const double* d_A11inv, const double* d_L11inv,
                           int N1, double* d_Ainv, double* d_Linv, int* h_info, void* stream) {
  return ppbo_pd_inverse_append_ex(ctx, d_A, N, d_A11inv, d_L11inv, nullptr, N1, d_Ainv, d_Linv, nullptr, h_info, stream);
}

int ppbo_pd_inverse_append_ex(ppbo_ctx* ctx, const double* d_A, int N, const double* d_A11inv, const double* d_L11inv,
                              const double* d_L11, int N1, double* d_Ainv, double* d_Linv, double* d_L, int* h_info,
                              void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_A && d_A11inv && d_L11inv && d_Ainv && d_Linv && N > 0, "matrix");
  PPBO_REQUIRE(ctx, (d_L == nullptr) == (d_L11 == nullptr) && (d_L == nullptr || d_L != d_L11), "d_L11 / d_L go together");
  PPBO_REQUIRE(ctx, N1 > 0 && N1 < N && N - N1 <= 64, "append at most 64 rows to a non-empty block");
  PPBO_REQUIRE(ctx, d_Ainv != d_A11inv && d_Linv != d_L11inv, "in-place append is not supported");
  hipStream_t s = (hipStream_t)stream;
  const int k = N - N1;
  const size_t nk = (size_t)N1 * k, kk = (size_t)k * k;
  double* ws = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_APPEND, (4 * nk + 3 * kk) * sizeof(double));
  if (!ws) return (int)hipErrorOutOfMemory;
  double* Y = ws;            // [N1, k]  L11^-1 A12   (= L21^T of the bordered factor)
  double* Z = Y + nk;        // [k, N1]  Y^T L11^-1
  double* X = Z + nk;        // [k, N1]  L22^-1 Z     (-X is the new bottom-left block of L^-1)
  double* Wm = X + nk;       // [N1, k]  X^T L22^-1   (-Wm is the new border of A^-1)
  double* S = Wm + nk;       // [k, k]   Schur complement A22 - Y^T Y  ->  L22 (in place)
  double* L22i = S + kk;     // [k, k]   L22^-1
  double* BR = L22i + kk;    // [k, k]   L22^-T L22^-1
  const double* A12 = d_A + N1;          // rows 0..N1-1, columns N1..N-1 of A
  if (h_info) *h_info = 0;
  auto gemm = [&](const double* A, int lda, int ta, const double* B, int ldb, int tb, double* C, int ldc, int M, int Nn,
                  int K, double alpha, double beta) {
    GemmArgs g{};
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.M = M; g.N = Nn; g.K = K; g.alpha = alpha; g.beta = beta; g.tri_block = 1;
    return ppbo_gemm_launch(ctx, g, ta, tb, s);
  };
  if (int rc = gemm(d_L11inv, N1, 0, A12, N, 0, Y, k, N1, k, N1, 1.0, 0.0)) return rc;          // Y = L11^-1 A12
  PPBO_HIP_CHECK(ctx, hipMemcpy2DAsync(S, (size_t)k * sizeof(double), d_A + (size_t)N1 * N + N1, (size_t)N * sizeof(double),
                                       (size_t)k * sizeof(double), k, hipMemcpyDeviceToDevice, s));
  if (int rc = gemm(Y, k, 1, Y, k, 0, S, k, k, k, N1, -1.0, 1.0)) return rc;                    // S = A22 - Y^T Y
  int info = 0;
  if (int rc = ppbo_potrf(ctx, S, k, k, &info, stream)) {
    if (h_info) *h_info = info ? N1 + info : 0;
    return rc;
  }
  if (int rc = ppbo_trtri_async(ctx, S, k, k, L22i, k, s)) return rc;
  if (int rc = gemm(Y, k, 1, d_L11inv, N1, 0, Z, N1, k, N1, N1, 1.0, 0.0)) return rc;           // Z = Y^T L11^-1
  if (int rc = gemm(L22i, k, 0, Z, N1, 0, X, N1, k, N1, k, 1.0, 0.0)) return rc;                // X = L22^-1 Z
  if (int rc = gemm(X, N1, 1, L22i, k, 0, Wm, k, N1, k, k, 1.0, 0.0)) return rc;                // W = X^T L22^-1
  if (int rc = gemm(L22i, k, 1, L22i, k, 0, BR, k, k, k, k, 1.0, 0.0)) return rc;               // BR = L22^-T L22^-1
  // A^-1: top-left = A11^-1 + X^T X, borders -W / -W^T, corner BR
  PPBO_HIP_CHECK(ctx, hipMemcpy2DAsync(d_Ainv, (size_t)N * sizeof(double), d_A11inv, (size_t)N1 * sizeof(double),
                                       (size_t)N1 * sizeof(double), N1, hipMemcpyDeviceToDevice, s));
  if (int rc = gemm(X, N1, 1, X, N1, 0, d_Ainv, N, N1, N1, k, 1.0, 1.0)) return rc;
  // L^-1: top-left = L11^-1, top-right 0, bottom-left -X, corner L22^-1
  PPBO_HIP_CHECK(ctx, hipMemcpy2DAsync(d_Linv, (size_t)N * sizeof(double), d_L11inv, (size_t)N1 * sizeof(double),
                                       (size_t)N1 * sizeof(double), N1, hipMemcpyDeviceToDevice, s));
  append_border_kernel<<<(N + 3) / 4, 256, 0, s>>>(Wm, BR, X, L22i, N1, k, d_Ainv, d_Linv, N);
  if (d_L) {      // the bordered factor itself: [[L11, 0], [Y^T, L22]]
    PPBO_HIP_CHECK(ctx, hipMemcpy2DAsync(d_L, (size_t)N * sizeof(double), d_L11, (size_t)N1 * sizeof(double),
                                         (size_t)N1 * sizeof(double), N1, hipMemcpyDeviceToDevice, s));
    append_factor_kernel<<<(N + 3) / 4, 256, 0, s>>>(Y, S, N1, k, d_L, N);
  }
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

int ppbo_pd_inverse_ex(ppbo_ctx* ctx, const double* d_A, int N, double* d_Ainv, double* d_L, double* d_Linv,
                       int* h_info, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_A && d_Ainv && N > 0, "matrix");
  PPBO_REQUIRE(ctx, d_L != d_A && d_L != d_Ainv && d_Linv != d_Ainv, "outputs must not alias");
  hipStream_t s = (hipStream_t)stream;
  const size_t bytes = (size_t)N * N * sizeof(double);
  double* W = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_LINALG, 2 * bytes);
  if (!W) return (int)hipErrorOutOfMemory;
  double* L = d_L ? d_L : W;                 // factor straight into the caller's buffer when it wants the factor
  double* Li = d_Linv ? d_Linv : W + (size_t)N * N;
  PPBO_HIP_CHECK(ctx, hipMemcpyAsync(L, d_A, bytes, hipMemcpyDeviceToDevice, s));
  int* d_info = (int*)ppbo_workspace(ctx, ppbo_ctx::WS_SMALL, 4096);
  if (!d_info) return (int)hipErrorOutOfMemory;
  // factor, inverse of the factor and the product are enqueued behind each other; the info word is read once, at the
  // end (a failed factorization wastes the rest: the rare case) -- one host wait per call instead of two
  if (int rc = ppbo_potrf_async(ctx, L, N, N, d_info, s)) return rc;
  if (int rc = ppbo_trtri_async(ctx, L, N, N, Li, N, s)) return rc;
  if (int rc = ppbo_syrk_inverse_async(ctx, Li, N, d_Ainv, s)) return rc;   // A^-1 = Linv^T Linv
  int info = 0;
  PPBO_HIP_CHECK(ctx, hipMemcpyAsync(&info, d_info, sizeof(int), hipMemcpyDeviceToHost, s));
  PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
  if (h_info) *h_info = info;
  if (info != 0) return ppbo_set_error(ctx, PPBO_ERR_NOT_PD, "matrix is not positive definite (leading minor %d)", info);
  return 0;
}

int ppbo_pd_inverse_factors(ppbo_ctx* ctx, const double* d_A, int N, double* d_Ainv, double* d_Linv, int* h_info,
                            void* stream) {
  return ppbo_pd_inverse_ex(ctx, d_A, N, d_Ainv, nullptr, d_Linv, h_info, stream);
}

int ppbo_pd_inverse(ppbo_ctx* ctx, const double* d_A, int N, double* d_Ainv, int* h_info, void* stream) {
  return ppbo_pd_inverse_ex(ctx, d_A, N, d_Ainv, nullptr, nullptr, h_info, stream);
}

}  // extern "C"
