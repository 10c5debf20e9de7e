// fp64 MFMA tile engine (v_mfma_f64_16x16x4_f64) shared by the dense kernels.
//
// One 256-thread workgroup (4 wavefronts, 2x2) owns a 128x128 output tile; each
// wavefront owns 64x64 = 4x4 MFMA tiles (64 fp64 accumulators per lane).  The
// K dimension is consumed in chunks of 16 through double-buffered LDS with
// register-staged prefetch (global loads of chunk k+1 are in flight while the
// MFMAs of chunk k issue).
//
// Operand storage in HBM is described per operand:
//   KC  "k contiguous"  : element (r, k) at base[r*ld + k]   (row-major [rows][K])
//   RC  "row contiguous": element (r, k) at base[k*ld + r]   (row-major [K][rows])
// LDS images keep the source orientation and are padded so that the MFMA
// fragment read (lane l -> row l&15, k l>>4) is bank-conflict free:
//   KC image [128][16+2]  : dword bank = (36 r + 2 k) mod 64  -> 64 distinct banks / 32 lanes
//   RC image [16][128+16] : row stride 288 dwords = 32 mod 64 -> two k rows use disjoint halves
// f64 MFMA fragment maps (cdna_hip_programming.md s3): A[l&15][l>>4], B[l>>4][l&15],
// C/D: col = l&15, row = (l>>4) + 4*reg.
#pragma once
#include "common.h"

namespace gemm64 {

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int KC_LD = BK + 2;     // 18
constexpr int RC_LD = BM + 16;    // 144
constexpr int IMG = BM * KC_LD;   // 2304 doubles == BK * RC_LD
static_assert(BM * KC_LD == BK * RC_LD, "both images have the same footprint");
constexpr int LDS_DOUBLES = 4 * IMG;  // A,B x 2 buffers = 73,728 bytes

enum Layout { KC = 0, RC = 1 };

struct Stage {  // 8 doubles of one operand tile per thread
  double2 v[4];
};

// Load this thread's share of a [128 rows x 16 k] operand tile into registers.
template <int LAY>
__device__ __forceinline__ void load_tile(const double* __restrict__ base, int ld, int r0, int k0, int rows,
                                          int K, bool fast, Stage& st) {
  const int t = threadIdx.x;
  if (LAY == KC) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = p * 32 + (t >> 3), k = (t & 7) * 2;
      const double* src = base + (size_t)(r0 + r) * ld + (k0 + k);
      if (fast) {
        st.v[p] = *reinterpret_cast<const double2*>(src);
      } else {
        const bool rok = (r0 + r) < rows;
        st.v[p].x = (rok && (k0 + k) < K) ? src[0] : 0.0;
        st.v[p].y = (rok && (k0 + k + 1) < K) ? src[1] : 0.0;
      }
    }
  } else {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int k = p * 4 + (t >> 6), r = (t & 63) * 2;
      const double* src = base + (size_t)(k0 + k) * ld + (r0 + r);
      if (fast) {
        st.v[p] = *reinterpret_cast<const double2*>(src);
      } else {
        const bool kok = (k0 + k) < K;
        st.v[p].x = (kok && (r0 + r) < rows) ? src[0] : 0.0;
        st.v[p].y = (kok && (r0 + r + 1) < rows) ? src[1] : 0.0;
      }
    }
  }
}

template <int LAY>
__device__ __forceinline__ void store_tile(double* __restrict__ img, const Stage& st) {
  const int t = threadIdx.x;
  if (LAY == KC) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = p * 32 + (t >> 3), k = (t & 7) * 2;
      *reinterpret_cast<double2*>(img + r * KC_LD + k) = st.v[p];
    }
  } else {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int k = p * 4 + (t >> 6), r = (t & 63) * 2;
      *reinterpret_cast<double2*>(img + k * RC_LD + r) = st.v[p];
    }
  }
}

template <int LAY>
__device__ __forceinline__ double frag(const double* __restrict__ img, int r, int k) {
  return (LAY == KC) ? img[r * KC_LD + k] : img[k * RC_LD + r];
}

template <int LAY>
__device__ __forceinline__ bool tile_fast(const double* base, int ld, int r0, int k0, int rows, int K) {
  const bool inb = (r0 + BM <= rows) && (k0 + BK <= K);
  const bool al = ((ld & 1) == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0) && ((r0 & 1) == 0) &&
                  ((k0 & 1) == 0);
  return inb && al;
}

// acc[ti][tj] += A[m0.., kbeg..kend) * B[kbeg..kend), n0..]
template <int ALAY, int BLAY>
__device__ __forceinline__ void mainloop(const double* __restrict__ A, int lda, const double* __restrict__ B,
                                         int ldb, int M, int N, int K, int m0, int n0, int kbeg, int kend,
                                         double* __restrict__ lds, double4_t acc[4][4]) {
  double* Aimg[2] = {lds, lds + 2 * IMG};
  double* Bimg[2] = {lds + IMG, lds + 3 * IMG};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lk = lane >> 4;

  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk <= 0) return;
  Stage sa, sb;
  load_tile<ALAY>(A, lda, m0, kbeg, M, kend, tile_fast<ALAY>(A, lda, m0, kbeg, M, kend), sa);
  load_tile<BLAY>(B, ldb, n0, kbeg, N, kend, tile_fast<BLAY>(B, ldb, n0, kbeg, N, kend), sb);
  store_tile<ALAY>(Aimg[0], sa);
  store_tile<BLAY>(Bimg[0], sb);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = (kt + 1) < nk;
    if (more) {
      const int k0 = kbeg + (kt + 1) * BK;
      load_tile<ALAY>(A, lda, m0, k0, M, kend, tile_fast<ALAY>(A, lda, m0, k0, M, kend), sa);
      load_tile<BLAY>(B, ldb, n0, k0, N, kend, tile_fast<BLAY>(B, ldb, n0, k0, N, kend), sb);
    }
    const double* ai = Aimg[cur];
    const double* bi = Bimg[cur];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = frag<ALAY>(ai, wm * 64 + i * 16 + lr, kk * 4 + lk);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = frag<BLAY>(bi, wn * 64 + j * 16 + lr, kk * 4 + lk);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      store_tile<ALAY>(Aimg[cur ^ 1], sa);
      store_tile<BLAY>(Bimg[cur ^ 1], sb);
    }
    __syncthreads();
  }
}

__device__ __forceinline__ void zero_acc(double4_t acc[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};
}

// element (row, col) of accumulator acc[i][j][r] inside the 128x128 tile
__device__ __forceinline__ int acc_row(int i, int r) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return (wave >> 1) * 64 + i * 16 + (lane >> 4) + 4 * r;
}
__device__ __forceinline__ int acc_col(int j) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return (wave & 1) * 64 + j * 16 + (lane & 15);
}

}  // namespace gemm64
