// fp64 MFMA tile engine (v_mfma_f64_16x16x4_f64) shared by the dense kernels.
//
// A workgroup of WM x WN wavefronts owns a BM x BN output tile (BM = WM*TM*16,
// BN = WN*TN*16); each wavefront owns TM x TN MFMA tiles (4 fp64 accumulators per lane
// and tile).  The K dimension is consumed in chunks of 16 through double-buffered LDS
// with register-staged prefetch (global loads of chunk k+1 are in flight while the MFMAs
// of chunk k issue).  Measured on MI355X (tools/mfma_f64_bench.hip): one wavefront alone
// issues an fp64 MFMA only every ~138 cycles; the 64-cycle pipe fills only with several
// wavefronts per SIMD, so small per-wave tiles (few accumulator VGPRs, 4+ waves/SIMD)
// beat large ones.
//
// Operand storage in HBM is described per operand:
//   KC  "k contiguous"  : element (r, k) at base[r*ld + k]   (row-major [rows][K])
//   RC  "row contiguous": element (r, k) at base[k*ld + r]   (row-major [K][rows])
// LDS images keep the source orientation and are padded so that the MFMA
// fragment read (lane l -> row l&15, k l>>4) is bank-conflict free:
//   KC image [rows][16+2]  : dword bank = (36 r + 2 k) mod 64  -> 64 distinct banks / 32 lanes
//   RC image [16][rows+16] : row stride = 32 dwords mod 64 -> two k rows use disjoint halves
// f64 MFMA fragment maps (cdna_hip_programming.md s3): A[l&15][l>>4], B[l>>4][l&15],
// C/D: col = l&15, row = (l>>4) + 4*reg.
#pragma once
#include "common.h"

namespace gemm64 {

constexpr int BK = 16;
constexpr int KC_LD = BK + 2;  // 18

enum Layout { KC = 0, RC = 1 };

template <int WM_, int WN_, int TM_, int TN_>
struct Cfg {
  static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
  static constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  static constexpr int NT = 64 * WM * WN;
  // an image must hold either orientation: KC [rows][18] or RC [16][rows+16] (equal only at rows = 128)
  static constexpr int IMG_A = (BM * KC_LD > BK * (BM + 16)) ? BM * KC_LD : BK * (BM + 16);
  static constexpr int IMG_B = (BN * KC_LD > BK * (BN + 16)) ? BN * KC_LD : BK * (BN + 16);
  static constexpr int LDS_DOUBLES = 2 * (IMG_A + IMG_B);
  static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "tile must split evenly over the threads");
};

// All LDS traffic goes through this symbol with integer offsets so the compiler keeps the
// accesses in the LDS address space (ds_read_b64 / ds_write_b128).  Routing the image
// pointers through a runtime-indexed pointer array made hipcc fall back to flat_load with a
// combined vmcnt(0)/lgkmcnt(0) wait in front of every MFMA group (-45 % throughput).
extern __shared__ __attribute__((aligned(16))) double lds_dyn[];

// thread -> (row, k) of its p-th double2 inside a [ROWS x 16] operand tile
template <int LAY, int ROWS, int NT>
__device__ __forceinline__ void tile_coord(int p, int& r, int& k) {
  const int t = threadIdx.x;
  if (LAY == KC) { r = p * (NT / 8) + (t >> 3); k = (t & 7) * 2; }
  else { constexpr int TPR = ROWS / 2; k = p * (NT / TPR) + t / TPR; r = (t % TPR) * 2; }
}

template <int LAY, int ROWS, int NT, bool FAST>
__device__ __forceinline__ double2 load_elem(const double* __restrict__ base, int ld, int r0, int k0, int rows,
                                             int K, int p) {
  int r, k;
  tile_coord<LAY, ROWS, NT>(p, r, k);
  const double* src = (LAY == KC) ? base + (size_t)(r0 + r) * ld + (k0 + k) : base + (size_t)(k0 + k) * ld + (r0 + r);
  if (FAST) return *reinterpret_cast<const double2*>(src);
  double2 v;
  if (LAY == KC) {
    const bool rok = (r0 + r) < rows;
    v.x = (rok && (k0 + k) < K) ? src[0] : 0.0;
    v.y = (rok && (k0 + k + 1) < K) ? src[1] : 0.0;
  } else {
    const bool kok = (k0 + k) < K;
    v.x = (kok && (r0 + r) < rows) ? src[0] : 0.0;
    v.y = (kok && (r0 + r + 1) < rows) ? src[1] : 0.0;
  }
  return v;
}

template <int LAY, int ROWS, int NT>
__device__ __forceinline__ void store_elem(int img, int p, double2 v) {
  int r, k;
  tile_coord<LAY, ROWS, NT>(p, r, k);
  const int off = (LAY == KC) ? r * KC_LD + k : k * (ROWS + 16) + r;
  *reinterpret_cast<double2*>(&lds_dyn[img + off]) = v;
}

// whole K range [kbeg, kend) of this operand tile is in bounds and 16-byte aligned
template <int ROWS>
__device__ __forceinline__ bool tile_fast(const double* base, int ld, int r0, int kbeg, int kend, int rows) {
  const bool inb = (r0 + ROWS <= rows) && (((kend - kbeg) % BK) == 0);
  const bool al = ((ld & 1) == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0) && ((r0 & 1) == 0) &&
                  ((kbeg & 1) == 0);
  return inb && al;
}

// per-lane LDS offset (doubles) of fragment element (row r, k) relative to the image start
template <int LAY, int ROWS>
__device__ __forceinline__ int frag_off(int r, int k) {
  return (LAY == KC) ? r * KC_LD + k : k * (ROWS + 16) + r;
}

// k_wave_end: this wavefront's rows of A are zero for k >= k_wave_end (block-triangular A): its MFMAs beyond
// that depth are skipped (wave-uniform branch); the other workgroup on the CU uses the freed matrix-core slots.
template <class C, int ALAY, int BLAY, bool FAST>
__device__ __forceinline__ void mainloop_impl(const double* __restrict__ A, int lda, const double* __restrict__ B,
                                              int ldb, int M, int N, int m0, int n0, int kbeg, int kend,
                                              double4_t acc[C::TM][C::TN], int k_wave_end) {
  constexpr int BUF = C::IMG_A + C::IMG_B;   // buffer b: A image at b*BUF, B image at b*BUF + IMG_A
  constexpr int PA = C::BM * 8 / C::NT, PB = C::BN * 8 / C::NT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / C::WN, wn = wave % C::WN;
  const int lr = lane & 15, lk = lane >> 4;
  const int nk = (kend - kbeg + BK - 1) / BK;
  // lane-constant LDS offsets; everything else in the loop is an immediate
  const int a_lane = frag_off<ALAY, C::BM>(wm * C::TM * 16 + lr, lk);
  const int b_lane = C::IMG_A + frag_off<BLAY, C::BN>(wn * C::TN * 16 + lr, lk);
  constexpr int A_I = (ALAY == KC) ? 16 * KC_LD : 16;              // next 16-row tile
  constexpr int B_J = (BLAY == KC) ? 16 * KC_LD : 16;
  constexpr int A_K = (ALAY == KC) ? 4 : 4 * (C::BM + 16);         // next k-step of 4
  constexpr int B_K = (BLAY == KC) ? 4 : 4 * (C::BN + 16);
  // register staging (plain arrays, fully unrolled: they must stay in VGPRs)
  double2 ra[PA], rb[PB];
#pragma unroll
  for (int p = 0; p < PA; ++p) ra[p] = load_elem<ALAY, C::BM, C::NT, FAST>(A, lda, m0, kbeg, M, kend, p);
#pragma unroll
  for (int p = 0; p < PB; ++p) rb[p] = load_elem<BLAY, C::BN, C::NT, FAST>(B, ldb, n0, kbeg, N, kend, p);
#pragma unroll
  for (int p = 0; p < PA; ++p) store_elem<ALAY, C::BM, C::NT>(0, p, ra[p]);
#pragma unroll
  for (int p = 0; p < PB; ++p) store_elem<BLAY, C::BN, C::NT>(C::IMG_A, p, rb[p]);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = (kt & 1) * BUF;
    // Prefetch chunk kt+1 (the last iteration re-reads its own chunk: harmless, keeps the
    // loop body branch-free so the staging registers are never demoted to scratch).
    const int k0 = kbeg + ((kt + 1 < nk) ? (kt + 1) : kt) * BK;
#pragma unroll
    for (int p = 0; p < PA; ++p) ra[p] = load_elem<ALAY, C::BM, C::NT, FAST>(A, lda, m0, k0, M, kend, p);
#pragma unroll
    for (int p = 0; p < PB; ++p) rb[p] = load_elem<BLAY, C::BN, C::NT, FAST>(B, ldb, n0, k0, N, kend, p);
    // register double-buffered fragments: the LDS reads of k-step kk+1 are in flight while
    // the MFMAs of k-step kk issue
    const double* la = lds_dyn + cur + a_lane;
    const double* lb = lds_dyn + cur + b_lane;
    const int kt0 = kbeg + kt * BK;
    if (kt0 < k_wave_end) {
      const bool full = (kt0 + BK <= k_wave_end);
      double a0[C::TM], b0[C::TN], a1[C::TM], b1[C::TN];
#pragma unroll
      for (int i = 0; i < C::TM; ++i) a0[i] = la[i * A_I];
#pragma unroll
      for (int j = 0; j < C::TN; ++j) b0[j] = lb[j * B_J];
#pragma unroll
      for (int i = 0; i < C::TM; ++i) a1[i] = la[A_K + i * A_I];
#pragma unroll
      for (int j = 0; j < C::TN; ++j) b1[j] = lb[B_K + j * B_J];
#pragma unroll
      for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[i], b0[j], acc[i][j], 0, 0, 0);
      if (full || kt0 + 4 < k_wave_end) {
#pragma unroll
        for (int i = 0; i < C::TM; ++i) a0[i] = la[2 * A_K + i * A_I];
#pragma unroll
        for (int j = 0; j < C::TN; ++j) b0[j] = lb[2 * B_K + j * B_J];
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
          for (int j = 0; j < C::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[i], b1[j], acc[i][j], 0, 0, 0);
      }
      if (full || kt0 + 8 < k_wave_end) {
#pragma unroll
        for (int i = 0; i < C::TM; ++i) a1[i] = la[3 * A_K + i * A_I];
#pragma unroll
        for (int j = 0; j < C::TN; ++j) b1[j] = lb[3 * B_K + j * B_J];
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
          for (int j = 0; j < C::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[i], b0[j], acc[i][j], 0, 0, 0);
      }
      if (full || kt0 + 12 < k_wave_end) {
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
          for (int j = 0; j < C::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[i], b1[j], acc[i][j], 0, 0, 0);
      }
    }
    // the other buffer was last read in iteration kt-1 (barrier since then): safe to overwrite
#pragma unroll
    for (int p = 0; p < PA; ++p) store_elem<ALAY, C::BM, C::NT>(BUF - cur, p, ra[p]);
#pragma unroll
    for (int p = 0; p < PB; ++p) store_elem<BLAY, C::BN, C::NT>(BUF - cur + C::IMG_A, p, rb[p]);
    __syncthreads();
  }
}

// ---- the lean main loop (every tile of the launch in bounds and aligned) --------------------------------------
// On CDNA4 a wavefront's non-MFMA VALU instructions do not overlap with the fp64 MFMAs of its SIMD (measured: one
// extra VALU op per MFMA costs 7 % of the matrix-core rate), and the generic loop above spends ~30 of them per
// 16-deep chunk on 64-bit global addresses, LDS buffer toggling and per-lane tests of the wave-uniform K limit
// (PMC: 1.03 VALU per MFMA; 0.85 of the MFMA peak).  Here the loop body has none:
//   * operand tiles are fetched with buffer loads: the descriptor's base (SGPRs) walks along K on the SALU, the
//     per-thread byte offset is a 32-bit VGPR fixed before the loop;
//   * the loop is unrolled over the two LDS buffers, the B image comes first in a buffer so that every fragment of
//     either buffer is within the 16-bit immediate of ds_read_b64 from ONE lane-constant base per operand (reads are
//     volatile so that the load/store optimizer does not pair them into ds_read2 + a VALU base adjustment);
//   * the K limit of the wavefront is a scalar.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int LAY, int ROWS, int NT>
__device__ __forceinline__ unsigned elem_byte_off(int ld, int p) {
  int r, k;
  tile_coord<LAY, ROWS, NT>(p, r, k);
  return (unsigned)(((LAY == KC) ? r * ld + k : k * ld + r) * 8);
}

__device__ __forceinline__ double2 buffer_load2(const double* __restrict__ sbase, unsigned byte_off) {
  // raw buffer, stride 0, no bound (the host checked the tiles): word 3 = 32-bit data format (CDNA3/4 encoding)
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(sbase), 0, -1, 0x00020000);
  return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}

typedef const volatile double __attribute__((address_space(3))) * lds_vptr;
__device__ __forceinline__ double lds_vread(int off_doubles) {
  return *((lds_vptr)(&lds_dyn[0]) + off_doubles);      // stays a ds_read_b64: the cast keeps the LDS address space
}

template <class C, int ALAY, int BLAY, int CUR>
__device__ __forceinline__ void lean_chunk(const double* __restrict__ pa, const double* __restrict__ pb,
                                           const unsigned (&offA)[C::BM * 8 / C::NT], const unsigned (&offB)[C::BN * 8 / C::NT],
                                           int a_lane, int b_lane, int kleft, double4_t acc[C::TM][C::TN]) {
  constexpr int BUF = C::IMG_A + C::IMG_B;          // lean layout of buffer b: B image at b*BUF, A image behind it
  constexpr int PA = C::BM * 8 / C::NT, PB = C::BN * 8 / C::NT;
  constexpr int A_I = (ALAY == KC) ? 16 * KC_LD : 16;
  constexpr int B_J = (BLAY == KC) ? 16 * KC_LD : 16;
  constexpr int A_K = (ALAY == KC) ? 4 : 4 * (C::BM + 16);
  constexpr int B_K = (BLAY == KC) ? 4 : 4 * (C::BN + 16);
  constexpr int RA = CUR * BUF + C::IMG_B, RB = CUR * BUF;              // read from this buffer ...
  constexpr int WA = (1 - CUR) * BUF + C::IMG_B, WB = (1 - CUR) * BUF;  // ... stage the next chunk into the other
  double2 ra[PA], rb[PB];
#pragma unroll
  for (int p = 0; p < PA; ++p) ra[p] = buffer_load2(pa, offA[p]);
#pragma unroll
  for (int p = 0; p < PB; ++p) rb[p] = buffer_load2(pb, offB[p]);
  if (kleft > 0) {                     // scalar: this wavefront's rows of A are zero from k_wave_end on
    double a0[C::TM], b0[C::TN], a1[C::TM], b1[C::TN];
#pragma unroll
    for (int i = 0; i < C::TM; ++i) a0[i] = lds_vread(a_lane + RA + i * A_I);
#pragma unroll
    for (int j = 0; j < C::TN; ++j) b0[j] = lds_vread(b_lane + RB + j * B_J);
#pragma unroll
    for (int i = 0; i < C::TM; ++i) a1[i] = lds_vread(a_lane + RA + A_K + i * A_I);
#pragma unroll
    for (int j = 0; j < C::TN; ++j) b1[j] = lds_vread(b_lane + RB + B_K + j * B_J);
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
      for (int j = 0; j < C::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[i], b0[j], acc[i][j], 0, 0, 0);
    if (kleft > 4) {
#pragma unroll
      for (int i = 0; i < C::TM; ++i) a0[i] = lds_vread(a_lane + RA + 2 * A_K + i * A_I);
#pragma unroll
      for (int j = 0; j < C::TN; ++j) b0[j] = lds_vread(b_lane + RB + 2 * B_K + j * B_J);
#pragma unroll
      for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[i], b1[j], acc[i][j], 0, 0, 0);
    }
    if (kleft > 8) {
#pragma unroll
      for (int i = 0; i < C::TM; ++i) a1[i] = lds_vread(a_lane + RA + 3 * A_K + i * A_I);
#pragma unroll
      for (int j = 0; j < C::TN; ++j) b1[j] = lds_vread(b_lane + RB + 3 * B_K + j * B_J);
#pragma unroll
      for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[i], b0[j], acc[i][j], 0, 0, 0);
    }
    // the next chunk goes to the other buffer BEFORE the last MFMA group: the stores' latency and the wait at the
    // barrier run under those eight MFMAs instead of after them
#pragma unroll
    for (int p = 0; p < PA; ++p) store_elem<ALAY, C::BM, C::NT>(WA, p, ra[p]);
#pragma unroll
    for (int p = 0; p < PB; ++p) store_elem<BLAY, C::BN, C::NT>(WB, p, rb[p]);
    if (kleft > 12) {
#pragma unroll
      for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[i], b1[j], acc[i][j], 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int p = 0; p < PA; ++p) store_elem<ALAY, C::BM, C::NT>(WA, p, ra[p]);
#pragma unroll
    for (int p = 0; p < PB; ++p) store_elem<BLAY, C::BN, C::NT>(WB, p, rb[p]);
  }
  __syncthreads();
}

template <class C, int ALAY, int BLAY>
__device__ __forceinline__ void mainloop_lean(const double* __restrict__ A, int lda, const double* __restrict__ B,
                                              int ldb, int m0, int n0, int kbeg, int kend,
                                              double4_t acc[C::TM][C::TN], int k_wave_end) {
  constexpr int PA = C::BM * 8 / C::NT, PB = C::BN * 8 / C::NT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / C::WN, wn = wave % C::WN;
  const int lr = lane & 15, lk = lane >> 4;
  const int nk = (kend - kbeg) / BK;
  const int a_lane = frag_off<ALAY, C::BM>(wm * C::TM * 16 + lr, lk);
  const int b_lane = frag_off<BLAY, C::BN>(wn * C::TN * 16 + lr, lk);
  unsigned offA[PA], offB[PB];
#pragma unroll
  for (int p = 0; p < PA; ++p) offA[p] = elem_byte_off<ALAY, C::BM, C::NT>(lda, p);
#pragma unroll
  for (int p = 0; p < PB; ++p) offB[p] = elem_byte_off<BLAY, C::BN, C::NT>(ldb, p);
  // scalar tile origins and per-chunk strides (doubles)
  const double* pa = (ALAY == KC) ? A + (size_t)m0 * lda + kbeg : A + (size_t)kbeg * lda + m0;
  const double* pb = (BLAY == KC) ? B + (size_t)n0 * ldb + kbeg : B + (size_t)kbeg * ldb + n0;
  const size_t sa = (ALAY == KC) ? (size_t)BK : (size_t)BK * lda;
  const size_t sb = (BLAY == KC) ? (size_t)BK : (size_t)BK * ldb;
  int kleft = __builtin_amdgcn_readfirstlane(k_wave_end < kend ? k_wave_end : kend) - kbeg;   // depth this wavefront still multiplies
  {
    double2 ra[PA], rb[PB];
#pragma unroll
    for (int p = 0; p < PA; ++p) ra[p] = buffer_load2(pa, offA[p]);
#pragma unroll
    for (int p = 0; p < PB; ++p) rb[p] = buffer_load2(pb, offB[p]);
#pragma unroll
    for (int p = 0; p < PA; ++p) store_elem<ALAY, C::BM, C::NT>(C::IMG_B, p, ra[p]);
#pragma unroll
    for (int p = 0; p < PB; ++p) store_elem<BLAY, C::BN, C::NT>(0, p, rb[p]);
  }
  __syncthreads();
  // chunk kt computes from buffer kt & 1 and stages chunk kt + 1 (the last one re-reads itself: branch-free body)
  for (int kt = 0; kt < nk; kt += 2) {
    if (kt + 1 < nk) { pa += sa; pb += sb; }
    lean_chunk<C, ALAY, BLAY, 0>(pa, pb, offA, offB, a_lane, b_lane, kleft, acc);
    kleft -= BK;
    if (kt + 1 >= nk) break;
    if (kt + 2 < nk) { pa += sa; pb += sb; }
    lean_chunk<C, ALAY, BLAY, 1>(pa, pb, offA, offB, a_lane, b_lane, kleft, acc);
    kleft -= BK;
  }
}

// ALWAYS_FAST: the host has verified that every tile of the launch is in bounds and aligned,
// so the guarded (scalar, zero-filling) loop is not even compiled into the kernel.
template <class C, int ALAY, int BLAY, bool ALWAYS_FAST = false>
__device__ __forceinline__ void mainloop(const double* __restrict__ A, int lda, const double* __restrict__ B,
                                         int ldb, int M, int N, int m0, int n0, int kbeg, int kend,
                                         double4_t acc[C::TM][C::TN], int k_wave_end = 0x7fffffff) {
  if (kend <= kbeg) return;
  // the lean loop addresses a tile by 32-bit byte offsets: (tile rows or 16) * ld * 8 must stay below 2^31
  const bool small_ld = (size_t)((ALAY == KC) ? C::BM : BK) * (size_t)lda < ((size_t)1 << 28) &&
                        (size_t)((BLAY == KC) ? C::BN : BK) * (size_t)ldb < ((size_t)1 << 28);
  if (ALWAYS_FAST) {
    if (small_ld) mainloop_lean<C, ALAY, BLAY>(A, lda, B, ldb, m0, n0, kbeg, kend, acc, k_wave_end);
    else mainloop_impl<C, ALAY, BLAY, true>(A, lda, B, ldb, M, N, m0, n0, kbeg, kend, acc, k_wave_end);
    return;
  }
  const bool fast = tile_fast<C::BM>(A, lda, m0, kbeg, kend, M) && tile_fast<C::BN>(B, ldb, n0, kbeg, kend, N);
  if (fast && small_ld) mainloop_lean<C, ALAY, BLAY>(A, lda, B, ldb, m0, n0, kbeg, kend, acc, k_wave_end);
  else if (fast) mainloop_impl<C, ALAY, BLAY, true>(A, lda, B, ldb, M, N, m0, n0, kbeg, kend, acc, k_wave_end);
  else mainloop_impl<C, ALAY, BLAY, false>(A, lda, B, ldb, M, N, m0, n0, kbeg, kend, acc, k_wave_end);
}

template <class C>
__device__ __forceinline__ void zero_acc(double4_t acc[C::TM][C::TN]) {
#pragma unroll
  for (int i = 0; i < C::TM; ++i)
#pragma unroll
    for (int j = 0; j < C::TN; ++j) acc[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};
}

// element (row, col) of accumulator acc[i][j][r] inside the BM x BN tile
template <class C>
__device__ __forceinline__ int acc_row(int i, int r) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return ((wave / C::WN) * C::TM + i) * 16 + (lane >> 4) + 4 * r;
}
template <class C>
__device__ __forceinline__ int acc_col(int j) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return ((wave % C::WN) * C::TN + j) * 16 + (lane & 15);
}

}  // namespace gemm64
