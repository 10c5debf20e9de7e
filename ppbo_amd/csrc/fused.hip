// K2+K3+K4 in ONE launch for models of up to ~1024 rows (BASELINE configs C1, C2, C4 and every early iteration of a
// PPBO run: N = (m+1) n_q <= ~1000 for the first 40 queries, src/ppbo_settings.py:14).
//   reference: gp_model.py:441-452 (mu_Sigma_pred: K*, mu = K*' alpha, diag(Sigma_pred) = sigma_f^2 - k*' A k*)
//
// The three-launch form (predict.hip: kstar -> quadform -> score) writes K* [N, M] to HBM and reads it back once per
// 128-row panel of G; at N = 2048 that hides behind the matrix cores, at N = 512 it does not: K* build 0.024 ms,
// score 0.011 ms, three launch boundaries and the round trip cost as much as the contraction (0.132 ms per step at C2
// against a 0.058 ms MFMA bound).  Here a workgroup OWNS 32 candidates from their coordinates to their score:
//   1. K* phase: the workgroup forms its [rows, 32] block of K* in LDS, one panel of at most KP = 32 NW rows at a time
//      (NW = wavefronts per workgroup).  r^2 = |x|^2 + |c|^2 - 2 x.c with the x.c of a 16-row x 16-candidate tile on the
//      matrix cores (operands straight from global memory in fragment layout), exp on the vector ALUs; mu = K*' alpha and
//      the diagonal part of k*' Lambda k* accumulate on the way, the per-row parameters (|x|^2, alpha, Lambda) having
//      been staged once per panel in the panel's four padding columns.  The star-edge part of k*' Lambda k* follows from
//      the finished panel (thread = candidate x 16 rows, the observation row's value riding along in a register).
//   2. contraction phase (fp64 MFMA): Y = G K* with the B operand read from LDS and the A operand loaded STRAIGHT into
//      MFMA fragment registers from the TRANSPOSE of G (L2-resident): a wavefront owns strips of 32 rows of G; lane
//      (r, q) loads the 16 bytes Gt[k0 + 4 q + s][r0 + 2 r .. + 1] -- its .x feeds the strip's even rows (row tile 0),
//      its .y the odd rows (row tile 1) -- for s = 0..3, and the s-th group of four MFMAs takes K*[k0 + 4 q + s][.] as
//      its B fragment: the contraction index may be visited in any order as long as both operands agree, so there is
//      no LDS staging of G, no barrier and no VALU instruction in the main loop (one v_add per 32 MFMAs for the LDS
//      address).  A load instruction covers four rows of Gt x 256 contiguous bytes; in G's own row-major layout every
//      lane of a load touched another, 4 KB-distant row and the vector L1's tag rate set the pace (measured: 85 us of
//      contraction at C2 with the loads, 67 without).  Row stride of the panel = 36 doubles: the rows 4 q + s of lanes
//      q = 0, 1 (one ds_read_b64 lane group) fall on disjoint halves of the 64 banks.  sum Y^2 per candidate stays in
//      registers.
//   3. finish: sigma^2 = sigma_f^2 + k*' Lambda k* + |G k*|^2, the score, the block's best -- then the one-workgroup
//      argmax of predict.hip.  No K* in HBM, no slab pass: one launch (+ the transpose of G, 2-4 us) + the argmax.
// Rows beyond one panel: TWO passes split at a STAR boundary R1 = floor(KP / (m+1)) (m+1).  Pass 1 holds K* rows [0, R1)
// and contracts them with the "lower" strips (rows < R1: block-triangular G ends their K range inside the pass) and with
// the "upper" strips (rows >= R1), whose accumulators stay in registers while pass 2 forms K* rows [R1, N) and finishes
// them.  A wavefront owns one lower and one upper strip, paired short with long (equal K chunks per wavefront).
// Up to 256 + rows a workgroup is 8 wavefronts with a 256-row panel (76 KB of LDS): two workgroups share a CU and one's
// K* phase runs under the other's MFMAs.  Above that, 16 wavefronts and a 512-row panel (149 KB), one workgroup per CU.
// Larger models, the camphor kernel and the fp32-K* report keep the three-launch form.
#include <cstdio>
#include <type_traits>
#include <vector>

#include "gemm_f64.h"
#include "score.h"

namespace {

using gemm64::buffer_load2;
using gemm64::lds_dyn;
using gemm64::lds_vread;

constexpr int FS_BN = 32;      // candidates per workgroup
constexpr int FS_LD = 36;      // row stride of the K* panel (doubles), = 4 mod 8

struct FusedArgs {
  const double* X; int N, D; KernParams p;
  const double* alpha; const double* lam_diag; const double* lam_off; int mblk;
  const double* Xc; long long M;
  const double* Gt; int ldgt;            // Gt[k][i] = G[i][k] framed with zeros (ppbo_fused_transposed_G)
  int R1;                                // rows of the first pass (a star boundary; N when one pass holds every row)
  double sf2; int kind; double mustar; long long idx_base;
  double* mu_out; double* var_out; double* score_out; Best* blk_best;
  int ncu, delay;                        // CUs of the device; start delay of the odd dispatch rounds (x 3.4 us)
  unsigned long long* stamps;            // dbg bit 3: [blocks][16] s_memrealtime stamps of wavefront 0 at the phase boundaries
  int dbg;      // PPBO_FUSED_DBG (measurement only): bit 0 = no kernel evaluations, bit 1 = no contraction,
                // bit 2 = the contraction re-uses its first four chunks of G (no loads in its loop), bit 3 = phase stamps
};

// A fragments of one 16-deep chunk of a 16-row strip: f[s] = Gt[k0 + 4 q + s][r0 + r] for lane (r, q)
struct AFrag { double f[4]; };
__device__ __forceinline__ double fs_ld8(const double* __restrict__ sbase, unsigned voff, unsigned soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(sbase), 0, -1, 0x00020000);
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ AFrag fs_load(const double* __restrict__ pa, unsigned voff, unsigned row_bytes) {
  AFrag f;
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) f.f[s4] = fs_ld8(pa, voff, s4 * row_bytes);
  return f;
}

// one 16-deep chunk of a strip: 8 MFMAs (2 column tiles x 4 k-steps) on the A fragments, B fragments from the panel at lb
// (doubles).  The B reads run one k-step ahead of the MFMAs THROUGH the chunk boundaries: (b0, b1) come in loaded for this
// chunk's first step and go out loaded for the first step of the chunk at lb_next.
__device__ __forceinline__ void fs_chunk(const AFrag& f, int lb, int lb_next, double& b0, double& b1, double4_t (&acc)[2]) {
  double n0 = lds_vread(lb + FS_LD), n1 = lds_vread(lb + FS_LD + 16);
  acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.f[0], b0, acc[0], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.f[0], b1, acc[1], 0, 0, 0);
  b0 = lds_vread(lb + 2 * FS_LD); b1 = lds_vread(lb + 2 * FS_LD + 16);
  acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.f[1], n0, acc[0], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.f[1], n1, acc[1], 0, 0, 0);
  n0 = lds_vread(lb + 3 * FS_LD); n1 = lds_vread(lb + 3 * FS_LD + 16);
  acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.f[2], b0, acc[0], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.f[2], b1, acc[1], 0, 0, 0);
  b0 = lds_vread(lb_next); b1 = lds_vread(lb_next + 16);
  acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.f[3], n0, acc[0], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.f[3], n1, acc[1], 0, 0, 0);
  // the order above IS the schedule: the two reads of the NEXT k-step, then the two MFMAs of the current one (left alone,
  // the scheduler moves every read down to just above its MFMA and the LDS latency shows at two wavefronts per SIMD)
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // MFMA
  }
}

// acc += G[strip rows][kb, ke) K*[kb, ke)[32 candidates] for a strip of 16 rows; ke - kb a positive multiple of 16.
// gt = &Gt[0][r0] (wave-uniform: the buffer descriptor's base walks down the K rows on the SALU), ldgt = row stride of Gt
// (doubles), voff = this lane's byte offset (4 q ldgt + r) * 8, lb = this lane's panel offset (doubles) of chunk kb.
// FOUR chunks of G are in flight per wavefront: a chunk is only 8 MFMAs (512 matrix-core cycles; ~2000 with the other
// three wavefronts of the SIMD in between), less than an L2 round trip under load.  The scheduler sinks a load to just
// above its first use (register pressure), and into a branch when only the branch uses it: every refill is therefore
// ISSUED behind a scheduling fence, and the four-chunk loop has no conditional uses (a chunk count off a multiple of
// four is peeled off first, rotating the ring by moves).  Past the strip's end the refills re-read its last chunk (an L1
// hit), and the B prefetch of the last chunk reads the rows behind the panel's K range (inside the allocation: four
// rows of slack; the values are not used).
__device__ __forceinline__ void fs_strip(const double* __restrict__ gt, int ldgt, unsigned voff, int kb, int ke, int lb,
                                         double4_t (&acc)[2], const int dbg) {
  const unsigned row_bytes = (unsigned)ldgt * 8u;
  const size_t step = (size_t)16 * ldgt;   // one chunk down
  const double* pa = gt + (size_t)kb * ldgt;
  int left = (ke - kb) >> 4;               // chunks still to multiply
  int ahead = left - 1;                    // chunks beyond the one pa points at
  AFrag A0 = fs_load(pa, voff, row_bytes);
  if (ahead > 0) pa += step;
  --ahead;
  AFrag A1 = fs_load(pa, voff, row_bytes);
  if (ahead > 0) pa += step;
  --ahead;
  AFrag A2 = fs_load(pa, voff, row_bytes);
  if (ahead > 0) pa += step;
  --ahead;
  AFrag A3 = fs_load(pa, voff, row_bytes);
  double b0 = lds_vread(lb), b1 = lds_vread(lb + 16);
  __builtin_amdgcn_sched_barrier(0);
  while (left & 3) {
    fs_chunk(A0, lb, lb + 16 * FS_LD, b0, b1, acc);
    __builtin_amdgcn_sched_barrier(0);
    lb += 16 * FS_LD;
    A0 = A1; A1 = A2; A2 = A3;
    if (ahead > 0) pa += step;
    --ahead;
    A3 = fs_load(pa, voff, row_bytes);
    __builtin_amdgcn_sched_barrier(0);
    --left;
  }
  for (; left > 0; left -= 4) {
    fs_chunk(A0, lb, lb + 16 * FS_LD, b0, b1, acc);
    __builtin_amdgcn_sched_barrier(0);
    if (ahead > 0) pa += step;
    if (!(dbg & 4)) A0 = fs_load(pa, voff, row_bytes);
    __builtin_amdgcn_sched_barrier(0);
    fs_chunk(A1, lb + 16 * FS_LD, lb + 32 * FS_LD, b0, b1, acc);
    __builtin_amdgcn_sched_barrier(0);
    if (ahead > 1) pa += step;
    if (!(dbg & 4)) A1 = fs_load(pa, voff, row_bytes);
    __builtin_amdgcn_sched_barrier(0);
    fs_chunk(A2, lb + 32 * FS_LD, lb + 48 * FS_LD, b0, b1, acc);
    __builtin_amdgcn_sched_barrier(0);
    if (ahead > 2) pa += step;
    if (!(dbg & 4)) A2 = fs_load(pa, voff, row_bytes);
    __builtin_amdgcn_sched_barrier(0);
    fs_chunk(A3, lb + 48 * FS_LD, lb + 64 * FS_LD, b0, b1, acc);
    __builtin_amdgcn_sched_barrier(0);
    if (ahead > 3) pa += step;
    if (!(dbg & 4)) A3 = fs_load(pa, voff, row_bytes);
    __builtin_amdgcn_sched_barrier(0);
    ahead -= 4;
    lb += 64 * FS_LD;
  }
}

// Two strips a (short K range) and b (long) of one wavefront in ONE loop: while both run, a 16-deep chunk is 16 MFMAs on
// FOUR independent accumulators (a wavefront alone then keeps the matrix cores busy: the dependent-issue distance of an
// fp64 MFMA is ~138 cycles, four accumulators put 256 between two uses of one) that share every B fragment (half the LDS
// reads per MFMA); the rest of b's K range follows through fs_strip.  na <= nb chunks, both from K row kb on.  Two dual
// chunks (32 MFMAs = 2048 matrix-core cycles) of G are in flight.
struct AFrag2 { double fa[4], fb[4]; };
__device__ __forceinline__ AFrag2 fs_load2(const double* __restrict__ pa, const double* __restrict__ pb, unsigned voff,
                                           unsigned row_bytes) {
  AFrag2 f;
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) { f.fa[s4] = fs_ld8(pa, voff, s4 * row_bytes); f.fb[s4] = fs_ld8(pb, voff, s4 * row_bytes); }
  return f;
}
__device__ __forceinline__ void fs_chunk2(const AFrag2& f, int lb, int lb_next, double& b0, double& b1,
                                          double4_t (&acc)[2][2]) {
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const int nx = (s4 < 3) ? lb + (s4 + 1) * FS_LD : lb_next;
    const double n0 = lds_vread(nx), n1 = lds_vread(nx + 16);
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.fa[s4], b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.fa[s4], b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.fb[s4], b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.fb[s4], b1, acc[1][1], 0, 0, 0);
    b0 = n0; b1 = n1;
  }
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {         // the schedule: next step's two reads, then this step's four MFMAs
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
  }
}
__device__ __forceinline__ void fs_pair(const double* __restrict__ gta, const double* __restrict__ gtb, int ldgt,
                                        unsigned voff, int kb, int na, int nb, int lb, double4_t (&acc)[2][2],
                                        const int dbg) {
  const unsigned row_bytes = (unsigned)ldgt * 8u;
  const size_t step = (size_t)16 * ldgt;
  if (na > 0) {
    const double* pa = gta + (size_t)kb * ldgt;
    const double* pb = gtb + (size_t)kb * ldgt;
    int left = na, ahead = na - 1;
    AFrag2 D0 = fs_load2(pa, pb, voff, row_bytes);
    if (ahead > 0) { pa += step; pb += step; }
    --ahead;
    AFrag2 D1 = fs_load2(pa, pb, voff, row_bytes);
    double b0 = lds_vread(lb), b1 = lds_vread(lb + 16);
    __builtin_amdgcn_sched_barrier(0);
    if (left & 1) {
      fs_chunk2(D0, lb, lb + 16 * FS_LD, b0, b1, acc);
      __builtin_amdgcn_sched_barrier(0);
      lb += 16 * FS_LD;
      D0 = D1;
      if (ahead > 0) { pa += step; pb += step; }
      --ahead;
      D1 = fs_load2(pa, pb, voff, row_bytes);
      __builtin_amdgcn_sched_barrier(0);
      --left;
    }
    for (; left > 0; left -= 2) {
      fs_chunk2(D0, lb, lb + 16 * FS_LD, b0, b1, acc);
      __builtin_amdgcn_sched_barrier(0);
      if (ahead > 0) { pa += step; pb += step; }
      if (!(dbg & 4)) D0 = fs_load2(pa, pb, voff, row_bytes);
      __builtin_amdgcn_sched_barrier(0);
      fs_chunk2(D1, lb + 16 * FS_LD, lb + 32 * FS_LD, b0, b1, acc);
      __builtin_amdgcn_sched_barrier(0);
      if (ahead > 1) { pa += step; pb += step; }
      if (!(dbg & 4)) D1 = fs_load2(pa, pb, voff, row_bytes);
      __builtin_amdgcn_sched_barrier(0);
      ahead -= 2;
      lb += 32 * FS_LD;
    }
  }
  if (nb > na) fs_strip(gtb, ldgt, voff, kb + 16 * na, kb + 16 * nb, lb, acc[1], dbg);
}

// |row|^2 as the fma chain d = 0, 1, ... (the order every other kernel of the library uses), with the loads of eight
// dimensions in flight at once: written as `for d: s = fma(p[d], p[d], s)` every iteration waited for its own load --
// D dependent memory round trips per row, and the kernel's set-up phases were made of them (measured: 45 us of a 130 us
// step at C2).  Indices beyond the row are clamped and their values replaced by zeros after the load (no branches).
__device__ __forceinline__ double fs_sumsq(const double* __restrict__ p, int D) {
  double s = 0.0;
  for (int d0 = 0; d0 < D; d0 += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int d = d0 + u;
      const double x = p[d < D ? d : D - 1];
      v[u] = d < D ? x : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s = fma(v[u], v[u], s);
  }
  return s;
}

template <int KID, int NW>
__global__ __launch_bounds__(64 * NW, 4) void fused_score_kernel(const FusedArgs a) {
  constexpr int KP = 32 * NW, LD = FS_LD;
  // LDS: the K* panel [KP + 4][LD] (four rows of slack behind it for the B prefetch of a strip's last chunk); its four
  // padding columns 32..35 carry the panel rows' parameters -- |x_j|^2, alpha_j, Lambda_jj and (first pass) the star
  // edge lam_off,j -- staged once per pass by one coalesced sweep instead of fetched per use; behind the panel the
  // second pass's star edges (the first pass's are still being read by slower wavefronts when a fast one stages the
  // second pass)
  double* panel = lds_dyn;
  double* s_lo2 = lds_dyn + (KP + 4) * LD; // [KP]
  int* s_obs = reinterpret_cast<int*>(s_lo2 + KP);   // [KP] byte offset of the panel row that holds a row's star observation
  const int t = threadIdx.x, lane = t & 63;
  // wave-uniform, and PROVABLY so: anything derived from threadIdx is divergent to the compiler, which would wrap every
  // buffer load of a strip (descriptor base = the strip's first column of Gt) in a waterfall loop
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const int N = a.N, D = a.D, mblk = a.mblk, R1 = a.R1;
#define FSTAMP(k) do { if (a.stamps && t == 0) a.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  FSTAMP(0);
  const long long cand0 = (long long)blockIdx.x * FS_BN;
  // Two workgroups of 8 wavefronts share a CU.  Dispatched together they run in lockstep -- both in their K* phases (matrix
  // cores idle), then both contracting -- and nothing overlaps.  The workgroups of every second dispatch round (the ones
  // that join a CU whose first workgroup has just started) therefore begin a K* phase late: from then on one workgroup's
  // K* phase runs under the other's MFMAs.  dbg bits 4.. = the delay in units of s_sleep 127 (~3.4 us), measurement only.
  if (NW == 8 && ((blockIdx.x / a.ncu) & 1)) {
    for (int i = 0; i < a.delay; ++i) __builtin_amdgcn_s_sleep(127);
  }
  // this lane's two candidates as MFMA columns (column tile ct, column lr) and their |c|^2
  long long cc[2];
  double nc[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const long long cnd = cand0 + 16 * ct + lr;
    cc[ct] = cnd < a.M ? cnd : a.M - 1;
    nc[ct] = fs_sumsq(a.Xc + (size_t)cc[ct] * D, D);
  }
  // operand addressing of the K* tiles, fixed per lane for the whole kernel: candidates through a buffer descriptor on the
  // workgroup's 32 rows of Xc (rows past M read as zeros), design points through one on X (rows past N likewise); the
  // pass, tile and dimension offsets are wave-uniform and travel in the scalar offset / the immediate
  const long long nrow_c = (a.M - cand0 < FS_BN) ? (a.M - cand0) : FS_BN;
  const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.Xc + (size_t)cand0 * D), 0,
                                                                      (int)(nrow_c * D * 8), 0x00020000);
  const int voffc0 = (lr * D + lk) * 8, voffc1 = ((16 + lr) * D + lk) * 8, voffx = (lr * D + lk) * 8;
  // mu = K*' alpha and the diagonal part of k*' Lambda k* per column tile (lane = column lr, rows lk + 4 r: summed over
  // lk at the end), its star-edge part tl per (candidate, 16 rows) thread, sum Y^2 per column tile
  double mu0 = 0.0, mu1 = 0.0, td0 = 0.0, td1 = 0.0, tl = 0.0, qs0 = 0.0, qs1 = 0.0;
  double4_t accU[2][2];                    // the two upper strips' accumulators: alive from pass 1's contraction to pass 2's
  const __amdgpu_buffer_rsrc_t xr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.X), 0, (int)((size_t)N * D * 8), 0x00020000);
  // The strips of this wavefront (16 rows each): lower strips `wave` and 2 NW - 1 - wave of [0, R1), upper strips `wave`
  // and 2 NW - 1 - wave counted from R1 -- short paired with long in EACH pass (the passes end with barriers): equal K
  // chunks of the block-triangular G per wavefront.  A K range ends with the last star that reaches into the strip.
  const unsigned voff = (unsigned)((4 * lk * a.ldgt + lr) * 8);
  // sum of squares of a finished strip: acc[ct][r] is row r0 + lk + 4 r; rows at and beyond `hi` belong to the other kind
  // of strip (or to the zero frame) and are left out
  auto fold = [&](const double4_t (&acc)[2], int r0, int hi) {
    if (r0 + 16 <= hi) {                   // (uniform) the whole strip counts
#pragma unroll
      for (int r = 0; r < 4; ++r) { qs0 = fma(acc[0][r], acc[0][r], qs0); qs1 = fma(acc[1][r], acc[1][r], qs1); }
      return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool in = r0 + lk + 4 * r < hi;
      const double y0 = in ? acc[0][r] : 0.0, y1 = in ? acc[1][r] : 0.0;
      qs0 = fma(y0, y0, qs0);
      qs1 = fma(y1, y1, qs1);
    }
  };

  auto pass = [&](auto first_tag, const int P0, const int Pend) {     // K* rows [P0, Pend) in panel rows [0, Pend - P0)
    constexpr bool FIRST = decltype(first_tag)::value;
    // ---- parameters of the panel's rows, one coalesced sweep ----------------------------------------------------------
    if (t < KP) {
      const int j = P0 + t;
      // a row behind the pass's rows gets |x|^2 = 1e300: its kernel values come out as exact zeros (exp underflows, the RQ
      // denominator overflows) without a select per value; alpha = Lambda = 0 there
      double nx = 1e300, al = 0.0, ld = 0.0, lo = 0.0;
      const int rb = (P0 + t) % mblk;      // (a pass begins with a star: the observation row is in the panel)
      if (j < Pend) {
        al = a.alpha[j]; ld = a.lam_diag[j];
        lo = rb == 0 ? 0.0 : 2.0 * a.lam_off[j];           // an observation row has no edge to itself
        nx = fs_sumsq(a.X + (size_t)j * D, D);
      }
      panel[t * LD + 32] = nx; panel[t * LD + 33] = al; panel[t * LD + 34] = ld;
      if (FIRST) panel[t * LD + 35] = lo;
      else s_lo2[t] = lo;
      s_obs[t] = (t - rb) * LD * 8;
    }
    __syncthreads();                       // parameters staged; and (second pass) everybody is done with the first panel
    FSTAMP(FIRST ? 1 : 6);
    // ---- K* phase: r^2 = (|x|^2 + |c|^2) - 2 x.c, the reference's expansion (kernels.py:7-10), with the x.c of a
    // 16-row x 16-candidate tile on the matrix cores: the accumulator starts at |x|^2 + |c|^2, the B operand is -2 c.
    // A wavefront forms rows [32 wave, 32 wave + 32) of the panel: two row tiles x two column tiles, 16 dimensions per
    // round of operand loads (lane (r, q) holds X[r][16 g + 4 s + q], s = 0..3: 8-byte loads, 32 contiguous bytes per
    // row and k-step).  Dimensions beyond D: the B operand is an explicit zero, whatever finite value A finds there.
    // Panel rows at and beyond Pend - P0 are written as zeros (the contraction's last chunk reads up to 15 of them).
#pragma unroll 1
    for (int rt = 0; rt < 2; ++rt) {       // one row tile at a time: half the fragment registers
      const int jl = 32 * wave + 16 * rt;  // first panel row of the tile
      double4_t kt[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) kt[ct][r] = panel[(jl + lk + 4 * r) * LD + 32] + nc[ct];
      if (P0 + jl < Pend) {                // (uniform) a tile wholly behind the pass's rows is zeros
        for (int d0 = 0; d0 < D; d0 += 16) {
          double af[4], bf[2][4];
          const int sx = ((P0 + jl) * D + d0) * 8, sc = d0 * 8;      // wave-uniform parts of the byte offsets
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            af[s4] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(xr, voffx + 32 * s4, sx, 0));
            // -2 c, and an explicit zero beyond D (the load finds the next candidate's first coordinates there)
            const double mk = (d0 + 4 * s4 + lk < D) ? -2.0 : 0.0;
            bf[0][s4] = mk * __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(cr, voffc0 + 32 * s4, sc, 0));
            bf[1][s4] = mk * __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(cr, voffc1 + 32 * s4, sc, 0));
          }
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            if (d0 + 4 * s4 < D) {         // (uniform) whole k-steps beyond D are not run
#pragma unroll
              for (int ct = 0; ct < 2; ++ct)
                kt[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s4], bf[ct][s4], kt[ct], 0, 0, 0);
            }
          }
        }
      }
      // finish: lane (lr, lk) holds rows lk + 4 r of the tile, candidate column lr
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jr = jl + lk + 4 * r;
        const double al = panel[jr * LD + 33], ld = panel[jr * LD + 34];
        double k0 = 0.5, k1 = 0.5;
        if (!(a.dbg & 1)) {                // (rows behind the pass: |x|^2 = 1e300 was staged, the values are zeros)
          k0 = kern_finish<KID>(fmax(kt[0][r], 0.0), a.p);
          k1 = kern_finish<KID>(fmax(kt[1][r], 0.0), a.p);
        }
        mu0 = fma(al, k0, mu0);
        mu1 = fma(al, k1, mu1);
        td0 = fma(ld * k0, k0, td0);
        td1 = fma(ld * k1, k1, td1);
        panel[jr * LD + lr] = k0;
        panel[jr * LD + 16 + lr] = k1;
      }
    }
    __syncthreads();
    FSTAMP(FIRST ? 2 : 7);
    // ---- the star-edge part of k*' Lambda k* from the finished panel: sum_j (2 lam_off,j) k_j k_obs(j); thread =
    // (candidate, 16 consecutive rows).  2 lam_off (zero on observation rows and behind the pass's rows) and the panel
    // offset of a row's observation row were staged with the parameters: three vector instructions per element
    {
      const int c = t & 31, jl0 = 16 * (t >> 5);
      const char* pbytes = reinterpret_cast<const char*>(panel + c);
#pragma unroll 8
      for (int i = 0; i < 16; ++i) {
        const double kv = panel[(jl0 + i) * LD + c];
        const double lo2 = FIRST ? panel[(jl0 + i) * LD + 35] : s_lo2[jl0 + i];
        const double ko = *reinterpret_cast<const double*>(pbytes + s_obs[jl0 + i]);
        tl = fma(lo2 * kv, ko, tl);
      }
    }
    FSTAMP(FIRST ? 3 : 8);
    // ---- contraction phase: K range [P0, .) of this pass, panel row 0 = K* row P0 ---------------------------------------
    // strips a = `wave` (short K range) and b = 2 NW - 1 - wave (long) of either kind, as a pair (fs_pair)
    if (!(a.dbg & 2)) {
      const int span = (Pend - P0 + 15) & ~15;             // K* rows the panel holds, zeros included
      const int lb0 = (4 * lk) * LD + lr;
      if (FIRST) {
        const int ra = 16 * wave, rb = 16 * (2 * NW - 1 - wave);
        if (ra < R1) {
          const int na = (((((ra + 16 < R1 ? ra + 16 : R1) + mblk - 1) / mblk) * mblk + 15) & ~15) >> 4;   // kend <= R1: no star straddles R1
          const int nb = rb < R1 ? (((((rb + 16 < R1 ? rb + 16 : R1) + mblk - 1) / mblk) * mblk + 15) & ~15) >> 4 : 0;
          double4_t accL[2][2];
#pragma unroll
          for (int e = 0; e < 2; ++e) { accL[e][0] = double4_t{0.0, 0.0, 0.0, 0.0}; accL[e][1] = double4_t{0.0, 0.0, 0.0, 0.0}; }
          if (nb > 0) {
            fs_pair(a.Gt + ra, a.Gt + rb, a.ldgt, voff, 0, na, nb, lb0, accL, a.dbg);
            fold(accL[1], rb, R1);
          } else {
            fs_strip(a.Gt + ra, a.ldgt, voff, 0, 16 * na, lb0, accL[0], a.dbg);
          }
          fold(accL[0], ra, R1);
        }
      }
      if (R1 < N) {
        const int ra = R1 + 16 * wave, rb = R1 + 16 * (2 * NW - 1 - wave);
        if (ra < N) {
          const bool hb = rb < N;
          if (FIRST) {                     // K* rows [0, R1) (and zeros up to `span`): the same K range for every upper strip
#pragma unroll
            for (int e = 0; e < 2; ++e) { accU[e][0] = double4_t{0.0, 0.0, 0.0, 0.0}; accU[e][1] = double4_t{0.0, 0.0, 0.0, 0.0}; }
            if (hb) fs_pair(a.Gt + ra, a.Gt + rb, a.ldgt, voff, 0, span >> 4, span >> 4, lb0, accU, a.dbg);
            else fs_strip(a.Gt + ra, a.ldgt, voff, 0, span, lb0, accU[0], a.dbg);
          } else {
            const int na = ((((((ra + 16 < N ? ra + 16 : N) + mblk - 1) / mblk) * mblk) - P0 + 15) & ~15) >> 4;
            if (hb) {
              const int nb = ((((((rb + 16 < N ? rb + 16 : N) + mblk - 1) / mblk) * mblk) - P0 + 15) & ~15) >> 4;
              fs_pair(a.Gt + ra, a.Gt + rb, a.ldgt, voff, P0, na, nb, lb0, accU, a.dbg);
              fold(accU[1], rb, N);
            } else {
              fs_strip(a.Gt + ra, a.ldgt, voff, P0, P0 + 16 * na, lb0, accU[0], a.dbg);
            }
            fold(accU[0], ra, N);
          }
        }
      }
    }
  };
  pass(std::true_type{}, 0, R1);
  FSTAMP(4);
  if (R1 < N) pass(std::false_type{}, R1, N);
  FSTAMP(9);
  // ---- finish: per-candidate sums over the wavefronts (fixed order), variance, score, the block's best --------------
  __syncthreads();                         // the panel is free: reuse it
  double* red_q = lds_dyn;                 // [NW][32]
  double* red_m = lds_dyn + NW * 32;       // [NW][32]
  double* red_t = lds_dyn + 2 * NW * 32;   // [NW][32]
  double* red_d = lds_dyn + 3 * NW * 32;   // [NW][32]
  qs0 += __shfl_xor(qs0, 16, 64); qs0 += __shfl_xor(qs0, 32, 64);
  qs1 += __shfl_xor(qs1, 16, 64); qs1 += __shfl_xor(qs1, 32, 64);
  mu0 += __shfl_xor(mu0, 16, 64); mu0 += __shfl_xor(mu0, 32, 64);
  mu1 += __shfl_xor(mu1, 16, 64); mu1 += __shfl_xor(mu1, 32, 64);
  td0 += __shfl_xor(td0, 16, 64); td0 += __shfl_xor(td0, 32, 64);
  td1 += __shfl_xor(td1, 16, 64); td1 += __shfl_xor(td1, 32, 64);
  tl += __shfl_xor(tl, 32, 64);            // the wavefront's two row groups
  if (lk == 0) {
    red_q[wave * 32 + lr] = qs0; red_q[wave * 32 + 16 + lr] = qs1;
    red_m[wave * 32 + lr] = mu0; red_m[wave * 32 + 16 + lr] = mu1;
    red_d[wave * 32 + lr] = td0; red_d[wave * 32 + 16 + lr] = td1;
  }
  if (lane < 32) red_t[wave * 32 + lane] = tl;
  __syncthreads();
  FSTAMP(10);
  if (t >= 64) return;
  const long long cand = cand0 + t;
  Best b{0.0, -1};
  if (t < 32 && cand < a.M) {
    double m = 0.0, tt = 0.0, qq = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { m += red_m[w * 32 + t]; tt += red_d[w * 32 + t] + red_t[w * 32 + t]; qq += red_q[w * 32 + t]; }
    const double var = a.sf2 + tt + qq;
    double sc;
    if (a.kind == PPBO_SCORE_MEAN) sc = m;
    else if (a.kind == PPBO_SCORE_VARIANCE) sc = var;
    else {
      const double d = m - a.mustar;
      const double sd = sqrt(fmax(var, 0.0));
      if (sd > 0.0) {
        const double z = d / sd;
        sc = d * norm_cdf(z) + sd * 0.39894228040143267794 * exp(-0.5 * z * z);
      } else sc = fmax(d, 0.0);
    }
    if (a.mu_out) a.mu_out[cand] = m;
    if (a.var_out) a.var_out[cand] = var;
    if (a.score_out) a.score_out[cand] = sc;
    if (sc == sc) { b.val = sc; b.idx = a.idx_base + cand; }
  }
  if (!a.blk_best) return;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {       // lanes 32 .. 63 hold empty records
    Best other;
    other.val = __shfl_xor(b.val, o, 64);
    other.idx = __shfl_xor(b.idx, o, 64);
    b = best_merge(b, other);
  }
  if (t == 0) a.blk_best[blockIdx.x] = b;
  FSTAMP(11);
#undef FSTAMP
}

// the pass structure for a model of N rows with stars of mblk rows under a panel of KP rows: R1 = rows of the first pass
// (N when one pass holds everything), or -1 when two passes do not do
static int fused_split(int N, int mblk, int KP) {
  if (((N + 15) & ~15) <= KP) return N;
  const int R1 = (KP / mblk) * mblk;
  if (R1 <= 0) return -1;
  if (((N - R1 + 15) & ~15) > KP) return -1;                      // the second pass's rows fit one panel
  return R1;
}

template <int KID>
int fused_launch(ppbo_ctx* ctx, FusedArgs& a, hipStream_t s) {
  const unsigned grid = (unsigned)((a.M + FS_BN - 1) / FS_BN);
  const int R8 = fused_split(a.N, a.mblk, 256);
  if (R8 >= 0) {
    constexpr int NW = 8;
    a.R1 = R8;
    const size_t lds = (size_t)((32 * NW + 4) * FS_LD + 32 * NW) * sizeof(double) + (size_t)32 * NW * sizeof(int);
    ppbo_lds_limit(ctx, (const void*)fused_score_kernel<KID, NW>, (int)lds);
    fused_score_kernel<KID, NW><<<grid, 64 * NW, lds, s>>>(a);
  } else {
    constexpr int NW = 16;
    a.R1 = fused_split(a.N, a.mblk, 512);
    const size_t lds = (size_t)((32 * NW + 4) * FS_LD + 32 * NW) * sizeof(double) + (size_t)32 * NW * sizeof(int);
    ppbo_lds_limit(ctx, (const void*)fused_score_kernel<KID, NW>, (int)lds);
    fused_score_kernel<KID, NW><<<grid, 64 * NW, lds, s>>>(a);
  }
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

}  // namespace

namespace {
// Gt[k][i] = G[i][k] for i, k < N, zero elsewhere in [rows_t][ld_t]; 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_pad_kernel(const double* __restrict__ G, int N, double* __restrict__ Gt,
                                                            int rows_t, int ld_t) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int i0 = blockIdx.x * 32, k0 = blockIdx.y * 32;          // tile of G: rows i0.., columns k0..
  for (int r = ty; r < 32; r += 8) {
    const int i = i0 + r, k = k0 + tx;
    tile[r][tx] = (i < N && k < N) ? G[(size_t)i * N + k] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, i = i0 + tx;
    if (k < rows_t && i < ld_t) Gt[(size_t)k * ld_t + i] = tile[tx][r];
  }
}
}  // namespace

// The contraction reads G TRANSPOSED (see the head of this file).  Built per call in a ctx workspace: N + 16 .. 31 rows (the K
// index; a pass's last chunk reads up to 15 rows past N) of ldgt = N rounded up to 32, + 32, doubles (a strip reads 32
// columns from an even row index below N), zeros outside [N][N]; ~2 us at N = 512, ~4 us at N = 1024.
const double* ppbo_fused_transposed_G(ppbo_ctx* ctx, const ppbo_model* m, int* ldgt_out, hipStream_t s) {
  const int N = m->N, rows_t = ((N + 15) & ~15) + 16, ldgt = ((N + 31) & ~31) + 32;
  *ldgt_out = ldgt;
  if (m->d_Gt) return m->d_Gt;            // formed once per fit by the caller (ppbo_transposed_G)
  double* Gt = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_GPAD, (size_t)rows_t * ldgt * sizeof(double));
  if (!Gt) return nullptr;
  transpose_pad_kernel<<<dim3(ldgt / 32, (rows_t + 31) / 32), 256, 0, s>>>(m->d_G, N, Gt, rows_t, ldgt);
  return Gt;
}

extern "C" {

int ppbo_transposed_G_shape(int N, int* rows, int* ld) {
  if (N <= 0 || !rows || !ld) return -1;
  *rows = ((N + 15) & ~15) + 16;
  *ld = ((N + 31) & ~31) + 32;
  return 0;
}

int ppbo_transposed_G(ppbo_ctx* ctx, const double* d_G, int N, double* d_Gt, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_G && d_Gt && N > 0, "G / Gt / N");
  int rows_t = 0, ldgt = 0;
  (void)ppbo_transposed_G_shape(N, &rows_t, &ldgt);
  transpose_pad_kernel<<<dim3(ldgt / 32, (rows_t + 31) / 32), 256, 0, (hipStream_t)stream>>>(d_G, N, d_Gt, rows_t, ldgt);
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}

}  // extern "C"

// does the one-launch form take this model?  (a property of the MODEL, never of the candidate count: a sharded search
// and the unsharded one must score a candidate with the same arithmetic)
bool ppbo_fused_eligible(const ppbo_ctx* ctx, const ppbo_model* m) {
  if (ctx->fused_score == 0 || m->d_G == nullptr || m->kstar_fp32 || m->kernel_id == PPBO_KERNEL_CAMPHOR) return false;
  // Default (PPBO_FUSED=1): the shapes where the one-launch form measured faster than the three-launch one -- two
  // workgroups of 8 wavefronts per CU (up to ~500 rows: 1.1x at N = 512, 1.2-1.5x below) and up to 16 dimensions (one
  // round of operand loads per K* tile).  PPBO_FUSED=2 also takes the 16-wavefront form (up to ~1000 rows, one workgroup
  // per CU: nothing hides its K* phases -- 0.92-1.02x at N = 1024) and any D: profiles/r06_fused_score.txt.
  if (fused_split(m->N, m->m + 1, 256) >= 0 && m->D <= 16) return true;
  return ctx->fused_score >= 2 && (fused_split(m->N, m->m + 1, 256) >= 0 || fused_split(m->N, m->m + 1, 512) >= 0);
}

// scores M candidates in one launch; blk_best (device, (M + 31) / 32 records) receives the per-block bests when not NULL.
// Gt: the transpose of G framed with zeros (ppbo_fused_transposed_G), row stride ldgt.
int ppbo_fused_score(ppbo_ctx* ctx, const ppbo_model* m, const double* Gt, int ldgt, const double* d_Xc, long long M,
                     int score_kind, double mustar, long long idx_base, double* d_mu, double* d_var, double* d_score,
                     void* blk_best, hipStream_t s) {
  FusedArgs a;
  a.X = m->d_X; a.N = m->N; a.D = m->D; a.p = make_kern_params(m->kernel_id, m->theta);
  a.alpha = m->d_alpha; a.lam_diag = m->d_lam_diag; a.lam_off = m->d_lam_off; a.mblk = m->m + 1;
  a.Xc = d_Xc; a.M = M; a.Gt = Gt; a.ldgt = ldgt; a.R1 = m->N;
  a.sf2 = m->theta[2] * m->theta[2]; a.kind = score_kind; a.mustar = mustar; a.idx_base = idx_base;
  a.mu_out = d_mu; a.var_out = d_var; a.score_out = d_score; a.blk_best = (Best*)blk_best;
  a.dbg = ctx->fused_dbg & 7;
  a.stamps = nullptr;
  const long long nblk_dbg = (M + FS_BN - 1) / FS_BN;
  if (ctx->fused_dbg & 8) {                // phase stamps (measurement only): wavefront 0 of every workgroup
    a.stamps = (unsigned long long*)ppbo_workspace(ctx, ppbo_ctx::WS_SEARCH_ROWS, (size_t)nblk_dbg * 16 * sizeof(unsigned long long));
    if (a.stamps) (void)hipMemsetAsync(a.stamps, 0, (size_t)nblk_dbg * 16 * sizeof(unsigned long long), s);
  }
  a.ncu = ctx->n_cu > 0 ? ctx->n_cu : 256;
  a.delay = (ctx->fused_dbg >> 4) & 15;
  int rc;
  switch (m->kernel_id) {
    case PPBO_KERNEL_SE: rc = fused_launch<PPBO_KERNEL_SE>(ctx, a, s); break;
    case PPBO_KERNEL_RQ: rc = fused_launch<PPBO_KERNEL_RQ>(ctx, a, s); break;
    default: return ppbo_set_error(ctx, -1, "the one-launch scoring kernel takes the SE and RQ kernels");
  }
  if (rc == 0 && a.stamps) {               // mean phase lengths over the workgroups, in us (s_memrealtime ticks at 100 MHz)
    std::vector<unsigned long long> h((size_t)nblk_dbg * 16);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), a.stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    static const char* names[12] = {"start", "pass1 parameters", "pass1 K* tiles", "pass1 star edges", "pass1 contraction", "",
                                    "pass2 parameters", "pass2 K* tiles", "pass2 star edges", "pass2 contraction",
                                    "reductions", "score + best"};
    double sum[12] = {};
    unsigned long long t_min = ~0ull, t_max = 0;
    for (long long b = 0; b < nblk_dbg; ++b) {
      unsigned long long prev = h[b * 16];
      if (prev < t_min) t_min = prev;
      for (int k = 1; k < 12; ++k) {
        const unsigned long long v = h[b * 16 + k];
        if (v == 0) continue;
        sum[k] += (double)(v - prev) * 0.01;
        prev = v;
        if (v > t_max) t_max = v;
      }
    }
    fprintf(stderr, "[fused stamps] %lld workgroups, first start to last end %.1f us; mean per workgroup:", nblk_dbg, (double)(t_max - t_min) * 0.01);
    for (int k = 1; k < 12; ++k)
      if (names[k][0]) fprintf(stderr, " %s %.2f |", names[k], sum[k] / (double)nblk_dbg);
    fprintf(stderr, "\n");
  }
  return rc;
}
