// Score finishing shared by ppbo_predict and ppbo_rff_score: partial-slab sums,
// variance / score evaluation and a deterministic argmax (first index wins ties,
// np.argmax semantics; NaN scores never win).
#pragma once
#include "common.h"

namespace {

__device__ __forceinline__ double norm_cdf(double z) { return 0.5 * erfc(-z * 0.70710678118654752440); }

struct Best {
  double val;
  long long idx;
};
__device__ __forceinline__ Best best_merge(Best a, Best b) {
  // larger value wins; ties -> smaller index (np.argmax first-occurrence); idx<0 == empty
  if (b.idx < 0) return a;
  if (a.idx < 0) return b;
  if (b.val > a.val || (b.val == a.val && b.idx < a.idx)) return b;
  return a;
}
__device__ __forceinline__ Best block_best(Best b, Best* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Best other;
    other.val = __shfl_xor(b.val, o, 64);
    other.idx = __shfl_xor(b.idx, o, 64);
    b = best_merge(b, other);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) b = best_merge(b, sh[w]);
  }
  return b;  // valid in thread 0
}

// Geometry of the score pass: a workgroup finishes SC_CAND candidates; its SC_WAVES wavefronts split the partial slabs
// of those candidates between them (wavefront w sums slabs w, w + SC_WAVES, ...: every load of the pass is in flight at
// once -- the first version had one thread walk all 2 n_mu + n_slab slabs of its candidate, a dependent chain of up
// to 144 strided loads: 38 us for 8192 candidates, 27 us for 65536, on a pass that moves 9-25 MB), the wavefront
// sums meet in LDS and are added in wavefront order (a fixed order: results do not depend on timing).
// The launch-wide best is a SECOND, one-workgroup launch (argmax_final_kernel).  Round 4 tried to fold it into this
// kernel ("the last workgroup to retire merges the records": a device-scope ticket per workgroup): the tickets cost
// ~40 ns per workgroup whatever their layout -- one word, or 32 group words + a top word -- because every one of
// them is a release at device scope: 45-51 us for the 1024 workgroups of a 65536-candidate launch against 6 + 3 us
// for the two launches.
constexpr int SC_CAND = 64, SC_WAVES = 16, SC_THREADS = SC_CAND * SC_WAVES;
static inline int score_blocks(long long M) { return (int)((M + SC_CAND - 1) / SC_CAND); }

__global__ __launch_bounds__(SC_THREADS) void score_kernel(const double* __restrict__ mu_part, int n_mu,
                                                           const double* __restrict__ t_part,
                                                           const double* __restrict__ slab, int n_slab, int M,
                                                           double sf2, int kind, double mustar, long long idx_base,
                                                           double* __restrict__ mu_out, double* __restrict__ var_out,
                                                           double* __restrict__ score_out, Best* __restrict__ blk_best) {
  __shared__ double part[3][SC_WAVES][SC_CAND];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * SC_CAND + lane;
  double pm = 0.0, pt = 0.0, pq = 0.0;
  if (c < M) {
    for (int s = wave; s < n_mu; s += SC_WAVES) pm += mu_part[(size_t)s * M + c];
    if (slab) {
      for (int s = wave; s < n_mu; s += SC_WAVES) pt += t_part[(size_t)s * M + c];
      for (int s = wave; s < n_slab; s += SC_WAVES) pq += slab[(size_t)s * M + c];
    }
  }
  part[0][wave][lane] = pm;
  part[1][wave][lane] = pt;
  part[2][wave][lane] = pq;
  __syncthreads();
  if (wave != 0) return;                 // the rest is one wavefront's work
  Best b{0.0, -1};
  if (c < M) {
    double mu = 0.0, t = 0.0, q = 0.0;
#pragma unroll
    for (int w = 0; w < SC_WAVES; ++w) { mu += part[0][w][lane]; t += part[1][w][lane]; q += part[2][w][lane]; }
    const double var = slab ? sf2 + t + q : sf2;
    double sc;
    if (kind == PPBO_SCORE_MEAN) sc = mu;
    else if (kind == PPBO_SCORE_VARIANCE) sc = var;
    else {
      const double d = mu - mustar;
      const double sd = sqrt(fmax(var, 0.0));
      if (sd > 0.0) {
        const double z = d / sd;
        sc = d * norm_cdf(z) + sd * 0.39894228040143267794 * exp(-0.5 * z * z);
      } else sc = fmax(d, 0.0);
    }
    if (mu_out) mu_out[c] = mu;
    if (var_out) var_out[c] = var;
    if (score_out) score_out[c] = sc;
    if (sc == sc) { b.val = sc; b.idx = idx_base + c; }
  }
  if (!blk_best) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Best other;
    other.val = __shfl_xor(b.val, o, 64);
    other.idx = __shfl_xor(b.idx, o, 64);
    b = best_merge(b, other);
  }
  if (lane == 0) blk_best[blockIdx.x] = b;
}

// per-block records of one launch -> *out; with `record` also the 16-byte (value, GLOBAL index as a double: exact
// below 2^53; index + record_offset; (NaN, -1) when nothing scored) record that the sharded search all-gathers, and
// with `publish` the flag publish[0] raised to `epoch` AFTER the record with system-scope release semantics (the
// record may live in host-mapped memory: the host polls the flag instead of waiting for a copy and a stream
// synchronisation)
__global__ __launch_bounds__(256) void argmax_final_kernel(const Best* __restrict__ blk_best, int n,
                                                           Best* __restrict__ out, double* __restrict__ record = nullptr,
                                                           long long record_offset = 0,
                                                           unsigned long long* __restrict__ publish = nullptr,
                                                           unsigned long long epoch = 0) {
  __shared__ Best sh[4];
  Best b{0.0, -1};
  for (int i = threadIdx.x; i < n; i += blockDim.x) b = best_merge(b, blk_best[i]);
  b = block_best(b, sh);
  if (threadIdx.x == 0) {
    if (out) *out = b;
    if (record) {
      record[0] = b.idx < 0 ? NAN : b.val;
      record[1] = b.idx < 0 ? -1.0 : (double)(b.idx + record_offset);
    }
    if (publish) __hip_atomic_store(publish, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// per-chunk bests -> the 16-byte (value, global index as a double) record of a sharded search; one wavefront
__global__ __launch_bounds__(64) void best_record_kernel(const Best* __restrict__ chunk_best, int n, long long offset,
                                                         double* __restrict__ record,
                                                         unsigned long long* __restrict__ publish = nullptr,
                                                         unsigned long long epoch = 0) {
  Best b{0.0, -1};
  for (int i = threadIdx.x; i < n; i += 64) b = best_merge(b, chunk_best[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Best other;
    other.val = __shfl_xor(b.val, o, 64);
    other.idx = __shfl_xor(b.idx, o, 64);
    b = best_merge(b, other);
  }
  if (threadIdx.x == 0) {
    record[0] = b.idx < 0 ? NAN : b.val;
    record[1] = b.idx < 0 ? -1.0 : (double)(b.idx + offset);
    if (publish) __hip_atomic_store(publish, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}


// host: reduce per-chunk bests (device) to one (value, index); synchronises the stream
inline int merge_chunk_bests(ppbo_ctx* ctx, const Best* d_chunk_best, int n_chunks, double* h_best_val,
                             int64_t* h_best_idx, hipStream_t s) {
  if (!h_best_val && !h_best_idx) return 0;
  Best* hb = (Best*)ppbo_pinned(ctx, (size_t)n_chunks * sizeof(Best));
  if (!hb) return ppbo_set_error(ctx, (int)hipErrorOutOfMemory, "pinned staging");
  PPBO_HIP_CHECK(ctx, hipMemcpyAsync(hb, d_chunk_best, (size_t)n_chunks * sizeof(Best), hipMemcpyDeviceToHost, s));
  PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
  double bv = 0.0;
  long long bi = -1;
  for (int ch = 0; ch < n_chunks; ++ch) {
    if (hb[ch].idx < 0) continue;
    if (bi < 0 || hb[ch].val > bv || (hb[ch].val == bv && hb[ch].idx < bi)) { bv = hb[ch].val; bi = hb[ch].idx; }
  }
  if (h_best_val) *h_best_val = bv;
  if (h_best_idx) *h_best_idx = (int64_t)bi;
  return 0;
}

}  // namespace
