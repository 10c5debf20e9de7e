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

// Geometry of the score pass: a workgroup finishes SC_CAND candidates at a time; its SC_WAVES wavefronts split the partial
// slabs of those candidates between them (wavefront w sums slabs w, w + SC_WAVES, ...: every load of the pass is in flight at
// once -- the first version had one thread walk all 2 n_mu + n_slab slabs of its candidate, a dependent chain of up
// to 144 strided loads: 38 us for 8192 candidates, 27 us for 65536, on a pass that moves 9-25 MB), the wavefront
// sums meet in LDS and are added in wavefront order (a fixed order: results do not depend on timing).
constexpr int SC_CAND = 64, SC_WAVES = 16, SC_THREADS = SC_CAND * SC_WAVES;
// at most SC_MAX_BLOCKS workgroups (each walks its tiles): the "last workgroup" ticket is one device-scope atomic per
// workgroup on ONE address, and those serialise across the eight XCDs (~40 ns each: 1024 of them cost more than the pass)
constexpr int SC_MAX_BLOCKS = 256;
static inline int score_blocks(long long M) {
  const long long t = (M + SC_CAND - 1) / SC_CAND;
  return (int)(t < SC_MAX_BLOCKS ? t : SC_MAX_BLOCKS);
}

// Scores + per-block best + (when `counter` is given) the launch-wide best in the SAME launch: the last workgroup to
// retire -- the one whose ticket from `counter` is gridDim.x - 1 -- merges the per-block records and writes
// *final_out; with `record` it also writes the 16-byte (value, GLOBAL index as a double: exact below 2^53) record
// that the sharded search all-gathers (index + record_offset; an empty / all-NaN launch: (NaN, -1)) and, with
// `publish`, raises publish[0] to `epoch` AFTER the record with system-scope release semantics (the record may live
// in host-mapped memory: the host polls the flag instead of waiting for a copy and a stream synchronisation).  The
// merge is a max under a total order (value, then lower index), so the result does not depend on which workgroup
// comes last.  The ticket counter is zero whenever no launch is in flight: the last workgroup resets it.
__global__ __launch_bounds__(SC_THREADS) void score_kernel(const double* __restrict__ mu_part, int n_mu,
                                                           const double* __restrict__ t_part,
                                                           const double* __restrict__ slab, int n_slab, int M,
                                                           double sf2, int kind, double mustar, long long idx_base,
                                                           double* __restrict__ mu_out, double* __restrict__ var_out,
                                                           double* __restrict__ score_out, Best* __restrict__ blk_best,
                                                           unsigned* __restrict__ counter = nullptr,
                                                           Best* __restrict__ final_out = nullptr,
                                                           double* __restrict__ record = nullptr,
                                                           long long record_offset = 0,
                                                           unsigned long long* __restrict__ publish = nullptr,
                                                           unsigned long long epoch = 0) {
  __shared__ double part[3][SC_WAVES][SC_CAND];
  __shared__ Best sh[SC_WAVES];
  __shared__ unsigned s_ticket;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntiles = (M + SC_CAND - 1) / SC_CAND;
  Best b{0.0, -1};                       // wavefront 0: this lane's best over the tiles of the workgroup
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int c = tile * SC_CAND + lane;
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
    if (wave == 0 && c < M) {
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
      if (sc == sc) b = best_merge(b, Best{sc, idx_base + c});
    }
    __syncthreads();                     // `part` is rewritten by the next tile
  }
  if (!blk_best) return;
  if (wave == 0) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      Best other;
      other.val = __shfl_xor(b.val, o, 64);
      other.idx = __shfl_xor(b.idx, o, 64);
      b = best_merge(b, other);
    }
    if (lane == 0) {
      // the record goes out with device-scope atomics (it must be visible to a workgroup on another XCD, whose L2 is
      // not ours), then the ticket is drawn with release / acquire semantics
      __hip_atomic_store(&blk_best[blockIdx.x].val, b.val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&blk_best[blockIdx.x].idx, b.idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (counter) s_ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (!blk_best || !counter) return;
  __syncthreads();
  if (s_ticket != gridDim.x - 1) return;
  Best f{0.0, -1};
  for (int i = threadIdx.x; i < (int)gridDim.x; i += blockDim.x) {
    Best o;
    o.val = __hip_atomic_load(&blk_best[i].val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    o.idx = __hip_atomic_load(&blk_best[i].idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f = best_merge(f, o);
  }
  f = block_best(f, sh);
  if (threadIdx.x == 0) {
    if (final_out) *final_out = f;
    if (record) {
      record[0] = f.idx < 0 ? NAN : f.val;
      record[1] = f.idx < 0 ? -1.0 : (double)(f.idx + record_offset);
    }
    __hip_atomic_store(counter, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (publish) __hip_atomic_store(publish, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ __launch_bounds__(256) void argmax_final_kernel(const Best* __restrict__ blk_best, int n,
                                                           Best* __restrict__ out) {
  __shared__ Best sh[4];
  Best b{0.0, -1};
  for (int i = threadIdx.x; i < n; i += blockDim.x) b = best_merge(b, blk_best[i]);
  b = block_best(b, sh);
  if (threadIdx.x == 0) *out = b;
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
