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

__global__ __launch_bounds__(256) void score_kernel(const double* __restrict__ mu_part, int n_mu,
                                                    const double* __restrict__ t_part,
                                                    const double* __restrict__ slab, int n_slab, int M,
                                                    double sf2, int kind, double mustar, long long idx_base,
                                                    double* __restrict__ mu_out, double* __restrict__ var_out,
                                                    double* __restrict__ score_out, Best* __restrict__ blk_best) {
  __shared__ Best sh[4];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  Best b{0.0, -1};
  if (c < M) {
    double mu = 0.0;
    for (int s = 0; s < n_mu; ++s) mu += mu_part[(size_t)s * M + c];
    double var = sf2;
    if (slab) {
      double t = 0.0, q = 0.0;
      for (int s = 0; s < n_mu; ++s) t += t_part[(size_t)s * M + c];
      for (int s = 0; s < n_slab; ++s) q += slab[(size_t)s * M + c];
      var = sf2 + t + q;
    }
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
  b = block_best(b, sh);
  if (threadIdx.x == 0 && blk_best) blk_best[blockIdx.x] = b;
}

__global__ __launch_bounds__(256) void argmax_final_kernel(const Best* __restrict__ blk_best, int n,
                                                           Best* __restrict__ out) {
  __shared__ Best sh[4];
  Best b{0.0, -1};
  for (int i = threadIdx.x; i < n; i += blockDim.x) b = best_merge(b, blk_best[i]);
  b = block_best(b, sh);
  if (threadIdx.x == 0) *out = b;
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
