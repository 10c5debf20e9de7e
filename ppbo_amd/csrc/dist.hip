// (e) multi-GPU: the ONE collective of the path behind the C-ABI, for integrators that bind the library with
// ctypes and do not want torch.distributed: the per-shard (best score, global index) records are combined by an
// RCCL all-gather of 16 bytes per rank over xGMI and a local lowest-index tie-break (np.argmax semantics).
// The reference has no distributed code; this replaces a sharded form of mu_star's search
// (gp_model.py:415-437).  librccl is dlopen'ed on first use -- the library has no link-time dependency on it and
// shares the copy PyTorch has already loaded when there is one.
#include <dlfcn.h>

#include "common.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, PpboUniqueId, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

// resolved per ctx (no process-global state); dlopen reference-counts the shared object itself
int rccl_open(ppbo_ctx* ctx, Rccl& r) {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);   // LOCAL: never export RCCL's symbols into a host that carries its own copy
    if (r.handle) break;
  }
  if (!r.handle) return ppbo_set_error(ctx, -4, "librccl.so not found (%s)", dlerror());
  r.GetUniqueId = (int (*)(void*))dlsym(r.handle, "ncclGetUniqueId");
  r.CommInitRank = (int (*)(void**, int, PpboUniqueId, int))dlsym(r.handle, "ncclCommInitRank");
  r.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(r.handle, "ncclAllGather");
  r.CommDestroy = (int (*)(void*))dlsym(r.handle, "ncclCommDestroy");
  r.GetErrorString = (const char* (*)(int))dlsym(r.handle, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) {
    dlclose(r.handle);
    r.handle = nullptr;
    return ppbo_set_error(ctx, -4, "librccl.so lacks the expected entry points");
  }
  return 0;
}

void rccl_close(Rccl& r) {
  if (r.handle) dlclose(r.handle);       // drops the reference rccl_open took; the object stays while others hold it
  r.handle = nullptr;
}

// (value, index) records of all shards -> out[0] = best value, out[1] = its global index (as a double, exact below
// 2^53): larger value wins, ties go to the smaller index (np.argmax first-occurrence semantics), NaN values and
// negative indices (empty shards) never win; no valid record: (NaN, -1).  One wavefront.
__global__ __launch_bounds__(64) void argmax_combine_kernel(const double* __restrict__ rec, int W,
                                                            double* __restrict__ out) {
  double bv = 0.0, bi = -1.0;
  for (int r = threadIdx.x; r < W; r += 64) {
    const double v = rec[2 * r], i = rec[2 * r + 1];
    if (i < 0.0 || v != v) continue;
    if (bi < 0.0 || v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(bv, o, 64), oi = __shfl_xor(bi, o, 64);
    if (oi >= 0.0 && (bi < 0.0 || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
  }
  if (threadIdx.x == 0) { out[0] = bi < 0.0 ? NAN : bv; out[1] = bi; }
}

constexpr int NCCL_FLOAT64 = 8;   // ncclFloat64 (rccl.h)

}  // namespace

struct ppbo_dist_state {
  Rccl r;
  void* comm = nullptr;
  int rank = 0, world = 1;
};

extern "C" {

int ppbo_dist_unique_id(ppbo_ctx* ctx, void* h_id128) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, h_id128 != nullptr, "id buffer");
  Rccl r;
  if (int rc = rccl_open(ctx, r)) return rc;
  PpboUniqueId id;
  const int e = r.GetUniqueId(&id);
  const int rc = e != 0 ? ppbo_set_error(ctx, 2000 + e, "ncclGetUniqueId: %s", r.GetErrorString ? r.GetErrorString(e) : "?") : 0;
  rccl_close(r);
  if (rc) return rc;
  std::memcpy(h_id128, &id, sizeof(id));
  return 0;
}

int ppbo_dist_init(ppbo_ctx* ctx, const void* h_id128, int rank, int world) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, h_id128 != nullptr && world >= 1 && rank >= 0 && rank < world, "rank / world / id");
  PPBO_REQUIRE(ctx, ctx->dist == nullptr, "ppbo_dist_init called twice on this ctx");
  ppbo_dist_state* d = new (std::nothrow) ppbo_dist_state();
  if (!d) return -2;
  if (int rc = rccl_open(ctx, d->r)) { delete d; return rc; }
  PpboUniqueId id;
  std::memcpy(&id, h_id128, sizeof(id));
  const int e = d->r.CommInitRank(&d->comm, world, id, rank);    // binds to the current device = ctx->device (PPBO_ENTER)
  if (e != 0) {
    const int rc = ppbo_set_error(ctx, 2000 + e, "ncclCommInitRank: %s", d->r.GetErrorString ? d->r.GetErrorString(e) : "?");
    rccl_close(d->r);
    delete d;
    return rc;
  }
  d->rank = rank;
  d->world = world;
  ctx->dist = d;
  return 0;
}

int ppbo_dist_destroy(ppbo_ctx* ctx) {
  PPBO_ENTER(ctx);
  if (!ctx->dist) return 0;
  int rc = 0;
  if (ctx->dist->comm) {
    const int e = ctx->dist->r.CommDestroy(ctx->dist->comm);
    if (e != 0)
      rc = ppbo_set_error(ctx, 2000 + e, "ncclCommDestroy: %s", ctx->dist->r.GetErrorString ? ctx->dist->r.GetErrorString(e) : "?");
  }
  rccl_close(ctx->dist->r);
  delete ctx->dist;
  ctx->dist = nullptr;
  return rc;
}

int ppbo_argmax_allgather(ppbo_ctx* ctx, double local_val, int64_t local_global_idx, double* h_best_val,
                          int64_t* h_best_idx, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, ctx->dist != nullptr, "ppbo_dist_init has not been called on this ctx");
  PPBO_REQUIRE(ctx, h_best_val && h_best_idx, "outputs");
  ppbo_dist_state* d = ctx->dist;
  hipStream_t s = (hipStream_t)stream;
  const int W = d->world;
  double* dev = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_DIST, (size_t)(2 + 2 * W) * sizeof(double));
  double* host = (double*)ppbo_pinned(ctx, (size_t)(2 + 2 * W) * sizeof(double) + 64 * sizeof(double));
  if (!dev || !host) return ppbo_set_error(ctx, (int)hipErrorOutOfMemory, "collective staging");
  host += 64;                                            // the first 64 doubles of the pinned block belong to the fit
  host[0] = local_val;
  host[1] = (double)local_global_idx;                    // exact below 2^53
  PPBO_HIP_CHECK(ctx, hipMemcpyAsync(dev, host, 2 * sizeof(double), hipMemcpyHostToDevice, s));
  const int e = d->r.AllGather(dev, dev + 2, 2, NCCL_FLOAT64, d->comm, s);
  if (e != 0) return ppbo_set_error(ctx, 2000 + e, "ncclAllGather: %s", d->r.GetErrorString ? d->r.GetErrorString(e) : "?");
  argmax_combine_kernel<<<1, 64, 0, s>>>(dev + 2, W, dev);          // reduced on the device: ONE 16-byte record comes back
  PPBO_LAUNCH_CHECK(ctx);
  PPBO_HIP_CHECK(ctx, hipMemcpyAsync(host, dev, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
  *h_best_val = host[0];
  *h_best_idx = (int64_t)host[1];
  return 0;
}

int ppbo_argmax_combine(ppbo_ctx* ctx, const double* d_records, int world, double* h_best_val, int64_t* h_best_idx,
                        void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_records && world >= 1 && h_best_val && h_best_idx, "arguments");
  hipStream_t s = (hipStream_t)stream;
  double* dev = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_DIST, (size_t)(2 + 2 * world) * sizeof(double));
  double* host = (double*)ppbo_pinned(ctx, (size_t)(2 + 2 * world) * sizeof(double) + 64 * sizeof(double));
  if (!dev || !host) return ppbo_set_error(ctx, (int)hipErrorOutOfMemory, "collective staging");
  host += 64;
  argmax_combine_kernel<<<1, 64, 0, s>>>(d_records, world, dev);
  PPBO_LAUNCH_CHECK(ctx);
  PPBO_HIP_CHECK(ctx, hipMemcpyAsync(host, dev, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
  *h_best_val = host[0];
  *h_best_idx = (int64_t)host[1];
  return 0;
}

}  // extern "C"
