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

// resolved per ctx (no process-global state); dlopen reference-counts the shared object itself.
// RTLD_NODELETE: ncclGetUniqueId starts RCCL's bootstrap root on rank 0 -- a listening socket and a detached service
// thread INSIDE the library -- so the object must never be unmapped once it has been entered, whatever the reference
// count does between ppbo_dist_unique_id and ppbo_dist_init (in a host without torch ours may be the only reference).
int rccl_open(ppbo_ctx* ctx, Rccl& r) {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    // LOCAL: never export RCCL's symbols into a host that carries its own copy
    r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NODELETE);
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
  if (r.handle) dlclose(r.handle);       // drops the reference rccl_open took; RTLD_NODELETE keeps the object mapped
  r.handle = nullptr;
}

// (value, index) records of all shards -> out[0] = best value, out[1] = its global index (as a double, exact below
// 2^53): larger value wins, ties go to the smaller index (np.argmax first-occurrence semantics), NaN values and
// negative indices (empty shards) never win; no valid record: (NaN, -1).  One wavefront.
// With `publish`, `out` may be the ctx's host-mapped record: the flag is raised to `epoch` after it (system scope).
__global__ __launch_bounds__(64) void argmax_combine_kernel(const double* __restrict__ rec, int W,
                                                            double* __restrict__ out,
                                                            unsigned long long* __restrict__ publish = nullptr,
                                                            unsigned long long epoch = 0) {
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
  if (threadIdx.x == 0) {
    out[0] = bi < 0.0 ? NAN : bv;
    out[1] = bi;
    if (publish) __hip_atomic_store(publish, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
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

// gather the ranks' device records (d_record[2] on this rank), reduce, and hand ONE record to the host: no host value
// travels to the device first, and the reduced record is written straight into the ctx's host-mapped record by the
// reduction kernel (flag polled by the host) -- no device-to-host copy, no stream synchronisation.  Without a
// communicator (a single-process search) the record is published by a one-wavefront copy of the same kernel.
static int gather_reduce_readback(ppbo_ctx* ctx, const double* d_record, double* h_best_val, int64_t* h_best_idx,
                                  hipStream_t s) {
  ppbo_dist_state* d = ctx->dist;
  const int W = d ? d->world : 1;
  double* dev = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_DIST, (size_t)(4 + 2 * W) * sizeof(double));
  if (!dev) return ppbo_set_error(ctx, (int)hipErrorOutOfMemory, "collective staging");
  PpboHostRecord hr;
  if (int rc = ppbo_host_record(ctx, &hr)) return rc;
  const double* src = d_record;
  int n = 1;
  if (d) {      // also at world = 1: the communicator's all-gather is the path, not a special case
    const int e = d->r.AllGather(d_record, dev + 4, 2, NCCL_FLOAT64, d->comm, s);
    if (e != 0) return ppbo_set_error(ctx, 2000 + e, "ncclAllGather: %s", d->r.GetErrorString ? d->r.GetErrorString(e) : "?");
    src = dev + 4;
    n = W;
  }
  argmax_combine_kernel<<<1, 64, 0, s>>>(src, n, hr.d_rec, hr.d_flag, hr.epoch);
  PPBO_LAUNCH_CHECK(ctx);
  if (int rc = ppbo_host_record_wait(ctx, hr, s)) return rc;
  if (h_best_val) *h_best_val = hr.h_rec[0];
  if (h_best_idx) *h_best_idx = (int64_t)hr.h_rec[1];
  return 0;
}

int ppbo_argmax_allgather_record(ppbo_ctx* ctx, const double* d_record, double* h_best_val, int64_t* h_best_idx,
                                 void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, ctx->dist != nullptr, "ppbo_dist_init has not been called on this ctx");
  PPBO_REQUIRE(ctx, d_record && (h_best_val || h_best_idx), "record / outputs");
  return gather_reduce_readback(ctx, d_record, h_best_val, h_best_idx, (hipStream_t)stream);
}

int ppbo_search_sharded(ppbo_ctx* ctx, const ppbo_model* model, const double* d_Xc, int64_t M, int score_kind,
                        double mustar, int64_t index_offset, double* h_best_val, int64_t* h_best_idx, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, h_best_val || h_best_idx, "outputs");
  if (!ctx->dist) {
    // a single-process search: the last score workgroup writes the host-mapped record itself
    PpboHostRecord hr;
    if (int rc = ppbo_host_record(ctx, &hr)) return rc;
    if (int rc = ppbo_predict_record_publish(ctx, model, d_Xc, M, score_kind, mustar, index_offset, hr.d_rec, hr.d_flag,
                                             hr.epoch, (hipStream_t)stream))
      return rc;
    if (int rc = ppbo_host_record_wait(ctx, hr, (hipStream_t)stream)) return rc;
    if (h_best_val) *h_best_val = hr.h_rec[0];
    if (h_best_idx) *h_best_idx = (int64_t)hr.h_rec[1];
    return 0;
  }
  const int W = ctx->dist->world;
  double* dev = (double*)ppbo_workspace(ctx, ppbo_ctx::WS_DIST, (size_t)(4 + 2 * W) * sizeof(double));
  if (!dev) return ppbo_set_error(ctx, (int)hipErrorOutOfMemory, "collective staging");
  // this rank's (best score, global index) stays on the device: scoring, the all-gather, the reduction and the
  // publication of the 16-byte record are enqueued behind each other on ONE stream and the host waits once
  if (int rc = ppbo_predict_record(ctx, model, d_Xc, M, score_kind, mustar, index_offset, dev + 2, stream)) return rc;
  return gather_reduce_readback(ctx, dev + 2, h_best_val, h_best_idx, (hipStream_t)stream);
}

int ppbo_argmax_combine(ppbo_ctx* ctx, const double* d_records, int world, double* h_best_val, int64_t* h_best_idx,
                        void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, d_records && world >= 1 && h_best_val && h_best_idx, "arguments");
  hipStream_t s = (hipStream_t)stream;
  PpboHostRecord hr;
  if (int rc = ppbo_host_record(ctx, &hr)) return rc;
  argmax_combine_kernel<<<1, 64, 0, s>>>(d_records, world, hr.d_rec, hr.d_flag, hr.epoch);
  PPBO_LAUNCH_CHECK(ctx);
  if (int rc = ppbo_host_record_wait(ctx, hr, s)) return rc;
  *h_best_val = hr.h_rec[0];
  *h_best_idx = (int64_t)hr.h_rec[1];
  return 0;
}

}  // extern "C"
