// Context management and error plumbing of libppbo_hip.so.
#include <chrono>

#include "common.h"

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

int ppbo_set_error(ppbo_ctx* ctx, int code, const char* fmt, ...) {
  if (ctx) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    ctx->err = buf;
  }
  return code;
}

void* ppbo_workspace(ppbo_ctx* ctx, int slot, size_t bytes) {
  if (bytes == 0) bytes = 256;
  if (ctx->ws_bytes[slot] >= bytes) return ctx->ws[slot];
  if (ctx->ws[slot]) {
    (void)hipDeviceSynchronize();
    (void)hipFree(ctx->ws[slot]);
    ctx->ws[slot] = nullptr;
    ctx->ws_bytes[slot] = 0;
  }
  size_t want = bytes + bytes / 8;  // headroom so small growth does not realloc
  void* p = nullptr;
  if (hipMalloc(&p, want) != hipSuccess) {
    ppbo_set_error(ctx, (int)hipErrorOutOfMemory, "workspace slot %d: hipMalloc(%zu) failed", slot, want);
    return nullptr;
  }
  ctx->ws[slot] = p;
  ctx->ws_bytes[slot] = want;
  return p;
}

void* ppbo_pinned(ppbo_ctx* ctx, size_t bytes) {
  if (ctx->pinned_bytes >= bytes) return ctx->pinned;
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  ctx->pinned = nullptr;
  ctx->pinned_bytes = 0;
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  ctx->pinned = p;
  ctx->pinned_bytes = bytes;
  return p;
}

int ppbo_upload_async(ppbo_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, hipStream_t s) {
  if (bytes == 0) return 0;
  ppbo_ctx::UploadSlot& u = ctx->upload[ctx->upload_next++ & 3];
  if (!u.ev) PPBO_HIP_CHECK(ctx, hipEventCreateWithFlags(&u.ev, hipEventDisableTiming));
  if (u.used) PPBO_HIP_CHECK(ctx, hipEventSynchronize(u.ev));     // four uploads ago: long done in practice
  if (u.bytes < bytes) {
    if (u.p) (void)hipHostFree(u.p);
    u.p = nullptr; u.bytes = 0;
    const size_t want = bytes < 4096 ? 4096 : bytes + bytes / 4;
    if (hipHostMalloc(&u.p, want, hipHostMallocDefault) != hipSuccess)
      return ppbo_set_error(ctx, (int)hipErrorOutOfMemory, "pinned upload slot: hipHostMalloc(%zu) failed", want);
    u.bytes = want;
  }
  std::memcpy(u.p, h_src, bytes);
  PPBO_HIP_CHECK(ctx, hipMemcpyAsync(d_dst, u.p, bytes, hipMemcpyHostToDevice, s));
  PPBO_HIP_CHECK(ctx, hipEventRecord(u.ev, s));
  u.used = true;
  return 0;
}

int ppbo_host_record(ppbo_ctx* ctx, PpboHostRecord* out) {
  if (!ctx->hostrec) {
    void* h = nullptr;
    void* d = nullptr;
    if (hipHostMalloc(&h, 256, hipHostMallocMapped) != hipSuccess)
      return ppbo_set_error(ctx, (int)hipErrorOutOfMemory, "host-mapped result record: hipHostMalloc failed");
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
      (void)hipHostFree(h);
      return ppbo_set_error(ctx, (int)hipErrorOutOfMemory, "host-mapped result record: no device pointer");
    }
    std::memset(h, 0, 256);
    ctx->hostrec = (double*)h;
    ctx->hostrec_dev = (double*)d;
  }
  out->d_rec = ctx->hostrec_dev;
  out->d_flag = reinterpret_cast<unsigned long long*>(ctx->hostrec_dev + 2);
  out->h_rec = ctx->hostrec;
  out->epoch = ++ctx->hostrec_epoch;
  return 0;
}

int ppbo_host_record_wait(ppbo_ctx* ctx, const PpboHostRecord& r, hipStream_t s) {
  const volatile unsigned long long* flag = reinterpret_cast<const volatile unsigned long long*>(ctx->hostrec + 2);
  // the kernels in front of the publishing one take 0.5 ... 5 ms; poll for a bounded time (yielding the core once the
  // flag has been still for a while), then let the runtime wait
  PpboSpinWait spin;
  spin.limit_s = 2.0;
  for (;;) {
    const unsigned long long f = *flag;
    if (f == r.epoch) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return 0; }
    if (spin.idle(f)) break;
  }
  PPBO_HIP_CHECK(ctx, hipStreamSynchronize(s));
  if (*flag == r.epoch) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return 0; }
  return ppbo_set_error(ctx, (int)hipErrorUnknown, "the result record was never published (flag %llu, expected %llu)",
                        (unsigned long long)*flag, r.epoch);
}

void ppbo_lds_limit(ppbo_ctx* ctx, const void* kernel_fn, int bytes) {
  for (size_t k = 0; k < ctx->lds_raised.size(); ++k)
    if (ctx->lds_raised[k] == kernel_fn) {
      if (bytes <= ctx->lds_raised_bytes[k]) return;           // one call per kernel and size class suffices
      (void)hipFuncSetAttribute(kernel_fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      ctx->lds_raised_bytes[k] = bytes;
      return;
    }
  (void)hipFuncSetAttribute(kernel_fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  ctx->lds_raised.push_back(kernel_fn);
  ctx->lds_raised_bytes.push_back(bytes);
}

extern "C" {

int ppbo_abi_version(void) { return PPBO_ABI_VERSION; }

int ppbo_ctx_create(int device, ppbo_ctx** out) {
  if (!out) return -1;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return (int)e;
  if (device < 0 || device >= n) return -1;
  int prev = -1;
  (void)hipGetDevice(&prev);
  e = hipSetDevice(device);           // validates the device; the caller's current device is restored below
  if (e != hipSuccess) return (int)e;
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  ppbo_ctx* c = new (std::nothrow) ppbo_ctx();
  if (!c) return -2;
  c->device = device;
  c->fused_score = env_int("PPBO_FUSED", 1);
  c->gemm_big16 = env_int("PPBO_GEMM_BIG16", 1);
  c->fused_dbg = env_int("PPBO_FUSED_DBG", 0);
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
  }
  c->qf_variant = env_int("PPBO_QF_VARIANT", 4);
  if (c->qf_variant < 0 || c->qf_variant > 5) c->qf_variant = 0;
  c->qf_order = env_int("PPBO_QF_ORDER", -1);      // -1: by size (launch_quadform)
  c->line_y_chunk = env_int("PPBO_LINE_Y_CHUNK", 0);
  c->syrk_cfg = env_int("PPBO_SYRK_CFG", 0);
  c->fit_overlap = env_int("PPBO_FIT_OVERLAP", 1);
  c->fit_gf_from = env_int("PPBO_FIT_GF_FROM", 8);
  c->poll_limit_ms = env_int("PPBO_POLL_LIMIT_MS", 5000);
  c->potrf_gen = env_int("PPBO_POTRF_GEN", 3);
  c->rff_nt = env_int("PPBO_RFF_NT", 0);
  c->gram_variant = env_int("PPBO_GRAM_VARIANT", -1);
  c->rff_score_mfma = env_int("PPBO_RFF_SCORE_MFMA", 1);
  *out = c;
  return 0;
}

int ppbo_ctx_destroy(ppbo_ctx* ctx) {
  if (!ctx) return 0;
  if (ctx->dist) (void)ppbo_dist_destroy(ctx);
  {
  PpboDeviceGuard guard(ctx);
  (void)hipDeviceSynchronize();
  for (int i = 0; i < ppbo_ctx::WS_COUNT; ++i)
    if (ctx->ws[i]) (void)hipFree(ctx->ws[i]);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->hostrec) (void)hipHostFree(ctx->hostrec);
  for (auto& u : ctx->upload) {
    if (u.p) (void)hipHostFree(u.p);
    if (u.ev) (void)hipEventDestroy(u.ev);
  }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
  for (int i = 0; i < ppbo_ctx::PF_COUNT; ++i)
    for (auto& pr : ctx->pf_events[i]) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  }
  delete ctx;
  return 0;
}

static int pf_slot(const char* name) {
  static const char* names[] = {"gram", "kstar", "quadform", "score", "rff_project", "rff_score", "potrf",
                                "line_kstar", "line_y", "line_cov", "line_mc", "fused_score"};
  for (int i = 0; i < ppbo_ctx::PF_COUNT; ++i)
    if (std::strcmp(name, names[i]) == 0) return i;
  return -1;
}

int ppbo_profile_enable(ppbo_ctx* ctx, int on) {
  if (!ctx) return -1;
  ctx->profiling = (on != 0);
  return 0;
}

int ppbo_profile_reset(ppbo_ctx* ctx) {
  if (!ctx) return -1;
  for (int i = 0; i < ppbo_ctx::PF_COUNT; ++i) ctx->pf_used[i] = 0;
  return 0;
}

int ppbo_profile_read(ppbo_ctx* ctx, const char* name, double* h_total_ms, int* h_count) {
  if (!ctx || !name) return -1;
  PpboDeviceGuard guard(ctx);
  const int slot = pf_slot(name);
  if (slot < 0) return ppbo_set_error(ctx, -1, "unknown profile name %s", name);
  double tot = 0.0;
  for (size_t i = 0; i < ctx->pf_used[slot]; ++i) {
    auto& pr = ctx->pf_events[slot][i];
    PPBO_HIP_CHECK(ctx, hipEventSynchronize(pr.second));
    float ms = 0.f;
    PPBO_HIP_CHECK(ctx, hipEventElapsedTime(&ms, pr.first, pr.second));
    tot += ms;
  }
  if (h_total_ms) *h_total_ms = tot;
  if (h_count) *h_count = (int)ctx->pf_used[slot];
  return 0;
}

int ppbo_last_error(ppbo_ctx* ctx, char* buf, size_t n) {
  if (!ctx || !buf || n == 0) return -1;
  snprintf(buf, n, "%s", ctx->err.c_str());
  return 0;
}

}  // extern "C"
