// Shared host/device helpers for libppbo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <sched.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ppbo_hip.h"

struct PpboUniqueId { char internal[128]; };   // layout of ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
struct ppbo_dist_state;                          // dist.hip: the RCCL communicator of a ctx

// A ctx owns private device workspaces (grown on demand, freed on destroy) and
// the last error string.  No global mutable state.
struct ppbo_ctx {
  int device = 0;
  std::string err;
  // named workspace slots
  enum { WS_KSTAR = 0, WS_PART, WS_SCRATCH, WS_LINALG, WS_LINALG2, WS_VEC, WS_SMALL, WS_POTRF, WS_APPEND, WS_DIST, WS_LBFGS, WS_SEARCH, WS_SEARCH_SMALL, WS_LBFGS_U, WS_TRANSPOSE, WS_GPAD, WS_SEARCH_ROWS, WS_COUNT };
  void* ws[WS_COUNT] = {};
  size_t ws_bytes[WS_COUNT] = {};
  void* pinned = nullptr;  // small pinned host staging buffer
  size_t pinned_bytes = 0;
  // ring of pinned upload slots for small host arguments that travel by hipMemcpyAsync (ppbo_upload_async): a slot is
  // reused only after the event recorded behind its last copy has completed
  struct UploadSlot { void* p = nullptr; size_t bytes = 0; hipEvent_t ev = nullptr; bool used = false; };
  UploadSlot upload[4];
  unsigned upload_next = 0;
  // optional per-kernel event timing
  bool profiling = false;
  enum { PF_GRAM = 0, PF_KSTAR, PF_QUADFORM, PF_SCORE, PF_RFF_PROJECT, PF_RFF_SCORE, PF_POTRF, PF_LINE_KSTAR, PF_LINE_Y, PF_LINE_COV, PF_LINE_MC, PF_FUSED, PF_COUNT };
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pf_events[PF_COUNT];
  size_t pf_used[PF_COUNT] = {};
  // kernels whose dynamic-LDS limit has been raised on THIS ctx's device (hipFuncSetAttribute is per device)
  std::vector<const void*> lds_raised;
  std::vector<int> lds_raised_bytes;
  // tuning knobs, read from the environment once per ctx (ppbo_ctx_create); defaults = measured best
  int qf_variant = 4, qf_order = -1, potrf_gen = 3, rff_nt = 0, gram_variant = -1, rff_score_mfma = 1;
  // PPBO_FUSED: 1 (default) = models of up to 1024 rows are scored by the one-launch kernel of fused.hip, 0 = always the
  // three-launch form (kstar -> quadform -> score)
  int fused_score = 1;
  int gemm_big16 = 1;     // PPBO_GEMM_BIG16: large GEMMs on 16 wavefronts of 32 x 32 (default since round 6) instead of 8 of 32 x 64
  int n_cu = 0;           // compute units of the device (hipDeviceProp_t::multiProcessorCount)
  int fused_dbg = 0;      // PPBO_FUSED_DBG: measurement switches of fused.hip (results are wrong when set)
  // ppbo_gp_fit runs the triangular inverse and Sigma^-1 on a second stream beside the first evaluations of the f_MAP
  // search, which needs only L until the |grad_f| rule is armed (fit.hip)
  int fit_overlap = 1;    // PPBO_FIT_OVERLAP
  int fit_gf_from = 8;    // PPBO_FIT_GF_FROM: the first evaluation that may apply the |grad_f| rule in that mode
  // the host-polled searches let the runtime wait (hipStreamSynchronize) when their progress word has been still for
  // this long, and carry on if that advanced the search (PPBO_POLL_LIMIT_MS; tests set it to 0 to walk that path)
  int poll_limit_ms = 5000;
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int syrk_cfg = 0;       // PPBO_SYRK_CFG: tile configuration of Sigma^-1 = Linv^T Linv (0 = by size; 1 / 2 / 3 = 128 / 64 / 32)
  int line_y_chunk = 0;   // PPBO_LINE_Y_CHUNK: column tiles per chunk of Y = G K* in the line acquisition (0 = chosen by size)
  ppbo_dist_state* dist = nullptr;   // set by ppbo_dist_init
  // host-mapped (pinned, device-visible) result record: [0] value, [1] index as a double, [2] the epoch flag the
  // publishing kernel raises last; the host polls it (ppbo_host_record_wait)
  double* hostrec = nullptr;         // host address
  double* hostrec_dev = nullptr;     // the same memory as the device sees it
  unsigned long long hostrec_epoch = 0;
};

// Waiting on a host-mapped word without burning a core: spin with `pause` while the word keeps changing (a search
// slot is 5-50 us), yield the core once it has been still for a while (other contexts' threads -- evidence_batch runs
// eight -- and the Python interpreter get it), and report after `limit_s` seconds without a change.
struct PpboSpinWait {
  unsigned long long last = ~0ull;
  unsigned still = 0;
  std::chrono::steady_clock::time_point t_last = std::chrono::steady_clock::now();
  double limit_s = 5.0;
  void reset() { still = 0; t_last = std::chrono::steady_clock::now(); }
  // true: the word has not changed for limit_s seconds
  bool idle(unsigned long long w) {
    if (w != last) { last = w; reset(); return false; }
    ++still;
    if (still < 2048) { __builtin_ia32_pause(); return false; }
    sched_yield();
    if ((still & 0xff) == 0 &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t_last).count() > limit_s) return true;
    return false;
  }
};

struct PpboHostRecord {
  double* d_rec;                     // device view of the record (2 doubles)
  unsigned long long* d_flag;        // device view of the flag
  const volatile double* h_rec;      // host view
  unsigned long long epoch;          // the value the flag takes when the record is complete
};
// a fresh epoch on the ctx's host-mapped record (allocated on first use)
int ppbo_host_record(ppbo_ctx* ctx, PpboHostRecord* out);
// spin until the kernel enqueued on `s` has raised the flag to the record's epoch; falls back to a stream
// synchronisation (and reports an error if the flag still is not there: the kernel did not run to completion)
int ppbo_host_record_wait(ppbo_ctx* ctx, const PpboHostRecord& r, hipStream_t s);
// fused.hip: the one-launch scoring path (K*, contraction, score in one kernel) for models of up to 1024 rows
struct ppbo_model;
bool ppbo_fused_eligible(const ppbo_ctx* ctx, const ppbo_model* m);
const double* ppbo_fused_transposed_G(ppbo_ctx* ctx, const ppbo_model* m, int* ldgt_out, hipStream_t s);
int ppbo_fused_score(ppbo_ctx* ctx, const ppbo_model* m, const double* Gt, int ldgt, const double* d_Xc, long long M,
                     int score_kind, double mustar, long long idx_base, double* d_mu, double* d_var, double* d_score,
                     void* blk_best, hipStream_t s);
// predict.hip, for dist.hip: one shard's scoring passes with the record published to host-mapped memory
int ppbo_predict_record_publish(ppbo_ctx* ctx, const ppbo_model* model, const double* d_Xc, int64_t M, int score_kind,
                                double mustar, int64_t index_offset, double* d_record, unsigned long long* d_flag,
                                unsigned long long epoch, hipStream_t s);

// Every extern "C" entry runs on its ctx's device and leaves the caller's current device as it found it.
struct PpboDeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit PpboDeviceGuard(const ppbo_ctx* c) {
    if (c && hipGetDevice(&prev) == hipSuccess && prev != c->device) switched = (hipSetDevice(c->device) == hipSuccess);
  }
  ~PpboDeviceGuard() { if (switched) (void)hipSetDevice(prev); }
  PpboDeviceGuard(const PpboDeviceGuard&) = delete;
  PpboDeviceGuard& operator=(const PpboDeviceGuard&) = delete;
};
#define PPBO_ENTER(ctx)      \
  if (!(ctx)) return -1;     \
  PpboDeviceGuard _ppbo_dev_guard(ctx)

// RAII bracket: records two events around a launch when profiling is on
struct PpboProfScope {
  ppbo_ctx* ctx; int slot; hipStream_t s; hipEvent_t stop = nullptr;
  PpboProfScope(ppbo_ctx* c, int slot_, hipStream_t s_) : ctx(c), slot(slot_), s(s_) {
    if (!ctx || !ctx->profiling) return;
    auto& v = ctx->pf_events[slot];
    if (ctx->pf_used[slot] == v.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      v.emplace_back(a, b);
    }
    auto& pr = v[ctx->pf_used[slot]++];
    (void)hipEventRecord(pr.first, s);
    stop = pr.second;
  }
  ~PpboProfScope() { if (stop) (void)hipEventRecord(stop, s); }
};

int ppbo_set_error(ppbo_ctx* ctx, int code, const char* fmt, ...);
// returns a device pointer of at least `bytes` (contents undefined); nullptr on failure
void* ppbo_workspace(ppbo_ctx* ctx, int slot, size_t bytes);
void* ppbo_pinned(ppbo_ctx* ctx, size_t bytes);
// copies h_src (any host memory; consumed before the call returns) to d_dst through a pinned slot of the ctx, truly
// asynchronously on `s`: neither blocks the host on the stream nor relies on how the runtime stages pageable memory
int ppbo_upload_async(ppbo_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, hipStream_t s);
// raise a kernel's dynamic-LDS limit to `bytes` once per ctx (i.e. once per device)
void ppbo_lds_limit(ppbo_ctx* ctx, const void* kernel_fn, int bytes);

#define PPBO_HIP_CHECK(ctx, expr)                                                      \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess)                                                              \
      return ppbo_set_error((ctx), (int)_e, "%s failed: %s (%s:%d)", #expr,            \
                            hipGetErrorString(_e), __FILE__, __LINE__);                \
  } while (0)

#define PPBO_LAUNCH_CHECK(ctx) PPBO_HIP_CHECK(ctx, hipGetLastError())

#define PPBO_REQUIRE(ctx, cond, msg)                                    \
  do {                                                                  \
    if (!(cond)) return ppbo_set_error((ctx), -1, "invalid argument: %s", msg); \
  } while (0)

// ---- kernel-function parameters (host-prepared, passed by value) ------------
struct KernParams {
  double sf2;      // sigma_f^2
  double c0;       // SE: 0.5/l^2     RQ: 1/(4 l^2)    camphor: 2/l^2
  double c1;       // camphor: 0.5/(l+0.05)^2
};

static inline KernParams make_kern_params(int kernel_id, const double theta[3]) {
  KernParams p;
  const double l = theta[1], sf = theta[2];
  p.sf2 = sf * sf;
  p.c1 = 0.0;
  if (kernel_id == PPBO_KERNEL_SE) p.c0 = 0.5 / (l * l);
  else if (kernel_id == PPBO_KERNEL_RQ) p.c0 = 1.0 / (4.0 * l * l);
  else { p.c0 = 2.0 / (l * l); p.c1 = 0.5 / ((l + 0.05) * (l + 0.05)); }
  return p;
}

#ifdef __HIPCC__
typedef double double4_t __attribute__((ext_vector_type(4)));

// Write-through ("sc1") global stores for bulk results that the NEXT kernel reads: a plain store leaves the
// line dirty in the XCD's L2 and the whole dirty footprint is written back when the kernel ends -- serial
// time on a chain of dependent launches (measured: 3-4 us per Cholesky step with 16 MB of dirty tiles).
// hipcc does not count an asm store in vmcnt; that only makes its waits for earlier loads conservative.
__device__ __forceinline__ void store_through(double* p, double v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
typedef double double2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_through2(double* p, double x, double y) {
  const double2_t v = {x, y};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// the value lane ^ 1 holds (DPP quad_perm [1,0,3,2]): two 32-bit moves, no LDS
__device__ __forceinline__ double lane_xor1(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// 64-lane wavefront reductions (gfx950: wave = 64)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// The same sum without the LDS crossbar: __shfl_xor compiles to ds_bpermute_b32, and a workgroup of 16 wavefronts
// that all reduce through it is bound by the CU's one LDS pipe (measured: 12 us for 25 sums x 16 waves).  DPP moves
// are plain VALU instructions: xor 1, xor 2 (quad_perm), row_half_mirror, row_mirror give every lane its row-of-16
// sum; the four row sums are combined through scalar registers (v_readlane).  Fixed order: bitwise reproducible.
__device__ __forceinline__ double dpp_add(double v, const int ctrl_selector) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  switch (ctrl_selector) {
    case 0: lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true); break;
    case 1: lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true); break;
    case 2: lo = __builtin_amdgcn_mov_dpp(lo, 0x141, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x141, 0xF, 0xF, true); break;
    default: lo = __builtin_amdgcn_mov_dpp(lo, 0x140, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x140, 0xF, 0xF, true); break;
  }
  return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_sum_dpp(double v) { return dpp_add(dpp_add(v, 0), 1); }
// sum over each row of 16 lanes (every lane of the row ends with the row's sum): four DPP steps, no LDS, no readlane
__device__ __forceinline__ double row16_sum_dpp(double v) { return dpp_add(dpp_add(dpp_add(dpp_add(v, 0), 1), 2), 3); }
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v = dpp_add(dpp_add(dpp_add(dpp_add(v, 0), 1), 2), 3);
  double r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    r[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * k),
                            __builtin_amdgcn_readlane(__double2loint(v), 16 * k));
  return (r[0] + r[1]) + (r[2] + r[3]);
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// exp(x) for x <= 0 (every covariance kernel's exponent): Cody-Waite reduction by ln2, degree-11 near-minimax
// polynomial on |r| <= ln2/2 (fit error 3e-18, tools/expfit.py), scaling by v_ldexp_f64.  18 VALU
// instructions and no special-case branches (the library exp spends as many again on range checks);
// measured against 50-digit arithmetic: <= 1 ulp.
__device__ __forceinline__ double exp_nonpos(double x) {
  x = fmax(x, -750.0);                       // exp underflows to 0 below -745.2; keeps (int)n in range
  const double n = __builtin_rint(x * 1.4426950408889634);
  double r = __builtin_fma(n, -6.93147180369123816490e-01, x);
  r = __builtin_fma(n, -1.90821492927058770002e-10, r);
  double q = 0x1.af632a0f7e2cep-26;
  q = __builtin_fma(q, r, 0x1.28b4101c77212p-22);
  q = __builtin_fma(q, r, 0x1.71ddf56d8deb5p-19);
  q = __builtin_fma(q, r, 0x1.a01991a10d9aep-16);
  q = __builtin_fma(q, r, 0x1.a01a01b1461c5p-13);
  q = __builtin_fma(q, r, 0x1.6c16c1880029fp-10);
  q = __builtin_fma(q, r, 0x1.111111110f21ep-7);
  q = __builtin_fma(q, r, 0x1.555555554f0bap-5);
  q = __builtin_fma(q, r, 0x1.555555555555ap-3);
  q = __builtin_fma(q, r, 0x1.0000000000011p-1);
  q = __builtin_fma(q, r, 1.0);
  q = __builtin_fma(q, r, 1.0);
  return ldexp(q, (int)n);
}

// Kernel value from accumulated per-dimension terms.
//   SE / RQ : s = sum_d (x_d - y_d)^2
//   camphor : s = c0 * sum_{d in 0,1,3,4,5} sin^2(pi |dx_d|) + c1 * dx_2^2  (already scaled)
template <int KID>
__device__ __forceinline__ double kern_finish(double s, const KernParams& p) {
  if (KID == PPBO_KERNEL_SE) return p.sf2 * exp_nonpos(-p.c0 * s);
  if (KID == PPBO_KERNEL_RQ) {
    const double t = 1.0 + s * p.c0;
    return p.sf2 / (t * t);
  }
  return p.sf2 * exp_nonpos(-s);
}

template <int KID>
__device__ __forceinline__ double kern_term(double dx, int d, const KernParams& p) {
  if (KID == PPBO_KERNEL_CAMPHOR) {
    if (d == 2) return p.c1 * dx * dx;
    const double sn = sinpi(fabs(dx));
    return p.c0 * sn * sn;
  }
  return dx * dx;
}
#endif
