// Development harness: single-launch blocked triangular solve with inter-workgroup hand-off through memory
// (NaN-sentinel polling with cache-bypassing loads).  Measures the hop latency that decides whether replacing the
// per-trial partial triangular inverse by solves on L can pay.  Not part of the library.
// Build: hipcc -O3 -w --offload-arch=gfx950 tools/dev/trsv_dev.hip -o tools/dev/trsv_dev.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__device__ __forceinline__ void store_through(double* p, double v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ double load_bypass(const double* p) {      // sc0 sc1: system scope, misses every cache level
  double v;
  asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// forward solve L y = b with B x B diagonal blocks whose inverses are given (Dinv[k] is B x B row-major).
// One workgroup per block row, 256 threads.  y must be NaN-filled on entry; bounded spinning (gives up after
// `spin_limit` polls and sets *err).
template <int B>
__global__ __launch_bounds__(256) void trsv_fwd(const double* __restrict__ L, int ld, int N, const double* __restrict__ Dinv,
                                                const double* __restrict__ b, double* __restrict__ y, int spin_limit,
                                                int* __restrict__ err) {
  constexpr int TPR = 256 / B;            // threads per row (4 at B = 64, 2 at B = 128)
  constexpr int CPT = B / TPR;            // columns per thread
  __shared__ double xs[B];
  __shared__ double vs[B];
  const int k = blockIdx.x, t = threadIdx.x;
  const int r = t / TPR, part = t % TPR;
  const int row = k * B + r;
  double acc = 0.0;
  double lt[CPT];
  // prefetch tile 0
  if (k > 0) {
#pragma unroll
    for (int c = 0; c < CPT; ++c) lt[c] = L[(size_t)row * ld + part * CPT + c];
  }
  for (int j = 0; j < k; ++j) {
    // wait for x_j: threads 0..B-1 poll one element each
    if (t < B) {
      double v = load_bypass(y + j * B + t);
      int spins = 0;
      while (v != v) {
        if (++spins > spin_limit) { *err = 1; v = 0.0; break; }
        __builtin_amdgcn_s_sleep(1);
        v = load_bypass(y + j * B + t);
      }
      xs[t] = v;
    }
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < CPT; ++c) s += lt[c] * xs[part * CPT + c];
    acc += s;
    if (j + 1 < k) {
#pragma unroll
      for (int c = 0; c < CPT; ++c) lt[c] = L[(size_t)row * ld + (size_t)(j + 1) * B + part * CPT + c];
    }
    __syncthreads();
  }
  // reduce over the TPR lanes of a row
#pragma unroll
  for (int o = 1; o < TPR; o <<= 1) acc += __shfl_xor(acc, o, 64);
  if (part == 0) vs[r] = b[row] - acc;
  __syncthreads();
  // y_k = Dinv_k v: thread = (row r, part): partial dot over CPT columns
  double s = 0.0;
  const double* dr = Dinv + (size_t)k * B * B + (size_t)r * B + part * CPT;
#pragma unroll
  for (int c = 0; c < CPT; ++c) s += dr[c] * vs[part * CPT + c];
#pragma unroll
  for (int o = 1; o < TPR; o <<= 1) s += __shfl_xor(s, o, 64);
  if (part == 0) store_through(y + row, s);
}

template <int B>
float run(const double* L, int N, const double* Dinv, const double* b, double* y, int* err, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) {
    hipMemsetAsync(y, 0xFF, (size_t)N * 8);
    trsv_fwd<B><<<N / B, 256>>>(L, N, N, Dinv, b, y, 2000000, err);
  }
  hipDeviceSynchronize();
  float tot = 0.f;
  for (int i = 0; i < reps; ++i) {
    hipMemsetAsync(y, 0xFF, (size_t)N * 8);
    hipEventRecord(e0);
    trsv_fwd<B><<<N / B, 256>>>(L, N, N, Dinv, b, y, 2000000, err);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms;
  }
  return tot / reps * 1e3f;
}

template <int B>
void test(int N) {
  std::vector<double> hL((size_t)N * N, 0.0), hb(N), hD((size_t)N * B);
  srand(1);
  for (int i = 0; i < N; ++i) {
    for (int j = 0; j < i; ++j) hL[(size_t)i * N + j] = (rand() / (double)RAND_MAX - 0.5) / N;
    hL[(size_t)i * N + i] = 1.0 + rand() / (double)RAND_MAX;
    hb[i] = rand() / (double)RAND_MAX;
  }
  // host inverses of the diagonal blocks (forward substitution per column)
  for (int k = 0; k < N / B; ++k)
    for (int c = 0; c < B; ++c) {
      std::vector<double> x(B, 0.0);
      for (int rI = 0; rI < B; ++rI) {
        double v = (rI == c) ? 1.0 : 0.0;
        for (int q = 0; q < rI; ++q) v -= hL[(size_t)(k * B + rI) * N + k * B + q] * x[q];
        x[rI] = v / hL[(size_t)(k * B + rI) * N + k * B + rI];
      }
      for (int rI = 0; rI < B; ++rI) hD[(size_t)k * B * B + (size_t)rI * B + c] = x[rI];
    }
  std::vector<double> ref(N);
  for (int i = 0; i < N; ++i) {
    double v = hb[i];
    for (int j = 0; j < i; ++j) v -= hL[(size_t)i * N + j] * ref[j];
    ref[i] = v / hL[(size_t)i * N + i];
  }
  double *L, *D, *b, *y; int* err;
  hipMalloc(&L, hL.size() * 8); hipMalloc(&D, hD.size() * 8); hipMalloc(&b, N * 8); hipMalloc(&y, N * 8); hipMalloc(&err, 4);
  hipMemcpy(L, hL.data(), hL.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(D, hD.data(), hD.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), N * 8, hipMemcpyHostToDevice);
  hipMemset(err, 0, 4);
  const float us = run<B>(L, N, D, b, y, err, 50);
  std::vector<double> hy(N); int herr = 0;
  hipMemcpy(hy.data(), y, N * 8, hipMemcpyDeviceToHost);
  hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
  double e = 0.0;
  for (int i = 0; i < N; ++i) e = fmax(e, fabs(hy[i] - ref[i]));
  printf("N=%5d B=%3d: %7.2f us per solve (%5.2f us per hop)  max err %.2e  spin-timeout %d\n", N, B, us, us / (N / B), e, herr);
  hipFree(L); hipFree(D); hipFree(b); hipFree(y); hipFree(err);
}

int main() {
  for (int N : {512, 1024, 2048, 4096}) { test<64>(N); test<128>(N); }
  return 0;
}
