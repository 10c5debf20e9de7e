// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 (and 4x4x4) on gfx950 versus
// wavefronts per SIMD and independent accumulators per wavefront, plus the clock held.
// Build: hipcc -O3 -w --offload-arch=gfx950 tools/mfma_f64_bench.hip -o tools/mfma_f64_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC, int MINW, int VALU>
__global__ __launch_bounds__(256, MINW) void k(double* out, unsigned long long* cyc, int iters, double a0, double b0) {
  double4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0, 0, 0, 0};
  double a[2] = {a0 + threadIdx.x * 1e-9, a0 * 0.5 + threadIdx.x * 1e-9};
  double b[2] = {b0 - threadIdx.x * 1e-9, b0 * 0.25};
  double v = a0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 1], b[(i >> 1) & 1], acc[i], 0, 0, 0);
      if (VALU) { v = v * 1.0000001 + 0.5; }
    }
  }
  double s = v;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    cyc[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0;
    cyc[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0;
  }
}

template <int NACC, int MINW>
__global__ __launch_bounds__(256, MINW) void k4(double* out, unsigned long long* cyc, int iters, double a0, double b0) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    cyc[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0;
    cyc[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0;
  }
}

template <typename F>
void run(const char* tag, int nacc, double flop_per_inst, int blocks, int iters, F launch) {
  double* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
  (void)hipMalloc(&cyc, (size_t)blocks * 4 * 16);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) launch(blocks, out, cyc, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  launch(blocks, out, cyc, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)blocks * 8);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cpm, ghz;
  for (int i = 0; i < blocks * 4; ++i) {
    cpm.push_back((double)h[2 * i] / ((double)iters * nacc));
    ghz.push_back((double)h[2 * i] / ((double)h[2 * i + 1] * 10.0));
  }
  std::sort(cpm.begin(), cpm.end()); std::sort(ghz.begin(), ghz.end());
  const double wps = blocks / 256.0;
  double flops = (double)blocks * 4 * iters * nacc * flop_per_inst;
  printf("%-22s NACC=%2d waves/SIMD=%.0f  %.1f TFLOP/s | per-wave cyc/MFMA %.1f -> per-SIMD %.1f | clock %.2f GHz\n", tag, nacc, wps,
         flops / ms * 1e-9, cpm[cpm.size() / 2], cpm[cpm.size() / 2] / wps, ghz[ghz.size() / 2]);
  (void)hipFree(out); (void)hipFree(cyc);
}

#define RUN16(NACC, MINW, VALU, BLOCKS, ITERS) \
  run("16x16x4 w" #MINW " valu" #VALU, NACC, 2048.0, BLOCKS, ITERS, [](int bl, double* o, unsigned long long* c, int it) { k<NACC, MINW, VALU><<<bl, 256>>>(o, c, it, 1.0, 0.5); })
#define RUN4(NACC, MINW, BLOCKS, ITERS) \
  run("4x4x4 w" #MINW, NACC, 512.0, BLOCKS, ITERS, [](int bl, double* o, unsigned long long* c, int it) { k4<NACC, MINW><<<bl, 256>>>(o, c, it, 1.0, 0.5); })

int main() {
  RUN16(16, 1, 0, 256, 3000);
  RUN16(16, 2, 0, 512, 3000);
  RUN16(8, 2, 0, 512, 6000);
  RUN16(8, 3, 0, 768, 6000);
  RUN16(8, 4, 0, 1024, 6000);
  RUN16(4, 4, 0, 1024, 10000);
  RUN16(4, 6, 0, 1536, 10000);
  RUN16(4, 8, 0, 2048, 10000);
  RUN16(2, 8, 0, 2048, 20000);
  RUN16(16, 2, 1, 512, 3000);
  RUN16(8, 4, 1, 1024, 6000);
  RUN4(16, 1, 256, 20000);
  RUN4(16, 2, 512, 20000);
  RUN4(16, 4, 1024, 20000);
  RUN4(8, 8, 2048, 20000);
  return 0;
}
