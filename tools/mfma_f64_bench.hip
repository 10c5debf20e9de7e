// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 and the clock the chip
// holds under it.  Build: hipcc -O3 -w --offload-arch=gfx950 tools/mfma_f64_bench.hip -o tools/mfma_f64_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* cyc, int iters, double a0, double b0) {
  double4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    cyc[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0;
    cyc[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0;
  }
}

template <int NACC>
void run(int blocks, int iters) {
  double* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
  (void)hipMalloc(&cyc, (size_t)blocks * 4 * 16);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) k<NACC><<<blocks, 256>>>(out, cyc, iters, 1.0, 0.5);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<NACC><<<blocks, 256>>>(out, cyc, iters, 1.0, 0.5);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)blocks * 8);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cpm, ghz;
  for (int i = 0; i < blocks * 4; ++i) {
    cpm.push_back((double)h[2 * i] / ((double)iters * NACC));
    ghz.push_back((double)h[2 * i] / ((double)h[2 * i + 1] * 10.0) );  // memrealtime ticks at 100 MHz
  }
  std::sort(cpm.begin(), cpm.end()); std::sort(ghz.begin(), ghz.end());
  double flops = (double)blocks * 4 * iters * NACC * 2.0 * 16 * 16 * 4;
  printf("NACC=%2d blocks=%5d (%.0f waves/SIMD) %.3f ms %.1f TFLOP/s | per-wave cycles/MFMA median %.1f | clock median %.2f GHz\n",
         NACC, blocks, blocks / 256.0, ms, flops / ms * 1e-9, cpm[cpm.size() / 2], ghz[ghz.size() / 2]);
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  run<1>(256, 40000);
  run<2>(256, 20000);
  run<4>(256, 10000);
  run<8>(256, 5000);
  run<16>(256, 4000);
  run<4>(512, 10000);
  run<16>(512, 4000);
  run<4>(1024, 10000);
  run<16>(1024, 4000);
  run<4>(2048, 10000);
  return 0;
}
