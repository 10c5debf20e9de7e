// Micro-benchmark: issue rate and dependent latency of fp64 vector FMAs on gfx950 versus independent chains
// per wavefront and wavefronts per SIMD (the latency-bound kernels of this repo -- slab factor, K*, exp --
// are sized against these numbers).
// Build: hipcc -O3 -w --offload-arch=gfx950 tools/valu_f64_bench.hip -o tools/valu_f64_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NCH>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* cyc, int iters, double a0, double b0) {
  double v[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) v[i] = a0 + i * 1e-3 + threadIdx.x * 1e-9;
  const double m = b0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep)
#pragma unroll
      for (int i = 0; i < NCH; ++i) v[i] = __builtin_fma(v[i], m, 0.5);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NCH; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NCH>
void run(int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;     // 4 waves per block -> one per SIMD of a CU
  double* out; unsigned long long* cyc;
  hipMalloc(&out, (size_t)blocks * 256 * 8); hipMalloc(&cyc, (size_t)blocks * 4 * 8);
  const int iters = 2000;
  k<NCH><<<blocks, 256>>>(out, cyc, 10, 1.0, 0.999);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<NCH><<<blocks, 256>>>(out, cyc, iters, 1.0, 0.999);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double n_inst = (double)iters * 8 * NCH;            // per wave
  const double flops = (double)blocks * 4 * 64 * n_inst * 2;
  // s_memtime ticks at 100 MHz on this part: report wall-clock based cycles at 2.4 GHz
  const double cyc_per_inst_wave = ms * 1e-3 * 2.4e9 / n_inst;
  printf("chains/wave %2d  waves/SIMD %d : %7.2f TFLOP/s  | %.1f cycles per FMA per wave -> %.1f per SIMD issue slot\n", NCH,
         waves_per_simd, flops / (ms * 1e-3) / 1e12, cyc_per_inst_wave, cyc_per_inst_wave / waves_per_simd);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<1>(1); run<2>(1); run<4>(1); run<8>(1); run<16>(1);
  run<1>(2); run<2>(2); run<4>(2); run<8>(2);
  run<1>(4); run<2>(4); run<4>(4); run<8>(4);
  run<1>(8); run<2>(8); run<4>(8);
  return 0;
}
