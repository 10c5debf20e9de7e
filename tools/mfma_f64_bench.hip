// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (peak check for the
// roofline denominators in bench.py).  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_bench.hip -o tools/mfma_f64_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
  double4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks, int iters) {
  double* out;
  hipMalloc(&out, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<blocks, 256>>>(out, iters, 1.0, 0.5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<blocks, 256>>>(out, iters, 1.0, 0.5);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * NACC * 2.0 * 16 * 16 * 4;
  printf("NACC=%2d blocks=%5d iters=%d  %.3f ms  %.1f TFLOP/s\n", NACC, blocks, iters, ms, flops / ms * 1e-9);
  hipFree(out);
}

int main() {
  run<1>(256, 20000);
  run<4>(256, 5000);
  run<16>(256, 2000);
  run<16>(512, 2000);
  run<16>(1024, 2000);
  run<4>(1024, 5000);
  return 0;
}
