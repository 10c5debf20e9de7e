// Micro-benchmark (round 6): do the fp64 vector instructions of ONE wavefront run beside the fp64 MFMAs of ANOTHER wavefront
// on the same SIMD of gfx950?  A 512-thread workgroup per CU (2 wavefronts per SIMD): the even wavefronts of a SIMD loop
// over v_mfma_f64_16x16x4_f64 (8 independent accumulators), the odd ones over v_fma_f64 chains (8 independent chains), or
// over v_fma_f32 / v_add_u32 chains, or idle.  Reported: MFMA cycles per instruction per SIMD alone, with the VALU
// wavefront beside it, and the VALU wavefront's own cycles per instruction alone and beside the MFMAs.
// Build: hipcc -O3 -w --offload-arch=gfx950 tools/mfma_valu_coissue.hip -o tools/mfma_valu_coissue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double double4_t __attribute__((ext_vector_type(4)));

// role of a wavefront: 0 idle, 1 MFMA f64, 2 VALU f64 fma, 3 VALU f32 fma, 4 VALU u32 add
template <int ROLE_EVEN, int ROLE_ODD, int NACC = 8>
__global__ __launch_bounds__(512) void k(double* out, unsigned long long* cyc, int iters, double a0, double b0) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // scalar: the role branches are s_cbranch, an idle wavefront idles
  // wavefronts of a workgroup go to the SIMDs in a cyclic order: wavefronts w and w + 4 share a SIMD
  const int role = (wave < 4) ? ROLE_EVEN : ROLE_ODD;
  double4_t acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = double4_t{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  double v[8];
  float f[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { v[i] = a0 + i; f[i] = (float)(b0 + i); u[i] = threadIdx.x + i; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (role == 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i % NACC], 0, 0, 0);
    }
  } else if (role == 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __builtin_fma(v[i], 1.0000001, 0.5);
    }
  } else if (role == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = __builtin_fmaf(f[i], 1.0000001f, 0.5f);
    }
  } else if (role == 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) u[i] = u[i] * 3u + 7u;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i] + f[i] + u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <typename F>
void run(const char* tag, int iters_even, F launch) {
  const int blocks = 256;
  double* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, (size_t)blocks * 512 * 8);
  (void)hipMalloc(&cyc, (size_t)blocks * 8 * 8);
  for (int w = 0; w < 2; ++w) launch(blocks, out, cyc);
  (void)hipDeviceSynchronize();
  launch(blocks, out, cyc);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)blocks * 8);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> ev, od;
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < 8; ++w) (w < 4 ? ev : od).push_back((double)h[b * 8 + w] / ((double)iters_even * 8));
  std::sort(ev.begin(), ev.end()); std::sort(od.begin(), od.end());
  // s_memtime counts at 100 MHz-derived constant rate on gfx950? it is the shader clock counter on this part (see
  // tools/mfma_f64_bench.hip: cycles per MFMA alone = 64)
  printf("%-44s even wavefronts %.1f cycles / instruction, odd wavefronts %.1f\n", tag, ev[ev.size() / 2], od[od.size() / 2]);
  (void)hipFree(out); (void)hipFree(cyc);
}

#define RUN(TAG, RE, RO, IT) run(TAG, IT, [](int bl, double* o, unsigned long long* c) { k<RE, RO><<<bl, 512>>>(o, c, IT, 1.0, 0.5); })
#define RUNA(TAG, RE, RO, NA, IT) run(TAG, IT, [](int bl, double* o, unsigned long long* c) { k<RE, RO, NA><<<bl, 512>>>(o, c, IT, 1.0, 0.5); })

int main() {
  RUN("MFMA f64 alone (1 wavefront / SIMD)", 1, 0, 20000);
  RUN("MFMA f64 x 2 wavefronts / SIMD", 1, 1, 20000);
  RUN("VALU f64 fma alone", 2, 0, 20000);
  RUN("VALU f64 fma x 2 wavefronts / SIMD", 2, 2, 20000);
  RUN("MFMA f64 (even) beside VALU f64 fma (odd)", 1, 2, 20000);
  RUN("MFMA f64 (even) beside VALU f32 fma (odd)", 1, 3, 20000);
  RUN("MFMA f64 (even) beside VALU u32 mad (odd)", 1, 4, 20000);
  RUNA("MFMA f64, ONE accumulator (dependent chain), alone", 1, 0, 1, 20000);
  RUNA("MFMA f64 dependent chain (even) beside VALU f64 (odd)", 1, 2, 1, 20000);
  RUNA("MFMA f64 two accumulators (even) beside VALU f64 (odd)", 1, 2, 2, 20000);
  RUN("VALU f64 fma (even, older) beside MFMA f64 (odd)", 2, 1, 20000);
  RUN("VALU f32 fma alone", 3, 0, 20000);
  RUN("VALU u32 mad alone", 4, 0, 20000);
  return 0;
}
