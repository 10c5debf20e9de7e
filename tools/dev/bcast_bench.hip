// round 6: what does a lane broadcast cost a LONE wavefront?  (the Cholesky slab factor is ~300 v_readlane + ~280 fp64
// instructions and takes 2.4 us = ~10 cycles per instruction whatever their order)
// build: hipcc -O3 -w --offload-arch=gfx950 tools/dev/bcast_bench.hip -o tools/dev/bcast_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double lane_bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* t, int iters) {
  __shared__ double sh[64];
  const int lane = threadIdx.x;
  double a[16], src[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = lane * 0.001 + i; src[i] = 1.0 + 1e-9 * (lane + i); }
  sh[lane] = 1.0 + 1e-9 * lane;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) a[i] = fma(-src[i], lane_bcast(src[(i + 1) & 15], i), a[i]);               // 2 readlanes + fma
      if (MODE == 1) a[i] = fma(-src[i], sh[i + (it & 1)], a[i]);                                  // LDS broadcast read + fma
      if (MODE == 2) a[i] = fma(-src[i], src[(i + 1) & 15], a[i]);                                 // fma only
      if (MODE == 3) a[i] += __hiloint2double(0, __builtin_amdgcn_readlane(__double2loint(src[i]), i));   // 1 readlane + add
      if (MODE == 4) {                                                                             // 2 ds_bpermute + fma
        const int lo = __builtin_amdgcn_ds_bpermute(4 * i, __double2loint(src[(i + 1) & 15]));
        const int hi = __builtin_amdgcn_ds_bpermute(4 * i, __double2hiint(src[(i + 1) & 15]));
        a[i] = fma(-src[i], __hiloint2double(hi, lo), a[i]);
      }
    }
    if (MODE == 0 || MODE == 3 || MODE == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) src[i] += 1e-12;     // keep the broadcasts from being hoisted
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  out[lane] = s;
  if (lane == 0) t[0] = t1 - t0;
}
template <int MODE>
void run(const char* name, double* out, unsigned long long* t, int per) {
  const int iters = 2000;
  k<MODE><<<1, 64>>>(out, t, iters);
  k<MODE><<<1, 64>>>(out, t, iters);
  hipDeviceSynchronize();
  unsigned long long h;
  hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  const double us = h / 100.0;
  printf("%-34s %8.1f us for %d x 16 groups: %6.1f cycles per group (2.4 GHz), %5.1f per instruction (%d per group)\n", name, us,
         iters, us * 2400.0 / (iters * 16.0), us * 2400.0 / (iters * 16.0 * per), per);
}
// dependent chains of one wavefront: cycles per link
template <int MODE>
__global__ __launch_bounds__(64) void chain(double* out, unsigned long long* t, int iters) {
  const int lane = threadIdx.x;
  double x = 1.5 + 1e-3 * lane, v = 2.0 + 1e-3 * lane;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) x = fma(x, 0.999999, 1e-7);                                   // one fp64 fma
      if (MODE == 1) x = __builtin_amdgcn_rsq(x) + 1.0;                            // v_rsq_f64 + add
      if (MODE == 2) x = lane_bcast(x, i) * 1.0000001;                             // 2 v_readlane + mul
      if (MODE == 3) {                                                             // the slab's pivot chain
        const double d = lane_bcast(v, i);
        double y = __builtin_amdgcn_rsq(d);
        const double h = -0.5 * d;
        y = y * fma(h, y * y, 1.5);
        y = y * fma(h, y * y, 1.5);
        const double aj = v * y;
        v = fma(-aj, 1e-3 * y, v + 1.0);
      }
      if (MODE == 4) x = __builtin_amdgcn_rcp(x) + 1.0;                            // v_rcp_f64 + add
      if (MODE == 5) x = __builtin_sqrt(x) + 1.0;                                  // sqrt (library sequence) + add
      if (MODE == 6) x = (double)__builtin_amdgcn_rsqf((float)x) + 1.0;            // cvt + v_rsq_f32 + cvt + add
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  out[lane] = x + v;
  if (lane == 0) t[0] = t1 - t0;
}
template <int MODE>
void runc(const char* name, double* out, unsigned long long* t) {
  const int iters = 2000;
  chain<MODE><<<1, 64>>>(out, t, iters);
  chain<MODE><<<1, 64>>>(out, t, iters);
  hipDeviceSynchronize();
  unsigned long long h;
  hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  printf("chain: %-40s %7.1f cycles per link\n", name, h / 100.0 * 2400.0 / (iters * 16.0));
}
int main() {
  double* out; unsigned long long* t;
  hipMalloc(&out, 64 * 8); hipMalloc(&t, 8);
  run<2>("fma only", out, t, 1);
  run<0>("2 v_readlane + fma", out, t, 3);
  run<3>("1 v_readlane + add", out, t, 2);
  run<1>("ds_read_b64 (broadcast) + fma", out, t, 2);
  run<4>("2 ds_bpermute_b32 + fma", out, t, 3);
  runc<0>("fma", out, t);
  runc<1>("v_rsq_f64 + add", out, t);
  runc<4>("v_rcp_f64 + add", out, t);
  runc<5>("sqrt() + add", out, t);
  runc<6>("cvt, v_rsq_f32, cvt + add", out, t);
  runc<2>("2 v_readlane + mul", out, t);
  runc<3>("pivot: bcast, rsq, 2 Newton, scale, fma", out, t);
  return 0;
}
