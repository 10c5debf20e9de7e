// Micro-benchmark: how fast can gfx950 absorb a write-only N x N fp64 matrix (the Gram build's
// algorithmic traffic)?  Gives the floor the Gram kernel is judged against at each N.
// Build: hipcc -O3 -w --offload-arch=gfx950 tools/store_floor.hip -o tools/store_floor.bin
// Run under `rocprofv3 --kernel-trace --stats` for per-kernel durations; prints event-timed averages too.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// linear fill, grid-stride, 16 B per lane per store
template <int NT>
__global__ __launch_bounds__(256) void fill_linear(double2* __restrict__ p, size_t n2, double v) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    if (NT) __builtin_nontemporal_store(v, &p[i].x), __builtin_nontemporal_store(v, &p[i].y);
    else p[i] = make_double2(v, v);
  }
}

// tile fill: block -> TR x TC tile of an N x N row-major matrix, 32 lanes = one 512 B row segment
typedef double double2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_through2(double* p, double x, double y) {   // write-through: line leaves the L2 at once
  const double2_t v = {x, y};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <int TR, int TC, int WT = 0>
__global__ __launch_bounds__(256) void fill_tile(double* __restrict__ S, int N, double v) {
  const int ntc = N / TC;
  const int bi = blockIdx.x / ntc, bj = blockIdx.x % ntc;
  constexpr int LPR = TC / 2;            // lanes per row
  constexpr int RPP = 256 / LPR;         // rows per pass
  const int c2 = (threadIdx.x % LPR) * 2, r0 = threadIdx.x / LPR;
#pragma unroll
  for (int a = 0; a < TR / RPP; ++a) {
    const int r = a * RPP + r0;
    double* dst = S + (size_t)(bi * TR + r) * N + bj * TC + c2;
    if (WT) store_through2(dst, v, v + r);
    else *reinterpret_cast<double2*>(dst) = make_double2(v, v + r);
  }
}


// the Gram kernel's store pattern without its arithmetic: upper-triangle 64x64 tiles, wave w owns rows
// 16w..16w+15; direct pieces as 8 B/lane (16 lanes = one 128 B line, 4 rows per instruction), mirror
// pieces as 16 B/lane (8 lanes = one line, 8 rows per instruction)
template <int MODE>   // 0: as described; 1: both halves as 512 B row segments (32 lanes x 16 B)
__global__ __launch_bounds__(256) void fill_gramlike(double* __restrict__ S, int N, int nt, double v) {
  const int t = blockIdx.x;
  const double q = 2.0 * nt + 1.0;
  int bi = (int)floor((q - sqrt(q * q - 8.0 * (double)t)) * 0.5);
  while (bi > 0 && t < bi * nt - bi * (bi - 1) / 2) --bi;
  while (t >= (bi + 1) * nt - (bi + 1) * bi / 2) ++bi;
  const int bj = bi + (t - (bi * nt - bi * (bi - 1) / 2));
  const int i0 = bi * 64, j0 = bj * 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (MODE == 0) {
    const int lr = lane & 15, lk = lane >> 4;
    double* ddst = S + (size_t)(i0 + w * 16 + lk) * N + j0 + lr;
    const int mc = lane >> 3, mp = (lane & 7) * 2;
    double* mdst = S + (size_t)(j0 + mc) * N + i0 + w * 16 + mp;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ddst[(size_t)(4 * r) * N + j * 16] = v + r;
      if (bi == bj) continue;
#pragma unroll
      for (int h = 0; h < 2; ++h)
        *reinterpret_cast<double2*>(mdst + (size_t)(j * 16 + 8 * h) * N) = make_double2(v, v + h);
    }
  } else {
    const int c2 = (threadIdx.x & 31) * 2, r0 = threadIdx.x >> 5;
#pragma unroll
    for (int a = 0; a < 8; ++a)
      *reinterpret_cast<double2*>(S + (size_t)(i0 + a * 8 + r0) * N + j0 + c2) = make_double2(v, v + a);
    if (bi == bj) return;
#pragma unroll
    for (int a = 0; a < 8; ++a)
      *reinterpret_cast<double2*>(S + (size_t)(j0 + a * 8 + r0) * N + i0 + c2) = make_double2(v, v + a);
  }
}

// 128 x 128 super-tiles of the upper triangle (diagonal included), one workgroup per 32 x 128 strip of a super-tile:
// the direct strip as full 1 KB rows (64 lanes x 16 B), its mirror image (128 rows x 32 columns) as 256 B row pieces
// (16 lanes x 16 B, 4 rows per instruction).  WT: write-through stores.
template <int WT>
__global__ __launch_bounds__(256) void fill_strips(double* __restrict__ S, int N, int nt, double v) {
  const int t = blockIdx.x >> 2, strip = blockIdx.x & 3;
  const double q = 2.0 * nt + 1.0;
  int bi = (int)floor((q - sqrt(q * q - 8.0 * (double)t)) * 0.5);
  while (bi > 0 && t < bi * nt - bi * (bi - 1) / 2) --bi;
  while (t >= (bi + 1) * nt - (bi + 1) * bi / 2) ++bi;
  const int bj = bi + (t - (bi * nt - bi * (bi - 1) / 2));
  const int i0 = bi * 128 + 32 * strip, j0 = bj * 128;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    double* p = S + (size_t)(i0 + 8 * w + a) * N + j0 + 2 * lane;
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const d2_t val = {v, v + a};
    if (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(val) : "memory");
    else *reinterpret_cast<double2*>(p) = make_double2(v, v + a);
  }
  if (bi == bj) return;
  const int mr = lane >> 4, mc = (lane & 15) * 2;
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    double* p = S + (size_t)(j0 + 32 * w + 4 * a + mr) * N + i0 + mc;
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const d2_t val = {v, v + a};
    if (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(val) : "memory");
    else *reinterpret_cast<double2*>(p) = make_double2(v, v + a);
  }
}

template <typename F>
static void timeit(const char* tag, double bytes, F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) launch();
  hipDeviceSynchronize();
  const int reps = 50;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("%-34s %8.2f us/launch (back-to-back)  %7.0f GB/s  frac %.3f\n", tag, us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0);
}

int main() {
  for (int N : {2048, 4096, 8192}) {
    const size_t n = (size_t)N * N;
    double* d;
    hipMalloc(&d, n * 8);
    const double bytes = 8.0 * n;
    char tag[96];
    for (int g : {256, 512, 1024, 2048, 4096}) {
      snprintf(tag, sizeof tag, "N=%d linear grid=%d", N, g);
      timeit(tag, bytes, [&] { fill_linear<0><<<g, 256>>>((double2*)d, n / 2, 1.0); });
    }
    snprintf(tag, sizeof tag, "N=%d linear NT grid=1024", N);
    timeit(tag, bytes, [&] { fill_linear<1><<<1024, 256>>>((double2*)d, n / 2, 1.0); });
    snprintf(tag, sizeof tag, "N=%d tile 64x64", N);
    timeit(tag, bytes, [&] { fill_tile<64, 64><<<(N / 64) * (N / 64), 256>>>(d, N, 1.0); });
    snprintf(tag, sizeof tag, "N=%d tile 64x64 write-through", N);
    timeit(tag, bytes, [&] { fill_tile<64, 64, 1><<<(N / 64) * (N / 64), 256>>>(d, N, 1.0); });
    snprintf(tag, sizeof tag, "N=%d tile 32x64", N);
    timeit(tag, bytes, [&] { fill_tile<32, 64><<<(N / 32) * (N / 64), 256>>>(d, N, 1.0); });
    snprintf(tag, sizeof tag, "N=%d tile 32x128", N);
    timeit(tag, bytes, [&] { fill_tile<32, 128><<<(N / 32) * (N / 128), 256>>>(d, N, 1.0); });
    snprintf(tag, sizeof tag, "N=%d tile 16x256", N);
    timeit(tag, bytes, [&] { fill_tile<16, 256><<<(N / 16) * (N / 256), 256>>>(d, N, 1.0); });
    {
      const int nt = N / 64, nblk = nt * (nt + 1) / 2;
      snprintf(tag, sizeof tag, "N=%d gram-like 128B pieces", N);
      timeit(tag, bytes, [&] { fill_gramlike<0><<<nblk, 256>>>(d, N, nt, 1.0); });
      snprintf(tag, sizeof tag, "N=%d gram-like 512B pieces", N);
      timeit(tag, bytes, [&] { fill_gramlike<1><<<nblk, 256>>>(d, N, nt, 1.0); });
    }
    {
      const int nt = N / 128, nblk = nt * (nt + 1) / 2 * 4;
      snprintf(tag, sizeof tag, "N=%d strips 32x128 + mirror 256B", N);
      timeit(tag, bytes, [&] { fill_strips<0><<<nblk, 256>>>(d, N, nt, 1.0); });
      snprintf(tag, sizeof tag, "N=%d strips, write-through", N);
      timeit(tag, bytes, [&] { fill_strips<1><<<nblk, 256>>>(d, N, nt, 1.0); });
    }
    snprintf(tag, sizeof tag, "N=%d memset", N);
    timeit(tag, bytes, [&] { hipMemsetAsync(d, 0, n * 8, 0); });
    hipFree(d);
  }
  return 0;
}
