// Development harness for rff_project_kernel (csrc/rff.hip): variants that drop one phase each, and alternative
// tile shapes, timed over back-to-back launches.  Not part of the library.
// Build: hipcc -O3 -w --offload-arch=gfx950 tools/dev/rff_dev.hip -o tools/dev/rff_dev.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int TS = 64;

__device__ __forceinline__ void store_through(double* p, double v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

typedef double double2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_through2(double* p, double x, double y) {
  const double2_t v = {x, y};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ double swap_adjacent(double v) {      // value of lane ^ 1 (DPP quad_perm [1,0,3,2])
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

template <int TERMS>
__device__ __forceinline__ double rff_cos_fast(double x) {
  const double ax = fabs(x);
  const double kf = rint(fma(ax, 3.18309886183790671538e-01, -0.5));
  const double n = fma(2.0, kf, 1.0);
  double r = fma(-n, 1.57079632673412561417e+00, ax);
  r = fma(-n, 6.07710050630396597660e-11, r);
  r = fma(-n, 2.02226624871116645580e-21, r);
  if (TERMS >= 4) r = fma(-n, 8.47842766036889956997e-32, r);
  const double z = r * r;
  double q = -0x1.26805104f0fb2p-57;
  q = fma(q, z, 0x1.94fe99353aaa5p-49);
  q = fma(q, z, -0x1.ae7eb995a1519p-41);
  q = fma(q, z, 0x1.61246051b86e7p-33);
  q = fma(q, z, -0x1.ae64567d5b22ap-26);
  q = fma(q, z, 0x1.71de3a5569d7bp-19);
  q = fma(q, z, -0x1.a01a01a019fdbp-13);
  q = fma(q, z, 0x1.1111111111111p-7);
  q = fma(q, z, -0x1.5555555555555p-3);
  const double sn = fma(r * z, q, r);
  return (((int)kf) & 1) ? sn : -sn;
}

// VAR bit flags: 1 no stores, 2 no cos, 4 no MFMA, 8 plain stores, 32 = nothing but stores
template <int DP, int NT, int VAR>
__global__ __launch_bounds__(512) void rff_k(const double* __restrict__ X, int N, int D, const double* __restrict__ W,
                                             int F, const double* __restrict__ b, double scale,
                                             double* __restrict__ Phi) {
  constexpr int LD = DP + 2, Q = DP / 4;
  __shared__ __attribute__((aligned(16))) double Wa[TS * LD];
  __shared__ __attribute__((aligned(16))) double Xb[NT * TS * LD];
  const int f0 = blockIdx.y * TS, n0 = blockIdx.x * (TS * NT);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, w = wv & 3, jh = wv >> 2;
  const int lr = lane & 15, lk = lane >> 4;
  double af[Q], bv[4];
  if (VAR != 32) {
    const int part = threadIdx.x & 3;
#pragma unroll
    for (int p = 0; p < (NT + 2) / 2; ++p) {
      const int row = p * 128 + (threadIdx.x >> 2);
      if (row >= (NT + 1) * TS) break;
      const bool isw = row < TS;
      const double* src = isw ? W : X;
      double* dst = isw ? Wa : Xb;
      const int r = isw ? row : row - TS;
      const int g = (isw ? f0 : n0) + r, lim = isw ? F : N;
#pragma unroll
      for (int k = 0; k < Q; ++k) {
        const int d = part * Q + k;
        dst[r * LD + d] = (d < D && g < lim) ? src[(size_t)g * D + d] : 0.0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < Q; ++kk) af[kk] = Wa[(w * 16 + lr) * LD + kk * 4 + lk];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = f0 + w * 16 + lk + 4 * r;
      bv[r] = (f < F) ? b[f] : 0.0;
    }
  }
  double sink = 0.0;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 4 * t + 2 * jh + jj;
      double4_t acc = double4_t{0.1 * lane, 0.2, 0.3 * j, 0.4};
      if (VAR != 32 && !(VAR & 4)) {
        acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < Q; ++kk)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], Xb[(j * 16 + lr) * LD + kk * 4 + lk], acc, 0, 0, 0);
      } else if (VAR != 32) {
        acc[0] += Xb[(j * 16 + lr) * LD + lk]; acc[1] += af[0];
      }
      const int n = n0 + j * 16 + lr;
      double cv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (VAR == 32) cv[r] = acc[r];
        else if (VAR & 2) cv[r] = acc[r] + bv[r];
        else cv[r] = scale * rff_cos_fast<3>(acc[r] + bv[r]);
      }
      if (VAR & 512) {   // 16-byte stores: lane pairs trade one value so that each lane holds two adjacent columns of one row
        const bool odd = lane & 1;
        const int nb = n0 + j * 16 + (lr & ~1);
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
          const double got = swap_adjacent(odd ? cv[2 * rp] : cv[2 * rp + 1]);
          const double x0 = odd ? got : cv[2 * rp], x1 = odd ? cv[2 * rp + 1] : got;
          const int f = f0 + w * 16 + lk + 4 * (2 * rp + (odd ? 1 : 0));
          if (f < F && nb + 1 < N) store_through2(Phi + (size_t)f * N + nb, x0, x1);
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = f0 + w * 16 + lk + 4 * r;
        if (VAR & 1) sink += cv[r];
        else if ((VAR & 16) && r != 0) sink += cv[r];
        else if ((VAR & 64) && r > 1) sink += cv[r];
        else if (f < F && n < N) {
          if ((VAR & 256) && (VAR & 8)) Phi[((size_t)f * N + n) & 0x1ffff] = cv[r];              // plain, 1 MB footprint: L2 absorbs
          else if (VAR & 256) store_through(Phi + (((size_t)f * N + n) & 0x1ffff), cv[r]);   // same instruction stream, 1 MB footprint
          else if (VAR & 8) Phi[(size_t)f * N + n] = cv[r];
          else store_through(Phi + (size_t)f * N + n, cv[r]);
        }
      }
    }
  }
  if ((VAR & (1 | 16 | 64)) && sink == 1.2345) Phi[0] = sink;
}

// Wave-independent variant: no LDS, no barrier.  One wavefront = 16 feature rows x TPW tiles of 16 points; the W
// fragment lives in registers, the X fragment of the next tile is requested before the current tile's arithmetic.
template <int DP, int TPW, int VAR>
__global__ __launch_bounds__(256) void rff_w(const double* __restrict__ X, int N, int D, const double* __restrict__ W,
                                             int F, const double* __restrict__ b, double scale,
                                             double* __restrict__ Phi) {
  constexpr int Q = DP / 4;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int wave_id = blockIdx.x * 4 + wv;
  const int strips = (N + 16 * TPW - 1) / (16 * TPW);
  const int f0 = (wave_id / strips) * 16, n0 = (wave_id % strips) * (16 * TPW);
  if (f0 >= F) return;
  double af[Q], bv[4], xf[Q], xn[Q];
#pragma unroll
  for (int kk = 0; kk < Q; ++kk) {
    const int d = kk * 4 + lk;
    af[kk] = (d < D && f0 + lr < F) ? W[(size_t)(f0 + lr) * D + d] : 0.0;
    xf[kk] = (d < D && n0 + lr < N) ? X[(size_t)(n0 + lr) * D + d] : 0.0;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = (f0 + lk + 4 * r < F) ? b[f0 + lk + 4 * r] : 0.0;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int nn = n0 + 16 * (t + 1) + lr;
    if (t + 1 < TPW) {
#pragma unroll
      for (int kk = 0; kk < Q; ++kk) {
        const int d = kk * 4 + lk;
        xn[kk] = (d < D && nn < N) ? X[(size_t)nn * D + d] : 0.0;
      }
    }
    double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < Q; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], xf[kk], acc, 0, 0, 0);
    const int n = n0 + 16 * t + lr;
    double cv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cv[r] = (VAR & 2) ? acc[r] + bv[r] : scale * rff_cos_fast<3>(acc[r] + bv[r]);
    if (VAR & 512) {
      const bool odd = lane & 1;
      const int nb = n0 + 16 * t + (lr & ~1);
#pragma unroll
      for (int rp = 0; rp < 2; ++rp) {
        const double got = swap_adjacent(odd ? cv[2 * rp] : cv[2 * rp + 1]);
        const double x0 = odd ? got : cv[2 * rp], x1 = odd ? cv[2 * rp + 1] : got;
        const int f = f0 + lk + 4 * (2 * rp + (odd ? 1 : 0));
        if (f < F && nb + 1 < N) store_through2(Phi + (size_t)f * N + nb, x0, x1);
      }
    } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = f0 + lk + 4 * r;
      if (f < F && n < N) {
        if (VAR & 8) Phi[(size_t)f * N + n] = cv[r];
        else store_through(Phi + (size_t)f * N + n, cv[r]);
      }
    }
    }
#pragma unroll
    for (int kk = 0; kk < Q; ++kk) xf[kk] = xn[kk];
  }
}

template <int TPW, int VAR>
float runw(const double* X, int N, int D, const double* W, int F, const double* b, double* Phi, int reps) {
  const int strips = (N + 16 * TPW - 1) / (16 * TPW);
  const int waves = ((F + 15) / 16) * strips;
  dim3 grid((waves + 3) / 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) rff_w<20, TPW, VAR><<<grid, 256>>>(X, N, D, W, F, b, 0.01, Phi);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) rff_w<20, TPW, VAR><<<grid, 256>>>(X, N, D, W, F, b, 0.01, Phi);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3f;
}

// deferred stores: all tiles of the strip are computed into registers first, then stored in one burst
// (row segments of a workgroup reach memory together).  ORDER 0: tile-major, 1: row-major (r outer)
template <int DP, int NT, int ORDER>
__global__ __launch_bounds__(512) void rff_d(const double* __restrict__ X, int N, int D, const double* __restrict__ W,
                                             int F, const double* __restrict__ b, double scale,
                                             double* __restrict__ Phi) {
  constexpr int LD = DP + 2, Q = DP / 4;
  __shared__ __attribute__((aligned(16))) double Wa[TS * LD];
  __shared__ __attribute__((aligned(16))) double Xb[NT * TS * LD];
  const int f0 = blockIdx.y * TS, n0 = blockIdx.x * (TS * NT);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, w = wv & 3, jh = wv >> 2;
  const int lr = lane & 15, lk = lane >> 4;
  double af[Q], bv[4];
  {
    const int part = threadIdx.x & 3;
#pragma unroll
    for (int p = 0; p < (NT + 2) / 2; ++p) {
      const int row = p * 128 + (threadIdx.x >> 2);
      if (row >= (NT + 1) * TS) break;
      const bool isw = row < TS;
      const double* src = isw ? W : X;
      double* dst = isw ? Wa : Xb;
      const int r = isw ? row : row - TS;
      const int g = (isw ? f0 : n0) + r, lim = isw ? F : N;
#pragma unroll
      for (int k = 0; k < Q; ++k) {
        const int d = part * Q + k;
        dst[r * LD + d] = (d < D && g < lim) ? src[(size_t)g * D + d] : 0.0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < Q; ++kk) af[kk] = Wa[(w * 16 + lr) * LD + kk * 4 + lk];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = f0 + w * 16 + lk + 4 * r;
      bv[r] = (f < F) ? b[f] : 0.0;
    }
  }
  double cv[2 * NT][4];
#pragma unroll
  for (int i = 0; i < 2 * NT; ++i) {
    // this wave's tiles are CONTIGUOUS along n: j = jh * 2 * NT + i  (a row segment of 2 NT x 128 B per wave)
    const int j = jh * 2 * NT + i;
    double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < Q; ++kk)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], Xb[(j * 16 + lr) * LD + kk * 4 + lk], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) cv[i][r] = scale * rff_cos_fast<3>(acc[r] + bv[r]);
  }
  if (ORDER == 0) {
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = f0 + w * 16 + lk + 4 * r, n = n0 + (jh * 2 * NT + i) * 16 + lr;
        if (f < F && n < N) store_through(Phi + (size_t)f * N + n, cv[i][r]);
      }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 2 * NT; ++i) {
        const int f = f0 + w * 16 + lk + 4 * r, n = n0 + (jh * 2 * NT + i) * 16 + lr;
        if (f < F && n < N) store_through(Phi + (size_t)f * N + n, cv[i][r]);
      }
  }
}

template <int NT, int ORDER>
float rund(const double* X, int N, int D, const double* W, int F, const double* b, double* Phi, int reps) {
  dim3 grid((N + TS * NT - 1) / (TS * NT), (F + TS - 1) / TS);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) rff_d<20, NT, ORDER><<<grid, 512>>>(X, N, D, W, F, b, 0.01, Phi);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) rff_d<20, NT, ORDER><<<grid, 512>>>(X, N, D, W, F, b, 0.01, Phi);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3f;
}

template <int NT, int VAR>
float run(const double* X, int N, int D, const double* W, int F, const double* b, double* Phi, int reps) {
  dim3 grid((N + TS * NT - 1) / (TS * NT), (F + TS - 1) / TS);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) rff_k<20, NT, VAR><<<grid, 512>>>(X, N, D, W, F, b, 0.01, Phi);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) rff_k<20, NT, VAR><<<grid, 512>>>(X, N, D, W, F, b, 0.01, Phi);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 2048, F = argc > 2 ? atoi(argv[2]) : 4096, D = 20;
  std::vector<double> hX((size_t)N * D), hW((size_t)F * D), hb(F);
  srand(1);
  for (auto& v : hX) v = rand() / (double)RAND_MAX;
  for (auto& v : hW) v = (rand() / (double)RAND_MAX - 0.5) * 10.0;
  for (auto& v : hb) v = rand() / (double)RAND_MAX * 6.28;
  double *X, *W, *b, *Phi;
  hipMalloc(&X, hX.size() * 8); hipMalloc(&W, hW.size() * 8); hipMalloc(&b, hb.size() * 8);
  hipMalloc(&Phi, (size_t)F * N * 8);
  hipMemcpy(X, hX.data(), hX.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(W, hW.data(), hW.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), hb.size() * 8, hipMemcpyHostToDevice);
  const int reps = 200;
  for (int i = 0; i < 3000; ++i) rff_k<20, 2, 0><<<dim3(N / 128, F / 64), 512>>>(X, N, D, W, F, b, 0.01, Phi);   // clocks up
  hipDeviceSynchronize();
  printf("N=%d F=%d  bytes %.1f MB\n", N, F, (double)F * N * 8 / 1e6);
#define ROW(NT) \
  printf("NT=%d: full %6.2f | no stores %6.2f | no cos %6.2f | no mfma %6.2f | no cos+mfma %6.2f | stores only %6.2f | plain stores %6.2f us\n", NT, \
         run<NT, 0>(X, N, D, W, F, b, Phi, reps), run<NT, 1>(X, N, D, W, F, b, Phi, reps), run<NT, 2>(X, N, D, W, F, b, Phi, reps), \
         run<NT, 4>(X, N, D, W, F, b, Phi, reps), run<NT, 6>(X, N, D, W, F, b, Phi, reps), run<NT, 32>(X, N, D, W, F, b, Phi, reps), \
         run<NT, 8>(X, N, D, W, F, b, Phi, reps))
  ROW(1); ROW(2); ROW(4);
  printf("NT=2 stores folded into 1 MB: write-through %6.2f | plain %6.2f us\n", run<2, 256>(X, N, D, W, F, b, Phi, reps), run<2, 264>(X, N, D, W, F, b, Phi, reps));
  printf("16-byte stores: NT=1 %6.2f NT=2 %6.2f NT=4 %6.2f us\n", run<1, 512>(X, N, D, W, F, b, Phi, reps), run<2, 512>(X, N, D, W, F, b, Phi, reps), run<4, 512>(X, N, D, W, F, b, Phi, reps));
  printf("NT=2 quarter stores %6.2f | half stores %6.2f us\n", run<2, 16>(X, N, D, W, F, b, Phi, reps), run<2, 64>(X, N, D, W, F, b, Phi, reps));
  printf("deferred stores: NT=1 %6.2f/%6.2f  NT=2 %6.2f/%6.2f  NT=4 %6.2f/%6.2f us (tile-major/row-major)\n",
         rund<1, 0>(X, N, D, W, F, b, Phi, reps), rund<1, 1>(X, N, D, W, F, b, Phi, reps), rund<2, 0>(X, N, D, W, F, b, Phi, reps),
         rund<2, 1>(X, N, D, W, F, b, Phi, reps), rund<4, 0>(X, N, D, W, F, b, Phi, reps), rund<4, 1>(X, N, D, W, F, b, Phi, reps));
#define ROWW(TPW) printf("wave-independent TPW=%2d: full %6.2f | no cos %6.2f | 16-byte stores %6.2f us\n", TPW, \
    runw<TPW, 0>(X, N, D, W, F, b, Phi, reps), runw<TPW, 2>(X, N, D, W, F, b, Phi, reps), runw<TPW, 512>(X, N, D, W, F, b, Phi, reps))
  ROWW(2); ROWW(4); ROWW(8); ROWW(16); ROWW(32);
  return 0;
}
