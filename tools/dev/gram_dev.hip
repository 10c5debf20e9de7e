// Development harness for the Gram kernel: variants that drop one phase each (stores / exp / MFMA / loads)
// to see which phase the launch time is made of.  Not part of the library.
// Build: hipcc -O3 -w --offload-arch=gfx950 tools/dev/gram_dev.hip -o tools/dev/gram_dev.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef double double4_t __attribute__((ext_vector_type(4)));
struct KernParams { double sf2, c0, c1; };
constexpr int TS = 64;

__device__ __forceinline__ double exp_nonpos(double x) {
  x = fmax(x, -750.0);
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

// VAR bit flags: 1 no stores, 2 no exp, 4 no MFMA, 8 no X loads; 32 = nothing but stores
__device__ unsigned long long* g_trace = nullptr;

template <int DP, int VAR>
__global__ __launch_bounds__(256) void gram_v4(const double* __restrict__ X, int N, int D, KernParams p,
                                               double shrink, double* __restrict__ Sigma, int nt) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
  constexpr int LD = DP + 2, LT = 18;
  double* Xa = smem;
  double* Xb = Xa + TS * LD;
  double* na = Xb + TS * LD;
  double* nb = na + TS;
  double* Tw = nb + TS;
  const int t = blockIdx.x;
  const double q = 2.0 * nt + 1.0;
  int bi = (int)floor((q - sqrt(q * q - 8.0 * (double)t)) * 0.5);
  while (bi > 0 && t < bi * nt - bi * (bi - 1) / 2) --bi;
  while (t >= (bi + 1) * nt - (bi + 1) * bi / 2) ++bi;
  const int bj = bi + (t - (bi * nt - bi * (bi - 1) / 2));
  const int i0 = bi * TS, j0 = bj * TS;
  if (VAR != 32) {
    constexpr int Q = DP / 4;
    const int r = threadIdx.x >> 2, part = threadIdx.x & 3;
    double xa[Q], xb[Q];
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      const int d = part * Q + k;
      if (VAR & 8) { xa[k] = 0.001 * (r + d) + shrink; xb[k] = 0.002 * (r - d) + shrink; }
      else {
        xa[k] = (d < D && i0 + r < N) ? X[(size_t)(i0 + r) * D + d] : 0.0;
        xb[k] = (d < D && j0 + r < N) ? X[(size_t)(j0 + r) * D + d] : 0.0;
      }
    }
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      sa += xa[k] * xa[k];
      sb += xb[k] * xb[k];
      Xa[r * LD + part * Q + k] = xa[k];
      Xb[r * LD + part * Q + k] = -2.0 * xb[k];
    }
    sa += __shfl_xor(sa, 1, 64); sa += __shfl_xor(sa, 2, 64);
    sb += __shfl_xor(sb, 1, 64); sb += __shfl_xor(sb, 2, 64);
    if (part == 0) { na[r] = sa; nb[r] = sb; }
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  double af[DP / 4], nai[4];
  if (VAR != 32) {
#pragma unroll
    for (int kk = 0; kk < DP / 4; ++kk) af[kk] = Xa[(w * 16 + lr) * LD + kk * 4 + lk];
#pragma unroll
    for (int r = 0; r < 4; ++r) nai[r] = na[w * 16 + lk + 4 * r];
  }
  const double one_minus = 1.0 - shrink;
  const double scale = one_minus * p.sf2;
  const double diagv = one_minus * p.sf2 + shrink * p.sf2;
  double* Ts = Tw + w * 16 * LT;
  double* ddst = Sigma + (size_t)(i0 + w * 16 + lk) * N + j0 + lr;
  const int mc = lane >> 3, mp = (lane & 7) * 2;
  double* mdst = Sigma + (size_t)(j0 + mc) * N + i0 + w * 16 + mp;
  const bool do_store = !(VAR & 1) || (shrink == 123.0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double v[4];
    if (VAR == 32) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = scale + r;
    } else {
      double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
      if (!(VAR & 4)) {
#pragma unroll
        for (int kk = 0; kk < DP / 4; ++kk)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], Xb[(j * 16 + lr) * LD + kk * 4 + lk], acc, 0, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = af[r] * Xb[(j * 16 + lr) * LD + r * 4 + lk];
      }
      const double nbj = nb[j * 16 + lr];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double r2 = fmax(acc[r] + (nai[r] + nbj), 0.0);
        v[r] = (VAR & 2) ? scale * r2 : scale * exp_nonpos(-p.c0 * r2);
      }
      if (bi == bj && j == w) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (lk + 4 * r == lr) v[r] = diagv;
      }
    }
    if (do_store) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ddst[(size_t)(4 * r) * N + j * 16] = v[r];
    }
    if (bi == bj) continue;
    if (VAR & 16) {
      // mirror element (col, row): lanes lk = 0..3 of one lr give 4 consecutive doubles (32-byte pieces)
      double* sdst = Sigma + (size_t)(j0 + j * 16 + lr) * N + i0 + w * 16 + lk;
      if (do_store) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sdst[4 * r] = v[r];
      }
      continue;
    }
    if (VAR == 32) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        *reinterpret_cast<double2*>(mdst + (size_t)(j * 16 + 8 * h) * N) = make_double2(v[h], v[h + 2]);
      continue;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Ts[lr * LT + lk + 4 * r] = v[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = mc + 8 * h;
      const double2 m = *reinterpret_cast<const double2*>(Ts + c * LT + mp);
      double* dst = mdst + (size_t)(j * 16 + 8 * h) * N;
      if (do_store) *reinterpret_cast<double2*>(dst) = m;
      else if (m.x == 1234.5) dst[0] = m.y;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (g_trace && threadIdx.x == 0) {
    unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all bits
    unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));     // HW_REG_XCC_ID low bits
    g_trace[3 * blockIdx.x + 0] = t_start;
    g_trace[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    g_trace[3 * blockIdx.x + 2] = ((unsigned long long)xcc << 32) | hwid;
  }
}


// ---- one wave = one 32x32 tile, no block-level barrier, fragments straight from global ----
template <int DP, int VAR>
__global__ __launch_bounds__(64) void gram_w32(const double* __restrict__ X, int N, int D, KernParams p,
                                               double shrink, double* __restrict__ Sigma, int nt) {
  __shared__ __attribute__((aligned(16))) double Ts[16 * 18];
  constexpr int Q = DP / 4, LT = 18;
  // fold the upper triangle of an nt x nt tile grid into a (nt+1) x ceil(nt/2) rectangle
  const int c = blockIdx.x, tr = blockIdx.y;
  int bi, bj;
  if (c < nt - tr) { bi = tr; bj = tr + c; }
  else { bi = nt - 1 - tr; bj = bi + (c - (nt - tr)); if (bi == tr) return; }
  const int i0 = bi * 32, j0 = bj * 32;
  const int lane = threadIdx.x, lr = lane & 15, lk = lane >> 4;
  double af[2][Q], bf[2][Q], na[2], nb[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const double* ra = X + (size_t)(i0 + 16 * a + lr) * D + lk;
    const double* rb = X + (size_t)(j0 + 16 * a + lr) * D + lk;
#pragma unroll
    for (int kk = 0; kk < Q; ++kk) { af[a][kk] = ra[4 * kk]; bf[a][kk] = rb[4 * kk]; }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int kk = 0; kk < Q; ++kk) {
      sa = __builtin_fma(af[a][kk], af[a][kk], sa);
      sb = __builtin_fma(bf[a][kk], bf[a][kk], sb);
      bf[a][kk] *= -2.0;
    }
    sa += __shfl_xor(sa, 16, 64); sa += __shfl_xor(sa, 32, 64);
    sb += __shfl_xor(sb, 16, 64); sb += __shfl_xor(sb, 32, 64);
    na[a] = sa; nb[a] = sb;
  }
  double nai[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) nai[a][r] = __shfl(na[a], lk + 4 * r, 64);
  const double one_minus = 1.0 - shrink;
  const double scale = one_minus * p.sf2;
  const double diagv = one_minus * p.sf2 + shrink * p.sf2;
  double* tile = Sigma + (size_t)i0 * N + j0;       // uniform
  double* mtile = Sigma + (size_t)j0 * N + i0;      // uniform
  const unsigned doff = (unsigned)lk * N + lr;
  const int mc = lane >> 3, mp = (lane & 7) * 2;
  const unsigned moff = (unsigned)mc * N + mp;
  const bool do_store = !(VAR & 1) || (shrink == 123.0);
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < Q; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a][kk], bf[b][kk], acc, 0, 0, 0);
      double v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double r2 = fmax(acc[r] + (nai[a][r] + nb[b]), 0.0);
        v[r] = scale * exp_nonpos(-p.c0 * r2);
      }
      if (bi == bj && a == b) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (lk + 4 * r == lr) v[r] = diagv;
      }
      if (do_store) {
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[doff + (unsigned)(16 * a + 4 * r) * N + 16 * b] = v[r];
      }
      if (bi == bj) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) Ts[lr * LT + lk + 4 * r] = v[r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const double2 m = *reinterpret_cast<const double2*>(Ts + (mc + 8 * h) * LT + mp);
        double* dst = mtile + (moff + (unsigned)(16 * b + 8 * h) * N + 16 * a);
        if (do_store) *reinterpret_cast<double2*>(dst) = m;
        else if (m.x == 1234.5) dst[0] = m.y;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

__global__ __launch_bounds__(256) void empty_kernel(double* out, int n) {
  extern __shared__ double sm[];
  if (n == -1) { sm[threadIdx.x] = 1.0; __syncthreads(); out[blockIdx.x] = sm[255 - threadIdx.x]; }
}
// index math + barrier only
__global__ __launch_bounds__(256) void index_kernel(double* out, int nt, int n) {
  extern __shared__ double sm[];
  const int t = blockIdx.x;
  const double q = 2.0 * nt + 1.0;
  int bi = (int)floor((q - sqrt(q * q - 8.0 * (double)t)) * 0.5);
  while (bi > 0 && t < bi * nt - bi * (bi - 1) / 2) --bi;
  while (t >= (bi + 1) * nt - (bi + 1) * bi / 2) ++bi;
  const int bj = bi + (t - (bi * nt - bi * (bi - 1) / 2));
  sm[threadIdx.x] = bi;
  __syncthreads();
  if (n == -1 || sm[255 - threadIdx.x] + bj == -5.0) out[blockIdx.x] = bj;
}

template <typename F>
static double timeit(F launch) {
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
  return ms * 1e3 / reps;
}

int main() {
  const int D = 20;
  constexpr int DP = 20;
  const size_t lds = (size_t)(2 * TS * (DP + 2) + 2 * TS + 4 * 16 * 18) * sizeof(double);
  for (int N : {2048, 4096, 8192}) {
    std::vector<double> hx((size_t)N * D);
    for (auto& x : hx) x = rand() / (double)RAND_MAX;
    double *dX, *dS;
    hipMalloc(&dX, hx.size() * 8);
    hipMalloc(&dS, (size_t)N * N * 8);
    hipMemcpy(dX, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
    KernParams p{0.25, 0.5 / 0.09, 0.0};
    const int nt = N / 64, nblk = nt * (nt + 1) / 2;
    const double bytes = 8.0 * N * N + 8.0 * N * D;
#define RUN(V, NAME)                                                                               \
    {                                                                                              \
      const double us = timeit([&] { gram_v4<DP, V><<<nblk, 256, lds>>>(dX, N, D, p, 1e-6, dS, nt); }); \
      printf("N=%d %-28s %8.2f us  frac %.3f\n", N, NAME, us, bytes / us * 1e-3 / 8000.0);        \
    }
    {
      double us = timeit([&] { empty_kernel<<<nblk, 256, lds>>>(dS, 0); });
      printf("N=%d %-28s %8.2f us\n", N, "empty, 32.7 KB LDS", us);
      us = timeit([&] { empty_kernel<<<nblk, 256, 0>>>(dS, 0); });
      printf("N=%d %-28s %8.2f us\n", N, "empty, no LDS", us);
      us = timeit([&] { empty_kernel<<<nblk * 4, 64, 0>>>(dS, 0); });
      printf("N=%d %-28s %8.2f us\n", N, "empty, 4x blocks of 64", us);
      us = timeit([&] { index_kernel<<<nblk, 256, lds>>>(dS, nt, 0); });
      printf("N=%d %-28s %8.2f us\n", N, "index math + barrier", us);
    }
    {
      const int nt32 = N / 32;
      dim3 g(nt32 + 1, (nt32 + 1) / 2);
      double us = timeit([&] { gram_w32<DP, 0><<<g, 64>>>(dX, N, D, p, 1e-6, dS, nt32); });
      printf("N=%d %-28s %8.2f us  frac %.3f\n", N, "w32 (wave = 32x32 tile)", us, bytes / us * 1e-3 / 8000.0);
      us = timeit([&] { gram_w32<DP, 1><<<g, 64>>>(dX, N, D, p, 1e-6, dS, nt32); });
      printf("N=%d %-28s %8.2f us  frac %.3f\n", N, "w32, no stores", us, bytes / us * 1e-3 / 8000.0);
      // compare results with the 64x64 kernel
      std::vector<double> r0((size_t)N * N), r1((size_t)N * N);
      hipMemset(dS, 0, (size_t)N * N * 8);
      gram_v4<DP, 0><<<nblk, 256, lds>>>(dX, N, D, p, 1e-6, dS, nt);
      hipMemcpy(r0.data(), dS, r0.size() * 8, hipMemcpyDeviceToHost);
      hipMemset(dS, 0, (size_t)N * N * 8);
      gram_w32<DP, 0><<<g, 64>>>(dX, N, D, p, 1e-6, dS, nt32);
      hipMemcpy(r1.data(), dS, r1.size() * 8, hipMemcpyDeviceToHost);
      double md = 0; size_t asym = 0;
      for (size_t i = 0; i < r0.size(); ++i) { double d = fabs(r0[i] - r1[i]); if (d > md) md = d; }
      for (int i = 0; i < N; i += 7) for (int j = 0; j < N; ++j) if (r1[(size_t)i * N + j] != r1[(size_t)j * N + i]) ++asym;
      printf("N=%d w32 vs v4 max abs diff %.3e, asymmetric entries %zu\n", N, md, asym);
    }
    RUN(0, "production")
    RUN(32, "stores only")
    RUN(16, "mirror by 32B-piece stores")
    RUN(17, "  same, no stores")
    RUN(1, "no stores")
    RUN(3, "no stores, no exp")
    RUN(5, "no stores, no MFMA")
    RUN(9, "no stores, no X loads")
    RUN(7, "no stores, no exp, no MFMA")
    RUN(15, "no stores/exp/MFMA/loads")
    RUN(2, "no exp")
    RUN(4, "no MFMA")
    RUN(6, "no exp, no MFMA")
    for (int var : {0, 15, 1}) {
      unsigned long long* dT;
      hipMalloc(&dT, (size_t)nblk * 24);
      hipMemset(dT, 0, (size_t)nblk * 24);
      hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &dT, sizeof(dT));
      if (var == 0) gram_v4<DP, 0><<<nblk, 256, lds>>>(dX, N, D, p, 1e-6, dS, nt);
      else if (var == 15) gram_v4<DP, 15><<<nblk, 256, lds>>>(dX, N, D, p, 1e-6, dS, nt);
      else gram_v4<DP, 1><<<nblk, 256, lds>>>(dX, N, D, p, 1e-6, dS, nt);
      hipDeviceSynchronize();
      std::vector<unsigned long long> ht((size_t)nblk * 3);
      hipMemcpy(ht.data(), dT, ht.size() * 8, hipMemcpyDeviceToHost);
      unsigned long long* nul = nullptr;
      hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &nul, sizeof(nul));
      char fn[64];
      snprintf(fn, sizeof fn, "gpurun_out/gram_trace_N%d_v%d.bin", N, var);
      FILE* f = fopen(fn, "wb");
      if (f) { fwrite(ht.data(), 8, ht.size(), f); fclose(f); }
      hipFree(dT);
    }
    hipFree(dX); hipFree(dS);
  }
  return 0;
}
