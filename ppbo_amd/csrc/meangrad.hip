// f-2: posterior mean and its analytic gradient at M points, the device half of the batched multi-start
// refinement that replaces mu_star's differential evolution (gp_model.py:415-437).
//   mu(x) = sum_i alpha_i k(x, x_i),   d mu / d x_d = sum_i alpha_i dk/dx_d
//   SE      dk/dx_d = -(x_d - x_i,d) / l^2 * k                               (kernels.py:19-25)
//   RQ      dk/dx_d = -(x_d - x_i,d) / l^2 * k / (1 + r^2 / (4 l^2))         (kernels.py:27-34, alpha = 2)
//   camphor dk/dx_d = -(2 pi / l^2) sin(2 pi (x_d - x_i,d)) * k  (d != 2),   -(x_2 - x_i,2) / (l + 0.05)^2 * k
//                                                                            (kernels.py:36-53)
// One 256-thread workgroup per point: lanes stride over the N design rows with the point held in
// registers, accumulate mu and D gradient components, then a shuffle + LDS reduction.  M is small here
// (a few hundred ascent iterates), so the work per launch is M * N * D * ~6 flops -- microseconds.
#include "common.h"

namespace {

template <int KID, int DP>
__global__ __launch_bounds__(256) void mean_grad_kernel(const double* __restrict__ X, int N, int D, KernParams p,
                                                        const double* __restrict__ alpha,
                                                        const double* __restrict__ Xc, double* __restrict__ mu,
                                                        double* __restrict__ grad) {
  __shared__ double red[4][DP + 1];
  const int c = blockIdx.x;
  double xc[DP], g[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) {
    xc[d] = (d < D) ? Xc[(size_t)c * D + d] : 0.0;
    g[d] = 0.0;
  }
  double m = 0.0;
  for (int i = threadIdx.x; i < N; i += 256) {
    const double* __restrict__ xi = X + (size_t)i * D;
    double dx[DP], s = 0.0;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      dx[d] = (d < D) ? xc[d] - xi[d] : 0.0;
      s += kern_term<KID>(dx[d], d, p);
    }
    const double w = alpha[i] * kern_finish<KID>(s, p);
    m += w;
    if (KID == PPBO_KERNEL_CAMPHOR) {
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        if (d == 2) g[d] -= 2.0 * p.c1 * dx[d] * w;
        else if (d < 6) g[d] -= p.c0 * 3.14159265358979323846 * sinpi(2.0 * dx[d]) * w;
      }
    } else {
      const double coef = (KID == PPBO_KERNEL_SE) ? -2.0 * p.c0 * w : -4.0 * p.c0 * w / (1.0 + p.c0 * s);
#pragma unroll
      for (int d = 0; d < DP; ++d) g[d] += coef * dx[d];
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  m = wave_sum(m);
#pragma unroll
  for (int d = 0; d < DP; ++d) g[d] = wave_sum(g[d]);
  if (lane == 0) {
    red[wave][DP] = m;
#pragma unroll
    for (int d = 0; d < DP; ++d) red[wave][d] = g[d];
  }
  __syncthreads();
  if (threadIdx.x <= DP) {
    const int d = threadIdx.x;
    const double v = (red[0][d] + red[1][d]) + (red[2][d] + red[3][d]);
    if (d == DP) mu[c] = v;
    else if (d < D) grad[(size_t)c * D + d] = v;
  }
}

template <int KID>
void launch_mean_grad(const ppbo_model* m, const KernParams& p, const double* d_Xc, int M, double* d_mu,
                      double* d_grad, hipStream_t s) {
  if (KID == PPBO_KERNEL_CAMPHOR || m->D <= 8)
    mean_grad_kernel<KID, 8><<<M, 256, 0, s>>>(m->d_X, m->N, m->D, p, m->d_alpha, d_Xc, d_mu, d_grad);
  else if (m->D <= 24)
    mean_grad_kernel<KID, 24><<<M, 256, 0, s>>>(m->d_X, m->N, m->D, p, m->d_alpha, d_Xc, d_mu, d_grad);
  else
    mean_grad_kernel<KID, 64><<<M, 256, 0, s>>>(m->d_X, m->N, m->D, p, m->d_alpha, d_Xc, d_mu, d_grad);
}

}  // namespace

extern "C" int ppbo_mean_grad(ppbo_ctx* ctx, const ppbo_model* m, const double* d_Xc, int64_t M, double* d_mu,
                              double* d_grad, void* stream) {
  PPBO_ENTER(ctx);
  PPBO_REQUIRE(ctx, m != nullptr && m->d_X && m->d_alpha, "model X/alpha");
  PPBO_REQUIRE(ctx, m->N > 0 && m->D > 0 && m->D <= 64, "model sizes (D<=64)");
  PPBO_REQUIRE(ctx, m->kernel_id >= 0 && m->kernel_id <= 2, "kernel_id");
  PPBO_REQUIRE(ctx, m->kernel_id != PPBO_KERNEL_CAMPHOR || m->D == 6, "camphor kernel needs D == 6");
  PPBO_REQUIRE(ctx, d_Xc && d_mu && d_grad && M >= 0 && M < (1 << 30), "points / outputs");
  if (M == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const KernParams p = make_kern_params(m->kernel_id, m->theta);
  switch (m->kernel_id) {
    case PPBO_KERNEL_SE: launch_mean_grad<PPBO_KERNEL_SE>(m, p, d_Xc, (int)M, d_mu, d_grad, s); break;
    case PPBO_KERNEL_RQ: launch_mean_grad<PPBO_KERNEL_RQ>(m, p, d_Xc, (int)M, d_mu, d_grad, s); break;
    default: launch_mean_grad<PPBO_KERNEL_CAMPHOR>(m, p, d_Xc, (int)M, d_mu, d_grad, s); break;
  }
  PPBO_LAUNCH_CHECK(ctx);
  return 0;
}
