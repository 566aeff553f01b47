// Elementwise steps of the DDIM sampler with flip test-time augmentation
// (reference common/diffusionpose.py:147-169 model_predictions_fliping, :129-133 predict_noise_from_start,
// :244-254 DDIM update, :260-267 q_sample).  HBM-trivial (one (B,H,F,J,3) tensor per step); what matters is
// matching the reference's arithmetic: fp32 multiplies/adds WITHOUT fma contraction, and fp64 for
// predict_noise_from_start (its coefficients differ by 1e-9 relative at t=999: catastrophic in fp32).
#include "common.h"
#include "kernels.h"

namespace {

// torch.clamp semantics (diffusionpose.py:148,164): NaN stays NaN -- fminf / fmaxf alone would turn it into a bound and
// hide a broken denoiser output behind a plausible pose
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return x != x ? x : fminf(fmaxf(x, lo), hi); }

// xt2[0:B]  = clamp(img, +-1.1 s) / s
// xt2[B:2B] = the same with x negated and left/right joints swapped (perm[j] = source joint of j)
__global__ __launch_bounds__(256) void ddim_pre_kernel(const float* __restrict__ img, float* __restrict__ xt2,
                                                       const int* __restrict__ perm, float scale, size_t total,
                                                       int J) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // index over (b, h, f, j) rows of 3
  if (i >= total) return;
  const float lim = 1.1f * scale;
  const int j = (int)(i % J);
  const size_t src = i - j + perm[j];
  float a[3], f[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    a[c] = clampf(img[i * 3 + c], -lim, lim) / scale;
    f[c] = clampf(img[src * 3 + c], -lim, lim) / scale;
  }
  f[0] = -f[0];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    xt2[i * 3 + c] = a[c];
    xt2[(total + i) * 3 + c] = f[c];
  }
}

__global__ __launch_bounds__(256) void ddim_post_kernel(const float* __restrict__ pred2, const float* __restrict__ img,
                                                        const float* __restrict__ noise, const int* __restrict__ perm,
                                                        float scale, double sqrt_recip, double sqrt_recipm1,
                                                        float c_xstart, float c_noise, float sigma, int last,
                                                        float* __restrict__ x_start, size_t xs_bstride,
                                                        float* __restrict__ img_next, size_t total, size_t per_b_rows,
                                                        int J) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float lim = 1.1f * scale;
  const int j = (int)(i % J);
  const size_t src = total + (i - j + perm[j]);
  const size_t b = i / per_b_rows, r = i % per_b_rows;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float pf = pred2[src * 3 + c];
    if (c == 0) pf = -pf;
    const float pred = __fadd_rn(pred2[i * 3 + c], pf) / 2.0f;
    const float xs = clampf(__fmul_rn(pred, scale), -lim, lim);
    x_start[b * xs_bstride + r * 3 + c] = xs;
    if (!last) {
      const float x = img[i * 3 + c];
      const float pn = (float)((sqrt_recip * (double)x - (double)xs) / sqrt_recipm1);
      const float v = __fadd_rn(__fadd_rn(__fmul_rn(xs, c_xstart), __fmul_rn(c_noise, pn)),
                                __fmul_rn(sigma, noise[i * 3 + c]));
      img_next[i * 3 + c] = v;
    }
  }
}

// out = float( a[b] * (x0 * scale) + bb[b] * noise ) clamped to +-1.1 s, / s   (diffusionpose.py:290-306)
__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                       const double* __restrict__ a, const double* __restrict__ bb,
                                                       float scale, float* __restrict__ out, size_t total,
                                                       size_t per_b) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const size_t b = i / per_b;
  const double lim = 1.1 * (double)scale;
  double v = a[b] * (double)__fmul_rn(x0[i], scale) + bb[b] * (double)noise[i];
  v = fmin(fmax(v, -lim), lim) / (double)scale;
  out[i] = (float)v;
}

}  // namespace

int d3dp_launch_ddim_pre(const float* img, float* xt2, const int* perm, float scale, int B, int per_b, int J,
                         hipStream_t st) {
  const size_t total = (size_t)B * per_b / 3;
  hipLaunchKernelGGL(ddim_pre_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, img, xt2, perm, scale,
                     total, J);
  return 0;
}

int d3dp_launch_ddim_post(const float* pred2, const float* img, const float* noise, const int* perm, float scale,
                          double sqrt_recip, double sqrt_recipm1, float c_xstart, float c_noise, float sigma,
                          int last, float* x_start, size_t xs_bstride, float* img_next, int B, int per_b, int J,
                          hipStream_t st) {
  const size_t total = (size_t)B * per_b / 3;
  hipLaunchKernelGGL(ddim_post_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pred2, img, noise,
                     perm, scale, sqrt_recip, sqrt_recipm1, c_xstart, c_noise, sigma, last, x_start, xs_bstride,
                     img_next, total, (size_t)per_b / 3, J);
  return 0;
}

int d3dp_launch_q_sample(const float* x0, const float* noise, const double* a, const double* b, float scale,
                         float* out, int B, int per_b, hipStream_t st) {
  const size_t total = (size_t)B * per_b;
  hipLaunchKernelGGL(q_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x0, noise, a, b, scale,
                     out, total, (size_t)per_b);
  return 0;
}
