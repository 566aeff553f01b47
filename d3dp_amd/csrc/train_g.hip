// Row kernels of the TRAINING step at a run-time width (round 6; VERDICT r5 missing 4).
//
// The reference trains at any `-cs` its 8 heads divide (common/arguments.py:49, main.py:325 around mixste.py:215-225); train.hip
// instantiates its row kernels for C in {64, 128, 256, 512}, the widths whose Linears run on the split-fp16 matrix-core kernels.
// Any other width trains on the fp32 path of capi.hip (D3DP_TRAIN_IMPL=f32's path: gemm_f32_kernel forward / dgrad / split-K wgrad,
// fp32 row attention forward, VALU attention backward with a run-time head dim) through the kernels below: the formulas of
// train.hip's kernels (same order per row: two-pass statistics, 1 / C as a multiplied reciprocal, fma with gamma / beta; the
// LayerNorm backward in the same three steps), the width an argument, a lane's NVM slots masked behind it.  No operand rows, no
// absmax (the fp32 path has no operand scales).  [dgamma | dbeta] leave as per-workgroup partial rows like in train.hip: the
// fixed-order reduction of capi.hip adds them -- no float atomics here either.
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ int sample_of(int tok, int axis, int F, int J) {
  return axis == 0 ? tok / J : (tok / (F * J)) * J + tok % J;
}

// a lane's slice of a C-channel row: slot i = column i 64 + lane, live while that column exists (masked slots hold 0)
template <int NVM> struct GRow {
  static __device__ __forceinline__ void load(const float* p, int C, int lane, float (&v)[NVM]) {
#pragma unroll
    for (int i = 0; i < NVM; ++i) { const int c = i * 64 + lane; v[i] = c < C ? p[c] : 0.f; }
  }
  static __device__ __forceinline__ void store(float* p, int C, int lane, const float (&v)[NVM]) {
#pragma unroll
    for (int i = 0; i < NVM; ++i) { const int c = i * 64 + lane; if (c < C) p[c] = v[i]; }
  }
  static __device__ __forceinline__ void stats(const float (&v)[NVM], float eps, int C, int lane, float& mean, float& rstd) {
    const float rc = 1.0f / (float)C;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NVM; ++i) s += v[i];
    mean = wave_sum(s) * rc;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NVM; ++i) { const float d = (i * 64 + lane < C) ? v[i] - mean : 0.f; q = fmaf(d, d, q); }
    rstd = 1.0f / sqrtf(wave_sum(q) * rc + eps);
  }
  // (v - mean) rstd gamma + beta in the live slots, 0 behind them
  static __device__ __forceinline__ void affine(float (&v)[NVM], float mean, float rstd, const float (&w)[NVM], const float (&b)[NVM],
                                                int C, int lane) {
#pragma unroll
    for (int i = 0; i < NVM; ++i) v[i] = (i * 64 + lane < C) ? fmaf((v[i] - mean) * rstd, w[i], b[i]) : 0.f;
  }
};

constexpr int kRowBlocksG = 1024;

// x_out = x_in + m[sample] y ; xn = LN(x_out)                                                    (train.hip add_mask_ln_kernel)
template <int NVM>
__global__ __launch_bounds__(256) void add_mask_ln_g_kernel(const float* __restrict__ x_in, const float* __restrict__ y,
                                                            const float* __restrict__ mask, int axis, int F, int J,
                                                            const float* __restrict__ w, const float* __restrict__ b, float eps,
                                                            float* __restrict__ x_out, float* __restrict__ xn, int T, int C) {
  using R = GRow<NVM>;
  const int lane = threadIdx.x & 63;
  float wl[NVM], bl[NVM];
  R::load(w, C, lane, wl);
  R::load(b, C, lane, bl);
  for (int tok = blockIdx.x * 4 + (threadIdx.x >> 6); tok < T; tok += gridDim.x * 4) {
    const float m = mask ? mask[sample_of(tok, axis, F, J)] : 1.0f;
    float v[NVM], yy[NVM];
    R::load(x_in + (size_t)tok * C, C, lane, v);
    R::load(y + (size_t)tok * C, C, lane, yy);
#pragma unroll
    for (int i = 0; i < NVM; ++i) v[i] = v[i] + m * yy[i];
    R::store(x_out + (size_t)tok * C, C, lane, v);
    float mean, rstd;
    R::stats(v, eps, C, lane, mean, rstd);
    R::affine(v, mean, rstd, wl, bl, C, lane);
    if (xn) R::store(xn + (size_t)tok * C, C, lane, v);
  }
}

// x_out = x_in + m[sample] y ; x_next = LN_a(x_out) (+ pos[f]) ; xn = LN_b(x_next)              (train.hip add_mask_ln2_kernel)
template <int NVM>
__global__ __launch_bounds__(256) void add_mask_ln2_g_kernel(const float* __restrict__ x_in, const float* __restrict__ y,
                                                             const float* __restrict__ mask, int axis, int F, int J,
                                                             const float* __restrict__ wa, const float* __restrict__ ba, float eps_a,
                                                             const float* __restrict__ pos, const float* __restrict__ wb,
                                                             const float* __restrict__ bb, float eps_b, float* __restrict__ x_out,
                                                             float* __restrict__ x_next, float* __restrict__ xn, int T, int C) {
  using R = GRow<NVM>;
  const int lane = threadIdx.x & 63;
  float wal[NVM], bal[NVM], wbl[NVM], bbl[NVM];
  R::load(wa, C, lane, wal);
  R::load(ba, C, lane, bal);
#pragma unroll
  for (int i = 0; i < NVM; ++i) { wbl[i] = 0.f; bbl[i] = 0.f; }
  if (wb) { R::load(wb, C, lane, wbl); R::load(bb, C, lane, bbl); }
  for (int tok = blockIdx.x * 4 + (threadIdx.x >> 6); tok < T; tok += gridDim.x * 4) {
    const float m = mask ? mask[sample_of(tok, axis, F, J)] : 1.0f;
    float v[NVM], yy[NVM];
    R::load(x_in + (size_t)tok * C, C, lane, v);
    R::load(y + (size_t)tok * C, C, lane, yy);
#pragma unroll
    for (int i = 0; i < NVM; ++i) v[i] = v[i] + m * yy[i];
    R::store(x_out + (size_t)tok * C, C, lane, v);
    float mean, rstd;
    R::stats(v, eps_a, C, lane, mean, rstd);
    R::affine(v, mean, rstd, wal, bal, C, lane);
    if (pos) {
      float pp[NVM];
      R::load(pos + (size_t)((tok / J) % F) * C, C, lane, pp);
#pragma unroll
      for (int i = 0; i < NVM; ++i) v[i] += pp[i];
    }
    R::store(x_next + (size_t)tok * C, C, lane, v);
    if (wb && xn) {
      R::stats(v, eps_b, C, lane, mean, rstd);
      R::affine(v, mean, rstd, wbl, bbl, C, lane);
      R::store(xn + (size_t)tok * C, C, lane, v);
    }
  }
}

// y = LN(x) (+ pos[f])                                                                           (train.hip ln_pos_kernel)
template <int NVM>
__global__ __launch_bounds__(256) void ln_pos_g_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                       float eps, const float* __restrict__ pos, int F, int J, float* __restrict__ y,
                                                       int T, int C) {
  using R = GRow<NVM>;
  const int lane = threadIdx.x & 63;
  float wl[NVM], bl[NVM];
  R::load(w, C, lane, wl);
  R::load(b, C, lane, bl);
  for (int tok = blockIdx.x * 4 + (threadIdx.x >> 6); tok < T; tok += gridDim.x * 4) {
    float v[NVM];
    R::load(x + (size_t)tok * C, C, lane, v);
    float mean, rstd;
    R::stats(v, eps, C, lane, mean, rstd);
    R::affine(v, mean, rstd, wl, bl, C, lane);
    if (pos) {
      float pp[NVM];
      R::load(pos + (size_t)((tok / J) % F) * C, C, lane, pp);
#pragma unroll
      for (int i = 0; i < NVM; ++i) v[i] += pp[i];
    }
    R::store(y + (size_t)tok * C, C, lane, v);
  }
}

// LayerNorm backward, one LayerNorm or two chained ones per pass over the rows                  (train.hip ln_bwd2_kernel)
//   LN_b:  xhat = (xb - mean) rstd ; gg = dy gamma_b ; g = rstd (gg - mean(gg) - xhat mean(gg xhat)) + dres
//   LN_a (TWO): the same formula on (g, xa, gamma_a) ; dxm = mask[sample] dx
template <int NVM, bool TWO>
__global__ __launch_bounds__(256) void ln_bwd2_g_kernel(const float* dy, const float* __restrict__ xb, const float* __restrict__ wb,
                                                        float eps_b, const float* __restrict__ dres, float* __restrict__ g_out,
                                                        const float* __restrict__ xa, const float* __restrict__ wa, float eps_a,
                                                        float* __restrict__ dx, const float* __restrict__ mask, int axis, int F, int J,
                                                        float* dxm, float* __restrict__ part_b, float* __restrict__ part_a, int T, int C) {
  using R = GRow<NVM>;
  __shared__ float sg[4][NVM * 64], sb[4][NVM * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float rc = 1.0f / (float)C;
  float agb[NVM], abb[NVM], wbl[NVM];
  [[maybe_unused]] float aga[NVM], aba[NVM], wal[NVM];
  R::load(wb, C, lane, wbl);
#pragma unroll
  for (int i = 0; i < NVM; ++i) { agb[i] = 0.f; abb[i] = 0.f; }
  if constexpr (TWO) {
    R::load(wa, C, lane, wal);
#pragma unroll
    for (int i = 0; i < NVM; ++i) { aga[i] = 0.f; aba[i] = 0.f; }
  }
  for (int tok = blockIdx.x * 4 + wv; tok < T; tok += gridDim.x * 4) {
    float v[NVM], d[NVM];
    R::load(xb + (size_t)tok * C, C, lane, v);
    R::load(dy + (size_t)tok * C, C, lane, d);
    float mean, rstd;
    R::stats(v, eps_b, C, lane, mean, rstd);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NVM; ++i) {
      v[i] = (i * 64 + lane < C) ? (v[i] - mean) * rstd : 0.f;     // xhat
      agb[i] = fmaf(d[i], v[i], agb[i]);
      abb[i] += d[i];
      d[i] *= wbl[i];                                              // gg
      s1 += d[i];
      s2 = fmaf(d[i], v[i], s2);
    }
    s1 = wave_sum(s1) * rc;
    s2 = wave_sum(s2) * rc;
#pragma unroll
    for (int i = 0; i < NVM; ++i) d[i] = (i * 64 + lane < C) ? rstd * (d[i] - s1 - v[i] * s2) : 0.f;
    if (dres) {
      float dr[NVM];
      R::load(dres + (size_t)tok * C, C, lane, dr);
#pragma unroll
      for (int i = 0; i < NVM; ++i) d[i] += dr[i];
    }
    if constexpr (TWO) {
      if (g_out) R::store(g_out + (size_t)tok * C, C, lane, d);
      float va[NVM];
      R::load(xa + (size_t)tok * C, C, lane, va);
      R::stats(va, eps_a, C, lane, mean, rstd);
      s1 = 0.f; s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NVM; ++i) {
        va[i] = (i * 64 + lane < C) ? (va[i] - mean) * rstd : 0.f;
        aga[i] = fmaf(d[i], va[i], aga[i]);
        aba[i] += d[i];
        d[i] *= wal[i];
        s1 += d[i];
        s2 = fmaf(d[i], va[i], s2);
      }
      s1 = wave_sum(s1) * rc;
      s2 = wave_sum(s2) * rc;
#pragma unroll
      for (int i = 0; i < NVM; ++i) d[i] = (i * 64 + lane < C) ? rstd * (d[i] - s1 - va[i] * s2) : 0.f;
    }
    R::store(dx + (size_t)tok * C, C, lane, d);
    if (dxm) {
      const float k = mask ? mask[sample_of(tok, axis, F, J)] : 1.0f;
#pragma unroll
      for (int i = 0; i < NVM; ++i) d[i] *= k;
      R::store(dxm + (size_t)tok * C, C, lane, d);
    }
  }
  // this workgroup's [dgamma | dbeta] rows: the four waves' sums in the order 0, 1, 2, 3
  auto flush = [&](const float (&ag)[NVM], const float (&ab)[NVM], float* part) {
#pragma unroll
    for (int i = 0; i < NVM; ++i) { sg[wv][i * 64 + lane] = ag[i]; sb[wv][i * 64 + lane] = ab[i]; }
    __syncthreads();
    float* row = part + (size_t)blockIdx.x * 2 * C;
    for (int c = threadIdx.x; c < C; c += 256) {
      row[c] = ((sg[0][c] + sg[1][c]) + sg[2][c]) + sg[3][c];
      row[C + c] = ((sb[0][c] + sb[1][c]) + sb[2][c]) + sb[3][c];
    }
    __syncthreads();
  };
  flush(agb, abb, part_b);
  if constexpr (TWO) flush(aga, aba, part_a);
}

// pred[t, o] = sum_c z[t, c] W[o, c] + b[o]  (o < 3)                                             (train.hip head_linear_kernel)
__global__ __launch_bounds__(256) void head_linear_g_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ out, int T, int C) {
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= T) return;
  float acc[3] = {0.f, 0.f, 0.f};
  for (int c = lane; c < C; c += 64) {
    const float v = z[(size_t)tok * C + c];
#pragma unroll
    for (int o = 0; o < 3; ++o) acc[o] = fmaf(v, w[o * C + c], acc[o]);
  }
#pragma unroll
  for (int o = 0; o < 3; ++o) acc[o] = wave_sum(acc[o]) + b[o];
  if (lane < 3) out[(size_t)tok * 3 + lane] = lane == 0 ? acc[0] : (lane == 1 ? acc[1] : acc[2]);
}

// time MLP backward: one WAVE per hidden unit k of the 2 C                                        (train.hip time_mlp_bwd_kernel)
template <int NVM>
__global__ __launch_bounds__(256) void time_mlp_bwd_g_kernel(const int64_t* __restrict__ t, const float* __restrict__ freq,
                                                             const float* __restrict__ w1, const float* __restrict__ b1,
                                                             const float* __restrict__ w2, const float* __restrict__ dtemb,
                                                             float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                                             float* __restrict__ db2, int B, int C) {
  constexpr float kInvSqrt2 = 0.70710678118654752440f, kInvSqrt2Pi = 0.39894228040143267794f;
  const int half = C / 2;
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);     // hidden unit
  if (k >= 2 * C) return;
  float w1r[NVM], w2c[NVM], fr[NVM], g1[NVM], g2[NVM];
#pragma unroll
  for (int i = 0; i < NVM; ++i) {
    const int j = i * 64 + lane;
    const bool in = j < C;
    w1r[i] = in ? w1[(size_t)k * C + j] : 0.f;
    w2c[i] = in ? w2[(size_t)j * 2 * C + k] : 0.f;
    fr[i] = in ? freq[j < half ? j : j - half] : 0.f;
    g1[i] = 0.f; g2[i] = 0.f;
  }
  const float bias = b1[k];
  float gb1 = 0.f;
  for (int b = 0; b < B; ++b) {
    const float tv = (float)t[b];
    float e[NVM], dy[NVM];
    float pre = 0.f, dg = 0.f;
#pragma unroll
    for (int i = 0; i < NVM; ++i) {
      const int j = i * 64 + lane;
      const float a = tv * fr[i];
      e[i] = j < C ? (j < half ? sinf(a) : cosf(a)) : 0.f;
      dy[i] = j < C ? dtemb[(size_t)b * C + j] : 0.f;
      pre = fmaf(e[i], w1r[i], pre);
      dg = fmaf(dy[i], w2c[i], dg);
    }
    pre = wave_sum(pre) + bias;
    dg = wave_sum(dg);
    const float act = gelu_erf(pre);
    const float cdf = 0.5f * (1.0f + erff(pre * kInvSqrt2));
    const float pdf = kInvSqrt2Pi * expf(-0.5f * pre * pre);
    const float dpre = dg * (cdf + pre * pdf);
    gb1 += dpre;
#pragma unroll
    for (int i = 0; i < NVM; ++i) {
      g1[i] = fmaf(dpre, e[i], g1[i]);
      g2[i] = fmaf(dy[i], act, g2[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < NVM; ++i) {
    const int j = i * 64 + lane;
    if (j < C) {
      dw1[(size_t)k * C + j] += g1[i];
      dw2[(size_t)j * 2 * C + k] += g2[i];
    }
  }
  if (lane == 0) db1[k] += gb1;
  if (k * 64 < C) {                                      // db2[j] for j = k * 64 + lane
    const int j = k * 64 + lane;
    if (j < C) {
      float a = 0.f;
      for (int b = 0; b < B; ++b) a += dtemb[(size_t)b * C + j];
      db2[j] += a;
    }
  }
}

int row_blocks_g(int T) { return (T + 3) / 4 < kRowBlocksG ? (T + 3) / 4 : kRowBlocksG; }

}  // namespace

#define TRAIN_DISPATCH_G(C, ...)                                            \
  if ((C) < 1 || (C) > 1024) return -2;                                     \
  if ((C) <= 256) { constexpr int NVM = 4; __VA_ARGS__; }                   \
  else if ((C) <= 512) { constexpr int NVM = 8; __VA_ARGS__; }              \
  else { constexpr int NVM = 16; __VA_ARGS__; }

int d3dp_train_g_add_mask_ln(const float* x_in, const float* y, const float* mask, int axis, int F, int J, const float* w,
                             const float* b, float eps, float* x_out, float* xn, int T, int C, hipStream_t st) {
  if (!w || !b || !x_out) return -1;
  TRAIN_DISPATCH_G(C, hipLaunchKernelGGL((add_mask_ln_g_kernel<NVM>), dim3(row_blocks_g(T)), dim3(256), 0, st, x_in, y, mask, axis, F, J,
                                         w, b, eps, x_out, xn, T, C))
  return 0;
}
int d3dp_train_g_add_mask_ln2(const float* x_in, const float* y, const float* mask, int axis, int F, int J, const float* wa,
                              const float* ba, float eps_a, const float* pos, const float* wb, const float* bb, float eps_b,
                              float* x_out, float* x_next, float* xn, int T, int C, hipStream_t st) {
  if (!wa || !ba || !x_out || !x_next || (xn && !wb) || (wb && !bb)) return -1;
  TRAIN_DISPATCH_G(C, hipLaunchKernelGGL((add_mask_ln2_g_kernel<NVM>), dim3(row_blocks_g(T)), dim3(256), 0, st, x_in, y, mask, axis, F, J,
                                         wa, ba, eps_a, pos, wb, bb, eps_b, x_out, x_next, xn, T, C))
  return 0;
}
int d3dp_train_g_ln_pos(const float* x, const float* w, const float* b, float eps, const float* pos, int F, int J, float* y, int T,
                        int C, hipStream_t st) {
  TRAIN_DISPATCH_G(C, hipLaunchKernelGGL((ln_pos_g_kernel<NVM>), dim3(row_blocks_g(T)), dim3(256), 0, st, x, w, b, eps, pos, F, J, y, T, C))
  return 0;
}
int d3dp_train_g_ln_bwd(const float* dy, const float* xb, const float* wb, float eps_b, const float* dres, float* g_out,
                        const float* xa, const float* wa, float eps_a, float* dx, const float* mask, int axis, int F, int J,
                        float* dxm, float* part_b, float* part_a, int T, int C, int blocks, hipStream_t st) {
  if (xa) {
    TRAIN_DISPATCH_G(C, hipLaunchKernelGGL((ln_bwd2_g_kernel<NVM, true>), dim3(blocks), dim3(256), 0, st, dy, xb, wb, eps_b, dres, g_out, xa,
                                           wa, eps_a, dx, mask, axis, F, J, dxm, part_b, part_a, T, C))
  } else {
    TRAIN_DISPATCH_G(C, hipLaunchKernelGGL((ln_bwd2_g_kernel<NVM, false>), dim3(blocks), dim3(256), 0, st, dy, xb, wb, eps_b, dres, nullptr,
                                           nullptr, nullptr, 0.f, dx, mask, axis, F, J, dxm, part_b, nullptr, T, C))
  }
  return 0;
}
int d3dp_train_g_head_linear(const float* z, const float* w, const float* b, float* out, int T, int C, hipStream_t st) {
  hipLaunchKernelGGL(head_linear_g_kernel, dim3((T + 3) / 4), dim3(256), 0, st, z, w, b, out, T, C);
  return 0;
}
int d3dp_train_g_time_mlp_bwd(const int64_t* t, const float* freq, const float* w1, const float* b1, const float* w2,
                              const float* dtemb, float* dw1, float* db1, float* dw2, float* db2, int B, int C, hipStream_t st) {
  TRAIN_DISPATCH_G(C, hipLaunchKernelGGL((time_mlp_bwd_g_kernel<NVM>), dim3((2 * C + 3) / 4), dim3(256), 0, st, t, freq, w1, b1, w2, dtemb,
                                         dw1, db1, dw2, db2, B, C))
  return 0;
}
