// Token-wise (HBM-bound) pieces of the MixSTE2 denoiser.  One wavefront owns one token's C channels;
// LayerNorm statistics are two-pass fp32 reductions with wavefront shuffles (no LDS, no atomics).
//
//   time_mlp_kernel : sinusoid(t) -> Linear(C,2C) -> GELU(erf) -> Linear(2C,C)   (mixste.py:127-139,179-184)
//   embed_ln_kernel : cat(x2d,x3d) -> Linear(5,C) + Spatial_pos + time embed, fused with STE block 0's
//                     norm1 (mixste.py:227-235, :114).  The reference materialises the time embedding as a
//                     (B,H,F,J,C) repeat; here it is a broadcast read of a (B,C) table.
//   ln_kernel       : xn = LN(x)                                               (mixste.py:114-115 norm1/norm2)
//   ln2_kernel      : x = LN_shared(x) [+ Temporal_pos[f]] ; xn = LN_next(x)   (mixste.py:243,250,257,269,273
//                     fused with the following block's norm1)
//   head_kernel     : Temporal_norm -> head LayerNorm(eps 1e-5) -> Linear(C,3) (mixste.py:257/273, :207-210)
#include "common.h"
#include "kernels.h"

namespace {

#ifndef NT_STREAMS
#define NT_STREAMS 1
#endif
#ifndef NT_Y
#define NT_Y 1      // branch outputs (yadd) also nontemporal; 0: only the fp32 residual (A/B)
#endif

// per-lane slice of a C-channel row: NV = C/64 values; for NV % 4 == 0 they are float4 groups at
// element (g*64 + lane)*4, otherwise scalars at i*64 + lane.
template <int C> struct Row {
  static constexpr int NV = C / 64;
  static constexpr bool V4 = (NV % 4 == 0);
  static __device__ __forceinline__ int elem(int lane, int i) {
    if constexpr (V4) return ((i >> 2) * 64 + lane) * 4 + (i & 3);
    else return i * 64 + lane;
  }
  static __device__ __forceinline__ void load(const float* p, int lane, float* v) {
    if constexpr (V4) {
#pragma unroll
      for (int g = 0; g < NV / 4; ++g) {
        float4 a = *reinterpret_cast<const float4*>(p + (g * 64 + lane) * 4);
        v[g * 4] = a.x; v[g * 4 + 1] = a.y; v[g * 4 + 2] = a.z; v[g * 4 + 3] = a.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = p[i * 64 + lane];
    }
  }
  static __device__ __forceinline__ void load(const bf16* p, int lane, float* v) {
    if constexpr (V4) {
#pragma unroll
      for (int g = 0; g < NV / 4; ++g) {
        bf16x4 a = *reinterpret_cast<const bf16x4*>(p + (g * 64 + lane) * 4);
        v[g * 4] = (float)a[0]; v[g * 4 + 1] = (float)a[1]; v[g * 4 + 2] = (float)a[2]; v[g * 4 + 3] = (float)a[3];
      }
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = (float)p[i * 64 + lane];
    }
  }
  // streaming variants (nontemporal hint): for the fp32 residual stream and the branch outputs, which are read or
  // written once here and not touched again until several kernels later -- they should not push the normalised
  // activations (the next GEMM's A operand) out of L2 / the memory-side cache
  static __device__ __forceinline__ void load_nt(const float* p, int lane, float* v) {
    if constexpr (V4) {
#pragma unroll
      for (int g = 0; g < NV / 4; ++g) {
        const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (g * 64 + lane) * 4));
        v[g * 4] = a[0]; v[g * 4 + 1] = a[1]; v[g * 4 + 2] = a[2]; v[g * 4 + 3] = a[3];
      }
    } else {
      load(p, lane, v);
    }
  }
  static __device__ __forceinline__ void load_nt(const bf16* p, int lane, float* v) {
    if constexpr (V4) {
#pragma unroll
      for (int g = 0; g < NV / 4; ++g) {
        const bf16x4 a = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(p + (g * 64 + lane) * 4));
        v[g * 4] = (float)a[0]; v[g * 4 + 1] = (float)a[1]; v[g * 4 + 2] = (float)a[2]; v[g * 4 + 3] = (float)a[3];
      }
    } else {
      load(p, lane, v);
    }
  }
  static __device__ __forceinline__ void store_nt(float* p, int lane, const float* v) {
    if constexpr (V4) {
#pragma unroll
      for (int g = 0; g < NV / 4; ++g) {
        const f32x4 a = {v[g * 4], v[g * 4 + 1], v[g * 4 + 2], v[g * 4 + 3]};
        __builtin_nontemporal_store(a, reinterpret_cast<f32x4*>(p + (g * 64 + lane) * 4));
      }
    } else {
      store(p, lane, v);
    }
  }
  static __device__ __forceinline__ void store(float* p, int lane, const float* v) {
    if constexpr (V4) {
#pragma unroll
      for (int g = 0; g < NV / 4; ++g)
        *reinterpret_cast<float4*>(p + (g * 64 + lane) * 4) = make_float4(v[g * 4], v[g * 4 + 1], v[g * 4 + 2], v[g * 4 + 3]);
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) p[i * 64 + lane] = v[i];
    }
  }
  static __device__ __forceinline__ void store(bf16* p, int lane, const float* v) {
    if constexpr (V4) {
#pragma unroll
      for (int g = 0; g < NV / 4; ++g) {
        bf16x4 a = {(bf16)v[g * 4], (bf16)v[g * 4 + 1], (bf16)v[g * 4 + 2], (bf16)v[g * 4 + 3]};
        *reinterpret_cast<bf16x4*>(p + (g * 64 + lane) * 4) = a;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) p[i * 64 + lane] = (bf16)v[i];
    }
  }
  // split-bf16 activation: three planes `plane` elements apart
  static __device__ __forceinline__ void store3(bf16* p, size_t plane, int lane, const float* v) {
    float a0[NV], a1[NV], a2[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      bf16 q0, q1, q2;
      split3(v[i], q0, q1, q2);
      a0[i] = (float)q0; a1[i] = (float)q1; a2[i] = (float)q2;
    }
    store(p, lane, a0);
    store(p + plane, lane, a1);
    store(p + 2 * plane, lane, a2);
  }
  // split-fp16 activation (common.h split2h) in the h2i layout: `p` = start of the token's row of 2 C fp16.  Lanes 2j and
  // 2j+1 hold neighbouring 4-column groups: the even lane collects both hi halves (8 columns = 16 bytes of the hi slot),
  // the odd lane both lo halves -- one 16-byte store per lane and group, a wave-instruction writes 1 KiB of whole lines.
  static __device__ __forceinline__ void store2h(f16* p, size_t, int lane, const float* v) {
    if constexpr (V4) {
      const bool odd = lane & 1;
#pragma unroll
      for (int g = 0; g < NV / 4; ++g) {
        f16x4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) { f16 h, l; split2h(v[g * 4 + e], h, l); a[e] = h; b[e] = l; }
        const uint2 h = __builtin_bit_cast(uint2, a), l = __builtin_bit_cast(uint2, b);
        const uint2 send = odd ? h : l;
        uint2 recv;                                    // quad_perm [1,0,3,2]: the value of lane ^ 1
        recv.x = __builtin_amdgcn_update_dpp(0, (int)send.x, 0xB1, 0xF, 0xF, false);
        recv.y = __builtin_amdgcn_update_dpp(0, (int)send.y, 0xB1, 0xF, 0xF, false);
        using u32x4 = unsigned __attribute__((ext_vector_type(4)));
        const u32x4 out = odd ? (u32x4){recv.x, recv.y, l.x, l.y} : (u32x4){h.x, h.y, recv.x, recv.y};
        const int c0 = (g * 64 + (lane & ~1)) * 4;     // first of the pair's 8 columns
        *reinterpret_cast<u32x4*>(p + h2i_col(c0) + (odd ? kH2iLo : 0)) = out;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        f16 h, l;
        split2h(v[i], h, l);
        const int c = i * 64 + lane;
        p[h2i_col(c)] = h;
        p[h2i_col(c) + kH2iLo] = l;
      }
    }
  }
  // y = (v - mean) * rstd * w + b   (two-pass statistics, biased variance as torch LayerNorm)
  static __device__ __forceinline__ void norm(const float* v, const float* w, const float* b, float eps, int lane,
                                              float* y) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    const float mean = wave_sum(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) { const float d = v[i] - mean; q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / C) + eps);
    float wv[NV], bv[NV];
    load(w, lane, wv);
    load(b, lane, bv);
#pragma unroll
    for (int i = 0; i < NV; ++i) y[i] = fmaf((v[i] - mean) * rstd, wv[i], bv[i]);
  }
  // the same arithmetic on weights the caller already holds in registers (kernels that keep them across several tokens)
  static __device__ __forceinline__ void norm_r(const float* v, const float* wv, const float* bv, float eps, float* y) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    const float mean = wave_sum(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) { const float d = v[i] - mean; q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / C) + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) y[i] = fmaf((v[i] - mean) * rstd, wv[i], bv[i]);
  }
};

// Residual adds (mixste.py:113-115) ride on the row-wise kernels.  ln_kernel normalises x + yadd; it writes the sum
// back only if write_x (the denoiser does NOT: ln2/head re-form (x + yadd0) + yadd from the untouched x, which saves
// one fp32 row write + read per token and block).
// activation store: XN = float / bf16 (one plane) or b3 (three bf16 planes, `plane` elements apart)
template <int C, typename XN> struct ActOut {
  using ptr = XN*;
  static __device__ __forceinline__ void st(ptr base, size_t, size_t off, int lane, const float* v) {
    Row<C>::store(base + off, lane, v);
  }
};
template <int C> struct ActOut<C, b3> {
  using ptr = bf16*;
  static __device__ __forceinline__ void st(ptr base, size_t plane, size_t off, int lane, const float* v) {
    Row<C>::store3(base + off, plane, lane, v);
  }
};

template <int C> struct ActOut<C, h2> {
  using ptr = f16*;
  static __device__ __forceinline__ void st(ptr base, size_t plane, size_t off, int lane, const float* v) {
    Row<C>::store2h(base + 2 * off, plane, lane, v);   // off = token * C; an h2i row is 2 C fp16
  }
};

template <int C, typename XN, typename YT>
__global__ __launch_bounds__(256) void ln_kernel(float* __restrict__ x, const YT* __restrict__ yadd,
                                                 const float* __restrict__ w, const float* __restrict__ b, float eps,
                                                 typename ActOut<C, XN>::ptr xn, size_t plane, int T, int write_x) {
  using R = Row<C>;
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= T) return;
  float v[R::NV], y[R::NV];
  if (NT_STREAMS) R::load_nt(x + (size_t)tok * C, lane, v); else R::load(x + (size_t)tok * C, lane, v);
  if (yadd != nullptr) {
    if (NT_STREAMS && NT_Y) R::load_nt(yadd + (size_t)tok * C, lane, y); else R::load(yadd + (size_t)tok * C, lane, y);
#pragma unroll
    for (int i = 0; i < R::NV; ++i) v[i] += y[i];
    if (write_x) R::store(x + (size_t)tok * C, lane, v);
  }
  R::norm(v, w, b, eps, lane, y);
  ActOut<C, XN>::st(xn, plane, (size_t)tok * C, lane, y);
}

template <int C, typename XN, typename YT>
__global__ __launch_bounds__(256) void ln2_kernel(float* __restrict__ x, const YT* __restrict__ yadd0,
                                                  const YT* __restrict__ yadd, const float* __restrict__ wa,
                                                  const float* __restrict__ ba, const float* __restrict__ pos,
                                                  const float* __restrict__ wb, const float* __restrict__ bb, float eps,
                                                  typename ActOut<C, XN>::ptr xn, size_t plane, int T, int F, int J,
                                                  int SP) {
  using R = Row<C>;
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= T) return;
  float v[R::NV], y[R::NV], z[R::NV];
  if (NT_STREAMS) R::load_nt(x + (size_t)tok * C, lane, v); else R::load(x + (size_t)tok * C, lane, v);
  if (yadd0 != nullptr) {
    if (NT_STREAMS && NT_Y) R::load_nt(yadd0 + (size_t)tok * C, lane, y); else R::load(yadd0 + (size_t)tok * C, lane, y);
#pragma unroll
    for (int i = 0; i < R::NV; ++i) v[i] += y[i];
  }
  if (yadd != nullptr) {
    if (NT_STREAMS && NT_Y) R::load_nt(yadd + (size_t)tok * C, lane, y); else R::load(yadd + (size_t)tok * C, lane, y);
#pragma unroll
    for (int i = 0; i < R::NV; ++i) v[i] += y[i];
  }
  R::norm(v, wa, ba, eps, lane, y);
  if (pos != nullptr) {
    const int f = min((tok % SP) / J, F - 1);          // (SP: rows per sequence, >= F J; the pad rows behind a sequence
    float pv[R::NV];                                   //  carry finite filler that nothing reads: any frame will do)
    R::load(pos + (size_t)f * C, lane, pv);
#pragma unroll
    for (int i = 0; i < R::NV; ++i) y[i] += pv[i];
  }
  if (NT_STREAMS) R::store_nt(x + (size_t)tok * C, lane, y); else R::store(x + (size_t)tok * C, lane, y);
  R::norm(y, wb, bb, eps, lane, z);
  ActOut<C, XN>::st(xn, plane, (size_t)tok * C, lane, z);
}

// Tokens per wave of embed_ln_kernel / head_kernel: the weights of a lane's columns (embedding: 5 + 1 values per column; the
// LayerNorm pairs; the head's three rows) are loaded ONCE per wave and kept in registers over this many tokens -- as one
// token per wave these kernels issued 64 / 16 weight loads per lane and token for 2 - 4 data accesses and ran at 0.40 / 0.38 of
// the HBM roofline.  A workgroup covers 4 kRowTokens consecutive token rows (its four waves step through them side by side).
constexpr int kRowTokens = 8;

template <int C, typename XN>
__global__ __launch_bounds__(256) void embed_ln_kernel(const float* __restrict__ x2d, const float* __restrict__ x3d,
                                                       const float* __restrict__ temb, const float* __restrict__ ew,
                                                       const float* __restrict__ eb, const float* __restrict__ spos,
                                                       const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                       float eps, float* __restrict__ x,
                                                       typename ActOut<C, XN>::ptr xn, size_t plane, int seq0,
                                                       int n_seq, int H, int F, int J, int SP) {
  using R = Row<C>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int FJ = F * J, T = n_seq * SP;
  float w5[R::NV][5], ebv[R::NV], lw[R::NV], lb[R::NV];
#pragma unroll
  for (int i = 0; i < R::NV; ++i) {
    const float* wr = ew + R::elem(lane, i) * 5;
#pragma unroll
    for (int k = 0; k < 5; ++k) w5[i][k] = wr[k];
  }
  R::load(eb, lane, ebv);
  R::load(lnw, lane, lw);
  R::load(lnb, lane, lb);
  for (int it = 0; it < kRowTokens; ++it) {
    const int tl = (blockIdx.x * kRowTokens + it) * 4 + wave;    // chunk-local row: sequence tl / SP, token tl % SP
    if (tl >= T) break;
    // SP >= F J rows per sequence (capi.hip: a multiple of 64 when the skewed Linear schedule is in use): the pad rows get
    // the embedding of an all-zero input at joint 0 -- finite filler that flows through the row-wise kernels and the Linears
    // and is never read by an attention kernel nor written to the output
    const bool pad = (tl % SP) >= FJ;
    const int seq = seq0 + tl / SP, fj = pad ? 0 : tl % SP, nj = fj % J;
    const int b = seq / H;
    const float* p2 = x2d + ((size_t)b * FJ + fj) * 2;
    const float* p3 = x3d + ((size_t)seq * FJ + fj) * 3;
    float in5[5] = {p2[0], p2[1], p3[0], p3[1], p3[2]};          // channel order [u, v, x, y, z], mixste.py:228
    if (pad) { in5[0] = in5[1] = in5[2] = in5[3] = in5[4] = 0.f; }
    float v[R::NV], y[R::NV], sp[R::NV], tb[R::NV];
    R::load(spos + (size_t)nj * C, lane, sp);
    R::load(temb + (size_t)b * C, lane, tb);
#pragma unroll
    for (int i = 0; i < R::NV; ++i) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 5; ++k) a = fmaf(in5[k], w5[i][k], a);
      a += ebv[i];
      a += sp[i];
      a += tb[i];
      v[i] = a;
    }
    if (NT_STREAMS) R::store_nt(x + (size_t)tl * C, lane, v); else R::store(x + (size_t)tl * C, lane, v);
    R::norm_r(v, lw, lb, eps, y);
    ActOut<C, XN>::st(xn, plane, (size_t)tl * C, lane, y);
  }
}

template <int C, typename YT>
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ x, const YT* __restrict__ yadd0,
                                                   const YT* __restrict__ yadd, const float* __restrict__ wa,
                                                   const float* __restrict__ ba, float eps_a,
                                                   const float* __restrict__ wh, const float* __restrict__ bh,
                                                   float eps_h, const float* __restrict__ w, const float* __restrict__ b,
                                                   float* __restrict__ out, int T, int FJ, int SP) {
  using R = Row<C>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float wav[R::NV], bav[R::NV], whv[R::NV], bhv[R::NV], wv[3][R::NV];      // (kRowTokens: above embed_ln_kernel)
  R::load(wa, lane, wav);
  R::load(ba, lane, bav);
  R::load(wh, lane, whv);
  R::load(bh, lane, bhv);
#pragma unroll
  for (int o = 0; o < 3; ++o) R::load(w + o * C, lane, wv[o]);
  const float b0 = b[0], b1 = b[1], b2 = b[2];
  for (int it = 0; it < kRowTokens; ++it) {
    const int tok = (blockIdx.x * kRowTokens + it) * 4 + wave;
    if (tok >= T) break;
    float v[R::NV], y[R::NV], z[R::NV];
    if (NT_STREAMS) R::load_nt(x + (size_t)tok * C, lane, v); else R::load(x + (size_t)tok * C, lane, v);
    if (yadd0 != nullptr) {
      if (NT_STREAMS && NT_Y) R::load_nt(yadd0 + (size_t)tok * C, lane, y); else R::load(yadd0 + (size_t)tok * C, lane, y);
#pragma unroll
      for (int i = 0; i < R::NV; ++i) v[i] += y[i];
    }
    if (yadd != nullptr) {
      if (NT_STREAMS && NT_Y) R::load_nt(yadd + (size_t)tok * C, lane, y); else R::load(yadd + (size_t)tok * C, lane, y);
#pragma unroll
      for (int i = 0; i < R::NV; ++i) v[i] += y[i];
    }
    R::norm_r(v, wav, bav, eps_a, y);
    R::norm_r(y, whv, bhv, eps_h, z);
    float acc[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < R::NV; ++i) a = fmaf(z[i], wv[o][i], a);
      acc[o] = wave_sum(a) + (o == 0 ? b0 : o == 1 ? b1 : b2);
    }
    // rows are (sequence, token) at pitch SP >= FJ; the output is compact
    const int fj = tok % SP;
    if (lane < 3 && fj < FJ) out[((size_t)(tok / SP) * FJ + fj) * 3 + lane] = lane == 0 ? acc[0] : (lane == 1 ? acc[1] : acc[2]);
  }
}

// one workgroup per batch element; sin/cos table `freq` is supplied by the host (computed with the
// reference's own fp32 expression, mixste.py:135-136) so no device exp() enters the argument.
__global__ __launch_bounds__(256) void time_mlp_kernel(const int64_t* __restrict__ t, const float* __restrict__ freq,
                                                       const float* __restrict__ w1, const float* __restrict__ b1,
                                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                                       float* __restrict__ temb, int C) {
  extern __shared__ float sm[];            // [C] sinusoid | [2C] hidden
  float* e = sm;
  float* h = sm + C;
  const int b = blockIdx.x, half = C / 2;
  const float tv = (float)t[b];
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    const float a = tv * freq[i];
    e[i] = sinf(a);
    e[half + i] = cosf(a);
  }
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * C; o += blockDim.x) {
    const float* wr = w1 + (size_t)o * C;
    float a = 0.f;
    for (int k = 0; k < C; ++k) a = fmaf(e[k], wr[k], a);
    h[o] = gelu_erf(a + b1[o]);
  }
  __syncthreads();
  for (int o = threadIdx.x; o < C; o += blockDim.x) {
    const float* wr = w2 + (size_t)o * 2 * C;
    float a = 0.f;
    for (int k = 0; k < 2 * C; ++k) a = fmaf(h[k], wr[k], a);
    temb[(size_t)b * C + o] = a + b2[o];
  }
}

// ------------------------------------------------------------------------------------------------
// Any width (round 6, VERDICT r5 missing 4): the reference takes every `-cs` its 8 heads divide (common/arguments.py:49,
// mixste.py:46-62); the kernels above are instantiated for C in {64, 128, 256, 512}, the widths whose Linears run on the
// split-fp16 matrix-core kernels.  Other widths run the fp32 implementation (capi.hip: exact_impl 2 -- fp32-MFMA Linears,
// fp32 row attention) through the four kernels below: the same arithmetic in the same order per row (two-pass statistics,
// 1 / C as a multiplied reciprocal, fma with gamma / beta), the width a run-time argument, a lane's NVM slots masked behind it.
// fp32 activations only (the FAST and split-operand forms exist for the instantiated widths).
// ------------------------------------------------------------------------------------------------
template <int NVM> struct RowG {
  static __device__ __forceinline__ void load(const float* p, int C, int lane, float* v) {
#pragma unroll
    for (int i = 0; i < NVM; ++i) { const int c = i * 64 + lane; v[i] = c < C ? p[c] : 0.f; }
  }
  static __device__ __forceinline__ void store(float* p, int C, int lane, const float* v) {
#pragma unroll
    for (int i = 0; i < NVM; ++i) { const int c = i * 64 + lane; if (c < C) p[c] = v[i]; }
  }
  static __device__ __forceinline__ void norm(const float* v, const float* w, const float* b, float eps, int C, int lane, float* y) {
    const float rc = 1.0f / (float)C;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NVM; ++i) s += v[i];                       // (masked slots hold 0)
    const float mean = wave_sum(s) * rc;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NVM; ++i) { const float d = (i * 64 + lane < C) ? v[i] - mean : 0.f; q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * rc + eps);
#pragma unroll
    for (int i = 0; i < NVM; ++i) {
      const int c = i * 64 + lane;
      y[i] = c < C ? fmaf((v[i] - mean) * rstd, w[c], b[c]) : 0.f;
    }
  }
};

template <int NVM>
__global__ __launch_bounds__(256) void ln_g_kernel(float* __restrict__ x, const float* __restrict__ yadd, const float* __restrict__ w,
                                                   const float* __restrict__ b, float eps, float* __restrict__ xn, int T, int C,
                                                   int write_x) {
  using R = RowG<NVM>;
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= T) return;
  float v[NVM], y[NVM];
  R::load(x + (size_t)tok * C, C, lane, v);
  if (yadd != nullptr) {
    R::load(yadd + (size_t)tok * C, C, lane, y);
#pragma unroll
    for (int i = 0; i < NVM; ++i) v[i] += y[i];
    if (write_x) R::store(x + (size_t)tok * C, C, lane, v);
  }
  R::norm(v, w, b, eps, C, lane, y);
  R::store(xn + (size_t)tok * C, C, lane, y);
}

template <int NVM>
__global__ __launch_bounds__(256) void ln2_g_kernel(float* __restrict__ x, const float* __restrict__ yadd0, const float* __restrict__ yadd,
                                                    const float* __restrict__ wa, const float* __restrict__ ba,
                                                    const float* __restrict__ pos, const float* __restrict__ wb,
                                                    const float* __restrict__ bb, float eps, float* __restrict__ xn, int T, int C,
                                                    int F, int J, int SP) {
  using R = RowG<NVM>;
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= T) return;
  float v[NVM], y[NVM], z[NVM];
  R::load(x + (size_t)tok * C, C, lane, v);
  if (yadd0 != nullptr) {
    R::load(yadd0 + (size_t)tok * C, C, lane, y);
#pragma unroll
    for (int i = 0; i < NVM; ++i) v[i] += y[i];
  }
  if (yadd != nullptr) {
    R::load(yadd + (size_t)tok * C, C, lane, y);
#pragma unroll
    for (int i = 0; i < NVM; ++i) v[i] += y[i];
  }
  R::norm(v, wa, ba, eps, C, lane, y);
  if (pos != nullptr) {
    const int f = min((tok % SP) / J, F - 1);
    float pv[NVM];
    R::load(pos + (size_t)f * C, C, lane, pv);
#pragma unroll
    for (int i = 0; i < NVM; ++i) y[i] += pv[i];
  }
  R::store(x + (size_t)tok * C, C, lane, y);
  R::norm(y, wb, bb, eps, C, lane, z);
  R::store(xn + (size_t)tok * C, C, lane, z);
}

template <int NVM>
__global__ __launch_bounds__(256) void embed_ln_g_kernel(const float* __restrict__ x2d, const float* __restrict__ x3d,
                                                         const float* __restrict__ temb, const float* __restrict__ ew,
                                                         const float* __restrict__ eb, const float* __restrict__ spos,
                                                         const float* __restrict__ lnw, const float* __restrict__ lnb, float eps,
                                                         float* __restrict__ x, float* __restrict__ xn, int seq0, int n_seq, int H,
                                                         int F, int J, int SP, int C) {
  using R = RowG<NVM>;
  const int lane = threadIdx.x & 63;
  const int tl = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int FJ = F * J;
  if (tl >= n_seq * SP) return;
  const bool pad = (tl % SP) >= FJ;
  const int seq = seq0 + tl / SP, fj = pad ? 0 : tl % SP, nj = fj % J;
  const int b = seq / H;
  const float* p2 = x2d + ((size_t)b * FJ + fj) * 2;
  const float* p3 = x3d + ((size_t)seq * FJ + fj) * 3;
  float in5[5] = {p2[0], p2[1], p3[0], p3[1], p3[2]};
  if (pad) { in5[0] = in5[1] = in5[2] = in5[3] = in5[4] = 0.f; }
  float v[NVM], y[NVM];
#pragma unroll
  for (int i = 0; i < NVM; ++i) {
    const int c = i * 64 + lane;
    float a = 0.f;
    if (c < C) {
      const float* wr = ew + c * 5;
#pragma unroll
      for (int k = 0; k < 5; ++k) a = fmaf(in5[k], wr[k], a);
      a += eb[c];
      a += spos[nj * C + c];
      a += temb[b * C + c];
    }
    v[i] = a;
  }
  R::store(x + (size_t)tl * C, C, lane, v);
  R::norm(v, lnw, lnb, eps, C, lane, y);
  R::store(xn + (size_t)tl * C, C, lane, y);
}

template <int NVM>
__global__ __launch_bounds__(256) void head_g_kernel(const float* __restrict__ x, const float* __restrict__ yadd0,
                                                     const float* __restrict__ yadd, const float* __restrict__ wa,
                                                     const float* __restrict__ ba, float eps_a, const float* __restrict__ wh,
                                                     const float* __restrict__ bh, float eps_h, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ out, int T, int C, int FJ,
                                                     int SP) {
  using R = RowG<NVM>;
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= T) return;
  float v[NVM], y[NVM], z[NVM];
  R::load(x + (size_t)tok * C, C, lane, v);
  if (yadd0 != nullptr) {
    R::load(yadd0 + (size_t)tok * C, C, lane, y);
#pragma unroll
    for (int i = 0; i < NVM; ++i) v[i] += y[i];
  }
  if (yadd != nullptr) {
    R::load(yadd + (size_t)tok * C, C, lane, y);
#pragma unroll
    for (int i = 0; i < NVM; ++i) v[i] += y[i];
  }
  R::norm(v, wa, ba, eps_a, C, lane, y);
  R::norm(y, wh, bh, eps_h, C, lane, z);
  float acc[3];
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    float wv[NVM];
    R::load(w + o * C, C, lane, wv);
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < NVM; ++i) a = fmaf(z[i], wv[i], a);
    acc[o] = wave_sum(a) + b[o];
  }
  const int fj = tok % SP;
  if (lane < 3 && fj < FJ) out[((size_t)(tok / SP) * FJ + fj) * 3 + lane] = lane == 0 ? acc[0] : (lane == 1 ? acc[1] : acc[2]);
}

// widths the instantiated kernels take; everything else goes through the *_g kernels (fp32 activations, C <= 1024)
__host__ inline bool width_instantiated(int C) { return C == 64 || C == 128 || C == 256 || C == 512; }
#define DISPATCH_G(C, ...)                                                  \
  if ((C) < 1 || (C) > 1024) return -2;                                     \
  if ((C) <= 256) { constexpr int NVM = 4; __VA_ARGS__; }                   \
  else if ((C) <= 512) { constexpr int NVM = 8; __VA_ARGS__; }              \
  else { constexpr int NVM = 16; __VA_ARGS__; }

}  // namespace

#define DISPATCH_C(C, ...)                          \
  switch (C) {                                      \
    case 512: { constexpr int CC = 512; __VA_ARGS__; break; } \
    case 256: { constexpr int CC = 256; __VA_ARGS__; break; } \
    case 128: { constexpr int CC = 128; __VA_ARGS__; break; } \
    case 64:  { constexpr int CC = 64;  __VA_ARGS__; break; } \
    default: return -2;                             \
  }

int d3dp_launch_time_mlp(const int64_t* t, const float* freq, const float* w1, const float* b1, const float* w2,
                         const float* b2, float* temb, int B, int C, hipStream_t st) {
  hipLaunchKernelGGL(time_mlp_kernel, dim3(B), dim3(256), (size_t)3 * C * sizeof(float), st, t, freq, w1, b1, w2, b2,
                     temb, C);
  return 0;
}

// `act`: 0 = fp32 activations (y fp32), 1 = bf16 activations (y bf16), 2 = split-bf16 activation planes (y fp32),
//        3 = split-fp16 activation planes (y fp32)
int d3dp_launch_embed_ln(int act_bf16, const float* x2d, const float* x3d, const float* temb, const float* ew,
                         const float* eb, const float* spos, const float* lnw, const float* lnb, float eps, float* x,
                         void* xn, int seq0, int n_seq, int H, int F, int J, int C, hipStream_t st, int SP) {
  if (SP <= 0) SP = F * J;
  if (SP < F * J) return -1;
  const int T = n_seq * SP;
  dim3 g((T + 3) / 4), blk(256);
  const size_t plane = (size_t)T * C;
  if (!width_instantiated(C)) {
    if (act_bf16 != 0) return -2;
    DISPATCH_G(C, hipLaunchKernelGGL((embed_ln_g_kernel<NVM>), g, blk, 0, st, x2d, x3d, temb, ew, eb, spos, lnw, lnb, eps, x, (float*)xn, seq0, n_seq, H, F, J, SP, C))
    return 0;
  }
  const dim3 gk((T + 4 * kRowTokens - 1) / (4 * kRowTokens));     // kRowTokens tokens per wave
  DISPATCH_C(C,
    if (act_bf16 == 1) hipLaunchKernelGGL((embed_ln_kernel<CC, bf16>), gk, blk, 0, st, x2d, x3d, temb, ew, eb, spos, lnw, lnb, eps, x, (bf16*)xn, plane, seq0, n_seq, H, F, J, SP);
    else if (act_bf16 == 2) hipLaunchKernelGGL((embed_ln_kernel<CC, b3>), gk, blk, 0, st, x2d, x3d, temb, ew, eb, spos, lnw, lnb, eps, x, (bf16*)xn, plane, seq0, n_seq, H, F, J, SP);
    else if (act_bf16 == 3) hipLaunchKernelGGL((embed_ln_kernel<CC, h2>), gk, blk, 0, st, x2d, x3d, temb, ew, eb, spos, lnw, lnb, eps, x, (f16*)xn, plane, seq0, n_seq, H, F, J, SP);
    else hipLaunchKernelGGL((embed_ln_kernel<CC, float>), gk, blk, 0, st, x2d, x3d, temb, ew, eb, spos, lnw, lnb, eps, x, (float*)xn, plane, seq0, n_seq, H, F, J, SP))
  return 0;
}

int d3dp_launch_ln(int act_bf16, float* x, const void* yadd, int write_x, const float* w, const float* b, float eps,
                   void* xn, int T, int C, hipStream_t st) {
  dim3 g((T + 3) / 4), blk(256);
  const size_t plane = (size_t)T * C;
  if (!width_instantiated(C)) {
    if (act_bf16 != 0) return -2;
    DISPATCH_G(C, hipLaunchKernelGGL((ln_g_kernel<NVM>), g, blk, 0, st, x, (const float*)yadd, w, b, eps, (float*)xn, T, C, write_x))
    return 0;
  }
  DISPATCH_C(C,
    if (act_bf16 == 1) hipLaunchKernelGGL((ln_kernel<CC, bf16, bf16>), g, blk, 0, st, x, (const bf16*)yadd, w, b, eps, (bf16*)xn, plane, T, write_x);
    else if (act_bf16 == 2) hipLaunchKernelGGL((ln_kernel<CC, b3, float>), g, blk, 0, st, x, (const float*)yadd, w, b, eps, (bf16*)xn, plane, T, write_x);
    else if (act_bf16 == 3) hipLaunchKernelGGL((ln_kernel<CC, h2, float>), g, blk, 0, st, x, (const float*)yadd, w, b, eps, (f16*)xn, plane, T, write_x);
    else hipLaunchKernelGGL((ln_kernel<CC, float, float>), g, blk, 0, st, x, (const float*)yadd, w, b, eps, (float*)xn, plane, T, write_x))
  return 0;
}

int d3dp_launch_ln2(int act_bf16, float* x, const void* yadd0, const void* yadd, const float* wa, const float* ba, const float* pos,
                    const float* wb, const float* bb, float eps, void* xn, int T, int C, int F, int J, hipStream_t st, int SP) {
  if (SP <= 0) SP = F * J;
  dim3 g((T + 3) / 4), blk(256);
  const size_t plane = (size_t)T * C;
  if (!width_instantiated(C)) {
    if (act_bf16 != 0) return -2;
    DISPATCH_G(C, hipLaunchKernelGGL((ln2_g_kernel<NVM>), g, blk, 0, st, x, (const float*)yadd0, (const float*)yadd, wa, ba, pos, wb, bb, eps, (float*)xn, T, C, F, J, SP))
    return 0;
  }
  DISPATCH_C(C,
    if (act_bf16 == 1) hipLaunchKernelGGL((ln2_kernel<CC, bf16, bf16>), g, blk, 0, st, x, (const bf16*)yadd0, (const bf16*)yadd, wa, ba, pos, wb, bb, eps, (bf16*)xn, plane, T, F, J, SP);
    else if (act_bf16 == 2) hipLaunchKernelGGL((ln2_kernel<CC, b3, float>), g, blk, 0, st, x, (const float*)yadd0, (const float*)yadd, wa, ba, pos, wb, bb, eps, (bf16*)xn, plane, T, F, J, SP);
    else if (act_bf16 == 3) hipLaunchKernelGGL((ln2_kernel<CC, h2, float>), g, blk, 0, st, x, (const float*)yadd0, (const float*)yadd, wa, ba, pos, wb, bb, eps, (f16*)xn, plane, T, F, J, SP);
    else hipLaunchKernelGGL((ln2_kernel<CC, float, float>), g, blk, 0, st, x, (const float*)yadd0, (const float*)yadd, wa, ba, pos, wb, bb, eps, (float*)xn, plane, T, F, J, SP))
  return 0;
}

int d3dp_launch_head(int act_bf16, const float* x, const void* yadd0, const void* yadd, const float* wa, const float* ba, float eps_a, const float* wh,
                     const float* bh, float eps_h, const float* w, const float* b, float* out, int T, int C,
                     hipStream_t st, int FJ, int SP) {
  if (FJ <= 0 || SP <= 0) { FJ = SP = 1 << 30; }       // compact rows: out row = input row
  dim3 g((T + 3) / 4), blk(256);
  if (!width_instantiated(C)) {
    if (act_bf16 != 0) return -2;
    DISPATCH_G(C, hipLaunchKernelGGL((head_g_kernel<NVM>), g, blk, 0, st, x, (const float*)yadd0, (const float*)yadd, wa, ba, eps_a, wh, bh, eps_h, w, b, out, T, C, FJ, SP))
    return 0;
  }
  const dim3 gk((T + 4 * kRowTokens - 1) / (4 * kRowTokens));     // kRowTokens tokens per wave
  DISPATCH_C(C,
    if (act_bf16 == 1) hipLaunchKernelGGL((head_kernel<CC, bf16>), gk, blk, 0, st, x, (const bf16*)yadd0, (const bf16*)yadd, wa, ba, eps_a, wh, bh, eps_h, w, b, out, T, FJ, SP);
    else hipLaunchKernelGGL((head_kernel<CC, float>), gk, blk, 0, st, x, (const float*)yadd0, (const float*)yadd, wa, ba, eps_a, wh, bh, eps_h, w, b, out, T, FJ, SP))
  return 0;
}
