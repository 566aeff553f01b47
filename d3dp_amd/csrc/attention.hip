// Multi-head attention of the MixSTE2 blocks: softmax(q k^T * hd^-0.5) v per (sequence, head)
// (reference common/mixste.py:63-79; spatial sequences = 17 joints of one frame, temporal sequences =
// F frames of one joint).  Sequences are addressed through SeqMap strides in the ONE physical token
// layout (bh, f, n, c): the reference's physical transposes (mixste.py:244, 270, 274) never happen.
//
//   attn_rows_kernel          : fp32 VALU, one thread per query row, K/V of the (sequence, head) problem
//                               broadcast from LDS.  Used for EXACT mode (fp32 activations, both axes)
//                               and for the spatial axis in FAST mode (bf16 activations; 0.4 % of FLOPs
//                               and HBM-bound, so matrix cores would buy nothing).
//   attn_temporal2_bf16_kernel / attn_spatial_bf16_kernel : FAST mode on v_mfma_f32_16x16x32_bf16.  K and V rows
//                               (swizzled, row-major) live in LDS; each wave owns 16-query tiles and keeps a whole
//                               score row-block in registers (keys <= 256), so softmax is a plain two-pass fp32 softmax
//                               with wavefront shuffles -- no online rescaling.  S^T = K Q^T is computed transposed so
//                               the probabilities land directly in the A/B fragment layout of the P.V MFMA.
//   attn_temporal_f32_kernel  : EXACT mode temporal axis on the fp32 matrix cores.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "ta_common.h"

#ifndef D3DP_ATTN_PAIR
#define D3DP_ATTN_PAIR 1     // temporal split-fp16 kernel: both query tiles of a wave share one pass over the K image
#endif
// temporal split-fp16 kernel, two tiles per wave: the softmax of one tile between the matrix instructions of the other
// (tile 0's under tile 1's scores, tile 1's under tile 0's P.V); takes precedence over D3DP_ATTN_PAIR.  Same arithmetic in
// the same order per element: bit-identical results (the attention tests pass with it).  MEASURED (profiles/r04_attn_timeline.md):
// the score phase loses 2 k cycles, the P.V phase gains them, and a problem takes 30.55 k cycles in either form -- the time
// per problem is set by something both forms share, not by where the vector instructions sit.  Off.
#ifndef D3DP_ATTN_OVERLAP
#define D3DP_ATTN_OVERLAP 0
#endif
#ifndef D3DP_ATTN_OVL_SGB
#define D3DP_ATTN_OVL_SGB 0  // > 0: ask the scheduler for this many VALU instructions behind every MFMA of an overlapped region
#endif

namespace {

__device__ __forceinline__ int seq_base(const SeqMap& m, int s) {
  return (s / m.inner) * m.outer_stride + (s % m.inner) * m.inner_stride;
}

// ------------------------------------------------------------------------------------------------
// generic fp32-VALU attention: thread per query row
// ------------------------------------------------------------------------------------------------
template <typename T> struct Vec16;   // 16-byte vector of T
template <> struct Vec16<float> { static constexpr int N = 4; };
template <> struct Vec16<bf16> { static constexpr int N = 8; };

template <typename T, int N>
__device__ __forceinline__ void ld_vec(const T* p, float* v) {
  if constexpr (N == 4) {
    float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  } else {
    load8(reinterpret_cast<const bf16*>(p), v);
  }
}

// OUT3: write the result as three split-bf16 planes (`plane` elements apart) instead of T -- the A operand
// format of the bf16x3 EXACT-mode Linear.
// Any sequence length: a thread owns the query rows row, row + TPP, ... of its problem, and K / V pass through LDS in
// chunks of `kchunk` keys under the online softmax (one chunk = the whole sequence up to 256 tokens, which is every BASELINE
// configuration; longer clips -- `-f 351`, reference common/arguments.py:58 -- run here instead of being refused).
// amax (optional): absmax of the fp32 output, one atomicMax per workgroup (the training step's proj operand scale).
// GEN (round 6: any head dim the reference's `-cs` / 8 heads gives, common/arguments.py:49): HD is the register capacity, the
// head dim itself the run-time `hd_rt` (a multiple of the 16-byte vector, <= HD); the 16-byte slots behind it hold zeros in
// q / K / V and are not stored.  Instantiated for fp32 rows only (the fp32 implementation of the widths outside {64 .. 512}).
template <typename T, int HD, int TPP, int OUTS, bool GEN = false>   // OUTS: 0 = T out, 3 = three split-bf16 planes, 2 = two split-fp16 planes
__global__ __launch_bounds__(256) void attn_rows_kernel(const T* __restrict__ qkv, void* __restrict__ out_v, int n_prob,
                                                        SeqMap map, int C, int heads, size_t plane, int kchunk,
                                                        unsigned* __restrict__ amax, int hd_rt) {
  static_assert(!GEN || (OUTS == 0 && sizeof(T) == 4), "the run-time head dim form exists for fp32 rows");
  const int hd = GEN ? hd_rt : HD;
  constexpr int PPB = 256 / TPP;
  constexpr int VN = Vec16<T>::N;                 // elements per 16-byte vector
  constexpr int LDR = HD + VN;                    // padded LDS row (elements)
  constexpr int CH = (HD >= VN) ? HD / VN : 1;    // 16-byte chunks per row
  static_assert(HD % VN == 0, "head dim must be a multiple of the 16-byte vector");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);

  const int n = map.n_tok;
  const int tid = threadIdx.x;
  const int lp = tid / TPP, row = tid % TPP;
  const int pid = blockIdx.x * PPB + lp;
  const bool live = pid < n_prob;
  const int seq = live ? pid / heads : 0, head = live ? pid % heads : 0;
  const int base = seq_base(map, seq);
  T* Ks = smem + (size_t)lp * 2 * kchunk * LDR;
  T* Vs = Ks + (size_t)kchunk * LDR;
  const float scale = 1.0f / sqrtf((float)hd);
  float am = 0.f;

  for (int q0 = 0; q0 < n; q0 += TPP) {
    const bool act = live && q0 + row < n;
    const size_t tok = (size_t)(base + (act ? q0 + row : 0) * map.tok_stride);
    float q[HD], o[HD];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (!GEN || c * VN < hd) ld_vec<T, VN>(qkv + tok * 3 * C + head * hd + c * VN, q + c * VN);
      else {
#pragma unroll
        for (int e = 0; e < VN; ++e) q[c * VN + e] = 0.f;
      }
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < n; k0 += kchunk) {
      const int nk = min(kchunk, n - k0);
      if (k0 > 0 || q0 > 0) __syncthreads();          // every thread is done with the chunk the images still hold
      if (live) {
        for (int u = row; u < nk * CH; u += TPP) {
          const int j = u / CH, c = u % CH;
          const T* src = qkv + (size_t)(base + (k0 + j) * map.tok_stride) * 3 * C + C + head * hd + c * VN;
          const bool in = !GEN || c * VN < hd;
          *reinterpret_cast<float4*>(Ks + j * LDR + c * VN) = in ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4*>(Vs + j * LDR + c * VN) = in ? *reinterpret_cast<const float4*>(src + C) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      __syncthreads();
      if (!act) continue;
      // online softmax over groups of KB keys: KB independent score accumulators per pass (the dot products are latency
      // chains; with one wave per SIMD in the 243-key configuration nothing else hides them)
      constexpr int KB = 4;
      for (int j0 = 0; j0 < nk; j0 += KB) {
        float sc[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u) sc[u] = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
#pragma unroll
          for (int u = 0; u < KB; ++u) {
            const int j = min(j0 + u, nk - 1);
            float kv[VN];
            ld_vec<T, VN>(Ks + j * LDR + c * VN, kv);
#pragma unroll
            for (int e = 0; e < VN; ++e) sc[u] = fmaf(q[c * VN + e], kv[e], sc[u]);
          }
        }
        float gm = m;
#pragma unroll
        for (int u = 0; u < KB; ++u) {
          sc[u] = (j0 + u < nk) ? sc[u] * scale : -INFINITY;
          gm = fmaxf(gm, sc[u]);
        }
        if (gm > m) {
          const float f = expf(m - gm);
          l *= f;
#pragma unroll
          for (int d = 0; d < HD; ++d) o[d] *= f;
          m = gm;
        }
        float p[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u) { p[u] = expf(sc[u] - m); l += p[u]; }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
#pragma unroll
          for (int u = 0; u < KB; ++u) {
            const int j = min(j0 + u, nk - 1);
            float vv[VN];
            ld_vec<T, VN>(Vs + j * LDR + c * VN, vv);
#pragma unroll
            for (int e = 0; e < VN; ++e) o[c * VN + e] = fmaf(p[u], vv[e], o[c * VN + e]);
          }
        }
      }
    }
    if (!act) continue;
    const float inv = 1.0f / l;
    if constexpr (OUTS == 2) {
      f16* dst = reinterpret_cast<f16*>(out_v) + (size_t)tok * (2 * C);   // h2i row (common.h)
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) {
        f16x4 p0, p1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f16 a0, a1;
          split2h(o[c * 4 + e] * inv, a0, a1);
          p0[e] = a0; p1[e] = a1;
        }
        const int hc = h2i_col(head * HD + c * 4);
        *reinterpret_cast<f16x4*>(dst + hc) = p0;
        *reinterpret_cast<f16x4*>(dst + hc + kH2iLo) = p1;
      }
    } else if constexpr (OUTS == 3) {
      bf16* dst = reinterpret_cast<bf16*>(out_v) + tok * C + head * HD;
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) {
        bf16x4 p0, p1, p2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bf16 a0, a1, a2;
          split3(o[c * 4 + e] * inv, a0, a1, a2);
          p0[e] = a0; p1[e] = a1; p2[e] = a2;
        }
        *reinterpret_cast<bf16x4*>(dst + c * 4) = p0;
        *reinterpret_cast<bf16x4*>(dst + plane + c * 4) = p1;
        *reinterpret_cast<bf16x4*>(dst + 2 * plane + c * 4) = p2;
      }
    } else {
      T* dst = reinterpret_cast<T*>(out_v) + tok * C + head * hd;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (GEN && c * VN >= hd) continue;
        float r[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) { r[e] = o[c * VN + e] * inv; am = fmaxf(am, fabsf(r[e])); }
        if constexpr (VN == 4) *reinterpret_cast<float4*>(dst + c * 4) = make_float4(r[0], r[1], r[2], r[3]);
        else store8(reinterpret_cast<bf16*>(dst) + c * 8, r);
      }
    }
  }
  if (amax) {
    // (in the DYNAMIC allocation, behind the images: a static array would push a 160 KiB opt-in over the CU's LDS)
    float* part_amax = reinterpret_cast<float*>(smem + (size_t)PPB * 2 * kchunk * LDR);
    am = wave_max(am);
    if ((threadIdx.x & 63) == 0) part_amax[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0)
      atomicMax(amax, __float_as_uint(fmaxf(fmaxf(part_amax[0], part_amax[1]), fmaxf(part_amax[2], part_amax[3]))));
  }
}

template <typename T, int HD, int TPP, int OUTS, bool GEN = false>
int launch_rows(const void* qkv, void* out, int n_seq, SeqMap map, int C, int heads, size_t plane, unsigned* amax, hipStream_t st) {
  constexpr int PPB = 256 / TPP;
  constexpr int LDR = HD + Vec16<T>::N;
  const int n_prob = n_seq * heads;
  int kchunk = map.n_tok < 256 ? map.n_tok : 256;
  if (GEN) {                                           // a 128-wide head: the K / V images of 256 keys do not fit the CU's LDS
    const int fit = (int)((160 * 1024 - 16) / ((size_t)PPB * 2 * LDR * sizeof(T)));
    if (fit < 1) return -2;
    if (kchunk > fit) kchunk = fit >= 128 ? 128 : fit;
  }
  const size_t lds = (size_t)PPB * 2 * kchunk * LDR * sizeof(T) + 16;   // (+ the four absmax partials)
  if (lds > 160 * 1024) return -2;
  auto kern = attn_rows_kernel<T, HD, TPP, OUTS, GEN>;
  static PerDeviceOnce once;                          // (one per template instantiation = per kernel)
  if (once.get([&](int) { return d3dp_lds_opt_in(reinterpret_cast<const void*>(kern), 160 * 1024); }) < 0) return -3;
  hipLaunchKernelGGL(kern, dim3((n_prob + PPB - 1) / PPB), dim3(256), lds, st, (const T*)qkv, out, n_prob, map, C, heads,
                     plane, kchunk, amax, C / heads);
  return 0;
}

template <typename T, int OUTS>
int dispatch_rows(const void* qkv, void* out, int n_seq, SeqMap map, int C, int heads, size_t plane, unsigned* amax, hipStream_t st) {
  const int hd = C / heads;
  const bool small = map.n_tok <= 32;
  if (map.n_tok < 1) return -2;
#define ROWS_CASE(HD_)                                                                                          \
  case HD_: return small ? launch_rows<T, HD_, 32, OUTS>(qkv, out, n_seq, map, C, heads, plane, amax, st)       \
                         : launch_rows<T, HD_, 256, OUTS>(qkv, out, n_seq, map, C, heads, plane, amax, st);
  switch (hd) {
    ROWS_CASE(64) ROWS_CASE(32) ROWS_CASE(16) ROWS_CASE(8)
    default: break;
  }
#undef ROWS_CASE
  // any other head dim (a multiple of 4 up to 128): the run-time form, fp32 rows in and out
  if constexpr (std::is_same<T, float>::value && OUTS == 0) {
    if (hd < 4 || hd % 4 || hd > 128 || hd * heads != C) return -2;
#define ROWS_GEN(HD_)                                                                                                       \
  {                                                                                                                         \
    if (small) {                                                                                                            \
      const int r = launch_rows<T, HD_, 32, OUTS, true>(qkv, out, n_seq, map, C, heads, plane, amax, st);                   \
      if (r != -2) return r;                          /* (-2: eight problems' images do not fit the LDS) */                 \
    }                                                                                                                       \
    return launch_rows<T, HD_, 256, OUTS, true>(qkv, out, n_seq, map, C, heads, plane, amax, st);                           \
  }
    if (hd <= 16) ROWS_GEN(16)
    if (hd <= 32) ROWS_GEN(32)
    if (hd <= 64) ROWS_GEN(64)
    ROWS_GEN(128)
#undef ROWS_GEN
  }
  return -2;
}

// ------------------------------------------------------------------------------------------------
// FAST-mode attention on bf16 MFMA (head dim 64): no transposes anywhere.
//   K rows sit in LDS row-major (128 B per key) with the 16-byte-slot XOR swizzle ds_read_b128 wants;
//   V rows sit row-major too, swizzled at 32-byte-chunk granularity, and are consumed with
//   ds_read_b64_tr_b16: within each 16-lane group the instruction returns, to lane i, column i of the
//   4-key x 16-column block the group's lanes point at (lanes 4j..4j+3 -> key j) -- exactly the
//   "8 consecutive keys of one output channel" fragment the O^T = V^T P^T MFMA needs.  (Mapping measured on
//   gfx950 with tools/probe_tr.hip.)
// ------------------------------------------------------------------------------------------------
typedef __bf16 v4bf16 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int vchunk_swz(int key, int chunk) { return chunk ^ ((key >> 1) & 3); }

// Per-lane LDS base addresses of the K and V^T fragments.  Both swizzles depend only on the lane (not on the key
// tile), so every fragment read in the tile loop is  base register + compile-time immediate.
struct FragBases {
  const char* k0;      // K row (lane&15), d-slot  (lane>>4)      ; tile t at +t*2048
  const char* k1;      // K row (lane&15), d-slot 4+(lane>>4)
  const char* v[4];    // V row 4g+j, 32-B chunk dn (swizzled), + qd*8 ; key chunk c at +c*4096, second half +2048
};
__device__ __forceinline__ FragBases make_frag_bases(const char* KS, const char* VS, int lane) {
  FragBases fb;
  const int fi = lane & 15, fg = lane >> 4;
  const int sw = (fi >> 1) & 7;                       // ((16t + fi) >> 1) & 7 is independent of t
  fb.k0 = KS + fi * 128 + ((fg ^ sw) << 4);
  fb.k1 = KS + fi * 128 + (((4 + fg) ^ sw) << 4);
  const int j = fi >> 2, qd = fi & 3, key = 4 * fg + j;
  const int vs = (key >> 1) & 3;                      // ((32c [+16] + key) >> 1) & 3 is independent of c
#pragma unroll
  for (int dn = 0; dn < 4; ++dn) fb.v[dn] = VS + key * 128 + ((dn ^ vs) << 5) + qd * 8;
  return fb;
}

// V^T fragment (MFMA A operand) for output channels dn*16 + (lane&15), keys {32c + 4g + j} and {32c + 16 + 4g + j}
template <int C0>
__device__ __forceinline__ bf16x8 load_vt_frag(const char* vb) {
  const v4bf16 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) v4bf16*)(vb + C0 * 4096));
  const v4bf16 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) v4bf16*)(vb + C0 * 4096 + 2048));
  typedef __bf16 raw8 __attribute__((ext_vector_type(8)));     // (16-bit payloads: whatever `bf16` is in this build, common.h D3DP_FAST_F16)
  return __builtin_bit_cast(bf16x8, (raw8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
}

template <int NKT, int C0>
__device__ __forceinline__ void pv_chunks(const FragBases& fb, const bf16x8 (&pf)[NKT / 2], f32x4 (&o)[4]) {
  if constexpr (C0 < NKT / 2) {
#pragma unroll
    for (int dn = 0; dn < 4; ++dn)
      o[dn] = D3DP_MFMA_16x16x32_BF16(load_vt_frag<C0>(fb.v[dn]), pf[C0], o[dn]);
    if (C0 & 1) __builtin_amdgcn_sched_barrier(0);
    pv_chunks<NKT, C0 + 1>(fb, pf, o);
  }
}

// One 16-query tile against NKT 16-key tiles resident in LDS.  q0/q1: the tile's Q fragments (d 0..31 / 32..63).
// Returns O^T accumulators (4 channel tiles) and the softmax denominator of query (lane & 15).
template <int NKT>
__device__ __forceinline__ void attn_tile(const FragBases& fb, bf16x8 q0, bf16x8 q1, int n, int lane, f32x4 (&o)[4],
                                          float& denom) {
  const int fg = lane >> 4;
  const float cexp = 0.125f * 1.44269504088896340736f;   // hd^-0.5 * log2(e), hd = 64
  f32x4 s[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(fb.k0 + t * 2048);
    const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(fb.k1 + t * 2048);
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    a = D3DP_MFMA_16x16x32_BF16(k0, q0, a);
    a = D3DP_MFMA_16x16x32_BF16(k1, q1, a);
    s[t] = a;
    if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // bound the ds_read hoisting window (VGPR pressure)
  }
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    if (16 * (t + 1) > n) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * t + 4 * fg + r >= n) s[t][r] = -INFINITY;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[t][r]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float mc = mx * cexp;
  float sum = 0.f;
  bf16x8 pf[NKT / 2];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = __builtin_amdgcn_exp2f(fmaf(s[t][r], cexp, -mc));
      sum += p;
      pf[t >> 1][(t & 1) * 4 + r] = (bf16)p;
    }
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  denom = sum;
#pragma unroll
  for (int dn = 0; dn < 4; ++dn) o[dn] = (f32x4){0.f, 0.f, 0.f, 0.f};
  pv_chunks<NKT, 0>(fb, pf, o);
}

// rows [0, n) of K and V (128 B per row for this head) -> swizzled LDS images; rows [n, NK) of V zeroed.
template <int NK, int NTHREADS>
__device__ __forceinline__ void stage_kv(const bf16* __restrict__ kbase, size_t row_stride, int n, char* KS, char* VS,
                                         int tid, int C) {
  for (int idx = tid; idx < NK * 8; idx += NTHREADS) {
    const int row = idx >> 3, slot = idx & 7;
    float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
    if (row < n) {
      const bf16* src = kbase + (size_t)row * row_stride + slot * 8;
      kv = *reinterpret_cast<const float4*>(src);
      vv = *reinterpret_cast<const float4*>(src + C);
    }
    *reinterpret_cast<float4*>(KS + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4)) = kv;
    *reinterpret_cast<float4*>(VS + row * 128 + ((slot ^ (((row >> 1) & 3) << 1)) << 4)) = vv;
  }
}

template <int NKT>   // temporal axis: one workgroup per (sequence, head), 8 waves share the K/V images
__global__ __launch_bounds__(512, 4) void attn_temporal2_bf16_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                                     SeqMap map, int C, int heads) {
  constexpr int NK = 16 * NKT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* KS = smem;
  char* VS = smem + NK * 128;
  const int n = map.n_tok;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int seq = blockIdx.x / heads, head = blockIdx.x % heads;
  const int base = seq_base(map, seq);
  const int ts = map.tok_stride;
  const size_t ld = (size_t)3 * C;
  const bf16* qbase = qkv + (size_t)base * ld + (size_t)head * 64;
  // This wave's Q fragments for ALL of its query tiles are requested before K/V staging, so their HBM latency
  // overlaps the staging loads instead of being paid once per tile in the compute loop.
  const int fi = lane & 15, fg = lane >> 4;
  const int n_qt = (n + 15) >> 4;
  constexpr int QPW = (NKT + 7) / 8;                 // query tiles per wave
  bf16x8 qf[QPW][2];
#pragma unroll
  for (int i = 0; i < QPW; ++i) {
    const int q = min((wave + 8 * i) * 16 + fi, n - 1);
    const bf16* qsrc = qbase + (size_t)q * ts * ld + fg * 8;
    qf[i][0] = *reinterpret_cast<const bf16x8*>(qsrc);
    qf[i][1] = *reinterpret_cast<const bf16x8*>(qsrc + 32);
  }
  stage_kv<NK, 512>(qbase + C, (size_t)ts * ld, n, KS, VS, tid, C);
  const FragBases fb = make_frag_bases(KS, VS, lane);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < QPW; ++i) {
    const int qt = wave + 8 * i;
    if (qt >= n_qt) break;
    const int q = qt * 16 + fi;
    f32x4 o[4];
    float denom;
    attn_tile<NKT>(fb, qf[i][0], qf[i][1], n, lane, o, denom);
    if (q < n) {
      const float inv = 1.0f / denom;
      bf16* dst = out + (size_t)(base + q * ts) * C + head * 64 + fg * 4;
#pragma unroll
      for (int dn = 0; dn < 4; ++dn) {
        bf16x4 r = {(bf16)(o[dn][0] * inv), (bf16)(o[dn][1] * inv), (bf16)(o[dn][2] * inv), (bf16)(o[dn][3] * inv)};
        *reinterpret_cast<bf16x4*>(dst + dn * 16) = r;
      }
    }
  }
}

// spatial axis (<= 32 tokens per sequence): one WAVE per (sequence, head) with a private 8 KiB K/V image;
// a 256-thread workgroup covers 4 heads of one sequence.
__global__ __launch_bounds__(256) void attn_spatial_bf16_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                                int n_prob, SeqMap map, int C, int heads) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 8192];
  const int n = map.n_tok;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pid = blockIdx.x * 4 + wave;
  if (pid >= n_prob) return;
  const int seq = pid / heads, head = pid % heads;
  const int base = seq_base(map, seq);
  const int ts = map.tok_stride;
  const size_t ld = (size_t)3 * C;
  const bf16* qbase = qkv + (size_t)base * ld + (size_t)head * 64;
  char* KS = smem + wave * 8192;
  char* VS = KS + 4096;
  // the query fragments of both 16-row tiles are requested BEFORE the K/V staging so that their HBM latency overlaps
  // it (the kernel is latency/HBM-bound: one small problem per wave)
  const int fi = lane & 15, fg = lane >> 4;
  const int n_qt = (n + 15) >> 4;
  bf16x8 qf[2][2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const bf16* qsrc = qbase + (size_t)min(qt * 16 + fi, n - 1) * ts * ld + fg * 8;
    qf[qt][0] = *reinterpret_cast<const bf16x8*>(qsrc);
    qf[qt][1] = *reinterpret_cast<const bf16x8*>(qsrc + 32);
  }
  stage_kv<32, 64>(qbase + C, (size_t)ts * ld, n, KS, VS, lane, C);
  const FragBases fb = make_frag_bases(KS, VS, lane);
  // (wave-private LDS image: the LDS pipe executes one wave's accesses in order, no barrier needed)
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    if (qt >= n_qt) break;
    const int q = qt * 16 + fi;
    const bf16x8 q0 = qf[qt][0], q1 = qf[qt][1];
    f32x4 o[4];
    float denom;
    attn_tile<2>(fb, q0, q1, n, lane, o, denom);
    if (q < n) {
      const float inv = 1.0f / denom;
      bf16* dst = out + (size_t)(base + q * ts) * C + head * 64 + fg * 4;
#pragma unroll
      for (int dn = 0; dn < 4; ++dn) {
        bf16x4 r = {(bf16)(o[dn][0] * inv), (bf16)(o[dn][1] * inv), (bf16)(o[dn][2] * inv), (bf16)(o[dn][3] * inv)};
        *reinterpret_cast<bf16x4*>(dst + dn * 16) = r;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// EXACT-mode attention on the fp16 matrix cores: split-fp16 operands, three MFMA passes per product (the scheme of the
// EXACT Linear, gemm_x2.hip), same dataflow and LDS images as the bf16 kernels above with TWO planes per operand.
// Input: the PACKED qkv rows the qkv Linear writes in this mode (gemm_x2.hip, EPI_QKV_PACK), 12 C bytes per token:
//     q fp32 [C] | k hi [C] | k lo [C] | v hi [C] | v lo [C]      (fp16 planes of the value x 16)
// so K and V are already MFMA operands: the temporal kernel copies them HBM -> LDS with LDS-DMA (no registers, no VALU)
// and the query fragments are split in registers.  S^T = Kl.Qh + Kh.Ql + Kh.Qh (fp32 accumulate, scale 256 folded
//   into the exponent constant); two-pass fp32 softmax over the whole score row in registers; the probabilities are
//   split into fp16 pairs (x 1024) and O^T = Vl.Ph + Vh.Pl + Vh.Ph; the output leaves as fp32 or as the two fp16 planes
//   the proj Linear consumes.  5.3x fewer matrix-pipe cycles than the fp32-MFMA kernel it replaces (16x16x4 f32: 32
//   cycles for 2 Kflop; three 16x16x32 f16: 48 cycles for 16 Kflop).
// ------------------------------------------------------------------------------------------------
// (probabilities are split at scale 1024: the +10 in the exponent bias of the softmax below)

__device__ __forceinline__ f16x8 as_f16x8(bf16x8 v) { return __builtin_bit_cast(f16x8, v); }

// Operand scales of the split-fp16 attention kernels (kernel argument; wave-uniform).  q is split in registers at `q`; the
// k / v planes of the packed rows were written at that same scale by the qkv Linear; `cexp` = hd^-0.5 log2(e) / (q scale x
// k scale) folds both into the softmax exponent; `onorm` = 1 / v scale brings O^T back to its true scale; `oplane` = the scale
// at which a plane output is split (that of v: |o| <= max |v|).  Everything is 2^4-based unless capi.hip lowered the scale of
// a block whose proven operand range asks for it (d3dp_exact_range_bound).
// The true-scale value is formed FIRST and the (power-of-two, hence exact) plane scale applied inside the split on purpose:
// with  x = o * (1 / denom)  handed to the split directly, hipcc (ROCm 7.2) folds the fp16 conversion of the INEXACT product
// into v_fma_mixlo_f16 for the lo half -- f16(o * inv) rounded once from the exact product -- while the stored hi half is
// v_cvt_pk_f16_f32 of the fp32-rounded product: wherever the two roundings disagree (about 1 element in 2^13) hi + lo is off
// by a whole fp16 ulp of hi, and the denoiser's error against the reference tripled (gpurun c5: 6.4e-4 -> 1.8e-3 mm).  The
// exact scaling in between makes both halves round the same number.
struct X2Scales { float q, cexp, onorm, oplane; };
#ifndef D3DP_ATTN_STORE16
#define D3DP_ATTN_STORE16 1
#endif

// 8 consecutive fp32 -> one 16-byte slot of hi and one of lo (values x sc)
__device__ __forceinline__ void split8(const float4 a, const float4 b, f16x8& hi, f16x8& lo, float sc) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) { f16 h, l; split2h_scaled(v[e] * sc, h, l); hi[e] = h; lo[e] = l; }
}

// key chunks [C0, C1) of O^T += V^T P^T (32 keys per chunk).  Software-pipelined by hand: the fragments of chunk c+1 are
// read before the MFMAs of chunk c, and the twelve MFMAs of a chunk go pass by pass over the four channel tiles, so that
// two MFMAs on the same accumulator are four apart (issued back to back each waits out the previous one's latency:
// with two waves per SIMD the kernel spent 39 % of its wave cycles in such issue stalls).
template <int C0>
__device__ __forceinline__ void load_v_frags_x2(const FragBases& fb, int plane, f16x8 (&vh)[4], f16x8 (&vl)[4]) {
#pragma unroll
  for (int dn = 0; dn < 4; ++dn) {
    vh[dn] = as_f16x8(load_vt_frag<C0>(fb.v[dn]));
    vl[dn] = as_f16x8(load_vt_frag<C0>(fb.v[dn] + plane));
  }
}
template <int NKT, int C0, int C1 = NKT / 2>
__device__ __forceinline__ void pv_chunks_x2_rec(const FragBases& fb, int plane, const f16x8 (&ph)[NKT / 2],
                                                 const f16x8 (&pl)[NKT / 2], f32x4 (&o)[4], const f16x8 (&vh)[4],
                                                 const f16x8 (&vl)[4]) {
  f16x8 nh[4], nl[4];
  if constexpr (C0 + 1 < C1) load_v_frags_x2<C0 + 1>(fb, plane, nh, nl);
  __builtin_amdgcn_sched_barrier(0);                   // (the reads stay in front of the MFMAs that cover their latency)
#pragma unroll
  for (int dn = 0; dn < 4; ++dn) o[dn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl[dn], ph[C0], o[dn], 0, 0, 0);
#pragma unroll
  for (int dn = 0; dn < 4; ++dn) o[dn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[dn], pl[C0], o[dn], 0, 0, 0);
#pragma unroll
  for (int dn = 0; dn < 4; ++dn) o[dn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[dn], ph[C0], o[dn], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (C0 + 1 < C1) pv_chunks_x2_rec<NKT, C0 + 1, C1>(fb, plane, ph, pl, o, nh, nl);
}
template <int NKT, int C0, int C1 = NKT / 2>
__device__ __forceinline__ void pv_chunks_x2(const FragBases& fb, int plane, const f16x8 (&ph)[NKT / 2],
                                             const f16x8 (&pl)[NKT / 2], f32x4 (&o)[4]) {
  if constexpr (C0 < C1) {
    f16x8 vh[4], vl[4];
    load_v_frags_x2<C0>(fb, plane, vh, vl);
    pv_chunks_x2_rec<NKT, C0, C1>(fb, plane, ph, pl, o, vh, vl);
  }
}

// the same product one channel tile after the other (8 fragment registers live instead of 32: the spatial kernel, whose
// occupancy -- five waves per SIMD at <= 96 registers -- matters more to it than MFMA issue order)
template <int C0>
__device__ __forceinline__ void pv_chunk_x2_seq(const FragBases& fb, int plane, const f16x8& ph, const f16x8& pl, f32x4 (&o)[4]) {
#pragma unroll
  for (int dn = 0; dn < 4; ++dn) {
    const f16x8 vh = as_f16x8(load_vt_frag<C0>(fb.v[dn]));
    const f16x8 vl = as_f16x8(load_vt_frag<C0>(fb.v[dn] + plane));
    o[dn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph, o[dn], 0, 0, 0);
    o[dn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl, o[dn], 0, 0, 0);
    o[dn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph, o[dn], 0, 0, 0);
  }
}

// -DD3DP_ATTN_MIXLO=1 (measurement build): the lo halves of a pair of probabilities by v_fma_mixlo_f16 / v_fma_mixhi_f16 --
// f16(y - hi) rounded once from the exact difference, which is what convert-back, subtract and convert compute in three
// instructions per element, written straight into the packed register: 3 vector instructions per pair instead of 6.
#ifndef D3DP_ATTN_MIXLO
#define D3DP_ATTN_MIXLO 0
#endif
__device__ __forceinline__ unsigned x2_lo_pair_mix(float y0, float y1, unsigned hpair) {
  unsigned d;
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(d) : "v"(y0), "v"(y1), "v"(hpair));
  return d;
}
// Softmax of one 16-query tile's score row s (S^T fragments: lane holds keys 16 t + 4 fg + r of query lane & 15) and
// the split of the probabilities into fp16 pairs (x 1024); `denom` = 1024 x the softmax denominator of query (lane & 15).
template <int NKT, bool MASK_ANY_TILE>
__device__ __forceinline__ void softmax_split_x2(f32x4 (&s)[NKT], int n, int lane, f16x8 (&ph)[NKT / 2], f16x8 (&pl)[NKT / 2],
                                                 float& denom, float cexp) {
  const int fg = lane >> 4;                            // cexp = hd^-0.5 log2(e) / (q scale x k scale), X2Scales
  // keys >= n are masked.  The usual case (n in the last key tile: MASK_ANY_TILE = false, chosen by the launcher) touches
  // that tile only; the per-element selects of the general form were a third of this kernel's VALU instructions, and
  // their 64 loop-invariant lane masks cost registers.
  if constexpr (!MASK_ANY_TILE) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (16 * (NKT - 1) + 4 * fg + r >= n) s[NKT - 1][r] = -INFINITY;
  } else {
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * t + 4 * fg + r >= n) s[t][r] = -INFINITY;
  }
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[t][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  // p * 1024 = exp2(s cexp - mx cexp + 10): one fma and the bare v_exp_f32 per element (arguments <= 10; a probability
  // below 2^-126 becomes 0 instead of a denormal); the denominator is accumulated at the same scale.
  const float nb = fmaf(-mx, cexp, 10.0f);
  float sum4[2] = {0.f, 0.f};                          // two independent chains
#if D3DP_ATTN_MIXLO
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    using f16x2v = _Float16 __attribute__((ext_vector_type(2)));
    unsigned hp[2], lp[2];
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      const float y0 = __builtin_amdgcn_exp2f(fmaf(s[t][r], cexp, nb)), y1 = __builtin_amdgcn_exp2f(fmaf(s[t][r + 1], cexp, nb));
      sum4[0] += y0;
      sum4[1] += y1;
      const f16x2v h2 = {(f16)y0, (f16)y1};
      hp[r >> 1] = __builtin_bit_cast(unsigned, h2);
      lp[r >> 1] = x2_lo_pair_mix(y0, y1, hp[r >> 1]);
    }
    using u32x4v = unsigned __attribute__((ext_vector_type(4)));
    u32x4v hv = __builtin_bit_cast(u32x4v, ph[t >> 1]), lv = __builtin_bit_cast(u32x4v, pl[t >> 1]);
    hv[(t & 1) * 2] = hp[0]; hv[(t & 1) * 2 + 1] = hp[1];
    lv[(t & 1) * 2] = lp[0]; lv[(t & 1) * 2 + 1] = lp[1];
    ph[t >> 1] = __builtin_bit_cast(f16x8, hv);
    pl[t >> 1] = __builtin_bit_cast(f16x8, lv);
  }
#else
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float y = __builtin_amdgcn_exp2f(fmaf(s[t][r], cexp, nb));
      sum4[r & 1] += y;
      const f16 h = (f16)y;
      ph[t >> 1][(t & 1) * 4 + r] = h;
      pl[t >> 1][(t & 1) * 4 + r] = (f16)fmaf(y, 1.0f, -(float)h);
    }
  }
#endif
  float sum = sum4[0] + sum4[1];
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  denom = sum;                                         // = 1024 x the softmax denominator
}

// ---- the same softmax in pieces, for the overlapped form of the temporal kernel (D3DP_ATTN_OVERLAP) ------------------------
// mask + row maximum of a score row -> the exponent bias nb (p x 1024 = exp2(s cexp + nb))
template <int NKT, bool MASK_ANY_TILE>
__device__ __forceinline__ float x2_mask_rowmax(f32x4 (&s)[NKT], int n, int lane, float cexp) {
  const int fg = lane >> 4;
  if constexpr (!MASK_ANY_TILE) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (16 * (NKT - 1) + 4 * fg + r >= n) s[NKT - 1][r] = -INFINITY;
  } else {
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * t + 4 * fg + r >= n) s[t][r] = -INFINITY;
  }
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[t][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  return fmaf(-mx, cexp, 10.0f);
}
// key tiles t, t + 1 (t even) of a masked score row -> one split-fp16 B operand pair; the two partial sums continue the
// chains of softmax_split_x2 (same order of additions: slices taken in ascending t give the same denominator bit for bit)
__device__ __forceinline__ void x2_softmax_slice(const f32x4& sa, const f32x4& sb, float nb, float cexp, f16x8& ph, f16x8& pl,
                                                 float (&sum2)[2]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float y = __builtin_amdgcn_exp2f(fmaf(sa[r], cexp, nb));
    sum2[r & 1] += y;
    const f16 h = (f16)y;
    ph[r] = h;
    pl[r] = (f16)fmaf(y, 1.0f, -(float)h);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float y = __builtin_amdgcn_exp2f(fmaf(sb[r], cexp, nb));
    sum2[r & 1] += y;
    const f16 h = (f16)y;
    ph[4 + r] = h;
    pl[4 + r] = (f16)fmaf(y, 1.0f, -(float)h);
  }
}
__device__ __forceinline__ float x2_denominator(const float (&sum2)[2]) {
  float sum = sum2[0] + sum2[1];
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  return sum;
}
// S^T of one query tile (the loop of attn_scores_x2 below); with OVL the softmax slices of ANOTHER tile's finished score row
// `sp` ride between the MFMAs: the wave's VALU work runs in the shadow of its own matrix instructions instead of after them.
template <int NKT, bool OVL>
__device__ __forceinline__ void x2_scores_rows(const FragBases& fb, int plane, const f16x8 (&qh)[2], const f16x8 (&ql)[2],
                                               f32x4 (&s)[NKT], const f32x4 (&sp)[OVL ? NKT : 1], float nbp, float cexp,
                                               f16x8 (&php)[OVL ? NKT / 2 : 1], f16x8 (&plp)[OVL ? NKT / 2 : 1], float (&sump)[2]) {
  auto read_k = [&](int t, f16x8 (&k)[4]) {
    k[0] = *reinterpret_cast<const f16x8*>(fb.k0 + plane + t * 2048);   // lo, d 0..31
    k[1] = *reinterpret_cast<const f16x8*>(fb.k1 + plane + t * 2048);   // lo, d 32..63
    k[2] = *reinterpret_cast<const f16x8*>(fb.k0 + t * 2048);           // hi
    k[3] = *reinterpret_cast<const f16x8*>(fb.k1 + t * 2048);
  };
  f16x8 kc[2][4], kn[2][2];
  read_k(0, kc[0]);
  read_k(1, kc[1]);
#pragma unroll
  for (int t = 0; t < NKT; t += 2) {
    if (t + 2 < NKT) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        kn[j][0] = *reinterpret_cast<const f16x8*>(fb.k0 + plane + (t + 2 + j) * 2048);
        kn[j][1] = *reinterpret_cast<const f16x8*>(fb.k1 + plane + (t + 2 + j) * 2048);
      }
    }
    __builtin_amdgcn_sched_barrier(0);                 // (the reads stay in front of the MFMAs that cover their latency)
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[0][0], qh[0], a, 0, 0, 0);
    b = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[1][0], qh[0], b, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[0][1], qh[1], a, 0, 0, 0);
    b = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[1][1], qh[1], b, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[0][2], ql[0], a, 0, 0, 0);
    b = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[1][2], ql[0], b, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[0][3], ql[1], a, 0, 0, 0);
    b = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[1][3], ql[1], b, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[0][2], qh[0], a, 0, 0, 0);
    b = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[1][2], qh[0], b, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[0][3], qh[1], a, 0, 0, 0);
    b = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[1][3], qh[1], b, 0, 0, 0);
    if constexpr (OVL) {
      x2_softmax_slice(sp[t], sp[t + 1], nbp, cexp, php[t >> 1], plp[t >> 1], sump);
#if D3DP_ATTN_OVL_SGB
#pragma unroll
      for (int i = 0; i < 12; ++i) {                   // one matrix instruction, then its share of the slice's vector work
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, D3DP_ATTN_OVL_SGB, 0);
      }
#endif
    }
    s[t] = a; s[t + 1] = b;
    if (t + 2 < NKT) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        kc[j][0] = kn[j][0]; kc[j][1] = kn[j][1];
        kc[j][2] = *reinterpret_cast<const f16x8*>(fb.k0 + (t + 2 + j) * 2048);
        kc[j][3] = *reinterpret_cast<const f16x8*>(fb.k1 + (t + 2 + j) * 2048);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
// O^T += V^T P^T of one tile (the pipeline of pv_chunks_x2) with the softmax slices of the OTHER tile's score row `sp` between
// the MFMAs of each key chunk: chunk c of this product needs this tile's probabilities only, which are complete.
template <int NKT, int C0>
__device__ __forceinline__ void x2_pv_overlap_rec(const FragBases& fb, int plane, const f16x8 (&ph)[NKT / 2],
                                                  const f16x8 (&pl)[NKT / 2], f32x4 (&o)[4], const f16x8 (&vh)[4],
                                                  const f16x8 (&vl)[4], const f32x4 (&sp)[NKT], float nbp, float cexp,
                                                  f16x8 (&php)[NKT / 2], f16x8 (&plp)[NKT / 2], float (&sump)[2]) {
  f16x8 nh[4], nl[4];
  if constexpr (C0 + 1 < NKT / 2) load_v_frags_x2<C0 + 1>(fb, plane, nh, nl);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int dn = 0; dn < 4; ++dn) o[dn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl[dn], ph[C0], o[dn], 0, 0, 0);
#pragma unroll
  for (int dn = 0; dn < 4; ++dn) o[dn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[dn], pl[C0], o[dn], 0, 0, 0);
#pragma unroll
  for (int dn = 0; dn < 4; ++dn) o[dn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[dn], ph[C0], o[dn], 0, 0, 0);
  x2_softmax_slice(sp[2 * C0], sp[2 * C0 + 1], nbp, cexp, php[C0], plp[C0], sump);
#if D3DP_ATTN_OVL_SGB
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, D3DP_ATTN_OVL_SGB, 0);
  }
#endif
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (C0 + 1 < NKT / 2) x2_pv_overlap_rec<NKT, C0 + 1>(fb, plane, ph, pl, o, nh, nl, sp, nbp, cexp, php, plp, sump);
}
template <int NKT>
__device__ __forceinline__ void x2_pv_overlap(const FragBases& fb, int plane, const f16x8 (&ph)[NKT / 2], const f16x8 (&pl)[NKT / 2],
                                              f32x4 (&o)[4], const f32x4 (&sp)[NKT], float nbp, float cexp, f16x8 (&php)[NKT / 2],
                                              f16x8 (&plp)[NKT / 2], float (&sump)[2]) {
  f16x8 vh[4], vl[4];
  load_v_frags_x2<0>(fb, plane, vh, vl);
  x2_pv_overlap_rec<NKT, 0>(fb, plane, ph, pl, o, vh, vl, sp, nbp, cexp, php, plp, sump);
}

// Scores and softmax of one 16-query tile against NKT 16-key tiles resident in LDS.  fb: fragment bases into the K hi
// image (the lo image is `plane` bytes further).  qh/ql: the tile's query fragments (d 0..31, 32..63), values x 16.
// Returns the probabilities as split-fp16 B operands (x 1024) and 1024 x the softmax denominator of query (lane & 15).
template <int NKT, bool MASK_ANY_TILE>
__device__ __forceinline__ void attn_scores_x2(const FragBases& fb, int plane, const f16x8 (&qh)[2], const f16x8 (&ql)[2],
                                               int n, int lane, f16x8 (&ph)[NKT / 2], f16x8 (&pl)[NKT / 2], float& denom,
                                               float cexp) {
  // S^T, two key tiles at a time with their MFMAs interleaved (two independent accumulator chains) and the fragments
  // of the next pair already on their way from LDS (see pv_chunks_x2)
  f32x4 s[NKT];
  auto read_k = [&](int t, f16x8 (&k)[4]) {
    k[0] = *reinterpret_cast<const f16x8*>(fb.k0 + plane + t * 2048);   // lo, d 0..31
    k[1] = *reinterpret_cast<const f16x8*>(fb.k1 + plane + t * 2048);   // lo, d 32..63
    k[2] = *reinterpret_cast<const f16x8*>(fb.k0 + t * 2048);           // hi
    k[3] = *reinterpret_cast<const f16x8*>(fb.k1 + t * 2048);
  };
  // PF = 1: the lo-plane fragments of the NEXT pair are read before this pair's MFMAs (16 registers; prefetching all
  // four fragments per tile does not fit the 256 registers of the two-tile temporal kernel)
  f16x8 kc[2][4], kn[2][2];
  read_k(0, kc[0]);
  read_k(1, kc[1]);
#pragma unroll
  for (int t = 0; t < NKT; t += 2) {
    if (t + 2 < NKT) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        kn[j][0] = *reinterpret_cast<const f16x8*>(fb.k0 + plane + (t + 2 + j) * 2048);
        kn[j][1] = *reinterpret_cast<const f16x8*>(fb.k1 + plane + (t + 2 + j) * 2048);
      }
    }
    __builtin_amdgcn_sched_barrier(0);                 // (the reads stay in front of the MFMAs that cover their latency)
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[0][0], qh[0], a, 0, 0, 0);
    b = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[1][0], qh[0], b, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[0][1], qh[1], a, 0, 0, 0);
    b = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[1][1], qh[1], b, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[0][2], ql[0], a, 0, 0, 0);
    b = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[1][2], ql[0], b, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[0][3], ql[1], a, 0, 0, 0);
    b = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[1][3], ql[1], b, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[0][2], qh[0], a, 0, 0, 0);
    b = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[1][2], qh[0], b, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[0][3], qh[1], a, 0, 0, 0);
    b = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[1][3], qh[1], b, 0, 0, 0);
    s[t] = a; s[t + 1] = b;
    if (t + 2 < NKT) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        kc[j][0] = kn[j][0]; kc[j][1] = kn[j][1];
        kc[j][2] = *reinterpret_cast<const f16x8*>(fb.k0 + (t + 2 + j) * 2048);
        kc[j][3] = *reinterpret_cast<const f16x8*>(fb.k1 + (t + 2 + j) * 2048);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  softmax_split_x2<NKT, MASK_ANY_TILE>(s, n, lane, ph, pl, denom, cexp);
}

// The same for BOTH 16-query tiles of a wave at once (temporal kernel, two tiles per wave): every K fragment is read from
// LDS once for the two tiles (half the fragment reads) and the MFMAs run as FOUR independent accumulator chains (two query
// tiles x two key tiles), so two MFMAs on the same accumulator are four instructions apart instead of two -- half of this
// kernel's wave cycles were issue stalls behind dependent MFMAs with the matrix pipes 29 % busy (profiles/r03_attn_pmc.md).
template <int NKT, bool MASK_ANY_TILE>
__device__ __forceinline__ void attn_scores_x2_pair(const FragBases& fb, int plane, const f16x8 (&qh)[2][2],
                                                    const f16x8 (&ql)[2][2], int n, int lane, f16x8 (&ph)[2][NKT / 2],
                                                    f16x8 (&pl)[2][NKT / 2], float (&denom)[2], float cexp) {
  f32x4 s0[NKT], s1[NKT];
  auto read_k = [&](int t, f16x8 (&k)[4]) {
    k[0] = *reinterpret_cast<const f16x8*>(fb.k0 + plane + t * 2048);   // lo, d 0..31
    k[1] = *reinterpret_cast<const f16x8*>(fb.k1 + plane + t * 2048);   // lo, d 32..63
    k[2] = *reinterpret_cast<const f16x8*>(fb.k0 + t * 2048);           // hi
    k[3] = *reinterpret_cast<const f16x8*>(fb.k1 + t * 2048);
  };
  // No registers are left for a prefetch of the next pair's fragments (s0 + s1 hold 128).  -DD3DP_ATTN_KPIPE=1 is the
  // prefetch that needs none -- a pair's lo fragments are dead after its first eight MFMAs and its hi fragments after its
  // last, so the next pair's are requested into the same registers behind MFMA 8 and MFMA 24 instead of all eight in front
  // of the pair's MFMAs, where the compiler puts them.  Measured (gpurun a2, interleaved): no difference (625 against
  // 626 us for the micro-benchmark's repack + kernel) -- the other wave of the SIMD already covers those LDS round trips.
#ifndef D3DP_ATTN_KPIPE
#define D3DP_ATTN_KPIPE 0
#endif
  f16x8 kc[2][4];
#define X2_PAIR_TERM(KI, Q, D)                                                                  \
    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[0][KI], Q[0][D], a0, 0, 0, 0);               \
    b0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[1][KI], Q[0][D], b0, 0, 0, 0);               \
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[0][KI], Q[1][D], a1, 0, 0, 0);               \
    b1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[1][KI], Q[1][D], b1, 0, 0, 0);
  auto read_lo = [&](int t, f16x8 (&k)[4]) {
    k[0] = *reinterpret_cast<const f16x8*>(fb.k0 + plane + t * 2048);
    k[1] = *reinterpret_cast<const f16x8*>(fb.k1 + plane + t * 2048);
  };
  auto read_hi = [&](int t, f16x8 (&k)[4]) {
    k[2] = *reinterpret_cast<const f16x8*>(fb.k0 + t * 2048);
    k[3] = *reinterpret_cast<const f16x8*>(fb.k1 + t * 2048);
  };
#if D3DP_ATTN_KPIPE
  read_k(0, kc[0]);
  read_k(1, kc[1]);
#endif
#pragma unroll
  for (int t = 0; t < NKT; t += 2) {
#if !D3DP_ATTN_KPIPE
    read_k(t, kc[0]);
    read_k(t + 1, kc[1]);
#endif
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, b0 = a0, a1 = a0, b1 = a0;
    // K fragment index / query operand of the six product terms: Kl.Qh (d 0..31, 32..63), Kh.Ql, Kh.Qh
    X2_PAIR_TERM(0, qh, 0)
    X2_PAIR_TERM(1, qh, 1)
#if D3DP_ATTN_KPIPE
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < NKT) { read_lo(t + 2, kc[0]); read_lo(t + 3, kc[1]); }
    __builtin_amdgcn_sched_barrier(0);
#endif
    X2_PAIR_TERM(2, ql, 0)
    X2_PAIR_TERM(3, ql, 1)
    X2_PAIR_TERM(2, qh, 0)
    X2_PAIR_TERM(3, qh, 1)
#if D3DP_ATTN_KPIPE
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < NKT) { read_hi(t + 2, kc[0]); read_hi(t + 3, kc[1]); }
    __builtin_amdgcn_sched_barrier(0);
#endif
    s0[t] = a0; s0[t + 1] = b0; s1[t] = a1; s1[t + 1] = b1;
  }
#undef X2_PAIR_TERM
  softmax_split_x2<NKT, MASK_ANY_TILE>(s0, n, lane, ph[0], pl[0], denom[0], cexp);
  softmax_split_x2<NKT, MASK_ANY_TILE>(s1, n, lane, ph[1], pl[1], denom[1], cexp);
}

// this lane's query fragments (16 fp32 of row q: d = fg*8 .. +7 and 32 + fg*8 .. +7), split
__device__ __forceinline__ void load_q_x2(const float* qrow, int fg, f16x8 (&qh)[2], f16x8 (&ql)[2], float sc) {
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const float* p = qrow + half * 32 + fg * 8;
    split8(*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4), qh[half], ql[half], sc);
  }
}

// this lane's K fragments of key row `krow` (packed row: hi plane at krow, lo plane C halves further): already operands
__device__ __forceinline__ void load_k_x2(const f16* krow, int C, int fg, f16x8 (&kh)[2], f16x8 (&kl)[2]) {
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    kh[half] = *reinterpret_cast<const f16x8*>(krow + half * 32 + fg * 8);
    kl[half] = *reinterpret_cast<const f16x8*>(krow + C + half * 32 + fg * 8);
  }
}

// O^T accumulators -> out row `tok` (fp32 [T][C], or the h2i layout of the proj Linear's operand, common.h): lane holds
// channels col + dn*16 + (0..3), col = head*64 + 4 fg.  `inv` = onorm / (1024 x softmax denominator): o * inv is the TRUE-scale
// value; plane outputs are split at `oplane` (see X2Scales for why in this order)
template <int OUTS>
__device__ __forceinline__ void store_o_x2(const f32x4 (&o)[4], float inv, void* out_v, size_t tok, int C, int col,
                                           float oplane) {
#pragma unroll
  for (int dn = 0; dn < 4; ++dn) {
    const float r4[4] = {o[dn][0] * inv, o[dn][1] * inv, o[dn][2] * inv, o[dn][3] * inv};
    if constexpr (OUTS == 2) {
      f16x4 p0, p1;
#pragma unroll
      for (int e = 0; e < 4; ++e) { f16 a0, a1; split2h_scaled(r4[e] * oplane, a0, a1); p0[e] = a0; p1[e] = a1; }
#if D3DP_ATTN_STORE16
      // ONE 16-byte store per lane instead of two of 8 bytes (the store side of these latency-bound
      // kernels).  Lanes l and l ^ 16 hold neighbouring 4-channel groups of the same row (fg even / odd): v_permlane16_swap
      // exchanges the odd 16-lane rows of its first operand with the even rows of its second, so with (hi, lo) as operands the
      // even-row lane ends up with both hi halves -- 8 consecutive channels of the hi slot -- and the odd-row lane with both
      // lo halves.  (Both lanes share `lane & 15`, i.e. the caller's q < n predicate.)
      const uint2 h = __builtin_bit_cast(uint2, p0), l = __builtin_bit_cast(uint2, p1);
      const auto sx = __builtin_amdgcn_permlane16_swap(h.x, l.x, false, false);
      const auto sy = __builtin_amdgcn_permlane16_swap(h.y, l.y, false, false);
      using u32x4 = unsigned __attribute__((ext_vector_type(4)));
      const u32x4 out16 = {sx[0], sy[0], sx[1], sy[1]};  // even row: hi own | hi partner; odd row: lo partner | lo own
      const bool oddrow = (col >> 2) & 1;                // fg odd (col = head 64 + 4 fg)
      const int c8 = (col & ~7) + dn * 16;               // first of the pair's 8 channels
      f16* dst = reinterpret_cast<f16*>(out_v) + tok * (2 * C) + h2i_col(c8) + (oddrow ? kH2iLo : 0);
      *reinterpret_cast<u32x4*>(dst) = out16;
#else
      f16* dst = reinterpret_cast<f16*>(out_v) + tok * (2 * C) + h2i_col(col + dn * 16);
      *reinterpret_cast<f16x4*>(dst) = p0;
      *reinterpret_cast<f16x4*>(dst + kH2iLo) = p1;
#endif
    } else {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(out_v) + tok * C + col + dn * 16) = make_float4(r4[0], r4[1], r4[2], r4[3]);
    }
  }
}

// One LDS-DMA wave-instruction: lane l's 16 bytes at `g` -> LDS bytes [lds + 16 l, +16) (`lds` wave-uniform).  Issued as
// inline assembly on purpose: for the builtin the compiler (a) holds every later ds_read that might alias the destination
// behind vmcnt(0) and (b) marks the instruction as a FLAT access of two address spaces, after which it stops counting
// and turns EVERY vector-memory wait of the kernel into vmcnt(0) -- both would serialise the pipeline below, whose
// ordering is carried by explicit counted waits and barriers instead.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"          // (the expected "clobber list contains reserved registers" note for m0)
__device__ __forceinline__ void lds_dma16(const char* g, const char* lds) {
  const unsigned la = (unsigned)(__UINTPTR_TYPE__)LPTR(lds);
  // m0 is declared clobbered (the compiler answers with "clobber list contains reserved registers" and
  // re-materialises m0 before each of its own uses, of which these kernels have none): no instruction is added to the
  // hand-counted pipeline
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(la) : "memory", "m0");
}
#pragma clang diagnostic pop

// 16-byte global load the compiler does not track (it would wait for it with counts that ignore the LDS-DMA operations
// queued behind it, i.e. far too early): the caller waits with wait_vmcnt and then passes the registers through settle().
__device__ __forceinline__ f32x4 gload16_untracked(const float* p) {
  f32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
  return r;
}
__device__ __forceinline__ void settle(f32x4& r) { asm volatile("" : "+v"(r)); }

// s_waitcnt vmcnt(v) with the other counters left alone (gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8])
template <int V>
__device__ __forceinline__ void wait_vmcnt() { __builtin_amdgcn_s_waitcnt((V & 15) | (7 << 4) | (15 << 8) | ((V >> 4) << 14)); }

// temporal axis: PERSISTENT workgroups (one per CU: the four planes of 256 keys fill 128 KiB of LDS), 8 waves, each wave
// TPW = 1 or 2 16-query tiles of the current (sequence, head) problem (two waves per SIMD -> 256 registers per wave:
// the score row, the probabilities of both tiles and the prefetched queries of the next problem all stay in registers).
// K and V arrive by LDS-DMA straight from the packed qkv rows (8 rows x 128 B per wave-instruction, the image swizzle
// applied on the global side) and are double-buffered in TIME, not in space:
//     K(p+1) streams in while problem p multiplies P.V,   V(p+1) while problem p+1 computes its scores.
//   top:  vmcnt(V pieces) -> K(p), q(p) landed        barrier 1     scores + softmax
//         vmcnt(0)        -> V(p) landed              barrier 2     (every wave is done with the K image)  issue K(p+1)
//         P.V (q(p+1) requested on the way), store O  barrier 3     (every wave is done with the V image)  issue V(p+1)
// The vmcnt(2 PER) at the top is exact: vector memory operations of a wave retire in order and the 2 PER youngest are
// the V pieces.
// measurement build (-DD3DP_ATTN_STAMP=1): s_memtime at the phase boundaries of the first 16 problems of workgroups 0..3,
// every wave's lane 0 -> d3dp_attn_stamps[wg][wave][problem][8] (read back by d3dp_debug_attn_stamps)
#ifndef D3DP_ATTN_STAMP
#define D3DP_ATTN_STAMP 0
#endif
// timing probes of the temporal split-fp16 kernel (results INVALID): 1 = the output stores, 2 = the query loads, 4 = the K / V
// LDS-DMA sit behind a condition that is false at run time (the arithmetic stays)
#ifndef D3DP_ATTN_PROBE
#define D3DP_ATTN_PROBE 0
#endif
#define ATTN_PROBE_OFF(bit) (!(D3DP_ATTN_PROBE & (bit)) || sc.q == -12345.f)
#if D3DP_ATTN_STAMP
__device__ unsigned long long d3dp_attn_stamps[4 * 8 * 16 * 8];
#define ATTN_STAMP(k)                                                                                          \
  do {                                                                                                         \
    if (blockIdx.x < 4 && iter < 16 && lane == 0)                                                              \
      d3dp_attn_stamps[((blockIdx.x * 8 + wave) * 16 + iter) * 8 + (k)] = __builtin_readcyclecounter();        \
  } while (0)
#else
#define ATTN_STAMP(k) do { } while (0)
#endif
template <int NKT, int OUTS, bool MASK_ANY_TILE>
__global__ __launch_bounds__(512) void attn_temporal_x2_kernel(const float* __restrict__ qkv, void* __restrict__ out_v,
                                                               SeqMap map, int C, int heads, size_t plane_elems,
                                                               int n_prob, X2Scales sc) {
  constexpr int NW = 8;
  constexpr int TPW = (NKT + NW - 1) / NW;             // query tiles per wave
  constexpr int NK = 16 * NKT;
  constexpr int PLANE = NK * 128;
  constexpr int PER = (2 * NKT + NW - 1) / NW;         // DMA pieces per wave and plane
  __shared__ __attribute__((aligned(16))) char kimg[2 * PLANE];
  __shared__ __attribute__((aligned(16))) char vimg[2 * PLANE];
  const int n = map.n_tok;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ts = map.tok_stride;
  const size_t ldb = (size_t)12 * C;                   // bytes per packed token row
  const int n_qt = (n + 15) >> 4;
  const float inv_scale = sc.onorm;                    // O^T is scaled by (v scale) x 1024, the denominator by 1024

  // DMA piece `pc` of a plane = image rows 8 pc .. 8 pc + 7; a lane moves the 16-byte slot that belongs at position
  // (lane & 7) of row 8 pc + (lane >> 3): K slot s sits at s ^ ((row >> 1) & 7), V slot s at s ^ (((row >> 1) & 3) << 1)
  auto problem_row0 = [&](int p, int& head) -> const char* {
    const int seq = p / heads;
    head = p - seq * heads;
    return reinterpret_cast<const char*>(qkv) + (size_t)seq_base(map, seq) * ldb;
  };
  // (per-lane source offsets are re-derived at every use from a copy of the lane id the compiler cannot see through:
  // hoisted out of the problem loop they are ~40 live registers, which spills -- and scratch traffic shares vmcnt)
  auto opaque = [](int x) { asm volatile("" : "+v"(x)); return x; };
  auto issue_kv = [&](const char* row0, int head, int is_v) {
    const int l = opaque(lane);
    const char* g = row0 + (is_v ? 8 : 4) * C + head * 128;
    char* img = is_v ? vimg : kimg;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int pc = min(wave + j * NW, 2 * NKT - 1);  // (surplus pieces repeat the last one: same bytes, same place)
      const int row = pc * 8 + (l >> 3);
      const int sw = is_v ? (((row >> 1) & 3) << 1) : ((row >> 1) & 7);
      // rows >= n: a copy of the last row (finite values; their scores are masked, their probabilities are 0)
      const char* gj = g + (unsigned)(min(row, n - 1) * ts) * (unsigned)ldb + (((l & 7) ^ sw) << 4);
      if (ATTN_PROBE_OFF(4)) {
        lds_dma16(gj, img + pc * 1024);
        lds_dma16(gj + 2 * C, img + PLANE + pc * 1024);
      }
    }
  };
  // raw query fragments of the next problem.  (Not zero-initialised on purpose: the compiler would place the zeroing of
  // the lanes / tiles that load nothing right behind the untracked loads, inside their in-flight window, which the ISA scan
  // of tests/test_abi.py forbids.  A wave whose second tile does not exist -- 129..240 frames -- runs the paired score path
  // on whatever these registers hold and never stores the result.)
  f32x4 qr[TPW][4];
  auto load_q_raw = [&](const char* row0, int head) {
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      const int qt = wave + u * NW;
      if (qt < n_qt && ATTN_PROBE_OFF(2)) {
        const int l = opaque(lane);
        const int q = qt * 16 + (l & 15);
        const float* qrow = reinterpret_cast<const float*>(row0 + (size_t)min(q, n - 1) * ts * ldb) + head * 64 + (l >> 4) * 8;
        qr[u][0] = gload16_untracked(qrow);
        qr[u][1] = gload16_untracked(qrow + 4);
        qr[u][2] = gload16_untracked(qrow + 32);
        qr[u][3] = gload16_untracked(qrow + 36);
      }
    }
  };

  int p = blockIdx.x;
  if (p >= n_prob) return;
  // waves w and w + 4 share a SIMD and run the same phases between the same barriers: the static priority lets one of
  // them take the matrix pipe first, after which its softmax (VALU) runs beside the other's MFMAs (measured: 563 vs
  // 568 ms per step without it -- within noise; starting odd workgroups half a problem late changed nothing either)
  if ((wave >> 2) & 1) __builtin_amdgcn_s_setprio(1);
  int head;
  const char* row0 = problem_row0(p, head);
  issue_kv(row0, head, 0);
  load_q_raw(row0, head);
  issue_kv(row0, head, 1);
  [[maybe_unused]] int iter = 0;
  for (;; ++iter) {
    ATTN_STAMP(0);
    wait_vmcnt<2 * PER>();
    ATTN_STAMP(1);
    __builtin_amdgcn_s_barrier();                      // 1: the K image of problem p is complete
    ATTN_STAMP(2);
    f16x8 ph[TPW][NKT / 2], pl[TPW][NKT / 2];
    float denom[TPW];
    // (fragment bases re-derived per phase from the opaque lane id: the four V bases are not live during the scores)
    const FragBases fbk = make_frag_bases(kimg, vimg, opaque(lane));
    // (not with per-element masks in every key tile -- 225..240 frames on 16 key tiles: their lane masks push the overlapped
    //  form past 256 registers; those lengths keep the paired form)
    constexpr bool OVL = TPW == 2 && D3DP_ATTN_OVERLAP && !MASK_ANY_TILE;
    [[maybe_unused]] f32x4 s1[OVL ? NKT : 1];          // OVL: tile 1's masked score row, turned into probabilities during tile 0's P.V
    [[maybe_unused]] float nb1 = 0.f;
    if constexpr (OVL) {
      f16x8 qh[2][2], ql[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float4 qf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          settle(qr[u][i]);                            // (after the counted wait above)
          qf[i] = make_float4(qr[u][i][0], qr[u][i][1], qr[u][i][2], qr[u][i][3]);
        }
        split8(qf[0], qf[1], qh[u][0], ql[u][0], sc.q);
        split8(qf[2], qf[3], qh[u][1], ql[u][1], sc.q);
      }
      f32x4 s0[NKT];
      f32x4 none_s[1];
      f16x8 none_p[1];
      float sum0[2] = {0.f, 0.f};
      x2_scores_rows<NKT, false>(fbk, PLANE, qh[0], ql[0], s0, none_s, 0.f, sc.cexp, none_p, none_p, sum0);
      const float nb0 = x2_mask_rowmax<NKT, MASK_ANY_TILE>(s0, n, lane, sc.cexp);
      // tile 1's scores (a wave without a second tile computes them on whatever its registers hold and never stores the
      // result) with tile 0's softmax between the MFMAs
      x2_scores_rows<NKT, true>(fbk, PLANE, qh[1], ql[1], s1, s0, nb0, sc.cexp, ph[0], pl[0], sum0);
      denom[0] = x2_denominator(sum0);
      nb1 = x2_mask_rowmax<NKT, MASK_ANY_TILE>(s1, n, lane, sc.cexp);
    } else if constexpr (TPW == 2 && D3DP_ATTN_PAIR) {
      {                                                // both query tiles of the wave in one pass over K (a wave whose second
                                                       // tile does not exist computes it on whatever its registers hold and
                                                       // never stores it: one code path, no second register allocation)
        f16x8 qh[2][2], ql[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float4 qf[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            settle(qr[u][i]);                          // (after the counted wait above)
            qf[i] = make_float4(qr[u][i][0], qr[u][i][1], qr[u][i][2], qr[u][i][3]);
          }
          split8(qf[0], qf[1], qh[u][0], ql[u][0], sc.q);
          split8(qf[2], qf[3], qh[u][1], ql[u][1], sc.q);
        }
        attn_scores_x2_pair<NKT, MASK_ANY_TILE>(fbk, PLANE, qh, ql, n, lane, ph, pl, denom, sc.cexp);
      }
    } else {
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      denom[u] = 1.f;
      if (wave + u * NW < n_qt) {
        f16x8 qh[2], ql[2];
        float4 qf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          settle(qr[u][i]);                            // (after the counted wait above)
          qf[i] = make_float4(qr[u][i][0], qr[u][i][1], qr[u][i][2], qr[u][i][3]);
        }
        split8(qf[0], qf[1], qh[0], ql[0], sc.q);
        split8(qf[2], qf[3], qh[1], ql[1], sc.q);
        attn_scores_x2<NKT, MASK_ANY_TILE>(fbk, PLANE, qh, ql, n, lane, ph[u], pl[u], denom[u], sc.cexp);
      }
    }
    }
    ATTN_STAMP(3);
    wait_vmcnt<0>();
    ATTN_STAMP(4);
    __builtin_amdgcn_s_barrier();                      // 2: V image complete; nobody reads the K image any more
    ATTN_STAMP(5);
    const int pn = p + gridDim.x;
    const bool has_next = pn < n_prob;
    int head_n = 0;
    const char* row0_n = row0;
    if (has_next) {
      row0_n = problem_row0(pn, head_n);
      issue_kv(row0_n, head_n, 0);
    }
    const int tok0 = seq_base(map, p / heads);
    const FragBases fb = make_frag_bases(kimg, vimg, opaque(lane));
    if constexpr (OVL) {
      f32x4 o[4];
#pragma unroll
      for (int dn = 0; dn < 4; ++dn) o[dn] = (f32x4){0.f, 0.f, 0.f, 0.f};
      float sum1[2] = {0.f, 0.f};
      x2_pv_overlap<NKT>(fb, PLANE, ph[0], pl[0], o, s1, nb1, sc.cexp, ph[1], pl[1], sum1);
      denom[1] = x2_denominator(sum1);
      const int l = opaque(lane);
      const int q = wave * 16 + (l & 15);
      if (q < n && ATTN_PROBE_OFF(1))
        store_o_x2<OUTS>(o, inv_scale / denom[0], out_v, (size_t)(tok0 + q * ts), C, head * 64 + (l >> 4) * 4, sc.oplane);
    }
#pragma unroll
    for (int u = OVL ? 1 : 0; u < TPW; ++u) {
      // the next problem's queries: requested before the LAST tile's P.V (with two tiles the probabilities of both are
      // live during the first one's, and 32 more registers there would spill)
      if (u == TPW - 1 && has_next) load_q_raw(row0_n, head_n);
      const int qt = wave + u * NW;
      if (qt < n_qt) {
        f32x4 o[4];
#pragma unroll
        for (int dn = 0; dn < 4; ++dn) o[dn] = (f32x4){0.f, 0.f, 0.f, 0.f};
        pv_chunks_x2<NKT, 0>(fb, PLANE, ph[u], pl[u], o);
        const int l = opaque(lane);
        const int q = qt * 16 + (l & 15);
        if (q < n && ATTN_PROBE_OFF(1))
          store_o_x2<OUTS>(o, inv_scale / denom[u], out_v, (size_t)(tok0 + q * ts), C, head * 64 + (l >> 4) * 4, sc.oplane);
      }
    }
    if (!has_next) break;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ATTN_STAMP(6);
    __builtin_amdgcn_s_barrier();                      // 3: nobody reads the V image any more
    ATTN_STAMP(7);
    issue_kv(row0_n, head_n, 1);
    p = pn; row0 = row0_n; head = head_n;
  }
}

// ------------------------------------------------------------------------------------------------
// Temporal axis of clips LONGER than 256 frames (reference common/arguments.py:58 `-f`; round 6).  The
// persistent kernel above holds the K and V images of a whole sequence in LDS (256 keys = 128 KiB) and forms a query's softmax
// over the complete score row; beyond 256 frames round 5 sent both attentions to the chunked fp32 VALU row kernel -- measured at
// F = 351: 53 % of the sampler's time, ten times the cost per FLOP of the 243-frame path.  Here: the flash form on the same
// split-fp16 operands and packed qkv rows.  A work unit = one (sequence, head) problem x one group of eight 16-query tiles (one
// per wave); the keys pass through LDS in chunks of 128 (K and V hi / lo images in the one-swizzle layout of ta_common.h: 64 KiB,
// two workgroups per CU) under an online softmax in base-2 units -- running row maximum, denominator and O^T per query in
// registers, rescaled whenever a key pair raises the maximum -- exactly the forward kernel of the training step
// (train_attn.hip tattn_fwd_kernel) with the key range cut into chunks and its operands taken from the packed rows: K and V are
// ALREADY split (the qkv Linear's epilogue), so staging is sixteen-byte copies, no arithmetic.  Any length up to the library's
// 1024 frames; results differ from the two-pass kernel only in the rounding of the rescaled partial sums (fp32-class either way).
template <int OUTS>
__global__ __launch_bounds__(512) void attn_temporal_x2_long_kernel(const float* __restrict__ qkv, void* __restrict__ out_v,
                                                                    SeqMap map, int C, int heads, int groups, int n_work,
                                                                    X2Scales sc) {
  constexpr int NW = 8, NKC = 128, NKT = NKC / 16, PLANE = NKC * 128;
  __shared__ __attribute__((aligned(16))) char kimg[2 * PLANE];
  __shared__ __attribute__((aligned(16))) char vimg[2 * PLANE];
  const int n = map.n_tok;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, fg = lane >> 4;
  const size_t ldb = (size_t)12 * C, rsb = (size_t)map.tok_stride * ldb;   // bytes per packed token row / between two tokens of a sequence
  using u32x4 = unsigned __attribute__((ext_vector_type(4)));
  // rows k0 .. k0 + NKC - 1 of one packed operand (byte offset `off` inside a row: hi plane; lo plane 2 C bytes further) -> its
  // hi / lo images; keys >= n: zero rows (their scores are masked, their probabilities 0: V must be finite there)
  auto stage = [&](const char* row0, int off, int k0, char* img) {
#pragma unroll
    for (int i0 = 0; i0 < NKC * 8; i0 += 2 * 512) {
      u32x4 h[2], l[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = i0 + u * 512 + tid, row = idx >> 3, slot = idx & 7;
        h[u] = (u32x4){0u, 0u, 0u, 0u}; l[u] = h[u];
        if (k0 + row < n) {
          const char* p = row0 + (size_t)(k0 + row) * rsb + off + slot * 16;
          h[u] = *reinterpret_cast<const u32x4*>(p);
          l[u] = *reinterpret_cast<const u32x4*>(p + 2 * C);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = i0 + u * 512 + tid, row = idx >> 3, slot = idx & 7;
        const int o = row * 128 + ((slot ^ ta_sw(row)) << 4);
        *reinterpret_cast<u32x4*>(img + o) = h[u];
        *reinterpret_cast<u32x4*>(img + PLANE + o) = l[u];
      }
    }
  };
  const TAFrag fk = ta_frag(kimg, lane), fv = ta_frag(vimg, lane);
  for (int unit = blockIdx.x; unit < n_work; unit += gridDim.x) {
    const int prob = unit / groups, group = unit - prob * groups;
    const int seq = prob / heads, head = prob - seq * heads;
    const int tok0 = seq_base(map, seq);
    const char* row0 = reinterpret_cast<const char*>(qkv) + (size_t)tok0 * ldb;
    const int qt = group * NW + wave;
    const bool active = qt * 16 < n;                     // (wave-uniform)
    const int q = qt * 16 + fi;
    f16x8 qh[2], ql[2];
    if (active)
      ta_load_row_op(reinterpret_cast<const float*>(row0 + (size_t)min(q, n - 1) * rsb) + head * 64, fg, sc.q, qh, ql);
    float mrun = -INFINITY, lrun = 0.f;                  // running row maximum (base-2 logit units), 1024 x running denominator
    f32x4 o[4];
#pragma unroll
    for (int dn = 0; dn < 4; ++dn) o[dn] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < n; k0 += NKC) {
      __syncthreads();                                   // every wave is done with the previous chunk's (or unit's) images
      stage(row0, 4 * C + head * 128, k0, kimg);
      stage(row0, 8 * C + head * 128, k0, vimg);
      __syncthreads();
      if (!active) continue;
#pragma unroll
      for (int t = 0; t < NKT; t += 2) {
        if (k0 + 16 * t >= n) break;                     // (uniform: the rest of the chunk lies behind the sequence)
        f32x4 a, b;
        ta_rows_pair<PLANE>(fk, t, qh, ql, a, b);        // S^T [key][query], raw: x sc.cexp = logits in base-2 units
        float sv[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { sv[r] = a[r] * sc.cexp; sv[4 + r] = b[r] * sc.cexp; }
        if (k0 + 16 * (t + 2) > n) {                     // (uniform: only the sequence's last pair holds keys >= n)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (k0 + 16 * t + 4 * fg + r >= n) sv[r] = -INFINITY;
            if (k0 + 16 * (t + 1) + 4 * fg + r >= n) sv[4 + r] = -INFINITY;
          }
        }
        float mx = sv[0];
#pragma unroll
        for (int e = 1; e < 8; ++e) mx = fmaxf(mx, sv[e]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx);              // (finite from the first pair on: key 0 exists)
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
        mrun = mnew;
        float psum = 0.f;
        f16x8 ph, pl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float y = __builtin_amdgcn_exp2f(sv[e] - mnew + 10.0f);   // p x 1024
          psum += y;
          const f16 hh = (f16)y;
          ph[e] = hh;
          pl[e] = (f16)fmaf(y, 1.0f, -(float)hh);
        }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        lrun = fmaf(lrun, alpha, psum);
#pragma unroll
        for (int dn = 0; dn < 4; ++dn) o[dn] *= alpha;
        ta_tr_chunk<PLANE>(fv, t >> 1, ph, pl, o);       // O^T[d][query] += V^T P^T
      }
    }
    // o = (v scale) x 1024 x sum_j p_j v_j against the running maximum; lrun = 1024 x sum_j p_j
    if (active && q < n)
      store_o_x2<OUTS>(o, sc.onorm / lrun, out_v, (size_t)(tok0 + q * map.tok_stride), C, head * 64 + fg * 4, sc.oplane);
  }
}

// ------------------------------------------------------------------------------------------------
// -DD3DP_ATTN_W16=1 (measurement build; not part of the default library): the temporal split-fp16 kernel as SIXTEEN waves
// of at most 128 registers, one 16-query tile each, the score row in two halves of 128 keys with an online softmax --
// the form VERDICT r3 item 3 named beside progressive DMA.  Four waves per SIMD instead of two cover each other's LDS and
// MFMA latencies (profiles/r04_attn_timeline.md: that, not memory and not the vector-instruction count, is what the
// two-wave form lacks); the price is the order S0, P0.V, S1, P1.V: K(p+1) can only stream in under P1.V and V(p+1) under
// the next problem's S0, every wave reads the whole K / V image for one tile instead of two, and the probabilities of
// the first half are formed against the first half's row maximum and rescaled (one more rounding on that half of O and
// of the denominator: not bit-identical to the two-pass kernel).  256 keys (241..256 frames) only.
#ifndef D3DP_ATTN_W16
#define D3DP_ATTN_W16 0
#endif
#if D3DP_ATTN_W16
// S^T of one query tile against HT key tiles, ONE key tile (four fragments, one accumulator chain of six MFMAs) at a time and
// no register prefetch: the other three waves of the SIMD cover the fragment reads and the chain's latency, and 128
// registers have room for neither a second tile's fragments nor a prefetch
template <int HT>
__device__ __forceinline__ void x2_scores_lean(const FragBases& fb, int plane, const f16x8 (&qh)[2], const f16x8 (&ql)[2],
                                               f32x4 (&s)[HT]) {
#pragma unroll
  for (int t = 0; t < HT; ++t) {
    const f16x8 kl0 = *reinterpret_cast<const f16x8*>(fb.k0 + plane + t * 2048);   // lo, d 0..31
    const f16x8 kl1 = *reinterpret_cast<const f16x8*>(fb.k1 + plane + t * 2048);   // lo, d 32..63
    const f16x8 kh0 = *reinterpret_cast<const f16x8*>(fb.k0 + t * 2048);           // hi
    const f16x8 kh1 = *reinterpret_cast<const f16x8*>(fb.k1 + t * 2048);
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl0, qh[0], a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl1, qh[1], a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh0, ql[0], a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh1, ql[1], a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh0, qh[0], a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh1, qh[1], a, 0, 0, 0);
    s[t] = a;
    if (t & 1) __builtin_amdgcn_sched_barrier(0);
  }
}
template <int OUTS>
__global__ __launch_bounds__(1024) void attn_temporal_x2_w16_kernel(const float* __restrict__ qkv, void* __restrict__ out_v,
                                                                    SeqMap map, int C, int heads, size_t plane_elems,
                                                                    int n_prob, X2Scales sc) {
  constexpr int NKT = 16, NW = 16, NK = 256, PLANE = NK * 128, HT = NKT / 2;   // HT: key tiles per half
  __shared__ __attribute__((aligned(16))) char kimg[2 * PLANE];
  __shared__ __attribute__((aligned(16))) char vimg[2 * PLANE];
  const int n = map.n_tok;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ts = map.tok_stride;
  const size_t ldb = (size_t)12 * C;
  const float inv_scale = sc.onorm;
  auto opaque = [](int x) { asm volatile("" : "+v"(x)); return x; };
  auto problem_row0 = [&](int p, int& head) -> const char* {
    const int seq = p / heads;
    head = p - seq * heads;
    return reinterpret_cast<const char*>(qkv) + (size_t)seq_base(map, seq) * ldb;
  };
  auto issue_kv = [&](const char* row0, int head, int is_v) {   // pieces wave and wave + 16 of both planes: 4 LDS-DMA
    const int l = opaque(lane);
    const char* g = row0 + (is_v ? 8 : 4) * C + head * 128;
    char* img = is_v ? vimg : kimg;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pc = wave + j * NW;
      const int row = pc * 8 + (l >> 3);
      const int sw = is_v ? (((row >> 1) & 3) << 1) : ((row >> 1) & 7);
      const char* gj = g + (unsigned)(min(row, n - 1) * ts) * (unsigned)ldb + (((l & 7) ^ sw) << 4);
      lds_dma16(gj, img + pc * 1024);
      lds_dma16(gj + 2 * C, img + PLANE + pc * 1024);
    }
  };
  // The queries by ORDINARY loads (a spill next to a load the compiler does not track would save stale data, and 128 registers
  // are spill territory): requested behind K(p+1), raw in 16 registers through the last P.V, split before the output stores.
  // The compiler waits for them with vmcnt(0), which also covers the older K(p+1) pieces -- landed long before -- and
  // nothing younger: V(p+1) is issued after that point.
  auto load_q = [&](const char* row0, int head, float4 (&qv)[4]) {
    const int l = opaque(lane);
    const int q = wave * 16 + (l & 15);
    const float* qrow = reinterpret_cast<const float*>(row0 + (size_t)min(q, n - 1) * ts * ldb) + head * 64 + (l >> 4) * 8;
    qv[0] = *reinterpret_cast<const float4*>(qrow);
    qv[1] = *reinterpret_cast<const float4*>(qrow + 4);
    qv[2] = *reinterpret_cast<const float4*>(qrow + 32);
    qv[3] = *reinterpret_cast<const float4*>(qrow + 36);
  };
  auto rowmax8 = [&](const f32x4 (&s)[HT]) {
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    return fmaxf(mx, __shfl_xor(mx, 32, 64));
  };

  int p = blockIdx.x;
  if (p >= n_prob) return;
  int head;
  const char* row0 = problem_row0(p, head);
  issue_kv(row0, head, 0);
  f16x8 qh[2], ql[2];
  {
    float4 qv[4];
    load_q(row0, head, qv);
    split8(qv[0], qv[1], qh[0], ql[0], sc.q);
    split8(qv[2], qv[3], qh[1], ql[1], sc.q);
  }
  issue_kv(row0, head, 1);
  for (;;) {
    wait_vmcnt<4>();                                   // K(p) landed; the four V pieces may be in flight
    __builtin_amdgcn_s_barrier();                      // 1: K image complete
    f32x4 s[HT];
    f16x8 ph[HT / 2], pl[HT / 2];
    f32x4 o[4];
#pragma unroll
    for (int dn = 0; dn < 4; ++dn) o[dn] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // ---- first half: keys 0..127   (fragment bases re-derived per phase from the opaque lane id: none stays live across phases)
    {
      const FragBases fb = make_frag_bases(kimg, vimg, opaque(lane));
      x2_scores_lean<HT>(fb, PLANE, qh, ql, s);
    }
    const float m0 = rowmax8(s);
    float lsum[2] = {0.f, 0.f};
    {
      const float nb = fmaf(-m0, sc.cexp, 10.0f);
#pragma unroll
      for (int t = 0; t < HT; t += 2) x2_softmax_slice(s[t], s[t + 1], nb, sc.cexp, ph[t >> 1], pl[t >> 1], lsum);
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                      // 2: V image complete
    {
      const FragBases fb = make_frag_bases(kimg, vimg, opaque(lane));
      pv_chunk_x2_seq<0>(fb, PLANE, ph[0], pl[0], o);
      pv_chunk_x2_seq<1>(fb, PLANE, ph[1], pl[1], o);
      pv_chunk_x2_seq<2>(fb, PLANE, ph[2], pl[2], o);
      pv_chunk_x2_seq<3>(fb, PLANE, ph[3], pl[3], o);
    }
    // ---- second half: keys 128..255 (the tail of the last tile masked)
    {
      const FragBases fb = make_frag_bases(kimg + HT * 2048, vimg, opaque(lane));
      x2_scores_lean<HT>(fb, PLANE, qh, ql, s);
    }
    {
      const int fg = lane >> 4;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * (NKT - 1) + 4 * fg + r >= n) s[HT - 1][r] = -INFINITY;
    }
    const float m = fmaxf(m0, rowmax8(s));
    __builtin_amdgcn_s_barrier();                      // 3: nobody reads the K image any more
    const int pn = p + gridDim.x;
    const bool has_next = pn < n_prob;
    int head_n = 0;
    const char* row0_n = row0;
    float4 qn[4];
    if (has_next) {
      row0_n = problem_row0(pn, head_n);
      issue_kv(row0_n, head_n, 0);
      load_q(row0_n, head_n, qn);
    }
    {
      const float resc = __builtin_amdgcn_exp2f((m0 - m) * sc.cexp);       // <= 1; exactly 1 where the first half holds the maximum
#pragma unroll
      for (int dn = 0; dn < 4; ++dn) o[dn] *= resc;
      lsum[0] *= resc; lsum[1] *= resc;
      const float nb = fmaf(-m, sc.cexp, 10.0f);
#pragma unroll
      for (int t = 0; t < HT; t += 2) x2_softmax_slice(s[t], s[t + 1], nb, sc.cexp, ph[t >> 1], pl[t >> 1], lsum);
    }
    {
      const FragBases fb = make_frag_bases(kimg, vimg + 4 * 4096, opaque(lane));
      pv_chunk_x2_seq<0>(fb, PLANE, ph[0], pl[0], o);
      pv_chunk_x2_seq<1>(fb, PLANE, ph[1], pl[1], o);
      pv_chunk_x2_seq<2>(fb, PLANE, ph[2], pl[2], o);
      pv_chunk_x2_seq<3>(fb, PLANE, ph[3], pl[3], o);
    }
    if (has_next) {                                    // (the current queries are dead: the next problem's take their registers)
      split8(qn[0], qn[1], qh[0], ql[0], sc.q);
      split8(qn[2], qn[3], qh[1], ql[1], sc.q);
    }
    {
      const float denom = x2_denominator(lsum);
      const int tok0 = seq_base(map, p / heads);
      const int l = opaque(lane);
      const int q = wave * 16 + (l & 15);
      if (q < n)
        store_o_x2<OUTS>(o, inv_scale / denom, out_v, (size_t)(tok0 + q * ts), C, head * 64 + (l >> 4) * 4, sc.oplane);
    }
    if (!has_next) break;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // 4: nobody reads the V image any more
    issue_kv(row0_n, head_n, 1);
    p = pn; row0 = row0_n; head = head_n;
  }
}
#endif

// spatial axis (<= 32 tokens per sequence): one WAVE per (sequence, head).  The K fragments of its two 16-key tiles are
// loaded straight from the packed rows into the MFMA operand layout (lane (key, g) <- 2 x 16 contiguous bytes of each K
// plane); only V goes through LDS (its fragments are transposed reads): a private 8 KiB image (V hi, V lo: 32 rows x
// 128 B each), so a 256-thread workgroup needs 32 KiB and five of them fit a CU -- this kernel is latency / HBM-bound
// (8 KB per token), and with K staged as well (16 KiB per wave, two workgroups per CU) it ran at 3.8 TB/s.
template <int OUTS>
__global__ __launch_bounds__(256) void attn_spatial_x2_kernel(const float* __restrict__ qkv, void* __restrict__ out_v,
                                                              int n_prob, SeqMap map, int C, int heads, size_t plane_elems,
                                                              X2Scales sc) {
  constexpr int PLANE = 32 * 128;
  __shared__ __attribute__((aligned(16))) char smem[4 * 2 * PLANE];
  const int n = map.n_tok;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pid = blockIdx.x * 4 + wave;
  if (pid >= n_prob) return;
  const int seq = pid / heads, head = pid % heads;
  const int base = seq_base(map, seq);
  const int ts = map.tok_stride;
  const size_t ldb = (size_t)12 * C;                   // bytes per packed token row: q fp32 | k hi | k lo | v hi | v lo
  const char* rows = reinterpret_cast<const char*>(qkv) + (size_t)base * ldb;
  char* img = smem + wave * 2 * PLANE;
  const int fi = lane & 15, fg = lane >> 4;
  const int n_qt = (n + 15) >> 4;
  // every global request of the problem is issued before anything is consumed: both query tiles, both key tiles, V
  f16x8 qh[2][2], ql[2][2], kh[2][2], kl[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const char* row = rows + (size_t)min(t * 16 + fi, n - 1) * ts * ldb;
    load_q_x2(reinterpret_cast<const float*>(row) + head * 64, fg, qh[t], ql[t], sc.q);
    load_k_x2(reinterpret_cast<const f16*>(row + 4 * C) + head * 64, C, fg, kh[t], kl[t]);
  }
  for (int idx = lane; idx < 32 * 8; idx += 64) {      // V rows -> hi / lo images (rows >= n zeroed)
    const int row = idx >> 3, slot = idx & 7;
    f16x8 vh = {}, vl = {};
    if (row < n) {
      const f16* src = reinterpret_cast<const f16*>(rows + (size_t)row * ts * ldb + 8 * C) + head * 64 + slot * 8;
      vh = *reinterpret_cast<const f16x8*>(src);
      vl = *reinterpret_cast<const f16x8*>(src + C);
    }
    const int vo = row * 128 + ((slot ^ (((row >> 1) & 3) << 1)) << 4);
    *reinterpret_cast<f16x8*>(img + vo) = vh;
    *reinterpret_cast<f16x8*>(img + PLANE + vo) = vl;
  }
  const FragBases fb = make_frag_bases(img, img, lane);   // only the V bases are used
  // (wave-private LDS image: the LDS pipe executes one wave's accesses in order, no barrier needed)
  const float inv_scale = sc.onorm;                    // O^T is scaled by (v scale) x 1024, `sum` by 1024
  const float cexp = sc.cexp;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    if (qt >= n_qt) break;
    const int q = qt * 16 + fi;
    f32x4 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {                      // S^T tile t: keys 16 t + 4 fg + r, query fi
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[t][0], qh[qt][0], a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[t][1], qh[qt][1], a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[t][0], ql[qt][0], a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[t][1], ql[qt][1], a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[t][0], qh[qt][0], a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[t][1], qh[qt][1], a, 0, 0, 0);
      s[t] = a;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (16 * t + 4 * fg + r >= n) s[t][r] = -INFINITY;
        mx = fmaxf(mx, s[t][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float nb = fmaf(-mx, cexp, 10.0f);          // p * 1024 = exp2(s cexp - mx cexp + 10), as attn_scores_x2
    float sum = 0.f;
    f16x8 ph[1], pl[1];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float y = __builtin_amdgcn_exp2f(fmaf(s[t][r], cexp, nb));
        sum += y;
        const f16 h = (f16)y;
        ph[0][t * 4 + r] = h;
        pl[0][t * 4 + r] = (f16)fmaf(y, 1.0f, -(float)h);
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    f32x4 o[4];
#pragma unroll
    for (int dn = 0; dn < 4; ++dn) o[dn] = (f32x4){0.f, 0.f, 0.f, 0.f};
    pv_chunk_x2_seq<0>(fb, PLANE, ph[0], pl[0], o);
    if (q < n) store_o_x2<OUTS>(o, inv_scale / sum, out_v, (size_t)(base + q * ts), C, head * 64 + fg * 4, sc.oplane);
  }
}

// fp32 qkv rows [T, 3C] -> the packed rows of the split-fp16 attention kernels (what the qkv Linear writes directly with
// EPI_QKV_PACK); used by d3dp_op_attention, whose C-ABI input is the plain fp32 layout.
__global__ void qkv_pack_x2_kernel(const float* __restrict__ src, char* __restrict__ dst, size_t T, int C, float sc) {
  const size_t total = T * (size_t)(3 * C / 4);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t t = i / (3 * C / 4);
    const int c = (int)(i - t * (3 * C / 4)) * 4, region = c / C, cn = c - region * C;
    const float4 v = *reinterpret_cast<const float4*>(src + t * 3 * C + c);
    char* row = dst + t * 12 * C;
    if (region == 0) {
      *reinterpret_cast<float4*>(row + cn * 4) = v;
    } else {
      const float a[4] = {v.x, v.y, v.z, v.w};
      f16x4 ph, pl;
#pragma unroll
      for (int e = 0; e < 4; ++e) { f16 h, l; split2h_scaled(a[e] * sc, h, l); ph[e] = h; pl[e] = l; }
      *reinterpret_cast<f16x4*>(row + region * 4 * C + cn * 2) = ph;
      *reinterpret_cast<f16x4*>(row + region * 4 * C + 2 * C + cn * 2) = pl;
    }
  }
}

template <int NKT, int OUTS>
int launch_temporal_x2(const void* qkv, void* out, int n_seq, SeqMap map, int C, int heads, size_t plane, X2Scales sc,
                       hipStream_t st) {
  constexpr int NK = 16 * NKT;
  const size_t lds = (size_t)NK * 128 * 4;             // static LDS of the kernel (K and V images, two planes each)
  constexpr int NW = 8;
  // (n inside the last key tile -- F = 243, 27 -- needs masking in that tile only)
  auto kern = map.n_tok > 16 * (NKT - 1) ? attn_temporal_x2_kernel<NKT, OUTS, false> : attn_temporal_x2_kernel<NKT, OUTS, true>;
  static PerDeviceOnce once;
  const int n_wg = once.get([&](int dev) {
    const int per_cu = (int)((160 * 1024) / lds) < 2048 / (NW * 64) ? (int)((160 * 1024) / lds) : 2048 / (NW * 64);
    const int cus = d3dp_cu_count(dev);
    return cus < 0 ? cus : cus * per_cu;               // persistent: as many workgroups as fit the chip
  });
  if (n_wg < 0) return -3;
  const int n_prob = n_seq * heads;
#if D3DP_ATTN_W16
  if constexpr (NKT == 16) {
    if (map.n_tok > 16 * (NKT - 1)) {
      hipLaunchKernelGGL((attn_temporal_x2_w16_kernel<OUTS>), dim3(n_prob < n_wg ? n_prob : n_wg), dim3(1024), 0, st,
                         (const float*)qkv, out, map, C, heads, plane, n_prob, sc);
      return 0;
    }
  }
#endif
  hipLaunchKernelGGL(kern, dim3(n_prob < n_wg ? n_prob : n_wg), dim3(NW * 64), 0, st, (const float*)qkv, out, map, C,
                     heads, plane, n_prob, sc);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// EXACT-mode temporal attention on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: bitwise an fp32 fmaf chain).
// Same dataflow as the bf16 kernel -- S^T = K Q^T so the probabilities are already the B operand of O^T = V^T P^T --
// with fp32 K/V images in LDS (row strides 65 / 68 floats: conflict-free ds_read_b32 fragments), the whole score
// row-block in registers, two-pass fp32 softmax.  Output: fp32, or three split-bf16 planes (OUT3) for the bf16x3
// Linear that follows.
// ------------------------------------------------------------------------------------------------
constexpr int LDKF = 65, LDVF = 68;

// A workgroup = eight waves = eight consecutive query tiles of one (sequence, head) problem; grid = problems x ceil(tiles / 8):
// 544 problems of 16 tiles (the configs[4] training batch) quantise to three rounds of one 139-KiB workgroup per CU, 1,088
// half problems to 4.25 half rounds, and two waves per SIMD cover each other's LDS waits (four waves walking four tiles each:
// 178 us per launch).
constexpr int TF32_WAVES = 8;
template <int NKT, int OUTS>
__global__ __launch_bounds__(TF32_WAVES * 64) void attn_temporal_f32_kernel(const float* __restrict__ qkv, void* __restrict__ out_v,
                                                                         SeqMap map, int C, int heads, size_t plane, int groups) {
  constexpr int NK = 16 * NKT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* KS = reinterpret_cast<float*>(smem);
  float* VS = KS + NK * LDKF + 3;          // keep V rows 16-byte aligned (NK*65 + 3 is a multiple of 4 for NK % 16 == 0)
  static_assert((16 * LDKF + 0) % 1 == 0, "");
  const int n = map.n_tok;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int prob = blockIdx.x / groups, group = blockIdx.x % groups;
  const int seq = prob / heads, head = prob % heads;
  const int base = seq_base(map, seq);
  const int ts = map.tok_stride;
  const size_t ld = (size_t)3 * C;
  const float* qbase = qkv + (size_t)base * ld + (size_t)head * 64;
  const int fi = lane & 15, fg = lane >> 4;
  const int n_qt = (n + 15) >> 4;

  // stage K (scalar stores, stride 65) and V (float4 stores, stride 68); zero the padding rows of V
  for (int idx = tid; idx < NK * 16; idx += TF32_WAVES * 64) {
    const int row = idx >> 4, c4 = (idx & 15) * 4;
    float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
    if (row < n) {
      const float* src = qbase + (size_t)row * ts * ld + c4;
      kv = *reinterpret_cast<const float4*>(src + C);
      vv = *reinterpret_cast<const float4*>(src + 2 * C);
    }
    float* kd = KS + row * LDKF + c4;
    kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
    *reinterpret_cast<float4*>(VS + row * LDVF + c4) = vv;
  }
  __syncthreads();
  const float cexp = 0.125f * 1.44269504088896340736f;
  for (int qt = group * TF32_WAVES + wave; qt < n_qt; qt = n_qt) {   // (one tile per wave)
    const int q = qt * 16 + fi;
    // contraction index of MFMA step kk for lane group g is d = 16 g + kk (K fragments use the same map)
    const float* qsrc = qbase + (size_t)min(q, n - 1) * ts * ld + fg * 16;
    float qv[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 a = *reinterpret_cast<const float4*>(qsrc + c * 4);
      qv[c * 4] = a.x; qv[c * 4 + 1] = a.y; qv[c * 4 + 2] = a.z; qv[c * 4 + 3] = a.w;
    }
    f32x4 s[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      const float* kr = KS + (t * 16 + fi) * LDKF + fg * 16;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) a = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[kk], qv[kk], a, 0, 0, 0);
      s[t] = a;                                   // S^T[key = 16t + 4 fg + r][query fi]
      __builtin_amdgcn_sched_barrier(0);          // bound the ds_read hoisting window (VGPR pressure)
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      if (16 * (t + 1) > n) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * t + 4 * fg + r >= n) s[t][r] = -INFINITY;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[t][r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // exp(x) with x = (s - max) / 8 in natural units: exp2f keeps one rounding of the scaled argument
        s[t][r] = exp2f((s[t][r] - mx) * cexp);
        sum += s[t][r];
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    f32x4 o[4];
#pragma unroll
    for (int dn = 0; dn < 4; ++dn) o[dn] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* vr = VS + (t * 16 + 4 * fg + r) * LDVF + fi;   // V[key][dn*16 + fi]
#pragma unroll
        for (int dn = 0; dn < 4; ++dn) o[dn] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[dn * 16], s[t][r], o[dn], 0, 0, 0);
        if (r == 3) __builtin_amdgcn_sched_barrier(0);
      }
    if (q < n) {
      const float inv = 1.0f / sum;
      const size_t off = (size_t)(base + q * ts) * C + head * 64 + fg * 4;
#pragma unroll
      for (int dn = 0; dn < 4; ++dn) {
        float r4[4] = {o[dn][0] * inv, o[dn][1] * inv, o[dn][2] * inv, o[dn][3] * inv};
        if constexpr (OUTS == 2) {                      // h2i row (common.h)
          f16* dst = reinterpret_cast<f16*>(out_v) + (size_t)(base + q * ts) * (2 * C) + h2i_col(head * 64 + fg * 4 + dn * 16);
          f16x4 p0, p1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f16 a0, a1;
            split2h(r4[e], a0, a1);
            p0[e] = a0; p1[e] = a1;
          }
          *reinterpret_cast<f16x4*>(dst) = p0;
          *reinterpret_cast<f16x4*>(dst + kH2iLo) = p1;
        } else if constexpr (OUTS == 3) {
          bf16* dst = reinterpret_cast<bf16*>(out_v) + off + dn * 16;
          bf16x4 p0, p1, p2;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            bf16 a0, a1, a2;
            split3(r4[e], a0, a1, a2);
            p0[e] = a0; p1[e] = a1; p2[e] = a2;
          }
          *reinterpret_cast<bf16x4*>(dst) = p0;
          *reinterpret_cast<bf16x4*>(dst + plane) = p1;
          *reinterpret_cast<bf16x4*>(dst + 2 * plane) = p2;
        } else {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(out_v) + off + dn * 16) = make_float4(r4[0], r4[1], r4[2], r4[3]);
        }
      }
    }
  }
}

template <int NKT, int OUTS>
int launch_temporal_f32(const void* qkv, void* out, int n_seq, SeqMap map, int C, int heads, size_t plane, hipStream_t st) {
  constexpr int NK = 16 * NKT;
  const size_t lds = (size_t)(NK * LDKF + 3 + NK * LDVF) * 4 + 16;
  auto kern = attn_temporal_f32_kernel<NKT, OUTS>;
  static PerDeviceOnce once;                          // (one per template instantiation = per kernel)
  if (once.get([&](int) { return d3dp_lds_opt_in(reinterpret_cast<const void*>(kern), 160 * 1024); }) < 0) return -3;
  const int groups = ((map.n_tok + 15) / 16 + TF32_WAVES - 1) / TF32_WAVES;
  hipLaunchKernelGGL(kern, dim3(n_seq * heads * groups), dim3(TF32_WAVES * 64), lds, st, (const float*)qkv, out, map, C, heads, plane,
                     groups);
  return 0;
}

template <int NKT>
int launch_temporal2(const void* qkv, void* out, int n_seq, SeqMap map, int C, int heads, hipStream_t st) {
  constexpr int NK = 16 * NKT;
  const size_t lds = (size_t)NK * 256;
  auto kern = attn_temporal2_bf16_kernel<NKT>;
  static PerDeviceOnce once;                          // (one per template instantiation = per kernel)
  if (once.get([&](int) { return d3dp_lds_opt_in(reinterpret_cast<const void*>(kern), 160 * 1024); }) < 0) return -3;
  hipLaunchKernelGGL(kern, dim3(n_seq * heads), dim3(512), lds, st, (const bf16*)qkv, (bf16*)out, map, C, heads);
  return 0;
}

}  // namespace

// act: 0 = fp32 in/out, 1 = bf16 in/out, 2 = fp32 in, split-bf16 planes out, 3 = fp32 in, split-fp16 planes out
// (amax: fp32 output only -- its absmax, see the kernel)
int d3dp_launch_attn_rows(int act, const void* qkv, void* out, int n_seq, SeqMap map, int C, int heads, hipStream_t st,
                          unsigned* amax) {
  const size_t plane = (size_t)n_seq * map.n_tok * C;
  if (act != 0 && amax) return -1;
  if (act == 1) return dispatch_rows<bf16, 0>(qkv, out, n_seq, map, C, heads, plane, nullptr, st);
  if (act == 2) return dispatch_rows<float, 3>(qkv, out, n_seq, map, C, heads, plane, nullptr, st);
  if (act == 3) return dispatch_rows<float, 2>(qkv, out, n_seq, map, C, heads, plane, nullptr, st);
  return dispatch_rows<float, 0>(qkv, out, n_seq, map, C, heads, plane, amax, st);
}

int d3dp_launch_attn_temporal_bf16(const void* qkv, void* out, int n_seq, SeqMap map, int C, int heads,
                                   hipStream_t st) {
  if (C / heads != 64 || map.n_tok > 256 || map.n_tok < 1) return -2;
  const int n = map.n_tok;
  if (n <= 32) return launch_temporal2<2>(qkv, out, n_seq, map, C, heads, st);
  if (n <= 64) return launch_temporal2<4>(qkv, out, n_seq, map, C, heads, st);
  if (n <= 128) return launch_temporal2<8>(qkv, out, n_seq, map, C, heads, st);
  return launch_temporal2<16>(qkv, out, n_seq, map, C, heads, st);
}

// EXACT-mode temporal axis on the fp32 matrix cores (head dim 64); act 0 -> fp32 out, 2 -> split-bf16 planes out,
// 3 -> split-fp16 planes out
int d3dp_launch_attn_temporal_f32(int act, const void* qkv, void* out, int n_seq, SeqMap map, int C, int heads,
                                  hipStream_t st) {
  if (C / heads != 64 || map.n_tok > 256 || map.n_tok < 1 || (act != 0 && act != 2 && act != 3)) return -2;
  const size_t plane = (size_t)n_seq * map.n_tok * C;
  const int n = map.n_tok;
#define TF32_CASE(NKT_)                                                                                       \
  return act == 3 ? launch_temporal_f32<NKT_, 2>(qkv, out, n_seq, map, C, heads, plane, st)                   \
       : act == 2 ? launch_temporal_f32<NKT_, 3>(qkv, out, n_seq, map, C, heads, plane, st)                   \
                  : launch_temporal_f32<NKT_, 0>(qkv, out, n_seq, map, C, heads, plane, st);
  if (n <= 32) { TF32_CASE(2) }
  if (n <= 64) { TF32_CASE(4) }
  if (n <= 128) { TF32_CASE(8) }
  TF32_CASE(16)
#undef TF32_CASE
}

// EXACT-mode attention on the fp16 matrix cores (split-fp16 operands; head dim 64) over PACKED qkv rows (see above).
// act 0 -> fp32 out, 3 -> two fp16 planes out (the EXACT Linear's operand format).  axis 0 with <= 32 tokens per sequence:
// one wave per problem; otherwise the persistent whole-sequence kernel (<= 256 tokens) or the chunked-key form (longer).
void d3dp_launch_qkv_pack_x2(const float* src, void* dst, size_t T, int C, float act_scale, hipStream_t st) {
  const size_t total = T * (size_t)(3 * C / 4);
  const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(qkv_pack_x2_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, (char*)dst, T, C, act_scale);
}

// act_scale: the power of two q, k, v (and, for act == 3, the output planes) are scaled by -- kActScale unless the block's
// proven operand range asked for less (capi.hip).  With 16 the arithmetic is bit for bit that of the constant-scale kernels.
int d3dp_launch_attn_x2(int act, int axis, const void* qkv, void* out, int n_seq, SeqMap map, int C, int heads,
                        float act_scale, hipStream_t st) {
  if (C / heads != 64 || map.n_tok < 1 || (act != 0 && act != 3) || !(act_scale > 0.f)) return -2;
  const size_t plane = (size_t)n_seq * map.n_tok * C;
  const int n = map.n_tok;
  const X2Scales sc = {act_scale, 0.125f * 1.44269504088896340736f / (act_scale * act_scale),
                       1.0f / act_scale, act_scale};
  if (axis == 0 && n <= 32) {                          // (more than 32 joints: the kernels below take any SeqMap)
    const int n_prob = n_seq * heads;
    if (act == 3) hipLaunchKernelGGL((attn_spatial_x2_kernel<2>), dim3((n_prob + 3) / 4), dim3(256), 0, st, (const float*)qkv, out, n_prob, map, C, heads, plane, sc);
    else hipLaunchKernelGGL((attn_spatial_x2_kernel<0>), dim3((n_prob + 3) / 4), dim3(256), 0, st, (const float*)qkv, out, n_prob, map, C, heads, plane, sc);
    return 0;
  }
  if (n > 256) {                                       // long clips: keys in chunks of 128 under an online softmax (see the kernel)
    const int groups = ((n + 15) / 16 + 7) / 8, n_work = n_seq * heads * groups;
    static PerDeviceOnce once;
    const int cus = once.get([&](int dev) { return d3dp_cu_count(dev); });
    if (cus < 0) return -3;
    const dim3 grid(n_work < 2 * cus ? n_work : 2 * cus), blk(512);
    if (act == 3) hipLaunchKernelGGL((attn_temporal_x2_long_kernel<2>), grid, blk, 0, st, (const float*)qkv, out, map, C, heads, groups, n_work, sc);
    else hipLaunchKernelGGL((attn_temporal_x2_long_kernel<0>), grid, blk, 0, st, (const float*)qkv, out, map, C, heads, groups, n_work, sc);
    return 0;
  }
#define X2_CASE(NKT_)                                                                                         \
  return act == 3 ? launch_temporal_x2<NKT_, 2>(qkv, out, n_seq, map, C, heads, plane, sc, st)                \
                  : launch_temporal_x2<NKT_, 0>(qkv, out, n_seq, map, C, heads, plane, sc, st);
  if (n <= 32) { X2_CASE(2) }
  if (n <= 64) { X2_CASE(4) }
  if (n <= 128) { X2_CASE(8) }
  X2_CASE(16)
#undef X2_CASE
}

// spatial axis on MFMA (bf16, head dim 64, <= 32 tokens per sequence)
int d3dp_launch_attn_spatial_bf16(const void* qkv, void* out, int n_seq, SeqMap map, int C, int heads, hipStream_t st) {
  if (C / heads != 64 || map.n_tok > 32 || map.n_tok < 1) return -2;
  const int n_prob = n_seq * heads;
  hipLaunchKernelGGL(attn_spatial_bf16_kernel, dim3((n_prob + 3) / 4), dim3(256), 0, st, (const bf16*)qkv, (bf16*)out,
                     n_prob, map, C, heads);
  return 0;
}

#if D3DP_ATTN_STAMP
extern "C" __attribute__((visibility("default"))) int d3dp_debug_attn_stamps(unsigned long long* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(d3dp_attn_stamps), sizeof(unsigned long long) * 4 * 8 * 16 * 8);
}
#endif
