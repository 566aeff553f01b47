// Device helpers shared by the split-fp16 attention kernels that keep their K / V (Q / dO) operands as hi / lo IMAGES in LDS under
// ONE swizzle serving row reads and transposed reads alike: the training step's attention (train_attn.hip) and the long-clip
// temporal attention of inference (attention.hip, attn_temporal_x2_long_kernel: more than 256 frames).
// LDS image of a [n][64] matrix: two planes (hi | lo) of 128-byte rows, 16-byte slot s of row r at s ^ (((r >> 1) & 3) << 1) --
// conflict-free for the transposed fragment reads (ds_read_b64_tr_b16: 4 rows x 32 B per 16 lanes) AND for the row fragment
// reads (ds_read_b128, whose lane groups pair rows {0-3, 12-15} of one slot with rows {4-11} of the neighbouring one).
#pragma once
#include "common.h"
#include "kernels.h"

namespace {

constexpr float kLog2e = 1.44269504088896340736f;
typedef __bf16 v4bf16_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int ta_seq_base(const SeqMap& m, int s) {
  return (s / m.inner) * m.outer_stride + (s % m.inner) * m.inner_stride;
}
__device__ __forceinline__ int ta_sw(int row) { return ((row >> 1) & 3) << 1; }

// scale of a split operand from its absmax slot (as gemm_x2.hip dyn_scale): the largest magnitude lands in [2^13, 2^14)
__device__ __forceinline__ float ta_scale(const unsigned* amax) {
  const float m = __uint_as_float(amax[0]);
  if (!(m > 0.f) || !(m < INFINITY)) return 1.0f;
  int e;
  frexpf(m, &e);
  return ldexpf(1.0f, 14 - e);
}
__device__ __forceinline__ float ta_opaque(float x) { asm volatile("" : "+v"(x)); return x; }

__device__ __forceinline__ void ta_split8(const float4 a, const float4 b, f16x8& hi, f16x8& lo, float sc) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) { f16 h, l; split2h_scaled(v[e] * sc, h, l); hi[e] = h; lo[e] = l; }
}

// rows [0, n) x 64 channels of an fp32 matrix (row stride `rs` floats) -> the hi / lo images (values x sc); rows [n, NK)
// zero.  Four 16-byte slots per thread and pass, all loads of a pass in flight before the first conversion.
template <int NK, int NT>
__device__ __forceinline__ void ta_stage(const float* __restrict__ src, size_t rs, int n, float sc, char* img, int tid) {
  constexpr int PLANE = NK * 128;
  constexpr int ITEMS = NK * 8;                        // 16-byte slots of one plane
#ifdef D3DP_TA_PROBE                                   // timing probe (results INVALID): 1 = no staging at all, 2 = loads without the split
  if (D3DP_TA_PROBE == 1 && sc != -12345.f) return;
#endif
  for (int i0 = 0; i0 < ITEMS; i0 += 4 * NT) {
    float4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = i0 + u * NT + tid, row = idx >> 3, slot = idx & 7;
      a[u] = make_float4(0.f, 0.f, 0.f, 0.f); b[u] = a[u];
      if (idx < ITEMS && row < n) {
        const float* p = src + (size_t)row * rs + slot * 8;
        a[u] = *reinterpret_cast<const float4*>(p);
        b[u] = *reinterpret_cast<const float4*>(p + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = i0 + u * NT + tid, row = idx >> 3, slot = idx & 7;
      if (idx < ITEMS) {
        f16x8 hi, lo;
#if defined(D3DP_TA_PROBE) && D3DP_TA_PROBE == 2
        hi = __builtin_bit_cast(f16x8, a[u]); lo = __builtin_bit_cast(f16x8, b[u]);
        if (sc == -12345.f)
#endif
        ta_split8(a[u], b[u], hi, lo, sc);
        const int off = row * 128 + ((slot ^ ta_sw(row)) << 4);
        *reinterpret_cast<f16x8*>(img + off) = hi;
        *reinterpret_cast<f16x8*>(img + PLANE + off) = lo;
      }
    }
  }
}

// The same for NS matrices at once: every matrix's loads of a pass are in flight before the first conversion -- ONE exposed memory round
// trip per chunk instead of one per operand (round 6: with the operands staged one after the other the loads' latency, not the
// split arithmetic, was 20 - 40 % of the training attention kernels: D3DP_TA_PROBE builds).  rows [0, n) valid for all of them.
template <int NK, int NT, int NS>
__device__ __forceinline__ void ta_stage_many(const float* const (&src)[NS], const size_t (&rs)[NS], const float (&sc)[NS],
                                              char* const (&img)[NS], int n, int tid) {
  constexpr int PLANE = NK * 128;
  constexpr int ITEMS = NK * 8;
#ifdef D3DP_TA_PROBE
  if (D3DP_TA_PROBE == 1 && sc[0] != -12345.f) return;
#endif
  for (int i0 = 0; i0 < ITEMS; i0 += 4 * NT) {
    float4 a[NS][4], b[NS][4];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = i0 + u * NT + tid, row = idx >> 3, slot = idx & 7;
        a[s][u] = make_float4(0.f, 0.f, 0.f, 0.f); b[s][u] = a[s][u];
        if (idx < ITEMS && row < n) {
          const float* p = src[s] + (size_t)row * rs[s] + slot * 8;
          a[s][u] = *reinterpret_cast<const float4*>(p);
          b[s][u] = *reinterpret_cast<const float4*>(p + 4);
        }
      }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = i0 + u * NT + tid, row = idx >> 3, slot = idx & 7;
        if (idx < ITEMS) {
          f16x8 hi, lo;
#if defined(D3DP_TA_PROBE) && D3DP_TA_PROBE == 2
          hi = __builtin_bit_cast(f16x8, a[s][u]); lo = __builtin_bit_cast(f16x8, b[s][u]);
          if (sc[0] == -12345.f)
#endif
          ta_split8(a[s][u], b[s][u], hi, lo, sc[s]);
          const int off = row * 128 + ((slot ^ ta_sw(row)) << 4);
          *reinterpret_cast<f16x8*>(img[s] + off) = hi;
          *reinterpret_cast<f16x8*>(img[s] + PLANE + off) = lo;
        }
      }
  }
}

// per-lane fragment addresses inside an image (hi plane; the lo plane is PLANE bytes further)
//   r0 / r1 : ROW fragment -- image row (16 t + lane & 15), channels 8 fg .. + 7 (r0) and 32 + 8 fg .. + 7 (r1); tile t at + t 2048
//   t[dn]   : TRANSPOSED fragment -- channel dn 16 + (lane & 15), image rows 32 c + 4 fg + {0..3} (first read) and + 16 (second,
//             2048 bytes further); chunk c at + c 4096.  (attention.hip make_frag_bases, V image)
struct TAFrag { const char* r0; const char* r1; const char* t[4]; };
__device__ __forceinline__ TAFrag ta_frag(const char* img, int lane) {
  TAFrag f;
  const int fi = lane & 15, fg = lane >> 4;
  const int sw = ta_sw(fi);                            // (independent of the tile: rows advance in multiples of 16)
  f.r0 = img + fi * 128 + ((fg ^ sw) << 4);
  f.r1 = img + fi * 128 + (((4 + fg) ^ sw) << 4);
  const int j = fi >> 2, qd = fi & 3, row = 4 * fg + j;
  const int vs = (row >> 1) & 3;
#pragma unroll
  for (int dn = 0; dn < 4; ++dn) f.t[dn] = img + row * 128 + ((dn ^ vs) << 5) + qd * 8;
  return f;
}
__device__ __forceinline__ f16x8 ta_tr(const char* p) {
  const v4bf16_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) v4bf16_t*)(p));
  const v4bf16_t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) v4bf16_t*)(p + 2048));
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  return __builtin_bit_cast(f16x8, (bf16x8_t){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
}
#define TA_MFMA(A, B, ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16((A), (B), (ACC), 0, 0, 0)

// the two 16-row tiles t, t + 1 of  (image rows) x (a register operand): acc_a / acc_b [row 16 t' + 4 fg + r][column lane & 15]
//   = sum_d img[row][d] x[column][d]  -- lo.hi + hi.lo + hi.hi over the two 32-deep halves of d.
template <int PLANE>
__device__ __forceinline__ void ta_rows_pair(const TAFrag& f, int t, const f16x8 (&xh)[2], const f16x8 (&xl)[2], f32x4& a, f32x4& b) {
  const char* p0 = f.r0 + t * 2048;
  const char* p1 = f.r1 + t * 2048;
  const f16x8 al0 = *reinterpret_cast<const f16x8*>(p0 + PLANE), al1 = *reinterpret_cast<const f16x8*>(p1 + PLANE);
  const f16x8 bl0 = *reinterpret_cast<const f16x8*>(p0 + 2048 + PLANE), bl1 = *reinterpret_cast<const f16x8*>(p1 + 2048 + PLANE);
  const f16x8 ah0 = *reinterpret_cast<const f16x8*>(p0), ah1 = *reinterpret_cast<const f16x8*>(p1);
  const f16x8 bh0 = *reinterpret_cast<const f16x8*>(p0 + 2048), bh1 = *reinterpret_cast<const f16x8*>(p1 + 2048);
  a = (f32x4){0.f, 0.f, 0.f, 0.f}; b = a;
  TA_MFMA(al0, xh[0], a); TA_MFMA(bl0, xh[0], b);
  TA_MFMA(al1, xh[1], a); TA_MFMA(bl1, xh[1], b);
  TA_MFMA(ah0, xl[0], a); TA_MFMA(bh0, xl[0], b);
  TA_MFMA(ah1, xl[1], a); TA_MFMA(bh1, xl[1], b);
  TA_MFMA(ah0, xh[0], a); TA_MFMA(bh0, xh[0], b);
  TA_MFMA(ah1, xh[1], a); TA_MFMA(bh1, xh[1], b);
}
// acc[dn][channel dn 16 + 4 fg + i][column] += sum over the 32 image rows of chunk c of img[row][channel] y[row][column]
// (y as a split register operand in the k order of the transposed fragments: rows 4 fg + r of tile 2 c, then of tile 2 c + 1)
template <int PLANE>
__device__ __forceinline__ void ta_tr_chunk(const TAFrag& f, int c, const f16x8& yh, const f16x8& yl, f32x4 (&acc)[4]) {
  f16x8 th[4], tl[4];
#pragma unroll
  for (int dn = 0; dn < 4; ++dn) { th[dn] = ta_tr(f.t[dn] + c * 4096); tl[dn] = ta_tr(f.t[dn] + c * 4096 + PLANE); }
#pragma unroll
  for (int dn = 0; dn < 4; ++dn) TA_MFMA(tl[dn], yh, acc[dn]);
#pragma unroll
  for (int dn = 0; dn < 4; ++dn) TA_MFMA(th[dn], yl, acc[dn]);
#pragma unroll
  for (int dn = 0; dn < 4; ++dn) TA_MFMA(th[dn], yh, acc[dn]);
}

// this lane's 16 values of a [.][64] fp32 row as a split register operand (column operand of ta_rows_pair): channels
// 8 fg .. + 7 and 32 + 8 fg .. + 7
__device__ __forceinline__ void ta_load_row_op(const float* row, int fg, float sc, f16x8 (&h)[2], f16x8 (&l)[2]) {
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const float* p = row + half * 32 + fg * 8;
    ta_split8(*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4), h[half], l[half], sc);
  }
}

// eight values (two tiles x four rows) -> one split register operand at the power of two 2^(140 - eb) (eb: biased exponent
// the caller keeps >= that of the largest magnitude: |y| 2^(140 - eb) < 2^14)
__device__ __forceinline__ void ta_split_run(const float (&y)[8], int eb, f16x8& h, f16x8& l) {
  const float sc = __uint_as_float((unsigned)(267 - eb) << 23);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = ta_opaque(y[e] * sc);              // (opaque: both halves must round the SAME number)
    const f16 hh = (f16)v;
    h[e] = hh;
    l[e] = (f16)(v - (float)hh);
  }
}
// biased exponent of max |y[e]| over the lane's eight values and the four lanes that share its column
__device__ __forceinline__ int ta_exp_of_max(const float (&y)[8]) {
  float m = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(y[e]));
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  return (int)(__float_as_uint(m) >> 23);
}

}  // namespace
