// Shared device helpers for the D3DP denoiser kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WAVE 64

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---- wave64 reductions (butterfly; every lane ends with the result) -------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact (erf) GELU as torch.nn.GELU() default: 0.5 x (1 + erf(x / sqrt(2)))
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// XCD-aware bijective remap of a 1-D grid: blocks that are consecutive in the LOGICAL order land on
// the same XCD (hardware places block b on XCD b % 8), so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int NX = 8;
  int xcd = bid % NX, idx = bid / NX;
  int q = nblk / NX, r = nblk % NX;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// activation storage type per numerics mode
template <typename T> struct Act;
template <> struct Act<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Act<bf16> {
  static __device__ __forceinline__ float ld(const bf16* p) { return (float)*p; }
  static __device__ __forceinline__ void st(bf16* p, float v) { *p = (bf16)v; }
};

// 8 consecutive activations <-> 8 floats
__device__ __forceinline__ void load8(const float* p, float* v) {
  float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16* p, float* v) {
  bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
__device__ __forceinline__ void store8(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16* p, const float* v) {
  bf16x8 a;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (bf16)v[i];
  *reinterpret_cast<bf16x8*>(p) = a;
}

// ---- split-bf16 ("bf16x3") representation of an fp32 value: x = x0 + x1 + x2 exactly (3 x 8 significand bits),
// each part a bf16.  Products of parts are exact in fp32, so a GEMM over the six leading part-pairs on the bf16
// matrix cores reproduces an fp32 GEMM to fp32 accuracy (measured: mean |err| 0.8e-7 vs 2.1e-7 for an fp32 fmaf
// chain at K = 512) at 1/6 of the bf16 MFMA rate = 2.6x the fp32 MFMA rate.
__device__ __forceinline__ void split3(float x, bf16& p0, bf16& p1, bf16& p2) {
  p0 = (bf16)x;
  const float r1 = x - (float)p0;
  p1 = (bf16)r1;
  const float r2 = r1 - (float)p1;
  p2 = (bf16)r2;
}
struct b3 {};   // tag: activation stored as three bf16 planes [3][T][C]

// ---- split-fp16 ("f16x2") representation: x ~ hi + lo * 2^-11 with hi = fp16(x), lo = fp16((x - hi) * 2^11)
// (2 x 11 significand bits; |x - hi - lo 2^-11| <= 2^-22 |x|).  The three leading products hi.hi + (hi.lo + lo.hi) 2^-11
// on v_mfma_f32_16x16x32_f16 reproduce an fp32 Linear to fp32 class (CPU study tools/err_budget_split.py: the dropped
// lo.lo term and the representation error stay below the fp32 accumulation noise of the reference itself) at 1/3 of
// the fp16 MFMA rate -- twice the six-pass split-bf16 scheme, with 4 instead of 6 bytes per operand element.
// lo is pre-scaled by 2^11 so that it stays a NORMAL fp16 number wherever hi is one; a value whose hi would be an
// fp16 subnormal is carried by lo alone (hi = 0), so the scheme does not depend on how the matrix cores treat fp16
// subnormals.  Valid for |x| < 65504 (LayerNorm outputs, GELU hidden activations, attention outputs, scaled weights).
typedef _Float16 f16;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f;
__device__ __forceinline__ void split2h(float x, f16& hi, f16& lo) {
  f16 h = (f16)x;
  if (fabsf(x) < 6.103515625e-05f) h = (f16)0.0f;
  hi = h;
  lo = (f16)((x - (float)h) * kLoScale);
}
struct h2 {};   // tag: activation stored as two fp16 planes [2][T][C] (hi, lo * 2^11)
