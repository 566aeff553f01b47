// Shared device helpers for the D3DP denoiser kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// -DD3DP_FAST_F16=1 (`make fastf16`: lib/variants/libd3dp_fastf16.so, NOT the product library): FAST mode's 2-byte operand type is
// IEEE fp16 instead of bf16 -- 11 significand bits instead of 8 at the same MFMA rate (reported beside
// the bf16 figure, no parity claim).  Every value FAST mode stores in 2 bytes lies inside fp16's range for this model (LayerNorm
// outputs <= 23, |w| < 1, probabilities, GELU outputs; the residual stream stays fp32), so the type is swapped and nothing is scaled.
#ifndef D3DP_FAST_F16
#define D3DP_FAST_F16 0
#endif
#if D3DP_FAST_F16
typedef _Float16 bf16;
typedef _Float16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));
#define D3DP_MFMA_16x16x32_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#else
typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define D3DP_MFMA_16x16x32_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#endif
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

// The same function with erf as ONE rational x P(x^2) / Q(x^2) on [-4, 4] (|erf| = 1 beyond, in fp32) instead of
// libm's branching erff: 21 VALU instructions instead of 41 per element in the EXACT fc1 epilogue, where they run with
// the matrix pipes idle.  Coefficients: the single-precision erf of Eigen / XLA; measured against fp64 on N(0,1) inputs
// (tools/erf_check.py): mean |error| of the GELU 2.2e-8, max 6.1e-7 -- torch's own fp32 GELU has 1.8e-8 and 9.9e-7.
// erf(z) by the same rational (z clamped to [-4, 4], where |erf| = 1 in fp32)
__device__ __forceinline__ float erf_rational(float zin) {
  const float z = __builtin_amdgcn_fmed3f(zin, -4.0f, 4.0f), z2 = z * z;
  float p = -2.72614225801306e-10f;
  p = fmaf(p, z2, 2.77068142495902e-08f);
  p = fmaf(p, z2, -2.10102402082508e-06f);
  p = fmaf(p, z2, -5.69250639462346e-05f);
  p = fmaf(p, z2, -7.34990630326855e-04f);
  p = fmaf(p, z2, -2.95459980854025e-03f);
  p = fmaf(p, z2, -1.60960333262415e-02f) * z;
  float q = -1.45660718464996e-05f;
  q = fmaf(q, z2, -2.13374055278905e-04f);
  q = fmaf(q, z2, -1.68282697438203e-03f);
  q = fmaf(q, z2, -7.37332916720468e-03f);
  q = fmaf(q, z2, -1.42647390514189e-02f);
  const float r = __builtin_amdgcn_rcpf(q);
  const float e = p * r;
  return fmaf(fmaf(-q, e, p), r, e);                   // one Newton step on the quotient
}
__device__ __forceinline__ float gelu_erf_rational(float x) {
  const float z = __builtin_amdgcn_fmed3f(x * 0.70710678118654752440f, -4.0f, 4.0f), z2 = z * z;
  float p = -2.72614225801306e-10f;
  p = fmaf(p, z2, 2.77068142495902e-08f);
  p = fmaf(p, z2, -2.10102402082508e-06f);
  p = fmaf(p, z2, -5.69250639462346e-05f);
  p = fmaf(p, z2, -7.34990630326855e-04f);
  p = fmaf(p, z2, -2.95459980854025e-03f);
  p = fmaf(p, z2, -1.60960333262415e-02f) * z;
  float q = -1.45660718464996e-05f;
  q = fmaf(q, z2, -2.13374055278905e-04f);
  q = fmaf(q, z2, -1.68282697438203e-03f);
  q = fmaf(q, z2, -7.37332916720468e-03f);
  q = fmaf(q, z2, -1.42647390514189e-02f);
  const float r = __builtin_amdgcn_rcpf(q);
  float e = p * r;
  e = fmaf(fmaf(-q, e, p), r, e);                      // one Newton step on the quotient
  const float hx = 0.5f * x;
  return fmaf(hx, e, hx);
}

// d GELU(v) / dv = Phi(v) + v phi(v) on the same rational erf; |.| <= 1.1290 (at v = sqrt 2)
__device__ __forceinline__ float gelu_grad_rational(float v) {
  const float cdf = fmaf(0.5f, erf_rational(v * 0.70710678118654752440f), 0.5f);
  const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(v * v * -0.72134752044448170368f);   // exp(-v^2 / 2) = 2^(-v^2 log2(e) / 2)
  return fmaf(v, pdf, cdf);
}
constexpr float kGeluGradMax = 1.13f;                  // (bound used for operand scales: dh gelu'(h) from the absmax of dh)

// two elements at a time on the packed fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32): the same arithmetic per
// element, half the instructions for the polynomials -- for epilogues, where no MFMA is in flight to be disturbed by them
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf_rational2(f32x2 x) {
  const f32x2 zs = x * 0.70710678118654752440f;
  const f32x2 z = {__builtin_amdgcn_fmed3f(zs.x, -4.0f, 4.0f), __builtin_amdgcn_fmed3f(zs.y, -4.0f, 4.0f)};
  const f32x2 z2 = z * z;
  auto c = [](float v) { return (f32x2){v, v}; };
  f32x2 p = c(-2.72614225801306e-10f);
  p = __builtin_elementwise_fma(p, z2, c(2.77068142495902e-08f));
  p = __builtin_elementwise_fma(p, z2, c(-2.10102402082508e-06f));
  p = __builtin_elementwise_fma(p, z2, c(-5.69250639462346e-05f));
  p = __builtin_elementwise_fma(p, z2, c(-7.34990630326855e-04f));
  p = __builtin_elementwise_fma(p, z2, c(-2.95459980854025e-03f));
  p = __builtin_elementwise_fma(p, z2, c(-1.60960333262415e-02f)) * z;
  f32x2 q = c(-1.45660718464996e-05f);
  q = __builtin_elementwise_fma(q, z2, c(-2.13374055278905e-04f));
  q = __builtin_elementwise_fma(q, z2, c(-1.68282697438203e-03f));
  q = __builtin_elementwise_fma(q, z2, c(-7.37332916720468e-03f));
  q = __builtin_elementwise_fma(q, z2, c(-1.42647390514189e-02f));
  const f32x2 r = {__builtin_amdgcn_rcpf(q.x), __builtin_amdgcn_rcpf(q.y)};
  f32x2 e = p * r;
  e = __builtin_elementwise_fma(__builtin_elementwise_fma(-q, e, p), r, e);
  const f32x2 hx = x * 0.5f;
  return __builtin_elementwise_fma(hx, e, hx);
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

// ---- split-fp16 ("f16x2") representation of an fp32 value: x 2^s = hi + lo with hi = fp16(x 2^s), lo = fp16(x 2^s - hi)
// (2 x 11 significand bits: |x 2^s - hi - lo| <= 2^-22 |x 2^s|).  The three leading products hi.hi + hi.lo + lo.hi on
// v_mfma_f32_16x16x32_f16 reproduce an fp32 Linear to fp32 class (CPU study tools/err_budget_split.py: the dropped lo.lo
// term and the representation error stay below the fp32 accumulation noise of the reference itself; measured on MI355X:
// mean error 1.3e-7 against 2.8e-7 for torch's fp32 matmul at K = 512) at 1/3 of the fp16 MFMA rate -- twice the
// six-pass split-bf16 scheme of round 1, with 4 instead of 6 bytes per operand element.
// The power-of-two pre-scale 2^s keeps lo a NORMAL fp16 number for every value that matters: activations use the fixed
// s = 4 (full 22 bits for 2^-7 <= |x| < 4094; smaller values keep an ABSOLUTE error <= 2^-29, far below 2^-22 of the
// O(1) values they are summed with; LayerNorm outputs, GELU hidden activations and attention outputs stay orders of
// magnitude below 4094), weight matrices get s from their largest element (capi.hip).  fp16 subnormal inputs are NOT
// flushed by the fp16 matrix cores (tools/probe_f16_denorm.py, tests/test_hip_parity.py), which the tail of the range
// relies on.
typedef _Float16 f16;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float kActScale = 16.0f, kActUnscale = 1.0f / 16.0f;
// HBM layout of a split-fp16 matrix [R][K] ("h2i": hi / lo interleaved per 128-byte line): row r = 2 K fp16 = K/32 blocks
// of 64 fp16, block kb = [hi of columns 32 kb .. 32 kb + 31 | lo of the same columns].  One 32-deep k-step of one row --
// both values of every element -- is ONE 128-byte line, which is what the Linear kernel's LDS-DMA loaders want
// (gemm_x2.hip: 8 whole lines per 1 KiB piece instead of 16 half lines).  h2i_col(c) = offset of column c's hi value
// inside its row; the lo value sits kH2iLo elements further.  A group of 4 (8) columns aligned to 4 (8) stays inside
// one block: 8 (16) contiguous bytes per plane.
__device__ __forceinline__ int h2i_col(int c) { return ((c >> 5) << 6) | (c & 31); }
constexpr int kH2iLo = 32;
__device__ __forceinline__ void split2h_scaled(float y, f16& hi, f16& lo) {   // y = x 2^s already
  hi = (f16)y;
  lo = (f16)(y - (float)hi);
}
__device__ __forceinline__ void split2h(float x, f16& hi, f16& lo) { split2h_scaled(x * kActScale, hi, lo); }
struct h2 {};   // tag: activation stored as two fp16 planes [2][T][C] (hi, lo of x * 16)
