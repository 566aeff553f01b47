// Attention of the TRAINING step on the fp16 matrix cores (reference common/mixste.py:63-82 inside
// main.py:387-401's forward / backward), head dim 64, sequences of up to 256 tokens (the temporal axis: 243 frames).
//
// Round 4 ran the temporal forward and both backward passes on the FP32 matrix cores (v_mfma_f32_16x16x4_f32: 1/16 of the fp16
// rate; 5.9 ms of the 27 ms configs[4] step).  Here every product runs on split-fp16 operands like the step's Linears
// (gemm_x2.hip): x 2^s = hi + lo, three v_mfma_f32_16x16x32_f16 passes lo.hi + hi.lo + hi.hi into one fp32 accumulator --
// fp32-class results at 5.3 x fewer matrix-pipe cycles.  Operand scales are DEVICE values, like the Linears': q, k, v share
// the power of two the qkv Linear's output absmax asks for, dO the one of the proj dgrad's output (both left by the GEMM
// epilogues, gemm_f16x2_dyn_kernel), probabilities use 2^10 (P <= 1), and dS -- whose magnitude no bound predicts within
// eight bits -- a RUNNING power of two per query (pass Q) / per key (pass KV): when a new key pair raises the row's
// largest |dS| the accumulators are rescaled by the (exact) ratio, as an online softmax rescales its output.
//
// Three kernels; a work unit = one (sequence, head) problem x one group of NW 16-row tiles, one tile per wave (NW = 8 for the
// long sequences: two groups per 243-frame problem; NW = 2 for the spatial axis' 17 joints: 128-thread workgroups, ten of them
// per CU); workgroups walk the units grid-stride, so a launch ends with at most a few thousand absmax atomics:
//   forward : K, V as split images in LDS; S^T = K Q^T per key pair, online softmax, O^T += V^T P^T.  Leaves
//             L_i = log2 sum_j exp(s_ij) per query (the softmax's log-sum-exp in base 2) for the backward pass.
//   pass Q  : K, V images; per key pair S^T = K Q^T, dP^T = V dO^T, P = exp2(s - L), dS^T = P^T (dP^T - D) / 8,
//             dQ^T += K^T dS^T.  Writes D_i = dO_i . O_i beside L_i.
//   pass KV : Q, dO images and (L, D); per query pair S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS.
// LDS image of a [n][64] matrix: two planes (hi | lo) of 128-byte rows, 16-byte slot s of row r at s ^ (((r >> 1) & 3) << 1)
// -- the V image of attention.hip: conflict-free for the transposed fragment reads (ds_read_b64_tr_b16: 4 rows x 32 B per 16
// lanes) AND for the row fragment reads (ds_read_b128, whose lane groups pair rows {0-3, 12-15} of one slot with rows {4-11}
// of the neighbouring one) -- so ONE image serves both uses of K (S^T and dQ^T) and of Q / dO in pass KV.
#include "common.h"
#include "kernels.h"
#include "ta_common.h"

namespace {

struct TAStat { float L, D; };                         // L = log2 sum_j exp(s_ij) (s = q.k / 8), D = dO_i . O_i

template <int NW>
__device__ __forceinline__ void ta_block_amax(float am, unsigned* amax, float* part) {
  am = wave_max(am);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = am;
  __syncthreads();
  if (threadIdx.x == 0 && amax) {
    float m = part[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, part[w]);
    atomicMax(amax, __float_as_uint(m));
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// forward
// (NKC: key tiles per LDS chunk.  Round 6: the images hold 128 keys at a time -- the online softmax runs over key pairs anyway --
//  so a 243-frame problem needs 64 KiB instead of 128 and TWO workgroups share a CU: four waves per SIMD instead of two)
template <int NKT, int NW, int NKC>
__global__ __launch_bounds__(NW * 64, 4) void tattn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                           TAStat* __restrict__ stats, SeqMap map, int C, int heads, int groups,
                                                           int n_work, const unsigned* __restrict__ amax_qkv,
                                                           unsigned* __restrict__ amax_out, f16* __restrict__ op, int T, int Tp,
                                                           float* __restrict__ op_unscale) {
  constexpr int NK = 16 * NKC, PLANE = NK * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kimg = smem;
  char* vimg = smem + 2 * PLANE;
  float* part = reinterpret_cast<float*>(smem + 4 * PLANE);
  const int n = map.n_tok;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t ld = (size_t)3 * C, rs = (size_t)map.tok_stride * ld;
  const float sq = ta_scale(amax_qkv);
  float am = 0.f;
  // op (round 6): the output ALSO as the split operand rows [Tp][2 C] of the proj Linear behind it (h2i; rows T .. Tp - 1 zero), at
  // the scale of q / k / v -- every output is a convex combination of v rows, so |o| <= max |v| <= the qkv absmax: the bound is
  // known before the first element is written and the operand pass between attention and proj disappears.  op_unscale[0] = 1 /
  // scale, amax_out[0] = the bound.
  if (op) {
    if (blockIdx.x == 0 && tid == 0) { op_unscale[0] = 1.0f / sq; if (amax_out) amax_out[0] = amax_qkv[0]; }
    const size_t n8 = (size_t)(Tp - T) * C / 4;        // 16-byte pieces of the pad rows (2 C fp16 each)
    uint4* z = reinterpret_cast<uint4*>(op + (size_t)T * 2 * C);
    for (size_t i = (size_t)blockIdx.x * (NW * 64) + tid; i < n8; i += (size_t)gridDim.x * (NW * 64)) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  for (int unit = blockIdx.x; unit < n_work; unit += gridDim.x) {
  const int prob = unit / groups, group = unit % groups;
  const int seq = prob / heads, head = prob % heads;
  const int base = ta_seq_base(map, seq);
  const float* p0 = qkv + (size_t)base * ld + (size_t)head * 64;
  const int qt = group * NW + wave;
  const bool active = qt * 16 < n;                     // (wave-uniform)
  const int fi = lane & 15, fg = lane >> 4;
  const int q = qt * 16 + fi;
  const size_t tok = (size_t)(base + min(q, n - 1) * map.tok_stride);
  f16x8 qh[2], ql[2];
  if (active) ta_load_row_op(qkv + tok * ld + head * 64, fg, sq, qh, ql);
  const TAFrag fk = ta_frag(kimg, lane), fv = ta_frag(vimg, lane);
  const float cexp = 0.125f * kLog2e / (sq * sq);     // raw accumulator -> logit in base-2 units
  float mrun = -INFINITY, lrun = 0.f;                  // running row maximum (base-2 logit units) and denominator
  f32x4 o[4];
#pragma unroll
  for (int dn = 0; dn < 4; ++dn) o[dn] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int kc = 0; kc < NKT && 16 * kc < n; kc += NKC) {
  if (unit != (int)blockIdx.x || kc != 0) __syncthreads();   // every wave is done with the previous chunk's / unit's images
  {
    const float* const src2[2] = {p0 + C + (size_t)kc * 16 * rs, p0 + 2 * C + (size_t)kc * 16 * rs};
    const size_t rs2[2] = {rs, rs};
    const float sc2[2] = {sq, sq};
    char* const img2[2] = {kimg, vimg};
    ta_stage_many<NK, NW * 64, 2>(src2, rs2, sc2, img2, n - 16 * kc, tid);
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int t = 0; t < NKC; t += 2) {
      if (16 * (kc + t) >= n) break;                   // (uniform: the rest of the chunk lies behind the sequence)
      f32x4 a, b;
      ta_rows_pair<PLANE>(fk, t, qh, ql, a, b);
      float s[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[r] = a[r] * cexp; s[4 + r] = b[r] * cexp; }
      if (16 * (kc + t + 2) > n) {                     // (uniform: only the last pair(s) hold keys >= n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (16 * (kc + t) + 4 * fg + r >= n) s[r] = -INFINITY;
          if (16 * (kc + t + 1) + 4 * fg + r >= n) s[4 + r] = -INFINITY;
        }
      }
      float mx = s[0];
#pragma unroll
      for (int e = 1; e < 8; ++e) mx = fmaxf(mx, s[e]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(mrun, mx);              // (finite from the first pair on: key 0 exists)
      const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
      mrun = mnew;
      float psum = 0.f;
      f16x8 ph, pl;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float y = __builtin_amdgcn_exp2f(s[e] - mnew + 10.0f);   // p x 1024
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
      ta_tr_chunk<PLANE>(fv, t >> 1, ph, pl, o);
    }
  }
  }
  if (active) {
    // o = (v scale) x 1024 x sum_j p_j v_j against the running maximum; lrun = 1024 x sum_j p_j
    const float inv = 1.0f / (sq * lrun);
    if (q < n) {
      float* dst = out + tok * C + head * 64 + fg * 4;     // O^T[d = dn 16 + 4 fg + i][query fi]
#pragma unroll
      for (int dn = 0; dn < 4; ++dn) {
        const float4 r4 = make_float4(o[dn][0] * inv, o[dn][1] * inv, o[dn][2] * inv, o[dn][3] * inv);
        am = fmaxf(am, fmaxf(fmaxf(fabsf(r4.x), fabsf(r4.y)), fmaxf(fabsf(r4.z), fabsf(r4.w))));
        *reinterpret_cast<float4*>(dst + dn * 16) = r4;
        if (op) {
          // (the true-scale value first, the exact power of two inside the split: both halves must round the SAME number)
          const float y[4] = {ta_opaque(r4.x * sq), ta_opaque(r4.y * sq), ta_opaque(r4.z * sq), ta_opaque(r4.w * sq)};
          f16x4 hi, lo;
#pragma unroll
          for (int e = 0; e < 4; ++e) { const f16 hh = (f16)y[e]; hi[e] = hh; lo[e] = (f16)(y[e] - (float)hh); }
          f16* d = op + tok * 2 * C + h2i_col(head * 64 + dn * 16 + fg * 4);
          *reinterpret_cast<f16x4*>(d) = hi;
          *reinterpret_cast<f16x4*>(d + kH2iLo) = lo;
        }
      }
      if (fg == 0) stats[(size_t)prob * n + q].L = mrun + __builtin_amdgcn_logf(lrun) - 10.0f;   // log2(sum exp2(s))
    }
  }
  }
  if (amax_out && !op) ta_block_amax<NW>(am, amax_out, part);
}

// ------------------------------------------------------------------------------------------------------------------------
// backward, pass Q: dQ (and D_i into the statistics)
template <int NKT, int NW, int NKC, int WPE>
__global__ __launch_bounds__(NW * 64, WPE) void tattn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                             const float* __restrict__ dout, float* __restrict__ dqkv,
                                                             TAStat* __restrict__ stats, SeqMap map, int C, int heads, int groups,
                                                             int n_work, const unsigned* __restrict__ amax_qkv,
                                                             const unsigned* __restrict__ amax_do, unsigned* __restrict__ amax_out) {
  constexpr int NK = 16 * NKC, PLANE = NK * 128;      // (the images hold NKC key tiles at a time: see tattn_fwd_kernel)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kimg = smem;
  char* vimg = smem + 2 * PLANE;
  float* part = reinterpret_cast<float*>(smem + 4 * PLANE);
  const int n = map.n_tok;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t ld = (size_t)3 * C, rs = (size_t)map.tok_stride * ld;
  const float sq = ta_scale(amax_qkv), sg = ta_scale(amax_do);
  float am = 0.f;
  for (int unit = blockIdx.x; unit < n_work; unit += gridDim.x) {
  const int prob = unit / groups, group = unit % groups;
  const int seq = prob / heads, head = prob % heads;
  const int base = ta_seq_base(map, seq);
  const float* p0 = qkv + (size_t)base * ld + (size_t)head * 64;
  const int qt = group * NW + wave;
  const bool active = qt * 16 < n;                     // (wave-uniform)
  const int fi = lane & 15, fg = lane >> 4;
  const int q = qt * 16 + fi;
  const size_t tok = (size_t)(base + min(q, n - 1) * map.tok_stride);
  f16x8 qh[2], ql[2], gh[2], gl[2];
  float D = 0.f, L = 0.f;
  if (active) {
    ta_load_row_op(qkv + tok * ld + head * 64, fg, sq, qh, ql);
    ta_load_row_op(dout + tok * C + head * 64, fg, sg, gh, gl);
    {
      const float* gs = dout + tok * C + head * 64;
      const float* os = o + tok * C + head * 64;
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int c4 = 0; c4 < 2; ++c4) {
          const float4 g = *reinterpret_cast<const float4*>(gs + half * 32 + fg * 8 + c4 * 4);
          const float4 ov = *reinterpret_cast<const float4*>(os + half * 32 + fg * 8 + c4 * 4);
          D = fmaf(g.x, ov.x, D); D = fmaf(g.y, ov.y, D); D = fmaf(g.z, ov.z, D); D = fmaf(g.w, ov.w, D);
        }
      D += __shfl_xor(D, 16, 64);
      D += __shfl_xor(D, 32, 64);
    }
    L = stats[(size_t)prob * n + min(q, n - 1)].L;
  }
  const TAFrag fk = ta_frag(kimg, lane), fv = ta_frag(vimg, lane);
  const float cexp = 0.125f * kLog2e / (sq * sq);
  const float cdp = 1.0f / (sq * sg);                  // raw dP accumulator -> true scale
  int eb = 20;                                         // running biased exponent of the row's largest |dS| (floor 2^-107)
  f32x4 dq[4];
#pragma unroll
  for (int dn = 0; dn < 4; ++dn) dq[dn] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int kc = 0; kc < NKT && 16 * kc < n; kc += NKC) {
  if (unit != (int)blockIdx.x || kc != 0) __syncthreads();
  {
    const float* const src2[2] = {p0 + C + (size_t)kc * 16 * rs, p0 + 2 * C + (size_t)kc * 16 * rs};
    const size_t rs2[2] = {rs, rs};
    const float sc2[2] = {sq, sq};
    char* const img2[2] = {kimg, vimg};
    ta_stage_many<NK, NW * 64, 2>(src2, rs2, sc2, img2, n - 16 * kc, tid);
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int t = 0; t < NKC; t += 2) {
      if (16 * (kc + t) >= n) break;                   // (uniform)
      f32x4 a, b, c, d;
      ta_rows_pair<PLANE>(fk, t, qh, ql, a, b);        // S^T  [key][query]
      ta_rows_pair<PLANE>(fv, t, gh, gl, c, d);        // dP^T [key][query]
      float ds[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pa = __builtin_amdgcn_exp2f(fmaf(a[r], cexp, -L)), pb = __builtin_amdgcn_exp2f(fmaf(b[r], cexp, -L));
        ds[r] = pa * (fmaf(c[r], cdp, -D)) * 0.125f;
        ds[4 + r] = pb * (fmaf(d[r], cdp, -D)) * 0.125f;
      }
      if (16 * (kc + t + 2) > n) {                     // (uniform: only the last pair(s) hold keys >= n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (16 * (kc + t) + 4 * fg + r >= n) ds[r] = 0.f;
          if (16 * (kc + t + 1) + 4 * fg + r >= n) ds[4 + r] = 0.f;
        }
      }
      const int en = max(eb, ta_exp_of_max(ds));
      if (en != eb) {                                  // (uniform over the four lanes of a column; exact rescale)
#pragma unroll
        for (int dn = 0; dn < 4; ++dn)
#pragma unroll
          for (int i = 0; i < 4; ++i) dq[dn][i] = ldexpf(dq[dn][i], eb - en);
        eb = en;
      }
      f16x8 sh, sl;
      ta_split_run(ds, eb, sh, sl);
      ta_tr_chunk<PLANE>(fk, t >> 1, sh, sl, dq);      // dQ^T[d][query] += K^T dS^T
    }
  }
  }
  if (active) {
    if (q < n) {
      const float un = 1.0f / sq;
      float* dst = dqkv + tok * ld + head * 64 + fg * 4;
#pragma unroll
      for (int dn = 0; dn < 4; ++dn) {
        const float4 r4 = make_float4(ldexpf(dq[dn][0], eb - 140) * un, ldexpf(dq[dn][1], eb - 140) * un,
                                      ldexpf(dq[dn][2], eb - 140) * un, ldexpf(dq[dn][3], eb - 140) * un);
        am = fmaxf(am, fmaxf(fmaxf(fabsf(r4.x), fabsf(r4.y)), fmaxf(fabsf(r4.z), fabsf(r4.w))));
        *reinterpret_cast<float4*>(dst + dn * 16) = r4;
      }
      if (fg == 0) stats[(size_t)prob * n + q].D = D;
    }
  }
  }
  if (amax_out) ta_block_amax<NW>(am, amax_out, part);
}

// ------------------------------------------------------------------------------------------------------------------------
// backward, pass KV: dK, dV
template <int NKT, int NW, int NKC, int WPE>
__global__ __launch_bounds__(NW * 64, WPE) void tattn_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                              float* __restrict__ dqkv, const TAStat* __restrict__ stats,
                                                              SeqMap map, int C, int heads, int groups, int n_work,
                                                              const unsigned* __restrict__ amax_qkv,
                                                              const unsigned* __restrict__ amax_do, unsigned* __restrict__ amax_out) {
  constexpr int NK = 16 * NKC, PLANE = NK * 128;      // (the images hold NKC QUERY tiles at a time)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* qimg = smem;
  char* gimg = smem + 2 * PLANE;
  float2* st = reinterpret_cast<float2*>(smem + 4 * PLANE);          // [NK] (L, D)
  float* part = reinterpret_cast<float*>(smem + 4 * PLANE + NK * 8);
  const int n = map.n_tok;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t ld = (size_t)3 * C, rs = (size_t)map.tok_stride * ld;
  const float sq = ta_scale(amax_qkv), sg = ta_scale(amax_do);
  float am = 0.f;
  for (int unit = blockIdx.x; unit < n_work; unit += gridDim.x) {
  const int prob = unit / groups, group = unit % groups;
  const int seq = prob / heads, head = prob % heads;
  const int base = ta_seq_base(map, seq);
  const int kt = group * NW + wave;
  const bool active = kt * 16 < n;                     // (wave-uniform)
  const int fi = lane & 15, fg = lane >> 4;
  const int key = kt * 16 + fi;
  const bool live = key < n;
  const size_t tok = (size_t)(base + min(key, n - 1) * map.tok_stride);
  f16x8 kh[2], kl[2], vh[2], vl[2];
  if (active) {
    ta_load_row_op(qkv + tok * ld + C + head * 64, fg, sq, kh, kl);
    ta_load_row_op(qkv + tok * ld + 2 * C + head * 64, fg, sq, vh, vl);
  }
  const TAFrag fq = ta_frag(qimg, lane), fgr = ta_frag(gimg, lane);
  const float cexp = 0.125f * kLog2e / (sq * sq);
  const float cdp = 1.0f / (sq * sg);
  int eb = 20;
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int dn = 0; dn < 4; ++dn) { dk[dn] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dn] = dk[dn]; }
#pragma unroll 1
  for (int qc = 0; qc < NKT && 16 * qc < n; qc += NKC) {
  if (unit != (int)blockIdx.x || qc != 0) __syncthreads();
  // (one operand after the other here: both at once -- ta_stage_many, as in the forward pass and pass Q -- costs this 200-register
  //  kernel more than the second round trip: 98 -> 105 us)
  ta_stage<NK, NW * 64>(qkv + (size_t)base * ld + (size_t)head * 64 + (size_t)qc * 16 * rs, rs, n - 16 * qc, sq, qimg, tid);
  ta_stage<NK, NW * 64>(dout + (size_t)base * C + (size_t)head * 64 + (size_t)qc * 16 * map.tok_stride * C, (size_t)map.tok_stride * C,
                        n - 16 * qc, sg, gimg, tid);
  for (int i = tid; i < NK; i += NW * 64) {
    // rows >= n: their Q and dO image rows are zero, so any FINITE p and dS contribute nothing: L = +large keeps p = 0
    float2 v = make_float2(1.0e30f, 0.f);
    if (16 * qc + i < n) { const TAStat s = stats[(size_t)prob * n + 16 * qc + i]; v = make_float2(s.L, s.D); }
    st[i] = v;
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int t = 0; t < NKC; t += 2) {
      if (16 * (qc + t) >= n) break;                   // (uniform)
      f32x4 a, b, c, d;
      ta_rows_pair<PLANE>(fq, t, kh, kl, a, b);        // S  [query][key]
      ta_rows_pair<PLANE>(fgr, t, vh, vl, c, d);       // dP [query][key]
      float ds[8];
      f16x8 ph, pl;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float2 sa = st[16 * t + 4 * fg + r], sb = st[16 * (t + 1) + 4 * fg + r];
        float pa = __builtin_amdgcn_exp2f(fmaf(a[r], cexp, -sa.x)), pb = __builtin_amdgcn_exp2f(fmaf(b[r], cexp, -sb.x));
        if (!live) { pa = 0.f; pb = 0.f; }
        ds[r] = pa * (fmaf(c[r], cdp, -sa.y)) * 0.125f;
        ds[4 + r] = pb * (fmaf(d[r], cdp, -sb.y)) * 0.125f;
        const float ya = ta_opaque(pa * 1024.0f), yb = ta_opaque(pb * 1024.0f);
        const f16 ha = (f16)ya, hb = (f16)yb;
        ph[r] = ha; pl[r] = (f16)(ya - (float)ha);
        ph[4 + r] = hb; pl[4 + r] = (f16)(yb - (float)hb);
      }
      ta_tr_chunk<PLANE>(fgr, t >> 1, ph, pl, dv);     // dV^T[d][key] += dO^T P
      const int en = max(eb, ta_exp_of_max(ds));
      if (en != eb) {
#pragma unroll
        for (int dn = 0; dn < 4; ++dn)
#pragma unroll
          for (int i = 0; i < 4; ++i) dk[dn][i] = ldexpf(dk[dn][i], eb - en);
        eb = en;
      }
      f16x8 sh, sl;
      ta_split_run(ds, eb, sh, sl);
      ta_tr_chunk<PLANE>(fq, t >> 1, sh, sl, dk);      // dK^T[d][key] += Q^T dS
    }
  }
  }
  if (active) {
    if (live) {
      const float uk = 1.0f / sq, uv = 1.0f / (sg * 1024.0f);
      float* dst = dqkv + tok * ld + C + head * 64 + fg * 4;
#pragma unroll
      for (int dn = 0; dn < 4; ++dn) {
        const float4 k4 = make_float4(ldexpf(dk[dn][0], eb - 140) * uk, ldexpf(dk[dn][1], eb - 140) * uk,
                                      ldexpf(dk[dn][2], eb - 140) * uk, ldexpf(dk[dn][3], eb - 140) * uk);
        const float4 v4 = make_float4(dv[dn][0] * uv, dv[dn][1] * uv, dv[dn][2] * uv, dv[dn][3] * uv);
        am = fmaxf(am, fmaxf(fmaxf(fabsf(k4.x), fabsf(k4.y)), fmaxf(fabsf(k4.z), fabsf(k4.w))));
        am = fmaxf(am, fmaxf(fmaxf(fabsf(v4.x), fabsf(v4.y)), fmaxf(fabsf(v4.z), fabsf(v4.w))));
        *reinterpret_cast<float4*>(dst + dn * 16) = k4;
        *reinterpret_cast<float4*>(dst + C + dn * 16) = v4;
      }
    }
  }
  }
  if (amax_out) ta_block_amax<NW>(am, amax_out, part);
}

// ------------------------------------------------------------------------------------------------------------------------
// backward of sequences of at most 32 tokens (the spatial axis: 17 joints) in ONE kernel (round 6).  As two launches -- pass Q
// with K / V images, then pass KV with Q / dO images -- a block's spatial backward read q, k, v and dO twice each and o once:
// 406 MB per launch pair, 112 us.  A sequence this short fits a workgroup whole: all four operands are staged once as images (32
// KiB), each wave takes one 16-row tile through pass Q as a query tile and through pass KV as a key tile -- its register operands
// are row-fragment reads of the very images the other wave's products contract over -- and D_i = dO_i . O_i goes from pass Q to
// pass KV through LDS instead of the statistics buffer.  Same products, same splits, same accumulation order as the two
// kernels: bit-identical gradients (test_attention_backward_on_matrix_cores_matches_the_valu_kernels covers n = 9, 17 through it).
__global__ __launch_bounds__(128) void tattn_bwd_small_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                              const float* __restrict__ dout, float* __restrict__ dqkv,
                                                              const TAStat* __restrict__ stats, SeqMap map, int C, int heads,
                                                              int n_work, const unsigned* __restrict__ amax_qkv,
                                                              const unsigned* __restrict__ amax_do, unsigned* __restrict__ amax_out) {
  constexpr int NK = 32, PLANE = NK * 128, NW = 2;
  __shared__ __attribute__((aligned(16))) char smem[8 * PLANE];
  __shared__ float2 st[NK];                              // (L, D) of the sequence's queries
  __shared__ float part[NW];
  char* qimg = smem;
  char* kimg = smem + 2 * PLANE;
  char* vimg = smem + 4 * PLANE;
  char* gimg = smem + 6 * PLANE;
  const int n = map.n_tok;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, fg = lane >> 4;
  const size_t ld = (size_t)3 * C, rs = (size_t)map.tok_stride * ld;
  const float sq = ta_scale(amax_qkv), sg = ta_scale(amax_do);
  const float cexp = 0.125f * kLog2e / (sq * sq);
  const float cdp = 1.0f / (sq * sg);                    // raw dP accumulator -> true scale
  const TAFrag fq = ta_frag(qimg, lane), fk = ta_frag(kimg, lane), fv = ta_frag(vimg, lane), fgr = ta_frag(gimg, lane);
  auto row_op = [&](const TAFrag& f, int t, f16x8 (&h)[2], f16x8 (&l)[2]) {   // tile t's rows as a split register operand
    h[0] = *reinterpret_cast<const f16x8*>(f.r0 + t * 2048);
    h[1] = *reinterpret_cast<const f16x8*>(f.r1 + t * 2048);
    l[0] = *reinterpret_cast<const f16x8*>(f.r0 + t * 2048 + PLANE);
    l[1] = *reinterpret_cast<const f16x8*>(f.r1 + t * 2048 + PLANE);
  };
  float am = 0.f;
  for (int unit = blockIdx.x; unit < n_work; unit += gridDim.x) {
    const int seq = unit / heads, head = unit - seq * heads;
    const int base = ta_seq_base(map, seq);
    const float* p0 = qkv + (size_t)base * ld + (size_t)head * 64;
    const float* g0 = dout + (size_t)base * C + (size_t)head * 64;
    if (unit != (int)blockIdx.x) __syncthreads();        // every wave is done with the previous unit's images
    {
      const float* const src4[4] = {p0, p0 + C, p0 + 2 * C, g0};
      const size_t rs4[4] = {rs, rs, rs, (size_t)map.tok_stride * C};
      const float sc4[4] = {sq, sq, sq, sg};
      char* const img4[4] = {qimg, kimg, vimg, gimg};
      ta_stage_many<NK, NW * 64, 4>(src4, rs4, sc4, img4, n, tid);
    }
    const int row = wave * 16 + fi;                      // this lane's query (pass Q) and key (pass KV)
    const bool live = row < n;
    const size_t tok = (size_t)(base + min(row, n - 1) * map.tok_stride);
    {
      // D_i = dO_i . O_i in the channel order of tattn_bwd_q_kernel; rows >= n: L = +large keeps p = 0 in pass KV
      float D = 0.f;
      const float* gs = dout + tok * C + head * 64;
      const float* os = o + tok * C + head * 64;
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int c4 = 0; c4 < 2; ++c4) {
          const float4 g = *reinterpret_cast<const float4*>(gs + half * 32 + fg * 8 + c4 * 4);
          const float4 ov = *reinterpret_cast<const float4*>(os + half * 32 + fg * 8 + c4 * 4);
          D = fmaf(g.x, ov.x, D); D = fmaf(g.y, ov.y, D); D = fmaf(g.z, ov.z, D); D = fmaf(g.w, ov.w, D);
        }
      D += __shfl_xor(D, 16, 64);
      D += __shfl_xor(D, 32, 64);
      if (fg == 0) st[row] = live ? make_float2(stats[(size_t)unit * n + row].L, D) : make_float2(1.0e30f, 0.f);
    }
    __syncthreads();
    if (wave * 16 < n) {
      // ---- pass Q: this wave's tile as QUERIES against the key pair (0, 1)
      {
        f16x8 qh[2], ql[2], gh[2], gl[2];
        row_op(fq, wave, qh, ql);
        row_op(fgr, wave, gh, gl);
        const float2 ld_ = st[row];
        f32x4 a, b, c, d;
        ta_rows_pair<PLANE>(fk, 0, qh, ql, a, b);        // S^T  [key][query]
        ta_rows_pair<PLANE>(fv, 0, gh, gl, c, d);        // dP^T [key][query]
        float ds[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pa = __builtin_amdgcn_exp2f(fmaf(a[r], cexp, -ld_.x)), pb = __builtin_amdgcn_exp2f(fmaf(b[r], cexp, -ld_.x));
          ds[r] = pa * (fmaf(c[r], cdp, -ld_.y)) * 0.125f;
          ds[4 + r] = pb * (fmaf(d[r], cdp, -ld_.y)) * 0.125f;
          if (4 * fg + r >= n) ds[r] = 0.f;
          if (16 + 4 * fg + r >= n) ds[4 + r] = 0.f;
        }
        const int eb = max(20, ta_exp_of_max(ds));
        f16x8 sh, sl;
        ta_split_run(ds, eb, sh, sl);
        f32x4 dq[4];
#pragma unroll
        for (int dn = 0; dn < 4; ++dn) dq[dn] = (f32x4){0.f, 0.f, 0.f, 0.f};
        ta_tr_chunk<PLANE>(fk, 0, sh, sl, dq);           // dQ^T[d][query] = K^T dS^T
        if (live) {
          const float un = 1.0f / sq;
          float* dst = dqkv + tok * ld + head * 64 + fg * 4;
#pragma unroll
          for (int dn = 0; dn < 4; ++dn) {
            const float4 r4 = make_float4(ldexpf(dq[dn][0], eb - 140) * un, ldexpf(dq[dn][1], eb - 140) * un,
                                          ldexpf(dq[dn][2], eb - 140) * un, ldexpf(dq[dn][3], eb - 140) * un);
            am = fmaxf(am, fmaxf(fmaxf(fabsf(r4.x), fabsf(r4.y)), fmaxf(fabsf(r4.z), fabsf(r4.w))));
            *reinterpret_cast<float4*>(dst + dn * 16) = r4;
          }
        }
      }
      // ---- pass KV: the same tile as KEYS against the query pair (0, 1)
      {
        f16x8 kh[2], kl[2], vh[2], vl[2];
        row_op(fk, wave, kh, kl);
        row_op(fv, wave, vh, vl);
        f32x4 a, b, c, d;
        ta_rows_pair<PLANE>(fq, 0, kh, kl, a, b);        // S  [query][key]
        ta_rows_pair<PLANE>(fgr, 0, vh, vl, c, d);       // dP [query][key]
        float ds[8];
        f16x8 ph, pl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float2 sa = st[4 * fg + r], sb = st[16 + 4 * fg + r];
          float pa = __builtin_amdgcn_exp2f(fmaf(a[r], cexp, -sa.x)), pb = __builtin_amdgcn_exp2f(fmaf(b[r], cexp, -sb.x));
          if (!live) { pa = 0.f; pb = 0.f; }
          ds[r] = pa * (fmaf(c[r], cdp, -sa.y)) * 0.125f;
          ds[4 + r] = pb * (fmaf(d[r], cdp, -sb.y)) * 0.125f;
          const float ya = ta_opaque(pa * 1024.0f), yb = ta_opaque(pb * 1024.0f);
          const f16 ha = (f16)ya, hb = (f16)yb;
          ph[r] = ha; pl[r] = (f16)(ya - (float)ha);
          ph[4 + r] = hb; pl[4 + r] = (f16)(yb - (float)hb);
        }
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dn = 0; dn < 4; ++dn) { dk[dn] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dn] = dk[dn]; }
        ta_tr_chunk<PLANE>(fgr, 0, ph, pl, dv);          // dV^T[d][key] = dO^T P
        const int eb = max(20, ta_exp_of_max(ds));
        f16x8 sh, sl;
        ta_split_run(ds, eb, sh, sl);
        ta_tr_chunk<PLANE>(fq, 0, sh, sl, dk);           // dK^T[d][key] = Q^T dS
        if (live) {
          const float uk = 1.0f / sq, uv = 1.0f / (sg * 1024.0f);
          float* dst = dqkv + tok * ld + C + head * 64 + fg * 4;
#pragma unroll
          for (int dn = 0; dn < 4; ++dn) {
            const float4 k4 = make_float4(ldexpf(dk[dn][0], eb - 140) * uk, ldexpf(dk[dn][1], eb - 140) * uk,
                                          ldexpf(dk[dn][2], eb - 140) * uk, ldexpf(dk[dn][3], eb - 140) * uk);
            const float4 v4 = make_float4(dv[dn][0] * uv, dv[dn][1] * uv, dv[dn][2] * uv, dv[dn][3] * uv);
            am = fmaxf(am, fmaxf(fmaxf(fabsf(k4.x), fabsf(k4.y)), fmaxf(fabsf(k4.z), fabsf(k4.w))));
            am = fmaxf(am, fmaxf(fmaxf(fabsf(v4.x), fabsf(v4.y)), fmaxf(fabsf(v4.z), fabsf(v4.w))));
            *reinterpret_cast<float4*>(dst + dn * 16) = k4;
            *reinterpret_cast<float4*>(dst + C + dn * 16) = v4;
          }
        }
      }
    }
  }
  if (amax_out) ta_block_amax<NW>(am, amax_out, part);
}

struct TAOperand { f16* op; int T, Tp; float* unscale; };   // forward only: the output also as the next Linear's operand rows
template <int NKT>
int ta_launch(int which, const float* qkv, const float* o, const float* dout, float* out, float* dqkv, void* stats, int n_seq,
              SeqMap map, int C, int heads, const unsigned* amax_qkv, const unsigned* amax_do, unsigned* amax_out,
              hipStream_t st, TAOperand po = {nullptr, 0, 0, nullptr}) {
  constexpr int NK = 16 * NKT, NW = NKT <= 2 ? 2 : (NKT <= 4 ? 4 : 8);
  constexpr int NKC = NKT > 8 ? 8 : NKT;               // forward: key tiles per LDS chunk
  // pass Q: chunked the same way, registers held to 128 (18 dwords of scratch per lane) for two workgroups per CU: 96 -> 92 us per
  // temporal launch.  pass KV (215 registers; at 128 it spills 41 dwords: 100 -> 122 us; chunked at its own register count: 102) keeps
  // whole sequences of Q / dO in LDS and one workgroup per CU.
  // Sequences longer than 256 tokens (training at `-f 351`, reference common/arguments.py:58): every kernel is chunked, so nothing
  // holds a whole sequence -- pass KV then takes its queries in chunks of 128 too (at its own register count: one workgroup per CU).
  constexpr int NKC_KV = NKT > 16 ? 8 : NKT;
  const size_t lds = (size_t)4 * NKC * 16 * 128 + 64, lds_kv = (size_t)4 * NKC_KV * 16 * 128 + NKC_KV * 16 * 8 + 64, lds_fwd = lds;
  (void)NK;
  static PerDeviceOnce once;
  if (once.get([&](int) {
        return d3dp_lds_opt_in(reinterpret_cast<const void*>(tattn_fwd_kernel<NKT, NW, NKC>), 160 * 1024) < 0 ? -3
               : d3dp_lds_opt_in(reinterpret_cast<const void*>(tattn_bwd_q_kernel<NKT, NW, NKC, 4>), 160 * 1024) < 0 ? -3
               : d3dp_lds_opt_in(reinterpret_cast<const void*>(tattn_bwd_kv_kernel<NKT, NW, NKC_KV, 2>), 160 * 1024);
      }) < 0) return -3;
  const int tiles = (map.n_tok + 15) / 16, groups = (tiles + NW - 1) / NW;
  const int n_work = n_seq * heads * groups;
  const dim3 grid(n_work < 2048 ? n_work : 2048), blk(NW * 64);   // (<= 2048 absmax atomics per launch)
  TAStat* s = reinterpret_cast<TAStat*>(stats);
  if (which == 0)
    hipLaunchKernelGGL((tattn_fwd_kernel<NKT, NW, NKC>), grid, blk, lds_fwd, st, qkv, out, s, map, C, heads, groups, n_work, amax_qkv, amax_out,
                       po.op, po.T, po.Tp, po.unscale);
  else if (NKT == 2) {
    // sequences of at most 32 tokens: both passes in ONE kernel (tattn_bwd_small_kernel); under the per-kernel profile it is timed as
    // pass Q and the pass-KV call is empty
    if (which != 3) {
      const dim3 g2(n_seq * heads < 4096 ? n_seq * heads : 4096);
      hipLaunchKernelGGL(tattn_bwd_small_kernel, g2, dim3(128), 0, st, qkv, o, dout, dqkv, (const TAStat*)s, map, C, heads, n_seq * heads,
                         amax_qkv, amax_do, amax_out);
    }
  } else {                                             // which: 1 = both passes, 2 = pass Q alone, 3 = pass KV alone (needs pass Q's D_i)
    if (which != 3)
      hipLaunchKernelGGL((tattn_bwd_q_kernel<NKT, NW, NKC, 4>), grid, blk, lds, st, qkv, o, dout, dqkv, s, map, C, heads, groups, n_work,
                         amax_qkv, amax_do, amax_out);
    if (which != 2)
      hipLaunchKernelGGL((tattn_bwd_kv_kernel<NKT, NW, NKC_KV, 2>), grid, blk, lds_kv, st, qkv, dout, dqkv, (const TAStat*)s, map, C, heads, groups,
                         n_work, amax_qkv, amax_do, amax_out);
  }
  return 0;
}

int ta_dispatch(int which, const float* qkv, const float* o, const float* dout, float* out, float* dqkv, void* stats, int n_seq,
                SeqMap map, int C, int heads, const unsigned* amax_qkv, const unsigned* amax_do, unsigned* amax_out,
                hipStream_t st, TAOperand po = {nullptr, 0, 0, nullptr}) {
  const int n = map.n_tok;
  if (C / heads != 64 || C % 4 != 0 || n < 1 || n > 1024 || !amax_qkv || !stats) return -2;
  if (po.op && (po.Tp < po.T || !po.unscale || C % 32 != 0)) return -1;
#define TA_CASE(NKT_) return ta_launch<NKT_>(which, qkv, o, dout, out, dqkv, stats, n_seq, map, C, heads, amax_qkv, amax_do, amax_out, st, po);
  if (n <= 32) { TA_CASE(2) }
  if (n <= 64) { TA_CASE(4) }
  if (n <= 128) { TA_CASE(8) }
  if (n <= 256) { TA_CASE(16) }
  if (n <= 512) { TA_CASE(32) }
  TA_CASE(64)
#undef TA_CASE
}

}  // namespace

size_t d3dp_train_attn_x2_stats_bytes(int n_seq, int n_tok, int heads) { return (size_t)n_seq * heads * n_tok * sizeof(TAStat); }

// out [T, C] = softmax(q k^T / 8) v per (sequence, head); stats: n_seq x heads x n_tok (L, D) pairs for the backward pass;
// amax_qkv: absmax slot of the whole qkv tensor (the qkv Linear's epilogue); amax_out (optional): absmax slot of `out`
// op (optional): the output also as split operand rows [Tp][2 C] (rows T .. Tp - 1 zero; T = all token rows of the batch) at the
// scale of q / k / v, op_unscale[0] = 1 / scale, amax_out[0] = the bound (the qkv absmax) -- see tattn_fwd_kernel
int d3dp_train_attn_x2_fwd(const float* qkv, float* out, void* stats, int n_seq, SeqMap map, int C, int heads,
                           const unsigned* amax_qkv, unsigned* amax_out, hipStream_t st, void* op, int T, int Tp,
                           float* op_unscale) {
  if (!qkv || !out) return -1;
  return ta_dispatch(0, qkv, nullptr, nullptr, out, nullptr, stats, n_seq, map, C, heads, amax_qkv, nullptr, amax_out, st,
                     TAOperand{(f16*)op, T, Tp, op_unscale});
}
// dqkv [T, 3C] = the gradient of (q | k | v) given dout [T, C]; o: the forward output; amax_do: absmax slot of dout;
// amax_out (optional): absmax slot of dqkv (both passes add to it)
// (part: 0 = both passes, 1 = pass Q alone (dq, D_i), 2 = pass KV alone (dk, dv; after pass Q): the library's per-kernel
//  profile times them apart)
int d3dp_train_attn_x2_bwd(const float* qkv, const float* o, const float* dout, float* dqkv, void* stats, int n_seq, SeqMap map,
                           int C, int heads, const unsigned* amax_qkv, const unsigned* amax_do, unsigned* amax_out,
                           hipStream_t st, int part) {
  if (!qkv || !o || !dout || !dqkv || !amax_do || part < 0 || part > 2) return -1;
  return ta_dispatch(1 + part, qkv, o, dout, nullptr, dqkv, stats, n_seq, map, C, heads, amax_qkv, amax_do, amax_out, st);
}
