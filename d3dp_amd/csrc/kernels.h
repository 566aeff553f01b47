// Internal launcher interface between the C-ABI layer (capi.hip) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

// "Once per DEVICE" cache of a launcher's set-up (the > 64 KiB dynamic-LDS opt-in of a kernel, which HIP records per device,
// or a number derived from the device's CU count): one process may drive several devices -- nn.DataParallel callers,
// one thread per device (reference main.py:242-248) -- so neither a process-wide `static bool` nor an unlocked one is right.
// get(fn): fn(dev) runs the first time the CURRENT device is seen and returns a value > 0 to cache (<= 0: error, not cached).
struct PerDeviceOnce {
  static constexpr int kMaxDev = 64;
  std::mutex mu;
  int val[kMaxDev] = {0};
  template <class F> int get(F&& fn) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return -3;
    std::lock_guard<std::mutex> lock(mu);
    if (val[dev] <= 0) {
      const int v = fn(dev);
      if (v <= 0) return v < 0 ? v : -3;
      val[dev] = v;
    }
    return val[dev];
  }
};
// multiProcessorCount of device `dev` (> 0) or -3
inline int d3dp_cu_count(int dev) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount < 1) return -3;
  return prop.multiProcessorCount;
}
// opt one kernel in to `bytes` of dynamic LDS on the current device: 1 or -3
inline int d3dp_lds_opt_in(const void* kern, int bytes) {
  return hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 1 : -3;
}

enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_PARTIAL = 3, EPI_QKV_PACK = 4, EPI_RESID_LN = 5, EPI_GELU_LN = 6 };

// ---- gemm.hip ----------------------------------------------------------------------------------
int d3dp_launch_linear_bf16_stream(int epi, int out_f32, const void* A, const void* W, const float* bias, void* out,
                                   int M, int N, int K, hipStream_t st);
int d3dp_launch_linear_bf16x3(int epi, const void* A3, const void* W3, const float* bias, float* outf, void* out3, int M,
                              int N, int K, hipStream_t st);
void d3dp_launch_split3(const float* src, void* dst, size_t n, hipStream_t st);
// ---- gemm_x2.hip (EXACT mode: split-fp16 operands, three fp16-MFMA passes) ------------------------
// EPI_RESID_LN: outf += ... as EPI_RESID, plus out2 = the SUM's split-fp16 operand (h2i, un-normalised) and aux =
//   [M][ceil(N/64)][2] (mean, M2) of every 64-column slice of every row (the next LayerNorm's statistics, in pieces);
// EPI_GELU_LN: a LayerNorm FOLDED into the Linear: A2 = un-normalised rows, W2 = W . diag(gamma), bias = [c2 | c1] (2 N floats:
//   c2 = W beta + b, c1 = row sums of W diag(gamma)), aux = [M + 256][2] (mean, rstd) per row:
//   out2 = split(GELU(rstd (A W'^T - mean c1) + c2)).
// `unscale` = 1 / (scale of the A operand x scale of the W operand); `oscale` = the power of two the plane outputs (EPI_GELU*,
// k / v of EPI_QKV_PACK, the operand copy of EPI_RESID_LN) are multiplied by before the hi / lo split (kActScale unless the
// proven range of that operand asks for less, capi.hip)
int d3dp_launch_linear_f16x2(int epi, const void* A2, const void* W2, const float* bias, float unscale, float oscale,
                             float* outf, void* out2, float* aux, unsigned* flag, int M, int N, int K, hipStream_t st,
                             int skew_d = 0, int pingpong = 0);
// pingpong: the two-team form of the kernel (gemm_f16x2_pp_kernel: one wave of a SIMD reads its fragments while the other
// multiplies); same results bit for bit (the k order and the arithmetic are unchanged).  OR-ed with X2_TILES_LAST_TO_FIRST the
// plain kernel walks its tiles from the last row of tiles to the first (the rows a following row kernel reads first are then the
// ones written last); same tiles, same results bit for bit
constexpr int X2_TILES_LAST_TO_FIRST = 0x100;
// skew_d (1, 2, 4): the row-class skewed schedule of gemm_x2.hip for EPI_QKV_PACK / EPI_GELU -- a tile's epilogue leaves under the
// next tile's k-loop, D k-steps per 16-row class; d3dp_x2_skew_applies says whether the launcher will use it for a shape
bool d3dp_x2_skew_applies(int epi, int M, int N, int K, int skew_d, int n_cu);
// whether gemm_x2.hip was built with -DD3DP_X2_VARIANTS=1 (the ping-pong / wide / skewed kernels and the norm2-folding
// epilogues: measured-negative experiments, not part of the product library)
bool d3dp_x2_variants_built();
// out[0] = max over rows n of  sum_k |W[n,k]| in_k + |bias[n]|,  out[1] = max_k in_k,  in_k = sq |gamma_k| + |beta_k|: the
// magnitude bound of a Linear fed by a LayerNorm over K channels (sq = sqrt(K - 1): |LN(x)_k| <= sq |gamma_k| + |beta_k| for
// ANY x), as the bit patterns of non-negative floats (integer max == float max; `out` pre-zeroed)
void d3dp_launch_rowbound(const float* W, const float* gamma, const float* beta, const float* bias, int N, int K,
                          unsigned* out, hipStream_t st);
// rowstat[M][2] = (mean, 1 / sqrt(var + eps)) from the slice statistics EPI_RESID_LN wrote (S = ceil(C / 64) slices of 64)
void d3dp_launch_ln_combine(const float* slices, float* rowstat, int M, int C, float eps, hipStream_t st);
// Wp[n][k] = W[n][k] gamma[k];  c12[0..N) = sum_k W[n][k] beta[k] + bias[n],  c12[N..2N) = sum_k Wp[n][k]
void d3dp_launch_fold_ln(const float* W, const float* gamma, const float* beta, const float* bias, float* Wp, float* c12,
                         int N, int K, hipStream_t st);
void d3dp_launch_split2(const float* src, void* dst, size_t n, float scale, hipStream_t st);
// ---- the training step's split-fp16 Linear (gemm_x2.hip, gemm_f16x2_dyn_kernel): operand scales live on the device ----
// out_z[M, N] (z = 0 .. Z-1, M N floats apart) = A2[M][2 Kfull] . W2[N][2 Kfull]^T over k-chunk z, x dynA[0] x dynW[0], + bias
// (bias may be null).  Kfull % (32 Z) == 0, N % 4 == 0.
// amax_out (optional, Z == 1): absmax slot of the output (bit pattern of a non-negative float, pre-zeroed; one atomicMax per workgroup)
// (amax_pos: the largest POSITIVE output value instead of the largest magnitude)
// (rem_blocks: Z == 1, M > 256, Kfull % 512 == 0 -- the M mod 256 rows behind the last whole tile go to 16 x 64 blocks spread over
//  all workgroups instead of a last row of mostly empty 256 x 128 tiles; ignored where the conditions do not hold)
int d3dp_launch_linear_f16x2_dyn(const void* A2, const void* W2, const float* bias, const float* dynA, const float* dynW,
                                 float* out, int M, int N, int Kfull, int Z, hipStream_t st, unsigned* amax_out = nullptr,
                                 int amax_pos = 0, int rem_blocks = 0);
// drow [Rpad][2 C] = split(GELU(src [R][C])) (rows R .. Rpad - 1 zero) at the scale max(GELU(pmax), 0.17) asks for (pmax: the
// largest positive value of src, left by the Linear that produced it); writes that bound to amax_out[0] and 1 / scale to unscale[0]
int d3dp_launch_gelu_rowprep(const float* src, void* drow, int R, int Rpad, int C, const unsigned* pmax, unsigned* amax_out,
                             float* unscale, hipStream_t st);
// src [R][C] fp32 -> h2i [R][2 Cpad] (zero columns behind C) at the power of two the tensor's absmax (amax[0], bits of a
// float >= 0, d3dp_launch_absmax) asks for; unscale[0] = 1 / that scale
void d3dp_launch_split2_dyn(const float* src, void* dst, int R, int C, int Cpad, const unsigned* amax, float* unscale,
                            hipStream_t st);
// the TRANSPOSE as a split operand: src [R][C] -> h2i [C][2 Rpad] (zero behind R; Rpad % 32 == 0)
void d3dp_launch_split2_t_dyn(const float* src, void* dst, int R, int C, int Rpad, const unsigned* amax, float* unscale,
                              hipStream_t st);
// out[i] = sum over z of part[z n + i], z ascending
// src [R][C] -> row form [R][2 C], transposed form [C][2 Rpad] and (colpart != null) its column sums as D3DP_DYPREP_ROWS
// partial rows of C floats (every row written; summed by d3dp_train_reduce_many) in one pass (gemm_x2.hip)
constexpr int D3DP_DYPREP_ROWS = 96;
int d3dp_launch_dyprep(const float* src, void* drow, void* dcol, float* colpart, int R, int C, int Rpad, const unsigned* amax,
                       float* unscale, hipStream_t st);
// the row form alone as a streaming pass: src [R][C] -> drow [Rpad][2 C] (rows R .. Rpad - 1 zero) + optional column sums as *rows
// (<= D3DP_ROWPREP_ROWS) partial rows of C floats; C % 32 == 0, C <= 1536
constexpr int D3DP_ROWPREP_ROWS = 512;
// (mask: optional per-sample scales applied to the rows first -- sample = r / J (axis 0) or (r / (F J)) J + r % J (axis 1))
// (gelu_pre: optional [R][C]; the operand is src x gelu'(gelu_pre) and amax[0] holds the absmax of SRC, see rowprep_kernel)
int d3dp_launch_rowprep(const float* src, void* drow, float* colpart, int* rows, int R, int Rpad, int C, const unsigned* amax,
                        float* unscale, hipStream_t st, const float* mask = nullptr, int axis = 0, int F = 1, int J = 1,
                        const float* gelu_pre = nullptr);
// drow = split(LayerNorm(src; w, b, eps)): the operand of a Linear fed by a LayerNorm, from the LayerNorm's INPUT (amax: the
// absmax of the LayerNorm's output, left by the kernel that produced src); C <= 512
int d3dp_launch_rowprep_ln(const float* src, const float* w, const float* b, float eps, void* drow, int R, int Rpad, int C,
                           const unsigned* amax, float* unscale, hipStream_t st);
// wgrad straight from the ROW forms (gemm_f16x2_tn_kernel): out_z[N, K] = sum_{t in chunk z} A2[t][n] W2[t][k] x dynA x dynW,
// Tp = Z NKz 32 rows per operand (rows beyond the real ones zero).  d3dp_tn_applies: N % 256 == 0 and K % 128 == 0.
bool d3dp_tn_applies(int N, int K);
int d3dp_launch_linear_f16x2_tn(const void* A2, const void* W2, const float* dynA, const float* dynW, float* out, int N, int K,
                                int Tp, int Z, hipStream_t st);
// ... several such products over the same Tp rows in one launch (the weight gradients of a block): one merged tile list, one Z
constexpr int D3DP_TN_MAX = 4;
struct D3dpTnProduct {
  const void* A2; const void* W2;                      // [Tp][2 N] and [Tp][2 K] row forms
  const float* dynA; const float* dynW;
  float* out;                                          // Z partial products, N K floats apart
  int N, K;
  int tiles_k, tile0;                                  // (filled by the launcher)
};
struct D3dpTnTable { D3dpTnProduct p[D3DP_TN_MAX]; int n; };
int d3dp_launch_linear_f16x2_tn_many(const D3dpTnProduct* prods, int n, int Tp, int Z, hipStream_t st);
struct D3dpSumTable { const float* part[D3DP_TN_MAX]; float* out[D3DP_TN_MAX]; size_t n4[D3DP_TN_MAX]; int Z; int n; };
void d3dp_launch_sum_partials_many(const D3dpSumTable& tb, hipStream_t st);   // out_p[i] = sum_z part_p[z n_p + i], z ascending
// the training step's weight operands in three launches (gemm_x2.hip): absmax -> slot, rows form [N][2 K] at rows_base + 2 off
// halves, transposed form [K][2 N] at cols_base + 2 off halves, unscale[slot] = 1 / scale
constexpr int D3DP_WPREP_MAX = 64;
struct D3dpWPrepItem { const float* w; int N, K, slot, pad; size_t off; };
struct D3dpWPrepTable { D3dpWPrepItem it[D3DP_WPREP_MAX]; int n; };
int d3dp_launch_wprep(const D3dpWPrepTable& tb, void* rows_base, void* cols_base, unsigned* amax, float* unscale, hipStream_t st);
void d3dp_launch_sum_partials(const float* part, float* out, size_t n, int Z, hipStream_t st);
void d3dp_launch_sum_partials_bias(const float* part, const float* bias, float* out, size_t n, int N, int Z, hipStream_t st,
                                   unsigned* amax = nullptr, int amax_pos = 0);
void d3dp_launch_absmax(const float* src, size_t n, unsigned* out, hipStream_t st);
// flag[0] |= 1 if any of x[0..n) is inf / nan
void d3dp_launch_nonfinite_flag(const float* x, size_t n, unsigned* flag, hipStream_t st);
int d3dp_launch_linear_f32_splitk(const float* A, const float* W, float* out, int M, int N, int K, hipStream_t st, float* part,
                                  size_t part_floats);
int d3dp_launch_linear_f32(int epi, const float* A, const float* W, const float* bias, float* out, int M, int N,
                           int K, hipStream_t st);

// ---- attention.hip -----------------------------------------------------------------------------
// qkv: [T, 3C] (q | k | v, each head-major hd-minor), out: [T, C].  A "sequence" s of length n_tok has
// token index  tok(s, i) = (s / inner) * outer_stride + (s % inner) * inner_stride + i * tok_stride.
//   spatial : n_tok = J, inner = 1,  outer_stride = J,   inner_stride = 0, tok_stride = 1   (s = bh*F + f)
//   temporal: n_tok = F, inner = J,  outer_stride = F*J, inner_stride = 1, tok_stride = J   (s = bh*J + n)
struct SeqMap { int n_tok, inner, outer_stride, inner_stride, tok_stride; };
// (any sequence length; amax: optional absmax slot of an fp32 output, one atomicMax per workgroup)
int d3dp_launch_attn_rows(int act_bf16, const void* qkv, void* out, int n_seq, SeqMap map, int C, int heads,
                          hipStream_t st, unsigned* amax = nullptr);
int d3dp_launch_attn_temporal_bf16(const void* qkv, void* out, int n_seq, SeqMap map, int C, int heads,
                                   hipStream_t st);
int d3dp_launch_attn_temporal_f32(int act, const void* qkv, void* out, int n_seq, SeqMap map, int C, int heads,
                                  hipStream_t st);
// split-fp16 attention: `qkv` = PACKED rows of 12 C bytes (q fp32 | k hi | k lo | v hi | v lo), written by the qkv Linear
// with EPI_QKV_PACK or from fp32 rows by d3dp_launch_qkv_pack_x2
// `act_scale`: the power of two the k / v planes were written at (q and, for plane output, o use the same)
int d3dp_launch_attn_x2(int act, int axis, const void* qkv, void* out, int n_seq, SeqMap map, int C, int heads,
                        float act_scale, hipStream_t st);
void d3dp_launch_qkv_pack_x2(const float* src, void* dst, size_t T, int C, float act_scale, hipStream_t st);
int d3dp_launch_attn_spatial_bf16(const void* qkv, void* out, int n_seq, SeqMap map, int C, int heads, hipStream_t st);

// ---- pointwise.hip -----------------------------------------------------------------------------
int d3dp_launch_time_mlp(const int64_t* t, const float* freq, const float* w1, const float* b1, const float* w2,
                         const float* b2, float* temb, int B, int C, hipStream_t st);
// x[T,C] = embed(x2d, x3d) + spos + temb ;  xn = LN1(x)
int d3dp_launch_embed_ln(int act_bf16, const float* x2d, const float* x3d, const float* temb, const float* ew,
                         const float* eb, const float* spos, const float* lnw, const float* lnb, float eps, float* x,
                         void* xn, int seq0, int n_seq, int H, int F, int J, int C, hipStream_t st, int SP = 0);
// (SP: rows per sequence in x / xn, >= F J; 0 = F J.  Rows F J .. SP - 1 of every sequence are finite filler.)
// xn = LN(x)
// (residual adds: ln normalises x + yadd (writing the sum back only if write_x); ln2 / head form (x + yadd0) + yadd;
//  yadd has the activation type: bf16 in FAST mode, fp32 in EXACT mode)
int d3dp_launch_ln(int act_bf16, float* x, const void* yadd, int write_x, const float* w, const float* b, float eps,
                   void* xn, int T, int C, hipStream_t st);
// x = LN_a(x) (+ pos[f]) in place ; xn = LN_b(x)   (shared Spatial/Temporal norm fused with the next block's norm1)
int d3dp_launch_ln2(int act_bf16, float* x, const void* yadd0, const void* yadd, const float* wa, const float* ba, const float* pos,
                    const float* wb, const float* bb, float eps, void* xn, int T, int C, int F, int J, hipStream_t st,
                    int SP = 0);
// out[T,3] = Linear(LN_head(LN_a(x)))
int d3dp_launch_head(int act_bf16, const float* x, const void* yadd0, const void* yadd, const float* wa, const float* ba, float eps_a, const float* wh,
                     const float* bh, float eps_h, const float* w, const float* b, float* out, int T, int C,
                     hipStream_t st, int FJ = 0, int SP = 0);   // (rows at pitch SP per sequence -> compact out rows; 0 = compact in)

// ---- sampler.hip -------------------------------------------------------------------------------
int d3dp_launch_ddim_pre(const float* img, float* xt2, const int* perm, float scale, int B, int per_b, int J,
                         hipStream_t st);
int d3dp_launch_ddim_post(const float* pred2, const float* img, const float* noise, const int* perm, float scale,
                          double sqrt_recip, double sqrt_recipm1, float c_xstart, float c_noise, float sigma,
                          int last, float* x_start, size_t xs_bstride, float* img_next, int B, int per_b, int J,
                          hipStream_t st);
int d3dp_launch_q_sample(const float* x0, const float* noise, const double* a, const double* b, float scale,
                         float* out, int B, int per_b, hipStream_t st);

// ---- jpma.hip ----------------------------------------------------------------------------------
int d3dp_launch_jpma(const float* pred, const float* traj, const float* cam, const float* gt2d, const float* gt3d,
                     float* agg, int* sel, float* err_sel, float* err_min, float* win, float* jbest, float* mean,
                     int h_offset, int B, int K, int H, int F, int J, int root_joint, int linear, hipStream_t st,
                     int h_inner = 0, size_t outer_stride = 0);   // (h_inner, outer_stride): see jpma_kernel; 0 = contiguous H

// ---- capi.hip helpers shared with caller.hip ---------------------------------------------------------------------
int d3dp_set_error(int code, const char* msg);        // records the message for d3dp_last_error(), returns code
int d3dp_check_launch(const char* what);              // hipGetLastError -> status
extern "C" int d3dp_clip_count(int32_t n, int32_t F);

// ---- train.hip (training step, fp32) -------------------------------------------------------------------------
// (amax: optional absmax slot of a kernel's result -- bit pattern of a non-negative float, pre-zeroed, one atomicMax per
//  workgroup -- so that the split-fp16 Linear that consumes the result needs no absmax pass of its own)
// x_out = x_in + mask[sample] y ; xn = LN(x_out)  (xn may be null: only amax, the absmax of LN(x_out), is produced)
int d3dp_train_add_mask_ln(const float* x_in, const float* y, const float* mask, int axis, int F, int J, const float* w,
                           const float* b, float eps, float* x_out, float* xn, unsigned* amax, int T, int C, hipStream_t st,
                           void* op = nullptr, int Tp = 0, float* op_unscale = nullptr);
// x_out = x_in + mask[sample] y ; x_next = LN_a(x_out) (+ pos[f]) ; xn = LN_b(x_next)  (wb null: no second norm; xn null with
// wb given: LN_b's output is not stored, only its absmax)
int d3dp_train_add_mask_ln2(const float* x_in, const float* y, const float* mask, int axis, int F, int J, const float* wa,
                            const float* ba, float eps_a, const float* pos, const float* wb, const float* bb, float eps_b,
                            float* x_out, float* x_next, float* xn, unsigned* amax, int T, int C, hipStream_t st, void* op = nullptr,
                            int Tp = 0, float* op_unscale = nullptr);
int d3dp_train_ln_pos(const float* x, const float* w, const float* b, float eps, const float* pos, int F, int J, float* y,
                      int T, int C, hipStream_t st);
// LayerNorm backward, one (xa == null) or two chained LayerNorms (y = LN_b(xb), xb = LN_a(xa) (+ pos)) in one pass:
//   g = LN_b-backward(dy) + dres (g_out: optional copy, two-norm form only);  dx = two ? LN_a-backward(g) : g;
//   dxm (optional) = mask[sample] dx, amax (optional) its absmax -- dy and dxm may alias.
// part_b / part_a: d3dp_train_ln_bwd_blocks(T) rows of [dgamma | dbeta] (2 C floats) per LayerNorm, to be summed in order by
// d3dp_train_reduce_many (no float atomics anywhere in the backward pass: bit-reproducible gradients).
constexpr int D3DP_LN_BWD_BLOCKS = 512;
int d3dp_train_ln_bwd_blocks(int T);
int d3dp_train_ln_bwd(const float* dy, const float* xb, const float* wb, float eps_b, const float* dres, float* g_out,
                      const float* xa, const float* wa, float eps_a, float* dx, const float* mask, int axis, int F, int J,
                      float* dxm, unsigned* amax, float* part_b, float* part_a, int T, int C, hipStream_t st);
// dst[i] (+)= sum_{p < count} part[p stride + i] in a fixed order, up to D3DP_REDUCE_MAX destinations per launch
constexpr int D3DP_REDUCE_MAX = 80;
struct D3dpReduceItem { const float* part; float* dst; unsigned n, count, stride, accumulate; };
struct D3dpReduceTable { D3dpReduceItem it[D3DP_REDUCE_MAX]; int count; };
int d3dp_train_reduce_many(const D3dpReduceTable& tb, hipStream_t st);
// (amax: optional absmax slot of the result, see train.hip block_amax_commit)
int d3dp_train_gelu_fwd(const float* x, float* y, size_t n, unsigned* amax, hipStream_t st);
int d3dp_train_gelu_bwd(const float* dh, const float* x, float* dpre, size_t n, unsigned* amax, hipStream_t st);
// column sums as *rows (<= max_rows <= 512) partial rows of C floats (summed by d3dp_train_reduce_many)
int d3dp_train_colsum(const float* in, float* part, int* rows, int max_rows, int T, int C, hipStream_t st);
// zero up to D3DP_ZERO_MAX fp32 buffers in one launch (the small gradient buffers of a backward pass)
constexpr int D3DP_ZERO_MAX = 192;
struct D3dpZeroTable { float* p[D3DP_ZERO_MAX]; unsigned n[D3DP_ZERO_MAX]; int count; };
int d3dp_train_zero_many(const D3dpZeroTable& tb, hipStream_t st);
// grouped row sums (mode 0: per joint, 1: per frame, 2: per batch element) as `slices` partial tables of groups x C floats
int d3dp_train_groupsum(const float* in, float* out, int T, int C, int mode, int F, int J, int slices, hipStream_t st);
int d3dp_train_transpose_pad(const float* in, float* out, int R, int C, int Rpad, hipStream_t st);
size_t d3dp_train_attn_stats_bytes(int n_seq, int n_tok, int heads);
int d3dp_train_attn_bwd(const float* qkv, const float* o, const float* dout, float* dqkv, void* stats, int n_seq,
                        SeqMap map, int C, int heads, hipStream_t st);
constexpr int D3DP_EMBED_BWD_ROWS = 64;
// ---- train_attn.hip: the training step's attention on split-fp16 operands (head dim 64, <= 256 tokens per sequence) ----
// stats: d3dp_train_attn_x2_stats_bytes per attention (the forward leaves the log-sum-exp of every query for the backward pass);
// amax_qkv / amax_do: absmax slots of the whole qkv tensor / of dout (left by the Linears that produced them); amax_out
// (optional): absmax slot of the result.
size_t d3dp_train_attn_x2_stats_bytes(int n_seq, int n_tok, int heads);
int d3dp_train_attn_x2_fwd(const float* qkv, float* out, void* stats, int n_seq, SeqMap map, int C, int heads,
                           const unsigned* amax_qkv, unsigned* amax_out, hipStream_t st, void* op = nullptr, int T = 0, int Tp = 0,
                           float* op_unscale = nullptr);
int d3dp_train_attn_x2_bwd(const float* qkv, const float* o, const float* dout, float* dqkv, void* stats, int n_seq, SeqMap map,
                           int C, int heads, const unsigned* amax_qkv, const unsigned* amax_do, unsigned* amax_out,
                           hipStream_t st, int part = 0);
int d3dp_train_embed_bwd(const float* dx, const float* x2d, const float* x3d, float* part, int T, int C, hipStream_t st);
int d3dp_train_head_linear(const float* z, const float* w, const float* b, float* out, int T, int C, hipStream_t st);
int d3dp_train_head_bwd(const float* g, const float* z, const float* w, float* dz, float* part, int* rows, int T, int C,
                        hipStream_t st);
// the time-embedding MLP of the training step (one wave per output unit, two launches); hidden: scratch of B x 2 C floats
int d3dp_train_time_mlp(const int64_t* t, const float* freq, const float* w1, const float* b1, const float* w2, const float* b2,
                        float* hidden, float* temb, int B, int C, hipStream_t st);
int d3dp_train_time_mlp_bwd(const int64_t* t, const float* freq, const float* w1, const float* b1, const float* w2,
                            const float* dtemb, float* dw1, float* db1, float* dw2, float* db2, int B, int C,
                            hipStream_t st);

// train_g.hip: the training step's row kernels at a run-time width (any C <= 1024; the fp32 path of capi.hip for the widths
// train.hip does not instantiate).  The d3dp_train_* launchers above forward to these when d3dp_width_instantiated(C) is false.
inline bool d3dp_width_instantiated(int C) { return C == 64 || C == 128 || C == 256 || C == 512; }
int d3dp_train_g_add_mask_ln(const float* x_in, const float* y, const float* mask, int axis, int F, int J, const float* w,
                             const float* b, float eps, float* x_out, float* xn, int T, int C, hipStream_t st);
int d3dp_train_g_add_mask_ln2(const float* x_in, const float* y, const float* mask, int axis, int F, int J, const float* wa,
                              const float* ba, float eps_a, const float* pos, const float* wb, const float* bb, float eps_b,
                              float* x_out, float* x_next, float* xn, int T, int C, hipStream_t st);
int d3dp_train_g_ln_pos(const float* x, const float* w, const float* b, float eps, const float* pos, int F, int J, float* y, int T,
                        int C, hipStream_t st);
int d3dp_train_g_ln_bwd(const float* dy, const float* xb, const float* wb, float eps_b, const float* dres, float* g_out,
                        const float* xa, const float* wa, float eps_a, float* dx, const float* mask, int axis, int F, int J,
                        float* dxm, float* part_b, float* part_a, int T, int C, int blocks, hipStream_t st);
int d3dp_train_g_head_linear(const float* z, const float* w, const float* b, float* out, int T, int C, hipStream_t st);
int d3dp_train_g_time_mlp_bwd(const int64_t* t, const float* freq, const float* w1, const float* b1, const float* w2,
                              const float* dtemb, float* dw1, float* db1, float* dw2, float* db2, int B, int C, hipStream_t st);
