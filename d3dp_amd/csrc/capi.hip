// C-ABI layer of libd3dp_hip.so (see include/d3dp_hip.h): context, weight packing, the denoiser schedule
// (which kernels run in which order over which buffers) and per-kernel HIP-event timing.
//
// Data layout in HBM (per internal pass over `n` (clip,hypothesis) sequences, Tc = n*F*J tokens, token order
// (sequence, frame, joint), channels fastest):
//   x    [Tc, C]   fp32   residual stream -- stays fp32 in both modes
//   (EXACT mode: bufA / the MLP hidden are three bf16 planes of the fp32 value, qkv and y are fp32)
//   bufA [Tc, C]   act    normalised input of the next GEMM / attention output      (act = bf16 FAST, fp32 EXACT)
//   bufB [Tc, 3C]  act    qkv; reused as the [Tc, 2C] MLP hidden
//   y1,y [Tc, C]   act    outputs of the residual-feeding Linears (proj, fc2); the norm pair after the block forms
//                          (x + y1) + y in registers, so x is read and written once per block
// The reference keeps two physical layouts and transposes between them 16 times per call
// (mixste.py:244,270,274); here spatial and temporal attention both index the single layout by stride.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/d3dp_hip.h"
#include "common.h"
#include "kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e__ = (expr);                                                               \
    if (e__ != hipSuccess) return fail(D3DP_EHIP, "%s: %s", #expr, hipGetErrorString(e__)); \
  } while (0)

#define LAUNCH_TRY(expr)                                                          \
  do {                                                                            \
    int r__ = (expr);                                                             \
    if (r__ != 0) return fail(r__ == -1 ? D3DP_EINVAL : D3DP_ENOTSUP, "%s -> %d", #expr, r__); \
  } while (0)

}  // namespace

int d3dp_set_error(int code, const char* msg) { return fail(code, "%s", msg); }
int d3dp_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(D3DP_EHIP, "%s: %s", what, hipGetErrorString(e));
  return D3DP_OK;
}

namespace {

enum ProfClass { P_QKV = 0, P_PROJ, P_FC1, P_FC2, P_ATTN_S, P_ATTN_T, P_LN, P_LN2, P_EMBED, P_HEAD, P_TIME, P_OTHER,
                 // the training step (d3dp_train_forward / d3dp_train_backward)
                 T_LINEAR, T_WGRAD, T_ATTN_FWD_S, T_ATTN_FWD_T, T_ATTN_BQ_S, T_ATTN_BQ_T, T_ATTN_BKV_S, T_ATTN_BKV_T, T_OPERAND,
                 T_LN_FWD, T_LN_BWD, T_OTHER,
                 P_EMPTY };                            // event pairs with nothing between them: what a scope adds to a launch's time
static_assert(P_EMPTY + 1 == D3DP_PROFILE_CLASSES, "include/d3dp_hip.h: D3DP_PROFILE_CLASSES");
const char* kClassNames[D3DP_PROFILE_CLASSES] = {"gemm_qkv", "gemm_proj", "gemm_fc1", "gemm_fc2", "attn_spatial",
                                                 "attn_temporal", "layernorm", "norm_pair", "embed_ln", "head",
                                                 "time_mlp", "other",
                                                 "train_linear", "train_wgrad", "train_attn_fwd_spatial", "train_attn_fwd_temporal",
                                                 "train_attn_bwd_q_spatial", "train_attn_bwd_q_temporal", "train_attn_bwd_kv_spatial",
                                                 "train_attn_bwd_kv_temporal", "train_operand_pass", "train_ln_fwd", "train_ln_bwd",
                                                 "train_other", "event_pair_overhead"};

struct BlockDev {
  const float *n1w, *n1b, *n2w, *n2b, *qkv_b, *proj_b, *fc1_b, *fc2_b;
  const void *qkv_w, *proj_w, *fc1_w, *fc2_w;   // bf16 (FAST); EXACT: 2 fp16 planes (default), 3 bf16 planes or fp32
  float qkv_u = 1.f, proj_u = 1.f, fc1_u = 1.f, fc2_u = 1.f;   // EXACT f16x2: 2^-s of the per-matrix pre-scale 2^s
  const float* fc1_c12 = nullptr;   // fold_ln: [c2 | c1] of norm2 folded into fc1 (fc1_w then holds W diag(gamma)); see run_block
  // EXACT f16x2: the power-of-two scales of this block's DATA-dependent split-fp16 operands -- q / k / v and the attention
  // output (s_kv), the MLP hidden (s_h) -- chosen at d3dp_set_weights from the range the weights PROVE for them: 2^4 when
  // the bound is below 4094, the largest smaller power of two that keeps bound x scale below fp16's 65504 otherwise
  // (LayerNorm outputs always use 2^4: their bound is sqrt(C-1) |gamma| + |beta|).
  float s_kv = kActScale, s_h = kActScale;
};

__global__ void to_bf16_kernel(const float* __restrict__ s, bf16* __restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) d[i] = (bf16)s[i];
}

void launch_to_bf16(const float* s, void* d, size_t n, hipStream_t st) {
  const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(to_bf16_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, s, (bf16*)d, n);
}

constexpr size_t kAlign = 256;
size_t align_up(size_t v) { return (v + kAlign - 1) / kAlign * kAlign; }

}  // namespace

struct d3dp_ctx {
  d3dp_cfg cfg{};
  int device = 0;
  bool weights_set = false;
  float range_bound = 0.f;       // d3dp_exact_range_bound (EXACT split-fp16 only)
  size_t ln_slice_floats = 0;    // fold_ln: floats of the slice-statistics part of the LN scratch (set per d3dp_denoise call)
  unsigned* d_flag = nullptr;    // device word: bit 0 = a d3dp_denoise output held inf / nan (d3dp_status)
  char* arena = nullptr;
  size_t arena_bytes = 0;
  const float *spos = nullptr, *tpos = nullptr, *ew = nullptr, *eb = nullptr, *freq = nullptr, *t1w = nullptr,
              *t1b = nullptr, *t3w = nullptr, *t3b = nullptr, *snw = nullptr, *snb = nullptr, *tnw = nullptr,
              *tnb = nullptr, *hnw = nullptr, *hnb = nullptr, *hw = nullptr, *hb = nullptr;
  std::vector<BlockDev> ste, tte;
  // profiling
  bool prof = false;
  struct Ev { hipEvent_t a, b; int cls; };
  std::vector<Ev> pool;
  size_t used = 0;
  int64_t counts[D3DP_PROFILE_CLASSES] = {0};
  double total_ms[D3DP_PROFILE_CLASSES] = {0};

  bool fast() const { return cfg.mode == D3DP_MODE_FAST; }
  // EXACT mode runs its Linears on split-fp16 operands (2 planes, 3 fp16-MFMA passes, gemm_x2.hip); activations that
  // feed a Linear are then two fp16 planes.  env D3DP_EXACT_IMPL=bf16x3 selects the round-1 six-pass split-bf16 kernels
  // and =f32 the plain fp32-MFMA kernels (bitwise an fp32 fmaf chain) -- both kept as cross-checks.
  int exact_impl = 0;   // 0 = f16x2, 1 = bf16x3, 2 = f32
  int exact_impl_req = 0;        // what D3DP_EXACT_IMPL asked for; d3dp_set_weights moves an f16x2 context to bf16x3 when a
  bool impl_fallback = false;    // LayerNorm's own output bound leaves the split-fp16 range (see there)
  bool train() const { return cfg.mode == D3DP_MODE_TRAIN; }
  bool exact() const { return !fast() && !train(); }
  bool x2() const { return exact() && exact_impl == 0; }
  // split-fp16 attention kernels: head dim 64.  Up to 256 frames (every BASELINE configuration) the temporal kernel holds a whole
  // sequence's K / V images in LDS; longer clips (`-f 351`, reference common/arguments.py:58, mixste.py:172) take the flash form of the
  // same arithmetic (attention.hip attn_temporal_x2_long_kernel: keys in chunks of 128 under an online softmax; round 5 ran both
  // attentions of such clips on the chunked fp32 VALU row kernel, ten times the cost per FLOP).  D3DP_LONG_ATTN=rows keeps that
  // kernel as a cross-check (read in d3dp_create).
  bool long_rows = false;
  bool x2_attn() const { return x2() && cfg.channels / cfg.heads == 64 && (cfg.frames <= 256 || !long_rows); }
  // proj / fc2 add into the residual stream in their epilogue (x += ...), so the row kernels read x alone
  bool fold_resid() const { return x2() && fold; }
  bool fold = true;
  // norm2 (mixste.py:115) has no kernel of its own: proj's epilogue leaves x + proj(...) a second time as fc1's split-fp16
  // operand, UN-normalised, with (mean, M2) of each 64-column slice of each row; fc1 runs on W diag(gamma) and applies
  // rstd (. - mean c1) + c2 in its epilogue (gemm_x2.hip EPI_RESID_LN / EPI_GELU_LN).  OFF by default (D3DP_FOLD_LN=1 turns it
  // on): measured on configs[2], same box, interleaved -- 50.06 / 49.86 hypothesis-clips/s folded against 49.83 / 49.70 with the
  // row kernel: the 284 ms/step of the LayerNorm kernel come back as +160 ms in proj (its tile epilogue now also splits, stores
  // the operand and reduces the statistics with the matrix pipes idle) and +90 ms in fc1 (profiles/r03_fold_ln_ab.md).
  bool fold_ln() const { return fold_resid() && fold_ln_on && 2 * cfg.hidden <= 2048; }
  bool fold_ln_on = false;
  // The EXACT qkv / fc1 Linears run the SKEWED schedule of gemm_x2.hip (a tile's epilogue spread under the k-loop of the
  // next): the order in which a token row sums its k-steps then depends on (row within its pass) / 16 mod 4.  Every sequence
  // therefore starts at a multiple of 64 rows -- seq_pitch() rows per sequence, F J rounded up, the rest finite filler -- so
  // that order is a function of the token's index within its sequence alone and results stay bit-identical whatever the batch
  // composition, pass split or rank count (the H-sharding contract, tests/test_hip_parity.py::test_full_size_properties).
  bool skew() const { return x2_attn() && skew_d > 0 && cfg.channels >= 128 * skew_d; }   // (K = C >= 4 D k-steps of 32)
  int skew_d = 0;                // D3DP_X2_SKEW=1|2|4: k-steps a parked row class takes to leave.  OFF by default: measured
                                 // 3 % slower on the whole step (gemm_x2.hip, DESIGN.md 7: the Linear is bound by its vector-memory
                                 // instruction rate, and an epilogue's stores cost the same wherever they issue)
  int seq_pitch() const {
    const int fj = cfg.frames * cfg.joints;
    return (pad_override < 0 ? skew() : (pad_override > 0 && x2_attn())) ? (fj + 63) / 64 * 64 : fj;
  }
  int pad_override = -1;         // D3DP_SEQ_PAD=0|1: measurement switch (pad without the skewed schedule, or the reverse)
  bool train_x2 = true;          // D3DP_TRAIN_IMPL=f32: the training Linears on the fp32 matrix cores (round-1 path, cross-check)
  // The backward pass runs the weight-gradient products (one merged TN launch per block, or one launch per Linear, and the sum of
  // their partial tiles) on a second stream beside the rest of the backward pass: a block's launch goes on while the next block's
  // dgrad products, attention and row kernels run (two operand / partial-tile sets, X2Train::use_set).  Forked and joined with events
  // on the caller's stream (nothing synchronises the host); D3DP_TRAIN_OVERLAP=0 keeps one stream.  Worth 0.2 ms of a 21 ms step
  // since the weight gradients are one launch per block: kept because it costs nothing.
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_done[2] = {nullptr, nullptr};
  bool train_overlap = true;
  int train_overlap_sets = 2;    // D3DP_TRAIN_OVERLAP=1: one operand set (every operand pass waits for the product before it)
  bool train_gelu_in_prep = true;// D3DP_TRAIN_GELU=pass: d h_pre by a pass of its own (gelu_bwd_kernel) instead of inside the fc1 gradients' operand pass
  bool train_ln_direct = true;  // D3DP_TRAIN_LN_OPERAND=pass: the qkv / fc1 operands by an operand pass behind the LayerNorm (round 5) instead of by its producer
  bool train_wgrad_merged = true;// D3DP_TRAIN_WGRAD=each: a launch (and 32 MB of partial tiles) per weight gradient instead of one per block
  bool train_tail_blocks = true; // D3DP_TRAIN_TAIL=split: the round-4 handling of a batch's last T mod 256 rows (an extra round of tiles,
                                 // or a split-K launch of their own) instead of the 16 x 64 blocks at the end of the product's kernel
  int train_attn_x2 = 2;         // the training step's attention on the split-fp16 kernels of train_attn.hip: 2 = both axes (default),
                                 // 1 = D3DP_TRAIN_ATTN=x2t: the temporal axis only, 0 = D3DP_TRAIN_ATTN=f32: neither (the round-4
                                 // fp32 kernels -- fp32-MFMA temporal forward and backward, VALU spatial forward: the cross-check)
  int pingpong = 0;              // D3DP_X2_PP=1: the ping-pong form of the EXACT Linear (gemm_x2.hip; bit-identical results;
                                 // measured 1.5-2 % SLOWER on the whole step, gpurun c8); 2 = D3DP_X2_WIDE=1: the 256 x 256
                                 // tile form (bit-identical; ties with the default, profiles/r04_gemm_probes.md section 4)
  bool x3() const { return exact() && exact_impl == 1; }
  int act() const { return fast() ? 1 : (x3() ? 2 : (x2() ? 3 : 0)); }   // code understood by the row-wise launchers
  size_t act_size() const { return fast() ? 2 : (x3() ? 6 : 4); }    // bytes per element of a Linear-input activation
  size_t wide_size() const { return fast() ? 2 : 4; }                // bytes per element of bufB (qkv fp32 = 12C; hidden planes <= 12C)
  size_t y_size() const { return fast() ? 2 : 4; }
  // (clip, hypothesis) sequences per internal pass: 15 (61,965 tokens) measured best for FAST (working set near the
  // 256 MiB memory-side cache); EXACT is compute-bound in its Linears and gains 1.5 % from 30 (fewer, fuller tile rounds)
  int chunk() const { return cfg.chunk_seqs != 0 ? std::abs(cfg.chunk_seqs) : (exact() ? 31 : 15); }   // (< 0: uniform passes, for A/B)
  // EXACT split-fp16 Linears are persistent kernels over 256 x 128 tiles on n_cu workgroups: a pass over n sequences costs
  // sum over the four Linears of ceil(row_tiles(n) * column_tiles / n_cu) tile rounds x k-depth, and a partly filled last
  // round costs a full one (uniform chunks of 30 lose 4.4 % of the Linear time to it).  plan() splits `total` sequences
  // into passes of at most chunk() that minimise that sum (dynamic programme; any split gives bit-identical results).
  int n_cu = 0;
  std::vector<int> plan_cache;
  int plan_total = -1;
  const std::vector<int>& plan(int total) {
    if (total == plan_total) return plan_cache;
    const int cap = std::min(chunk(), total), SP = seq_pitch();
    plan_cache.clear();
    plan_total = total;
    if (!x2() || cfg.chunk_seqs < 0 || n_cu <= 0) {    // uniform passes (FAST, cross-check implementations)
      for (int s0 = 0; s0 < total; s0 += cap) plan_cache.push_back(std::min(cap, total - s0));
      return plan_cache;
    }
    // per Linear (qkv, proj, fc1, fc2): column strips, k-depth, and whether it runs the skewed schedule -- there a
    // workgroup owns the row tiles of one row group inside one strip (ceil(R / Q) tiles, Q = n_cu / strips row groups)
    // plus the flush of 3 D k-steps; in the plain schedule tiles are dealt round robin (ceil(R strips / n_cu) rounds)
    const long tn[4] = {(3 * cfg.channels + 127) / 128, (cfg.channels + 127) / 128, (cfg.hidden + 127) / 128,
                        (cfg.channels + 127) / 128};
    const long kd[4] = {cfg.channels, cfg.channels, cfg.channels, cfg.hidden};
    const bool sk[4] = {skew(), false, skew(), false};
    std::vector<double> cost(cap + 1, 0.0);
    for (int n = 1; n <= cap; ++n) {
      const long R = ((long)n * SP + 255) / 256;
      for (int k = 0; k < 4; ++k) {
        const long Q = n_cu / tn[k];
        if (sk[k] && Q >= 1) cost[n] += ((double)((R + std::min(Q, R) - 1) / std::min(Q, R)) + 3.0 * skew_d * 32.0 / (double)kd[k]) * (double)kd[k];
        else cost[n] += (double)((R * tn[k] + n_cu - 1) / n_cu) * (double)kd[k];
      }
      cost[n] += 1e-3 * (double)kd[0];                   // (a pass has a fixed cost too: 7 launches per block)
    }
    std::vector<double> best(total + 1, 1e300);
    std::vector<int> pick(total + 1, 0);
    best[0] = 0.0;
    for (int b = 1; b <= total; ++b)
      for (int n = 1; n <= std::min(cap, b); ++n)
        if (best[b - n] + cost[n] < best[b]) { best[b] = best[b - n] + cost[n]; pick[b] = n; }
    for (int b = total; b > 0; b -= pick[b]) plan_cache.push_back(pick[b]);
    return plan_cache;
  }

  int flush_events() {
    for (size_t i = 0; i < used; ++i) {
      if (hipEventSynchronize(pool[i].b) != hipSuccess) return -1;
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, pool[i].a, pool[i].b) != hipSuccess) return -1;
      counts[pool[i].cls]++;
      total_ms[pool[i].cls] += ms;
    }
    used = 0;
    return 0;
  }
  // returns slot index or -1
  int begin(int cls, hipStream_t st) {
    if (!prof) return -1;
    if (used == pool.size()) {
      if (pool.size() >= 32768) { if (flush_events() != 0) return -1; }
      else {
        Ev e{};
        // (no system-scope fence at the event: the bracket must not add an L2 write-back of the kernel's output to the time it measures)
        if (hipEventCreateWithFlags(&e.a, hipEventDisableSystemFence) != hipSuccess || hipEventCreateWithFlags(&e.b, hipEventDisableSystemFence) != hipSuccess) return -1;
        pool.push_back(e);
      }
    }
    pool[used].cls = cls;
    (void)hipEventRecord(pool[used].a, st);
    return (int)used++;
  }
  void end(int slot, hipStream_t st) {
    if (slot >= 0) (void)hipEventRecord(pool[slot].b, st);
  }
};

namespace {

struct Scope {
  d3dp_ctx* c; int slot; hipStream_t st;
  Scope(d3dp_ctx* c_, int cls, hipStream_t st_) : c(c_), slot(c_ ? c_->begin(cls, st_) : -1), st(st_) {}
  ~Scope() { if (c) c->end(slot, st); }
};

// out = epi(A W^T + bias).  out_f32: fp32 output even in FAST mode (the Linear outputs that feed a residual add).
// EXACT f16x2: `a_scale` = the scale the A operand was written at, `o_scale` = the scale of a plane output (BlockDev).
int linear(d3dp_ctx* c, int cls, int epi, int out_f32, const void* A, const void* W, float wu, const float* bias, void* out,
           int M, int N, int K, hipStream_t st, void* out2 = nullptr, float* aux = nullptr, float a_scale = kActScale,
           float o_scale = kActScale) {
  Scope s(c, cls, st);
  if (c->fast()) return d3dp_launch_linear_bf16_stream(epi, out_f32, A, W, bias, out, M, N, K, st);
  if (c->x2()) {
    // the qkv Linear writes the packed rows of the split-fp16 attention kernels (K and V already as fp16 planes)
    if (cls == P_QKV && c->x2_attn()) epi = EPI_QKV_PACK;
    // qkv and fc1 (epilogues without loads) run the skewed schedule when the context has it on (seq_pitch() pads for it)
    const int skew_d = c->skew() && (epi == EPI_QKV_PACK || epi == EPI_GELU) ? c->skew_d : 0;
    // proj walks its tiles from the LAST row of tiles to the first: the norm2 row kernel that follows starts at row 0, on the
    // rows of x this launch wrote last (its first 6 us run warm: -9.6 % cycles at -1 % L2 fetches).  Same tiles, same arithmetic.
    // Measured on two boxes: layernorm class 276 -> 254 / 279 -> 253 ms per step, proj -5, step -0.45 %; the same order on
    // fc2 (norm pair +28 ms), fc1, qkv or the row kernels themselves: neutral or worse (profiles/r06_tile_order_ab.md)
    const int rev = cls == P_PROJ ? X2_TILES_LAST_TO_FIRST : 0;
    return d3dp_launch_linear_f16x2(epi, A, W, bias, wu / a_scale, o_scale, (float*)out, out2 ? out2 : out, aux, c->d_flag, M, N,
                                    K, st, skew_d, c->pingpong | rev);
  }
  if (c->x3()) return d3dp_launch_linear_bf16x3(epi, A, W, bias, (float*)out, out, M, N, K, st);
  return d3dp_launch_linear_f32(epi, (const float*)A, (const float*)W, bias, (float*)out, M, N, K, st);
}

// (sp: rows per (clip, hypothesis) sequence in the token buffers, >= F J; d3dp_ctx::seq_pitch)
SeqMap spatial_map(int F, int J, int sp = 0) { return sp > F * J ? SeqMap{J, F, sp, J, 1} : SeqMap{J, 1, J, 0, 1}; }
SeqMap temporal_map(int F, int J, int sp = 0) { return SeqMap{F, J, sp > F * J ? sp : F * J, 1, J}; }

int attention(d3dp_ctx* c, int axis, const void* qkv, void* out, int n_bh, float s_kv, hipStream_t st) {
  const d3dp_cfg& g = c->cfg;
  Scope s(c, axis == 0 ? P_ATTN_S : P_ATTN_T, st);
  if (c->x2_attn())                                    // split-fp16 operands on the fp16 matrix cores; packed qkv rows
    return d3dp_launch_attn_x2(3, axis, qkv, out, axis == 0 ? n_bh * g.frames : n_bh * g.joints,
                               axis == 0 ? spatial_map(g.frames, g.joints, c->seq_pitch())
                                         : temporal_map(g.frames, g.joints, c->seq_pitch()),
                               g.channels, g.heads, s_kv, st);
  if (axis == 0) {
    if (c->fast() && g.channels / g.heads == 64 && g.joints <= 32)
      return d3dp_launch_attn_spatial_bf16(qkv, out, n_bh * g.frames, spatial_map(g.frames, g.joints), g.channels,
                                           g.heads, st);
    return d3dp_launch_attn_rows(c->act(), qkv, out, n_bh * g.frames, spatial_map(g.frames, g.joints), g.channels,
                                 g.heads, st);
  }
  if (c->fast() && g.channels / g.heads == 64 && g.frames <= 256)
    return d3dp_launch_attn_temporal_bf16(qkv, out, n_bh * g.joints, temporal_map(g.frames, g.joints), g.channels,
                                          g.heads, st);
  if (c->exact() && g.channels / g.heads == 64 && g.frames <= 256)     // fp32 matrix cores
    return d3dp_launch_attn_temporal_f32(c->act(), qkv, out, n_bh * g.joints, temporal_map(g.frames, g.joints),
                                         g.channels, g.heads, st);
  return d3dp_launch_attn_rows(c->act(), qkv, out, n_bh * g.joints, temporal_map(g.frames, g.joints), g.channels,
                               g.heads, st);
}

// x = x + proj(attn(qkv(xn)));  x = x + fc2(gelu(fc1(LN2(x))))        (mixste.py:113-115)
// The two residual adds are not done by the GEMMs: each residual-feeding Linear writes y = A W^T + b (fp32) and the
// (activation type: bf16 in FAST mode -- one more bf16 rounding on the branch output, none on the fp32 residual
// stream itself) and the next row-wise kernel (LN2 here; the norm pair / head in the caller) performs x += y while it has the row in
// registers anyway.  On return y1 / y hold the proj / fc2 outputs that the CALLER's next kernel must add to x.
int run_block(d3dp_ctx* c, const BlockDev& w, int axis, float* x, void* y1, void* y, void* bufA, void* bufB, float* lnst,
              int n_bh, hipStream_t st) {
  const d3dp_cfg& g = c->cfg;
  const int Tc = n_bh * c->seq_pitch(), C = g.channels;
  // (s_kv / s_h differ from kActScale only in EXACT f16x2 contexts whose weights asked for it, and s_kv only with the x2
  //  attention kernels: the other attention kernels write their output planes at kActScale)
  const float s_o = c->x2_attn() ? w.s_kv : kActScale;
  LAUNCH_TRY(linear(c, P_QKV, EPI_BIAS, 0, bufA, w.qkv_w, w.qkv_u, w.qkv_b, bufB, Tc, 3 * C, C, st, nullptr, nullptr, kActScale, s_o));
  LAUNCH_TRY(attention(c, axis, bufB, bufA, n_bh, s_o, st));
  const bool fold = c->fold_resid();   // EXACT split-fp16 Linears: x += proj / fc2 inside their epilogues
  if (c->fold_ln()) {
    // norm2 folded into proj's epilogue (statistics, un-normalised operand -> y1) and fc1's (normalisation): no row kernel
    float* slices = lnst;                                                   // [Tc][C / 64][2]
    float* rowstat = lnst + (size_t)c->ln_slice_floats;                     // [Tc + 256][2]
    LAUNCH_TRY(linear(c, P_PROJ, EPI_RESID_LN, 0, bufA, w.proj_w, w.proj_u, w.proj_b, x, Tc, C, C, st, y1, slices, s_o));
    {
      Scope s(c, P_LN, st);
      d3dp_launch_ln_combine(slices, rowstat, Tc, C, g.eps_block, st);
    }
    LAUNCH_TRY(linear(c, P_FC1, EPI_GELU_LN, 0, y1, w.fc1_w, w.fc1_u, w.fc1_c12, bufB, Tc, g.hidden, C, st, bufB, rowstat, kActScale, w.s_h));
  } else {
  LAUNCH_TRY(linear(c, P_PROJ, fold ? EPI_RESID : EPI_BIAS, 0, bufA, w.proj_w, w.proj_u, w.proj_b, fold ? (void*)x : y1, Tc, C, C, st, nullptr, nullptr, s_o));
  {
    Scope s(c, P_LN, st);      // xn = LN2(x + y1); x itself stays untouched (the caller's norm pair adds y1 and y)
    LAUNCH_TRY(d3dp_launch_ln(c->act(), x, fold ? nullptr : y1, 0, w.n2w, w.n2b, g.eps_block, bufA, Tc, C, st));
  }
  LAUNCH_TRY(linear(c, P_FC1, EPI_GELU, 0, bufA, w.fc1_w, w.fc1_u, w.fc1_b, bufB, Tc, g.hidden, C, st, nullptr, nullptr, kActScale, w.s_h));
  }
  LAUNCH_TRY(linear(c, P_FC2, fold ? EPI_RESID : EPI_BIAS, 0, bufB, w.fc2_w, w.fc2_u, w.fc2_b, fold ? (void*)x : y, Tc, C, g.hidden, st, nullptr, nullptr, w.s_h));
  return 0;
}

}  // namespace

extern "C" {

int d3dp_abi_version(void) { return D3DP_ABI_VERSION; }
// test hook (include/d3dp_hip.h, "test hooks"): 1 if this library carries the experiment kernels of gemm_x2.hip
int d3dp_debug_x2_variants(void) { return d3dp_x2_variants_built() ? 1 : 0; }
// test hook (include/d3dp_hip.h, "test hooks"): the training step's split-fp16 Linear alone, out[M, N] = A[M, K] W[N, K]^T + bias on fp32
// device operands -- absmax, operand passes and gemm_f16x2_dyn_kernel as the step launches them.  tail: 0 = the rows behind the
// last whole 256-row tile as one more row of tiles, 1 = as 16 x 64 blocks at the end of the kernel.  amax_out: optional
// pre-zeroed device slot (amax_pos: see kernels.h).  Allocates its operand buffers and synchronises the stream.
int d3dp_debug_train_linear(const float* A, const float* W, const float* bias, float* out, int32_t M, int32_t N, int32_t K,
                            int32_t tail, unsigned* amax_out, int32_t amax_pos, void* stream) {
  if (!A || !W || !out || M < 1 || N < 4 || N % 4 || K < 32 || K % 32) return fail(D3DP_EINVAL, "d3dp_debug_train_linear: bad argument");
  hipStream_t st = (hipStream_t)stream;
  char* buf = nullptr;
  const size_t a_bytes = (size_t)M * K * 4, w_bytes = (size_t)N * K * 4;
  HIP_TRY(hipMalloc((void**)&buf, a_bytes + w_bytes + 64));
  unsigned* amax = reinterpret_cast<unsigned*>(buf + a_bytes + w_bytes);
  float* uns = reinterpret_cast<float*>(amax + 8);
  int rc = hipMemsetAsync(amax, 0, 64, st) == hipSuccess ? 0 : -3;
  if (!rc) {
    d3dp_launch_absmax(A, (size_t)M * K, amax, st);
    d3dp_launch_absmax(W, (size_t)N * K, amax + 1, st);
    d3dp_launch_split2_dyn(A, buf, M, K, K, amax, uns, st);
    d3dp_launch_split2_dyn(W, buf + a_bytes, N, K, K, amax + 1, uns + 1, st);
    rc = d3dp_launch_linear_f16x2_dyn(buf, buf + a_bytes, bias, uns, uns + 1, out, M, N, K, 1, st, amax_out, amax_pos, tail);
  }
  const hipError_t e = hipStreamSynchronize(st);
  (void)hipFree(buf);
  if (rc) return fail(rc == -3 ? D3DP_EHIP : D3DP_EINVAL, "d3dp_debug_train_linear: launch refused (%d)", rc);
  if (e != hipSuccess) return fail(D3DP_EHIP, "d3dp_debug_train_linear: %s", hipGetErrorString(e));
  return 0;
}
const char* d3dp_last_error(void) { return g_err.c_str(); }
const char* d3dp_profile_class_name(int32_t cls) {
  return (cls >= 0 && cls < D3DP_PROFILE_CLASSES) ? kClassNames[cls] : "";
}

int d3dp_create(const d3dp_cfg* cfg, d3dp_ctx** out) {
  if (!cfg || !out) return fail(D3DP_EINVAL, "d3dp_create: null argument");
  const d3dp_cfg& g = *cfg;
  // (frames > 256: EXACT mode's temporal attention takes the chunked-key flash kernel; FAST / TRAIN contexts the fp32 row kernel)
  if (g.frames < 1 || g.frames > 1024) return fail(D3DP_ENOTSUP, "frames=%d not in [1,1024]", g.frames);
  // (more than 32 joints: the spatial axis takes the whole-sequence attention kernels the temporal axis runs on, round 6)
  if (g.joints < 1 || g.joints > 256) return fail(D3DP_ENOTSUP, "joints=%d not in [1,256]", g.joints);
  if (g.heads < 1 || g.channels < 1 || g.channels % g.heads) return fail(D3DP_EINVAL, "heads=%d does not divide channels=%d", g.heads, g.channels);
  if (g.depth < 1) return fail(D3DP_EINVAL, "depth=%d", g.depth);
  if (g.mode != D3DP_MODE_EXACT && g.mode != D3DP_MODE_FAST && g.mode != D3DP_MODE_TRAIN)
    return fail(D3DP_EINVAL, "mode=%d", g.mode);
  const int hd = g.channels / g.heads;
  // The matrix-core kernels (split-fp16 / bf16 operands) and the row kernels around them are instantiated for the widths
  // {64, 128, 256, 512} with head dims {8, 16, 32, 64}: every configuration the reference publishes (`-cs 512`, README.md:33-39)
  // and its smaller powers of two.  The reference itself takes ANY `-cs` its 8 heads divide (common/arguments.py:49,
  // mixste.py:46-62): such a width runs EXACT mode on the fp32 implementation -- fp32-MFMA Linears (gemm_f32_kernel), the fp32
  // row attention with a run-time head dim, run-time-width row kernels (pointwise.hip *_g_kernel) -- i.e. the cross-check
  // implementation D3DP_EXACT_IMPL=f32 selects by hand for the instantiated widths: same tolerance, roughly a fifth of the
  // throughput.  A TRAIN context of such a width (the reference trains at any `-cs` too: main.py:325) takes the fp32 path of the
  // training step -- what D3DP_TRAIN_IMPL=f32 selects for the instantiated widths -- through the run-time-width row kernels of
  // train_g.hip and the VALU attention backward with a run-time head dim.  FAST contexts exist for the instantiated widths only.
  const bool width_inst = (g.channels == 64 || g.channels == 128 || g.channels == 256 || g.channels == 512) &&
                          (hd == 8 || hd == 16 || hd == 32 || hd == 64) && g.hidden >= 64 && g.hidden % 64 == 0;
  if (!width_inst) {
    if (g.mode == D3DP_MODE_FAST)
      return fail(D3DP_ENOTSUP, "channels=%d heads=%d hidden=%d: FAST contexts exist for channels in {64,128,256,512} with head dim in "
                                "{8,16,32,64} and hidden a multiple of 64; other widths run in D3DP_MODE_EXACT / D3DP_MODE_TRAIN (fp32 implementation)",
                  g.channels, g.heads, g.hidden);
    if (g.channels > 1024 || g.channels % 4 || hd % 4 || hd > 128 || g.hidden < 4 || g.hidden % 4)
      return fail(D3DP_ENOTSUP, "channels=%d heads=%d hidden=%d: the fp32 implementation takes channels <= 1024, head dim a multiple of 4 up to 128 "
                                "and hidden a multiple of 4", g.channels, g.heads, g.hidden);
    // (the fp32 attention backward holds two whole-sequence images of its capacity head dim + the statistics in the CU's 160 KiB)
    const int nmax = std::max(g.frames, g.joints), cap = hd <= 16 ? 16 : hd <= 32 ? 32 : hd <= 64 ? 64 : 128;
    if (g.mode == D3DP_MODE_TRAIN && (nmax > 256 || (size_t)nmax * (8 * (cap + 4) + 12) > 160 * 1024))
      return fail(D3DP_ENOTSUP, "channels=%d heads=%d frames=%d joints=%d: training at a width outside {64,128,256,512} runs the fp32 attention "
                                "backward, which holds a whole sequence in LDS (<= 256 tokens; <= 153 at head dims above 64)",
                  g.channels, g.heads, g.frames, g.joints);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(D3DP_EHIP, "no HIP device visible: libd3dp_hip has no CPU fallback");
  d3dp_ctx* c = new d3dp_ctx();
  c->cfg = g;
  const char* xf = getenv("D3DP_EXACT_IMPL");
  c->exact_impl = c->exact_impl_req = (xf && !strcmp(xf, "bf16x3")) ? 1 : (xf && !strcmp(xf, "f32")) ? 2 : 0;
  if (!width_inst) {
    if (c->exact_impl_req == 1) {
      delete c;
      return fail(D3DP_ENOTSUP, "D3DP_EXACT_IMPL=bf16x3 exists for channels in {64,128,256,512}; channels=%d runs the fp32 implementation", g.channels);
    }
    c->exact_impl = c->exact_impl_req = 2;
  }
  const char* lr = getenv("D3DP_LONG_ATTN");             // cross-check: clips > 256 frames on the fp32 row attention kernel
  c->long_rows = lr && !strcmp(lr, "rows");
  const char* nf = getenv("D3DP_NO_FOLD");               // cross-check: residual adds (and norm2) in the row kernels
  c->fold = !(nf && nf[0] == '1');
  const char* ti = getenv("D3DP_TRAIN_IMPL");
  c->train_x2 = !(ti && !strcmp(ti, "f32")) && width_inst;     // (a width outside the instantiated set: the fp32 path)
  const char* ov = getenv("D3DP_TRAIN_OVERLAP");
  c->train_overlap = !(ov && ov[0] == '0');
  c->train_overlap_sets = (ov && ov[0] == '1') ? 1 : 2;
  const char* tg = getenv("D3DP_TRAIN_GELU");
  c->train_gelu_in_prep = !(tg && !strcmp(tg, "pass"));
  const char* tw = getenv("D3DP_TRAIN_WGRAD");
  c->train_wgrad_merged = !(tw && !strcmp(tw, "each"));
  const char* tl = getenv("D3DP_TRAIN_LN_OPERAND");       // =pass: round 5's operand pass behind every LayerNorm (variants build only)
  c->train_ln_direct = !(tl && !strcmp(tl, "pass"));
  const char* tt = getenv("D3DP_TRAIN_TAIL");
  c->train_tail_blocks = !(tt && !strcmp(tt, "split"));
  const char* ta = getenv("D3DP_TRAIN_ATTN");
  c->train_attn_x2 = (ta && !strcmp(ta, "f32")) ? 0 : (ta && !strcmp(ta, "x2t")) ? 1 : 2;
  // D3DP_TRAIN_WGRAD=each / D3DP_TRAIN_TAIL=split / D3DP_TRAIN_GELU=pass were round 5's same-box A/B switches: they make the
  // configs[4] shapes take the launch forms the library keeps for the shapes its merged / fused forms do not cover (widths below
  // 256, contractions that are not a multiple of 16 k-steps).  Measured and superseded: like the experiment
  // kernels below they are honoured by the `make variants` library only and REFUSED here -- never ignored.  (What stays: the
  // cross-check implementations D3DP_TRAIN_IMPL=f32, D3DP_TRAIN_ATTN=f32|x2t, D3DP_TRAIN_ATTN_BWD=valu, D3DP_EXACT_IMPL, D3DP_NO_FOLD,
  // and the profiling switch D3DP_TRAIN_OVERLAP=0|1: one stream, so that no kernel's duration contains a wait for CUs.)
  if ((!c->train_gelu_in_prep || !c->train_wgrad_merged || !c->train_tail_blocks || !c->train_ln_direct) && !d3dp_x2_variants_built()) {
    delete c;
    return fail(D3DP_ENOTSUP, "D3DP_TRAIN_WGRAD=each / D3DP_TRAIN_TAIL=split / D3DP_TRAIN_GELU=pass / D3DP_TRAIN_LN_OPERAND=pass select superseded launch forms of the "
                              "training step that only the variants build honours (make -C d3dp_amd/csrc variants; "
                              "D3DP_LIB=d3dp_amd/lib/variants/libd3dp_variants.so)");
  }
  {
    // Measurement switches of experiments that were measured and not adopted: the row-class skewed schedule
    // (D3DP_X2_SKEW=1|2|4), the ping-pong (D3DP_X2_PP=1) and wide (D3DP_X2_WIDE=1) forms of the EXACT Linear, norm2 folded into
    // proj / fc1 (D3DP_FOLD_LN=1), the sequence padding the skewed schedule needs (D3DP_SEQ_PAD).  Their kernels exist only in a
    // library built with -DD3DP_X2_VARIANTS=1 (`make variants`): the product library refuses the request instead of ignoring it.
    const char* sk = getenv("D3DP_X2_SKEW");
    const char* pp = getenv("D3DP_X2_PP");
    const char* wide = getenv("D3DP_X2_WIDE");
    const char* pd = getenv("D3DP_SEQ_PAD");
    const char* nl = getenv("D3DP_FOLD_LN");
    int skew_d = 0, pingpong = 0, pad = -1;
    if (sk && (sk[0] == '0' || sk[0] == '1' || sk[0] == '2' || sk[0] == '4') && sk[1] == 0) skew_d = sk[0] - '0';
    if (pp && (pp[0] == '0' || pp[0] == '1') && pp[1] == 0) pingpong = pp[0] - '0';
    if (wide && wide[0] == '1' && wide[1] == 0) pingpong = 2;
    if (pd && (pd[0] == '0' || pd[0] == '1') && pd[1] == 0) pad = pd[0] - '0';
    const bool fold_ln = nl && nl[0] == '1';
    if ((skew_d || pingpong || pad > 0 || fold_ln) && !d3dp_x2_variants_built()) {
      delete c;
      return fail(D3DP_ENOTSUP, "D3DP_X2_SKEW / D3DP_X2_PP / D3DP_X2_WIDE / D3DP_SEQ_PAD / D3DP_FOLD_LN select experiment kernels that this "
                                "library was built without (make -C d3dp_amd/csrc variants; D3DP_LIB=d3dp_amd/lib/variants/libd3dp_variants.so)");
    }
    c->skew_d = skew_d; c->pingpong = pingpong; c->pad_override = pad; c->fold_ln_on = fold_ln;
  }
  HIP_TRY(hipGetDevice(&c->device));
  {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    c->n_cu = prop.multiProcessorCount;
  }
  if (hipMalloc((void**)&c->d_flag, sizeof(unsigned)) != hipSuccess || hipMemset(c->d_flag, 0, sizeof(unsigned)) != hipSuccess) {
    delete c;
    return fail(D3DP_EHIP, "d3dp_create: cannot allocate the status word");
  }
  *out = c;
  return D3DP_OK;
}

int d3dp_destroy(d3dp_ctx* c) {
  if (!c) return D3DP_OK;
  for (auto& e : c->pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  if (c->arena) (void)hipFree(c->arena);
  if (c->d_flag) (void)hipFree(c->d_flag);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  for (hipEvent_t e : c->ev_done)
    if (e) (void)hipEventDestroy(e);
  if (c->aux) (void)hipStreamDestroy(c->aux);
  delete c;
  return D3DP_OK;
}

int d3dp_set_weights(d3dp_ctx* c, const d3dp_weights* w, void* stream) {
  if (!c || !w || !w->ste || !w->tte) return fail(D3DP_EINVAL, "d3dp_set_weights: null argument");
  hipStream_t st = (hipStream_t)stream;
  const d3dp_cfg& g = c->cfg;
  const size_t C = g.channels, Hd = g.hidden, J = g.joints, F = g.frames;
  // ---- EXACT f16x2: the range the weights PROVE for every data-dependent split-fp16 operand (d3dp_exact_range_bound), on
  // the device (one wave per weight row; 4 floats per block come back), and from it the operand scales of every block.
  //   LayerNorm output  |LN(x)_k| <= sqrt(C-1) |gamma_k| + |beta_k|                         (operand of qkv / fc1: fixed 2^4)
  //   q, k, v           |.| <= max_n sum_k |Wqkv[n,k]| in_k + |b_n|  -- and the attention output <= max |v|      (s_kv)
  //   MLP hidden        |GELU(y)| <= |y| <= the same form with Wfc1                                               (s_h)
  // A bound below 4094 keeps 2^4 (full 22 bits down to 2^-7); above it the scale drops to the largest power of two with
  // bound x scale < 65504: nothing can overflow to inf (the fp32 reference is finite there, so must this be -- VERDICT r3
  // item 5) and the representation error stays <= max(2^-22 |x|, 2^-25 / scale) <= 2^-40 x bound in absolute terms, far below
  // the fp32 rounding of the sums these operands enter.  A LayerNorm whose own output bound reaches 4094 (|gamma| ~ 180),
  // or an out-of-range q/k/v bound where the split-fp16 attention kernels do not apply, switches the whole context to the
  // six-pass split-bf16 implementation, which has fp32's exponent range.
  c->range_bound = 0.f;
  c->impl_fallback = false;
  c->plan_total = -1;            // the pass plan depends on the implementation chosen below (x2-tuned or uniform passes)
  std::vector<float> blk_scale(4 * (size_t)g.depth, kActScale);      // [kind][d][s_kv, s_h]
  if (c->exact() && c->exact_impl_req == 0) {
    c->exact_impl = 0;
    const size_t nb = 2 * (size_t)g.depth;
    unsigned* dbound = nullptr;
    HIP_TRY(hipMalloc((void**)&dbound, nb * 6 * sizeof(unsigned)));
    struct Free { unsigned* p; ~Free() { (void)hipFree(p); } } free_dbound{dbound};
    HIP_TRY(hipMemsetAsync(dbound, 0, nb * 6 * sizeof(unsigned), st));
    for (int kind = 0; kind < 2; ++kind)
      for (int d = 0; d < g.depth; ++d) {
        const d3dp_block_weights& b = (kind == 0 ? w->ste : w->tte)[d];
        if (!b.norm1_w || !b.norm1_b || !b.qkv_w || !b.qkv_b || !b.norm2_w || !b.norm2_b || !b.fc1_w || !b.fc1_b || !b.proj_w ||
            !b.fc2_w)
          return fail(D3DP_EINVAL, "d3dp_set_weights: a weight pointer is null");
        unsigned* o = dbound + ((size_t)kind * g.depth + d) * 6;
        d3dp_launch_rowbound(b.qkv_w, b.norm1_w, b.norm1_b, b.qkv_b, 3 * (int)C, (int)C, o, st);       // o[0] q/k/v, o[1] LN1 out
        d3dp_launch_rowbound(b.fc1_w, b.norm2_w, b.norm2_b, b.fc1_b, (int)Hd, (int)C, o + 2, st);      // o[2] hidden, o[3] LN2 out
        d3dp_launch_absmax(b.proj_w, C * C, o + 4, st);                                                // o[4], o[5]: the two matrices
        d3dp_launch_absmax(b.fc2_w, C * Hd, o + 5, st);                                                // no bound reads (inf check)
      }
    std::vector<float> hb(nb * 6);
    HIP_TRY(hipMemcpyAsync(hb.data(), dbound, nb * 6 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const float kSafe = 65504.0f / kActScale;           // 4094: what 2^4 holds
    auto pick = [&](float bound) {                       // largest power of two <= 2^4 with bound x scale < 65504
      float sc = kActScale;
      while (!(bound * sc < 65504.0f) && sc > 1e-30f) sc *= 0.5f;
      return sc;
    };
    float worst = 0.f;
    bool fallback = false;
    for (size_t i = 0; i < nb; ++i) {
      const float bq = hb[6 * i], bl1 = hb[6 * i + 1], bh = hb[6 * i + 2], bl2 = hb[6 * i + 3];
      // Non-finite weights (a diverged checkpoint): ONE behaviour whichever tensor holds them -- the context moves
      // to the split-bf16 implementation, which has fp32's range and carries inf / nan through like the reference's fp32
      // kernels do; the output is then non-finite where the reference's is and d3dp_status reports it.  Never an error here.
      if (!(bq < INFINITY) || !(bh < INFINITY) || !(bl1 < INFINITY) || !(bl2 < INFINITY) || !(hb[6 * i + 4] < INFINITY) ||
          !(hb[6 * i + 5] < INFINITY)) {
        fallback = true;
        worst = INFINITY;
        continue;
      }
      worst = std::max(worst, std::max(std::max(bq, bh), std::max(bl1, bl2)));
      if (!(bl1 < kSafe) || !(bl2 < kSafe)) fallback = true;
      if (!(bq < kSafe) && !c->x2_attn()) fallback = true;
      blk_scale[2 * i] = pick(bq);
      blk_scale[2 * i + 1] = pick(bh);
    }
    c->range_bound = std::min(worst, 3.0e38f);
    if (fallback) {
      c->exact_impl = 1;                                 // split-bf16, six passes: no range limit
      c->impl_fallback = true;
      std::fill(blk_scale.begin(), blk_scale.end(), kActScale);
    }
  }
  const size_t ws = c->fast() ? 2 : (c->x3() ? 6 : 4);   // bytes per weight-matrix element (bf16 / 3 bf16 planes / fp32)
  // ---- arena layout ----
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
  struct Item { const void* src; size_t n; bool mat; size_t off; };
  std::vector<Item> items;
  auto add = [&](const void* src, size_t n, bool mat) {
    items.push_back({src, n, mat, take(n * (mat ? ws : 4))});
    return items.size() - 1;
  };
  const size_t i_spos = add(w->spatial_pos, J * C, false), i_tpos = add(w->temporal_pos, F * C, false);
  const size_t i_ew = add(w->embed_w, C * 5, false), i_eb = add(w->embed_b, C, false);
  const size_t i_fr = add(w->time_freq, C / 2, false);
  const size_t i_t1w = add(w->time1_w, 2 * C * C, false), i_t1b = add(w->time1_b, 2 * C, false);
  const size_t i_t3w = add(w->time3_w, 2 * C * C, false), i_t3b = add(w->time3_b, C, false);
  const size_t i_snw = add(w->spatial_norm_w, C, false), i_snb = add(w->spatial_norm_b, C, false);
  const size_t i_tnw = add(w->temporal_norm_w, C, false), i_tnb = add(w->temporal_norm_b, C, false);
  const size_t i_hnw = add(w->head_norm_w, C, false), i_hnb = add(w->head_norm_b, C, false);
  const size_t i_hw = add(w->head_w, 3 * C, false), i_hb = add(w->head_b, 3, false);
  struct BI { size_t v[13]; };
  std::vector<BI> bis;
  // fold_ln: fc1 runs on W diag(gamma2) with [c2 | c1] in place of its bias (computed into temporaries that live until the
  // synchronisation below)
  std::vector<float*> temps;
  struct TempFree { std::vector<float*>& t; ~TempFree() { for (float* p : t) (void)hipFree(p); } } temp_free{temps};
  for (int kind = 0; kind < 2; ++kind)
    for (int d = 0; d < g.depth; ++d) {
      const d3dp_block_weights& b = (kind == 0 ? w->ste : w->tte)[d];
      if (!b.norm1_w || !b.norm1_b || !b.qkv_w || !b.qkv_b || !b.proj_w || !b.proj_b || !b.norm2_w || !b.norm2_b || !b.fc1_w ||
          !b.fc1_b || !b.fc2_w || !b.fc2_b) return fail(D3DP_EINVAL, "d3dp_set_weights: a weight pointer is null");
      BI bi;
      bi.v[0] = add(b.norm1_w, C, false); bi.v[1] = add(b.norm1_b, C, false);
      bi.v[2] = add(b.qkv_w, 3 * C * C, true); bi.v[3] = add(b.qkv_b, 3 * C, false);
      bi.v[4] = add(b.proj_w, C * C, true); bi.v[5] = add(b.proj_b, C, false);
      bi.v[6] = add(b.norm2_w, C, false); bi.v[7] = add(b.norm2_b, C, false);
      if (c->fold_ln()) {
        float *wp = nullptr, *c12 = nullptr;
        HIP_TRY(hipMalloc((void**)&wp, Hd * C * 4)); temps.push_back(wp);
        HIP_TRY(hipMalloc((void**)&c12, 2 * Hd * 4)); temps.push_back(c12);
        d3dp_launch_fold_ln(b.fc1_w, b.norm2_w, b.norm2_b, b.fc1_b, wp, c12, (int)Hd, (int)C, st);
        bi.v[8] = add(wp, Hd * C, true); bi.v[9] = add(b.fc1_b, Hd, false); bi.v[12] = add(c12, 2 * Hd, false);
      } else {
        bi.v[8] = add(b.fc1_w, Hd * C, true); bi.v[9] = add(b.fc1_b, Hd, false); bi.v[12] = bi.v[9];
      }
      bi.v[10] = add(b.fc2_w, C * Hd, true); bi.v[11] = add(b.fc2_b, C, false);
      bis.push_back(bi);
    }
  for (auto& it : items)
    if (!it.src) return fail(D3DP_EINVAL, "d3dp_set_weights: a weight pointer is null");
  if (!c->arena || c->arena_bytes < off) {
    if (c->arena) HIP_TRY(hipFree(c->arena));
    c->arena = nullptr;
    HIP_TRY(hipMalloc((void**)&c->arena, off));
    c->arena_bytes = off;
  }
  // EXACT f16x2: every weight matrix is multiplied by 2^s, s = 14 - exponent(max |w|), before the hi/lo split (max |w| 2^s
  // in [2^13, 2^14): far from fp16 overflow, and the bulk of the matrix far above the subnormal range); the GEMM
  // multiplies its result by 2^-s.
  std::vector<float> unscale(items.size(), 1.f);
  if (c->x2()) {
    size_t n_mat = 0;
    for (auto& it : items) n_mat += it.mat;
    unsigned* dmax = nullptr;
    HIP_TRY(hipMalloc((void**)&dmax, n_mat * sizeof(unsigned)));
    HIP_TRY(hipMemsetAsync(dmax, 0, n_mat * sizeof(unsigned), st));
    size_t k = 0;
    for (auto& it : items)
      if (it.mat) d3dp_launch_absmax((const float*)it.src, it.n, dmax + k++, st);
    std::vector<float> hmax(n_mat);
    HIP_TRY(hipMemcpyAsync(hmax.data(), dmax, n_mat * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipFree(dmax));
    k = 0;
    for (size_t i = 0; i < items.size(); ++i) {
      if (!items[i].mat) continue;
      const float mx = hmax[k++];
      if (!(mx < INFINITY)) return fail(D3DP_EINVAL, "d3dp_set_weights: a weight matrix holds inf/nan");
      int e = 0;
      if (mx > 0.f) frexpf(mx, &e);                  // mx = f 2^e, f in [0.5, 1)
      unscale[i] = ldexpf(1.f, e - 14);
    }
  }
  for (size_t i = 0; i < items.size(); ++i) {
    auto& it = items[i];
    if (it.mat && c->fast()) launch_to_bf16((const float*)it.src, c->arena + it.off, it.n, st);
    else if (it.mat && c->x3()) d3dp_launch_split3((const float*)it.src, c->arena + it.off, it.n, st);
    else if (it.mat && c->x2()) d3dp_launch_split2((const float*)it.src, c->arena + it.off, it.n, 1.0f / unscale[i], st);
    else HIP_TRY(hipMemcpyAsync(c->arena + it.off, it.src, it.n * 4, hipMemcpyDeviceToDevice, st));
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  auto F32 = [&](size_t i) { return (const float*)(c->arena + items[i].off); };
  auto ANY = [&](size_t i) { return (const void*)(c->arena + items[i].off); };
  c->spos = F32(i_spos); c->tpos = F32(i_tpos); c->ew = F32(i_ew); c->eb = F32(i_eb); c->freq = F32(i_fr);
  c->t1w = F32(i_t1w); c->t1b = F32(i_t1b); c->t3w = F32(i_t3w); c->t3b = F32(i_t3b);
  c->snw = F32(i_snw); c->snb = F32(i_snb); c->tnw = F32(i_tnw); c->tnb = F32(i_tnb);
  c->hnw = F32(i_hnw); c->hnb = F32(i_hnb); c->hw = F32(i_hw); c->hb = F32(i_hb);
  c->ste.clear(); c->tte.clear();
  for (size_t k = 0; k < bis.size(); ++k) {
    const BI& bi = bis[k];
    BlockDev b{F32(bi.v[0]), F32(bi.v[1]), F32(bi.v[6]), F32(bi.v[7]), F32(bi.v[3]), F32(bi.v[5]), F32(bi.v[9]),
               F32(bi.v[11]), ANY(bi.v[2]), ANY(bi.v[4]), ANY(bi.v[8]), ANY(bi.v[10]),
               unscale[bi.v[2]], unscale[bi.v[4]], unscale[bi.v[8]], unscale[bi.v[10]], F32(bi.v[12]),
               blk_scale[2 * k], blk_scale[2 * k + 1]};
    (k < (size_t)g.depth ? c->ste : c->tte).push_back(b);
  }
  c->weights_set = true;
  return D3DP_OK;
}

int d3dp_set_weights_borrowed(d3dp_ctx* c, const d3dp_weights* w) {
  if (!c || !w || !w->ste || !w->tte) return fail(D3DP_EINVAL, "d3dp_set_weights_borrowed: null argument");
  if (!c->train()) return fail(D3DP_ESTATE, "d3dp_set_weights_borrowed needs a D3DP_MODE_TRAIN context (fp32 weights)");
  c->spos = w->spatial_pos; c->tpos = w->temporal_pos; c->ew = w->embed_w; c->eb = w->embed_b; c->freq = w->time_freq;
  c->t1w = w->time1_w; c->t1b = w->time1_b; c->t3w = w->time3_w; c->t3b = w->time3_b;
  c->snw = w->spatial_norm_w; c->snb = w->spatial_norm_b; c->tnw = w->temporal_norm_w; c->tnb = w->temporal_norm_b;
  c->hnw = w->head_norm_w; c->hnb = w->head_norm_b; c->hw = w->head_w; c->hb = w->head_b;
  const float* top[] = {c->spos, c->tpos, c->ew, c->eb, c->freq, c->t1w, c->t1b, c->t3w, c->t3b, c->snw, c->snb, c->tnw,
                        c->tnb, c->hnw, c->hnb, c->hw, c->hb};
  for (const float* p : top)
    if (!p) return fail(D3DP_EINVAL, "d3dp_set_weights_borrowed: a weight pointer is null");
  c->ste.clear(); c->tte.clear();
  for (int kind = 0; kind < 2; ++kind)
    for (int d = 0; d < c->cfg.depth; ++d) {
      const d3dp_block_weights& b = (kind == 0 ? w->ste : w->tte)[d];
      const void* all[] = {b.norm1_w, b.norm1_b, b.qkv_w, b.qkv_b, b.proj_w, b.proj_b, b.norm2_w, b.norm2_b, b.fc1_w,
                           b.fc1_b, b.fc2_w, b.fc2_b};
      for (const void* p : all)
        if (!p) return fail(D3DP_EINVAL, "d3dp_set_weights_borrowed: a weight pointer is null");
      BlockDev bd{b.norm1_w, b.norm1_b, b.norm2_w, b.norm2_b, b.qkv_b, b.proj_b, b.fc1_b, b.fc2_b, b.qkv_w, b.proj_w,
                  b.fc1_w, b.fc2_w};
      (kind == 0 ? c->ste : c->tte).push_back(bd);
    }
  c->weights_set = true;
  return D3DP_OK;
}

int d3dp_workspace_bytes(const d3dp_ctx* c, int32_t B, int32_t H, size_t* bytes) {
  if (!c || !bytes || B < 1 || H < 1) return fail(D3DP_EINVAL, "d3dp_workspace_bytes: bad argument");
  const d3dp_cfg& g = c->cfg;
  const size_t n = (size_t)std::min(c->chunk(), B * H);
  const size_t Tc = n * c->seq_pitch(), C = g.channels;
  const size_t wide = (size_t)std::max(3 * g.channels, g.hidden);
  *bytes = align_up((size_t)B * C * 4) + align_up(Tc * C * 4) + 2 * align_up(Tc * C * c->y_size()) +
           align_up(Tc * C * c->act_size()) + align_up(Tc * wide * c->wide_size()) +
           (c->fold_ln() ? align_up(Tc * ((C + 63) / 64) * 8) + align_up((Tc + 256) * 8) : 0);
  return D3DP_OK;
}

int d3dp_denoise(d3dp_ctx* c, const float* x2d, const float* x_t, const int64_t* t, float* out, int32_t B, int32_t H,
                 void* workspace, size_t workspace_bytes, void* stream) {
  if (!c || !x2d || !x_t || !t || !out || !workspace || B < 1 || H < 1)
    return fail(D3DP_EINVAL, "d3dp_denoise: bad argument");
  if (!c->weights_set) return fail(D3DP_ESTATE, "d3dp_denoise: weights not set");
  size_t need = 0;
  d3dp_workspace_bytes(c, B, H, &need);
  if (workspace_bytes < need) return fail(D3DP_ESTATE, "workspace %zu < required %zu bytes", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  const d3dp_cfg& g = c->cfg;
  const int C = g.channels, F = g.frames, J = g.joints, FJ = F * J, SP = c->seq_pitch();
  const int BH = B * H, chunk = std::min(c->chunk(), BH);
  const size_t Tmax = (size_t)chunk * SP;
  char* p = (char*)workspace;
  float* temb = (float*)p; p += align_up((size_t)B * C * 4);
  float* x = (float*)p;    p += align_up(Tmax * C * 4);
  void* y1 = p;            p += align_up(Tmax * C * c->y_size());
  void* y = p;             p += align_up(Tmax * C * c->y_size());
  void* bufA = p;          p += align_up(Tmax * C * c->act_size());
  void* bufB = p;          p += align_up(Tmax * (size_t)std::max(3 * g.channels, g.hidden) * c->wide_size());
  float* lnst = (float*)p;                               // fold_ln: slice statistics, then (mean, rstd) per row
  c->ln_slice_floats = align_up(Tmax * ((C + 63) / 64) * 8) / 4;

  {
    Scope s(c, P_TIME, st);
    LAUNCH_TRY(d3dp_launch_time_mlp(t, c->freq, c->t1w, c->t1b, c->t3w, c->t3b, temb, B, C, st));
  }
  int seq0 = 0;
  for (const int n : c->plan(BH)) {
    const int Tc = n * SP;
    {
      Scope s(c, P_EMBED, st);
      LAUNCH_TRY(d3dp_launch_embed_ln(c->act(), x2d, x_t, temb, c->ew, c->eb, c->spos, c->ste[0].n1w, c->ste[0].n1b,
                                      g.eps_block, x, bufA, seq0, n, H, F, J, C, st, SP));
    }
    const bool fold = c->fold_resid();
    for (int d = 0; d < g.depth; ++d) {
      int r = run_block(c, c->ste[d], 0, x, y1, y, bufA, bufB, lnst, n, st);
      if (r) return r;
      {
        Scope s(c, P_LN2, st);   // x += fc2 out; Spatial_norm (+ Temporal_pos after block 0); TTE block d's norm1
        LAUNCH_TRY(d3dp_launch_ln2(c->act(), x, fold ? nullptr : y1, fold ? nullptr : y, c->snw, c->snb, d == 0 ? c->tpos : nullptr, c->tte[d].n1w,
                                   c->tte[d].n1b, g.eps_block, bufA, Tc, C, F, J, st, SP));
      }
      r = run_block(c, c->tte[d], 1, x, y1, y, bufA, bufB, lnst, n, st);
      if (r) return r;
      if (d + 1 < g.depth) {
        Scope s(c, P_LN2, st);   // x += fc2 out; Temporal_norm; STE block d+1's norm1
        LAUNCH_TRY(d3dp_launch_ln2(c->act(), x, fold ? nullptr : y1, fold ? nullptr : y, c->tnw, c->tnb, nullptr, c->ste[d + 1].n1w, c->ste[d + 1].n1b,
                                   g.eps_block, bufA, Tc, C, F, J, st, SP));
      }
    }
    {
      Scope s(c, P_HEAD, st);    // x += fc2 out; Temporal_norm; head LayerNorm; Linear(C,3)
      LAUNCH_TRY(d3dp_launch_head(c->fast() ? 1 : 0, x, fold ? nullptr : y1, fold ? nullptr : y, c->tnw, c->tnb, g.eps_block, c->hnw, c->hnb, g.eps_head, c->hw, c->hb,
                                  out + (size_t)seq0 * FJ * 3, Tc, C, st, FJ, SP));
    }
    seq0 += n;
  }
  d3dp_launch_nonfinite_flag(out, (size_t)BH * FJ * 3, c->d_flag, st);    // 15.9 MB at B = 32, H = 20: microseconds
  HIP_TRY(hipGetLastError());
  return D3DP_OK;
}

int d3dp_exact_range_bound(const d3dp_ctx* c, float* bound) {
  if (!c || !bound) return fail(D3DP_EINVAL, "d3dp_exact_range_bound: null argument");
  if (!c->weights_set) return fail(D3DP_ESTATE, "d3dp_exact_range_bound: weights not set");
  *bound = c->range_bound;
  return D3DP_OK;
}

int d3dp_exact_scales(const d3dp_ctx* c, float* s_kv, float* s_hidden, int32_t* implementation) {
  if (!c) return fail(D3DP_EINVAL, "d3dp_exact_scales: null argument");
  if (!c->weights_set) return fail(D3DP_ESTATE, "d3dp_exact_scales: weights not set");
  const int dep = c->cfg.depth;
  for (int i = 0; i < 2 * dep; ++i) {
    const BlockDev& b = i < dep ? c->ste[i] : c->tte[i - dep];
    if (s_kv) s_kv[i] = b.s_kv;
    if (s_hidden) s_hidden[i] = b.s_h;
  }
  if (implementation) *implementation = c->exact() ? c->exact_impl : -1;
  return D3DP_OK;
}

int d3dp_status(d3dp_ctx* c, int32_t* nonfinite) {
  if (!c || !nonfinite) return fail(D3DP_EINVAL, "d3dp_status: null argument");
  unsigned v = 0;
  HIP_TRY(hipDeviceSynchronize());                     // every d3dp_denoise issued so far has written its verdict
  HIP_TRY(hipMemcpy(&v, c->d_flag, sizeof v, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(c->d_flag, 0, sizeof v));
  *nonfinite = (int32_t)(v != 0u);                     // bit 0: inf / nan in an output; bit 1: an operand left the split range
  return D3DP_OK;
}

int d3dp_ddim_pre(const float* img, float* xt2, const int32_t* perm, float scale, int32_t B, int32_t H, int32_t F,
                  int32_t J, void* stream) {
  if (!img || !xt2 || !perm || B < 1) return fail(D3DP_EINVAL, "d3dp_ddim_pre: bad argument");
  LAUNCH_TRY(d3dp_launch_ddim_pre(img, xt2, perm, scale, B, H * F * J * 3, J, (hipStream_t)stream));
  HIP_TRY(hipGetLastError());
  return D3DP_OK;
}

int d3dp_ddim_post(const float* pred2, const float* img, const float* noise, const int32_t* perm, float scale,
                   double sqrt_recip, double sqrt_recipm1, float c_xstart, float c_noise, float sigma, int32_t last,
                   float* x_start, size_t xs_bstride, float* img_next, int32_t B, int32_t H, int32_t F, int32_t J,
                   void* stream) {
  if (!pred2 || !perm || !x_start || B < 1) return fail(D3DP_EINVAL, "d3dp_ddim_post: bad argument");
  if (!last && (!img || !noise || !img_next)) return fail(D3DP_EINVAL, "d3dp_ddim_post: img/noise/img_next required");
  LAUNCH_TRY(d3dp_launch_ddim_post(pred2, img, noise, perm, scale, sqrt_recip, sqrt_recipm1, c_xstart, c_noise, sigma,
                                   last, x_start, xs_bstride, img_next, B, H * F * J * 3, J, (hipStream_t)stream));
  HIP_TRY(hipGetLastError());
  return D3DP_OK;
}

int d3dp_q_sample(const float* x0, const float* noise, const double* a, const double* s, float scale, float* out,
                  int32_t B, int32_t per_b, void* stream) {
  if (!x0 || !noise || !a || !s || !out || B < 1 || per_b < 1) return fail(D3DP_EINVAL, "d3dp_q_sample: bad argument");
  LAUNCH_TRY(d3dp_launch_q_sample(x0, noise, a, s, scale, out, B, per_b, (hipStream_t)stream));
  HIP_TRY(hipGetLastError());
  return D3DP_OK;
}

int d3dp_jpma(const float* pred, const float* traj, const float* cam, const float* gt2d, const float* gt3d, float* agg,
              int32_t* sel, float* err_sel, float* err_min, int32_t B, int32_t K, int32_t H, int32_t F, int32_t J,
              int32_t zero_root, void* stream) {
  if (!pred || !traj || !cam || !gt2d || !agg || B < 1 || K < 1 || H < 1)
    return fail(D3DP_EINVAL, "d3dp_jpma: bad argument");
  if ((err_sel || err_min) && !gt3d) return fail(D3DP_EINVAL, "d3dp_jpma: error outputs need gt3d");
  LAUNCH_TRY(d3dp_launch_jpma(pred, traj, cam, gt2d, gt3d, agg, sel, err_sel, err_min, nullptr, nullptr, nullptr, 0, B,
                              K, H, F, J, zero_root ? 0 : -1, 0, (hipStream_t)stream));
  HIP_TRY(hipGetLastError());
  return D3DP_OK;
}

int d3dp_jpma_gathered(const float* gathered, const float* traj, const float* cam, const float* gt2d, const float* gt3d,
                       float* agg, int32_t* sel, float* err_sel, float* err_min, int32_t R, int32_t B, int32_t K,
                       int32_t H_local, int32_t F, int32_t J, int32_t zero_root, void* stream) {
  if (!gathered || !traj || !cam || !gt2d || !agg || R < 1 || B < 1 || K < 1 || H_local < 1)
    return fail(D3DP_EINVAL, "d3dp_jpma_gathered: bad argument");
  if ((err_sel || err_min) && !gt3d) return fail(D3DP_EINVAL, "d3dp_jpma_gathered: error outputs need gt3d");
  LAUNCH_TRY(d3dp_launch_jpma(gathered, traj, cam, gt2d, gt3d, agg, sel, err_sel, err_min, nullptr, nullptr, nullptr, 0, B,
                              K, R * H_local, F, J, zero_root ? 0 : -1, 0, (hipStream_t)stream, H_local,
                              (size_t)B * K * H_local * F * J * 3));
  HIP_TRY(hipGetLastError());
  return D3DP_OK;
}

int d3dp_jpma_ex(const float* pred, const float* traj, const float* cam, const float* gt2d, const float* gt3d, float* agg,
                 int32_t* sel, float* err_sel, float* err_min, float* jbest, float* mean, int32_t B, int32_t K, int32_t H,
                 int32_t F, int32_t J, int32_t root_joint, int32_t linear_projection, void* stream) {
  if (!pred || !traj || !cam || !gt2d || B < 1 || K < 1 || H < 1 || root_joint >= J)
    return fail(D3DP_EINVAL, "d3dp_jpma_ex: bad argument");
  if ((err_sel || err_min || jbest) && !gt3d) return fail(D3DP_EINVAL, "d3dp_jpma_ex: error / J-Best outputs need gt3d");
  LAUNCH_TRY(d3dp_launch_jpma(pred, traj, cam, gt2d, gt3d, agg, sel, err_sel, err_min, nullptr, jbest, mean, 0, B, K, H,
                              F, J, root_joint, linear_projection, (hipStream_t)stream));
  HIP_TRY(hipGetLastError());
  return D3DP_OK;
}

int d3dp_jpma_winners(const float* pred, const float* traj, const float* cam, const float* gt2d, float* win,
                      int32_t h_offset, int32_t B, int32_t K, int32_t H, int32_t F, int32_t J, int32_t zero_root,
                      void* stream) {
  if (!pred || !traj || !cam || !gt2d || !win || B < 1 || K < 1 || H < 1 || h_offset < 0)
    return fail(D3DP_EINVAL, "d3dp_jpma_winners: bad argument");
  LAUNCH_TRY(d3dp_launch_jpma(pred, traj, cam, gt2d, nullptr, nullptr, nullptr, nullptr, nullptr, win, nullptr, nullptr,
                              h_offset, B, K, H, F, J, zero_root ? 0 : -1, 0, (hipStream_t)stream));
  HIP_TRY(hipGetLastError());
  return D3DP_OK;
}

int d3dp_op_linear(int32_t mode, int32_t epi, const void* A, const void* W, const float* bias, void* out, int32_t M,
                   int32_t N, int32_t K, void* stream) {
  if (!A || !W || !bias || !out) return fail(D3DP_EINVAL, "d3dp_op_linear: null argument");
  if (mode == D3DP_MODE_FAST) {
    // epi 0/1: the persistent streaming kernel the denoiser uses (epi | 16 selects its fp32-output form)
    const int e = epi & 3, f32 = (epi & 16) != 0;
    if (e == EPI_RESID || (epi & ~19)) return fail(D3DP_EINVAL, "d3dp_op_linear: FAST mode has epilogues 0, 1 and 0|16");
    LAUNCH_TRY(d3dp_launch_linear_bf16_stream(e, f32, A, W, bias, out, M, N, K, (hipStream_t)stream));
  }
  else if (mode == 2) {
    // split-bf16: A, W are three bf16 planes each (d3dp_op_split3); epi 0 -> fp32 out, epi 1 -> three bf16 planes out
    LAUNCH_TRY(d3dp_launch_linear_bf16x3(epi, A, W, bias, (float*)out, out, M, N, K, (hipStream_t)stream));
  }

  else LAUNCH_TRY(d3dp_launch_linear_f32(epi, (const float*)A, (const float*)W, bias, (float*)out, M, N, K, (hipStream_t)stream));
  HIP_TRY(hipGetLastError());
  return D3DP_OK;
}

int d3dp_op_split3(const float* src, void* dst, size_t n, void* stream) {
  if (!src || !dst) return fail(D3DP_EINVAL, "d3dp_op_split3: null argument");
  d3dp_launch_split3(src, dst, n, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return D3DP_OK;
}

int d3dp_op_linear_x2(int32_t epi, const void* A2, const void* W2, const float* bias, float w_scale, void* out, int32_t M,
                      int32_t N, int32_t K, void* stream) {
  if (!A2 || !W2 || !bias || !out || !(w_scale > 0.f)) return fail(D3DP_EINVAL, "d3dp_op_linear_x2: bad argument");
  const int skew_d = (epi >> 8) & 7;                     // epi | (D << 8), D = 1, 2, 4: the skewed schedule (epi 1 and 4)
  const int pingpong = (epi >> 12) & 1 ? 2 : (epi >> 11) & 1;   // epi | 2048: the ping-pong form, | 4096: the wide form (bit-identical results)
  if ((skew_d || pingpong) && !d3dp_x2_variants_built())
    return fail(D3DP_ENOTSUP, "d3dp_op_linear_x2: epi flags %d select an experiment kernel this library was built without "
                              "(make -C d3dp_amd/csrc variants)", epi & ~0xff);
  epi &= 255;
  if (epi == EPI_RESID_LN || epi == EPI_GELU_LN) return fail(D3DP_EINVAL, "d3dp_op_linear_x2: epilogues 5 / 6 are internal to d3dp_denoise");
  if (skew_d) {
    int dev = 0;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (!d3dp_x2_skew_applies(epi, M, N, K, skew_d, prop.multiProcessorCount))
      return fail(D3DP_EINVAL, "d3dp_op_linear_x2: the skewed schedule (D = %d) does not apply to epi %d, M = %d, N = %d, K = %d", skew_d, epi, M, N, K);
  }
  LAUNCH_TRY(d3dp_launch_linear_f16x2(epi, A2, W2, bias, kActUnscale / w_scale, kActScale, (float*)out, out, nullptr, nullptr, M, N, K, (hipStream_t)stream, skew_d, pingpong));
  HIP_TRY(hipGetLastError());
  return D3DP_OK;
}

int d3dp_op_split2(const float* src, void* dst, size_t n, float scale, void* stream) {
  if (!src || !dst) return fail(D3DP_EINVAL, "d3dp_op_split2: null argument");
  if (n % 32 != 0) return fail(D3DP_EINVAL, "d3dp_op_split2: n = %zu is not a multiple of 32 (the h2i layout is made of whole 32-column blocks)", n);
  d3dp_launch_split2(src, dst, n, scale, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return D3DP_OK;
}

int d3dp_op_attention(int32_t act_bf16, int32_t impl, int32_t axis, const void* qkv, void* out, int32_t n_bh, int32_t F,
                      int32_t J, int32_t C, int32_t heads, void* stream) {
  if (!qkv || !out || n_bh < 1) return fail(D3DP_EINVAL, "d3dp_op_attention: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (act_bf16 != 0 && act_bf16 != 1) return fail(D3DP_EINVAL, "act_bf16 must be 0 or 1");
  if (impl == 2) {       // EXACT mode: split-fp16 operands on the fp16 matrix cores, fp32 in / fp32 out
    if (act_bf16) return fail(D3DP_EINVAL, "split-fp16 attention takes fp32 activations");
    // the kernels read the packed rows the EXACT qkv Linear writes: repack the fp32 rows into a stream-ordered temporary
    void* packed = nullptr;
    const size_t T = (size_t)n_bh * F * J;
    HIP_TRY(hipMallocAsync(&packed, T * 12 * (size_t)C, st));
    d3dp_launch_qkv_pack_x2((const float*)qkv, packed, T, C, kActScale, st);
    const int rc = d3dp_launch_attn_x2(0, axis, packed, out, axis == 0 ? n_bh * F : n_bh * J,
                                       axis == 0 ? spatial_map(F, J) : temporal_map(F, J), C, heads, kActScale, st);
    HIP_TRY(hipFreeAsync(packed, st));
    LAUNCH_TRY(rc);
  } else if (axis == 0 && impl == 1) {
    if (!act_bf16) return fail(D3DP_EINVAL, "MFMA spatial attention needs bf16 activations");
    LAUNCH_TRY(d3dp_launch_attn_spatial_bf16(qkv, out, n_bh * F, spatial_map(F, J), C, heads, st));
  } else if (axis == 0) LAUNCH_TRY(d3dp_launch_attn_rows(act_bf16, qkv, out, n_bh * F, spatial_map(F, J), C, heads, st));
  else if (impl == 1 && !act_bf16) {
    LAUNCH_TRY(d3dp_launch_attn_temporal_f32(0, qkv, out, n_bh * J, temporal_map(F, J), C, heads, st));   // fp32 MFMA
  } else if (impl == 1) {
    if (!act_bf16) return fail(D3DP_EINVAL, "MFMA temporal attention needs bf16 activations");
    LAUNCH_TRY(d3dp_launch_attn_temporal_bf16(qkv, out, n_bh * J, temporal_map(F, J), C, heads, st));
  } else LAUNCH_TRY(d3dp_launch_attn_rows(act_bf16, qkv, out, n_bh * J, temporal_map(F, J), C, heads, st));
  HIP_TRY(hipGetLastError());
  return D3DP_OK;
}

int d3dp_op_layernorm(int32_t out_bf16, const float* x, const float* w, const float* b, float eps, void* out, int32_t T,
                      int32_t C, void* stream) {
  if (!x || !w || !b || !out) return fail(D3DP_EINVAL, "d3dp_op_layernorm: null argument");
  LAUNCH_TRY(d3dp_launch_ln(out_bf16, const_cast<float*>(x), nullptr, 0, w, b, eps, out, T, C, (hipStream_t)stream));
  HIP_TRY(hipGetLastError());
  return D3DP_OK;
}

int d3dp_op_to_bf16(const float* src, void* dst, size_t n, void* stream) {
  if (!src || !dst) return fail(D3DP_EINVAL, "d3dp_op_to_bf16: null argument");
  launch_to_bf16(src, dst, n, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return D3DP_OK;
}

int d3dp_profile_enable(d3dp_ctx* c, int32_t on) {
  if (!c) return fail(D3DP_EINVAL, "null ctx");
  c->prof = on != 0;
  c->used = 0;
  memset(c->counts, 0, sizeof c->counts);
  for (auto& v : c->total_ms) v = 0.0;
  return D3DP_OK;
}

int d3dp_profile_read(d3dp_ctx* c, int64_t* counts, double* total_ms) {
  if (!c || !counts || !total_ms) return fail(D3DP_EINVAL, "null argument");
  if (c->flush_events() != 0) return fail(D3DP_EHIP, "event read failed");
  for (int i = 0; i < D3DP_PROFILE_CLASSES; ++i) { counts[i] = c->counts[i]; total_ms[i] = c->total_ms[i]; }
  memset(c->counts, 0, sizeof c->counts);
  for (auto& v : c->total_ms) v = 0.0;
  return D3DP_OK;
}

}  // extern "C"

// ==================================================================================================================
// Training step: forward of the MixSTE2 train branch (mixste.py:215-225, H = 1) with saved
// activations, and the full backward.  Context must be created with D3DP_MODE_TRAIN (fp32 weights and activations;
// Linears on the fp32 matrix cores).  Not tuned -- correctness first: every Linear gradient is an fp32-MFMA GEMM over
// explicitly transposed operands, everything else is a row-wise fp32 kernel (train.hip).
// ==================================================================================================================
namespace {

struct TrainLayout {
  size_t T, Tpad, C, Hd, unit;      // unit = T*C floats
  // offsets in floats
  size_t temb, x_final, zero_bias, saved0, saved_stride, tmp0;
  // per-block saved tensors (offsets inside a block's slab)
  size_t o_xin, o_qkv, o_att, o_xmid, o_hpre, o_xout;
  // temporaries
  size_t xn, y, hid, dA, dB, dC, dqkv, dh, z, At, Xt, Wt, dtemb, stats;
  // split-fp16 operands of the training Linears (X2Train below): row form [rows][2 K] and transposed form [features][2 Tp]
  size_t op_a, op_w, op_at, op_xt, part, part_rem, slots;
  size_t op_a2, op_at2, part2;      // a second set of the dY operand and partial-tile regions: two weight-gradient products in flight
  size_t x_cols, x_block;           // every Linear's activation operand kept from the forward pass for its wgrad: the ROW form [Tp][2 K]
                                    // where the TN kernel applies (it is then also the forward product's operand), else the transposed [K][2 Tp]
  size_t w_rows, w_cols, w_block;   // every weight's split operands (row form / transposed form), prepared once per step: w_block floats per block
  size_t Tp_max;                    // columns (tokens, padded) of a transposed operand row
  size_t stats_x2, stats_x2_stride; // per block: (log-sum-exp, dO . O) of every query of the split-fp16 temporal attention
  size_t red, red_floats;           // partial sums of the backward pass (LayerNorm gammas / betas, biases, embedding side),
                                    // summed in a fixed order by d3dp_train_reduce_many at its end
  size_t total_floats;
};

constexpr int kGroupSlices = 32;     // slices of the grouped row sums (position / time embedding gradients)

TrainLayout train_layout(const d3dp_cfg& g, int B) {
  TrainLayout L{};
  L.T = (size_t)B * g.frames * g.joints;
  L.Tpad = (L.T + 15) / 16 * 16;
  L.C = g.channels; L.Hd = g.hidden; L.unit = L.T * L.C;
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += (n + 63) / 64 * 64; return o; };
  L.temb = take((size_t)B * L.C);
  L.x_final = take(L.unit);
  L.zero_bias = take(4 * L.C);
  L.o_xin = 0; L.o_qkv = L.unit; L.o_att = 4 * L.unit; L.o_xmid = 5 * L.unit; L.o_hpre = 6 * L.unit;
  L.o_xout = 6 * L.unit + L.T * L.Hd;
  L.saved_stride = (7 * L.unit + L.T * L.Hd + 63) / 64 * 64;
  L.saved0 = take(L.saved_stride * 2 * g.depth);
  L.xn = take(L.unit); L.y = take(L.unit); L.hid = take(L.T * L.Hd);
  L.dA = take(L.unit); L.dB = take(L.unit); L.dC = take(L.unit);
  L.dqkv = take(3 * L.unit); L.dh = take(L.T * L.Hd); L.z = take(L.unit);
  const size_t wide = std::max<size_t>(3 * L.C, L.Hd);
  L.At = take(wide * L.Tpad); L.Xt = take(wide * L.Tpad); L.Wt = take(wide * wide);
  L.dtemb = take((size_t)B * 2 * L.C);              // (the training forward's time-MLP hidden layer borrows it: B x 2 C)
  L.stats = take(d3dp_train_attn_stats_bytes(B * std::max(g.frames, g.joints), std::max(g.frames, g.joints), g.heads) / 4 + 64);
  {
    const size_t fmax = std::max<size_t>(3 * L.C, L.Hd), kmax = std::max<size_t>(L.C, L.Hd);
    L.Tp_max = ((L.T + 31) / 32 + 64) * 32;          // (room for rounding the k-steps up to a multiple of the split count)
    const size_t dy_all = 5 * L.C + L.Hd;            // the four dY of a block side by side (their weight gradients go out as ONE launch)
    L.op_a = take(L.Tp_max * dy_all);                // [Tp][2 N] fp16 = Tp N floats per dY (rows T .. Tp - 1: the TN wgrad's zero rows)
    L.op_w = take(fmax * kmax);                      // [N][2 K] or [K][2 N] fp16
    L.op_at = take(fmax * L.Tp_max);                 // dY^T: [N][2 Tp] fp16
    L.op_xt = take(kmax * L.Tp_max);                 // X^T:  [K][2 Tp] fp16
    L.part = take((size_t)(1024 + 64) * 256 * 128);  // split-K partial products: at most ~ (CUs + tiles) output tiles
    L.op_a2 = take(L.Tp_max * dy_all); L.op_at2 = take(fmax * L.Tp_max); L.part2 = take((size_t)(1024 + 64) * 256 * 128);
    L.part_rem = take((size_t)16 * 256 * fmax);      // ... of a forward / dgrad product's remainder rows (its own region: the
                                                     // weight-gradient product of the same dY runs concurrently on the second stream)
    L.slots = take(2 * 8192);                        // absmax words | 1 / scale per operand
    L.w_block = 4 * L.C * L.C + 2 * L.Hd * L.C;      // qkv | proj | fc1 | fc2 of one block ([N][2 K] fp16 = N K floats each)
    L.x_block = (3 * L.C + L.Hd) * L.Tp_max;         // qkv | proj | fc1 | fc2 inputs of one block
    L.x_cols = take(L.x_block * 2 * g.depth);
    L.w_rows = take(L.w_block * 2 * g.depth);
    L.w_cols = take(L.w_block * 2 * g.depth);
  }
  L.stats_x2_stride = (d3dp_train_attn_x2_stats_bytes(B * g.joints, g.frames, g.heads) / 4 + 63) / 64 * 64;   // (temporal axis: B J sequences of F tokens)
  L.stats_x2 = take(L.stats_x2_stride * 2 * g.depth);
  {
    const size_t lnb = (size_t)D3DP_LN_BWD_BLOCKS * 2 * L.C;         // one LayerNorm call's [dgamma | dbeta] rows
    L.red_floats = (size_t)(6 * g.depth + 2) * lnb                   // 3 LayerNorm backward calls per block + the head's
                   + (size_t)2 * g.depth * std::max(D3DP_DYPREP_ROWS, D3DP_ROWPREP_ROWS) * (5 * L.C + L.Hd + 256)   // bias gradients: the column sums of every dY
                   + (size_t)512 * (3 * L.C + 4) + (size_t)D3DP_EMBED_BWD_ROWS * 5 * L.C + (size_t)512 * L.C   // head, embedding
                   + (size_t)kGroupSlices * (g.joints + (size_t)B) * L.C + 4096;                                // spos, time embedding
    L.red = take(L.red_floats);
  }
  L.total_floats = off;
  return L;
}

const float* mask_ptr(const float* masks, const d3dp_cfg& g, int B, int blk, int branch) {
  if (!masks) return nullptr;
  const size_t smax = (size_t)B * std::max(g.frames, g.joints);
  return masks + ((size_t)blk * 2 + branch) * smax;
}

// The backward pass' deferred, fixed-order sums: kernels leave partial rows in the workspace's `red` region (bump-allocated
// here), `add` only RECORDS where they go -- an item may be recorded long before its partial rows exist (the shared norms' rows are
// written by one LayerNorm-backward call per block) -- and `flush`, called once every producer has been launched, walks the list
// in launches of <= D3DP_REDUCE_MAX destinations (d3dp_train_reduce_many).  (Round 5 flushed from inside `add` when the table
// was full: at depth >= 6 that reduced the shared norms' rows before the remaining blocks had written theirs -- ADVICE r5.)
// Nothing in the backward pass adds floats atomically: its gradients are bit-reproducible.
struct Reducer {
  float* base;
  size_t cap, used = 0;
  hipStream_t st;
  std::vector<D3dpReduceItem> items{};
  int rc = 0;
  float* take(size_t n) {
    n = (n + 63) / 64 * 64;
    if (used + n > cap) { rc = -1; return nullptr; }
    float* p = base + used;
    used += n;
    return p;
  }
  void add(const float* part, float* dst, size_t n, size_t count, size_t stride, bool accumulate = false) {
    if (!part || !dst || rc) { rc = rc ? rc : -1; return; }
    items.push_back(D3dpReduceItem{part, dst, (unsigned)n, (unsigned)count, (unsigned)stride, accumulate ? 1u : 0u});
  }
  void flush() {
    for (size_t i = 0; i < items.size() && rc == 0; i += D3DP_REDUCE_MAX) {
      D3dpReduceTable tb{};
      tb.count = (int)std::min<size_t>(D3DP_REDUCE_MAX, items.size() - i);
      for (int j = 0; j < tb.count; ++j) tb.it[j] = items[i + j];
      rc = d3dp_train_reduce_many(tb, st);
    }
    items.clear();
  }
};

int lin32(const float* A, const float* W, const float* bias, float* out, int M, int N, int K, hipStream_t st) {
  return d3dp_launch_linear_f32(EPI_BIAS, A, W, bias, out, M, N, K, st);
}

// The training step's Linears on the split-fp16 scheme of EXACT inference (three fp16-MFMA passes, fp32-class; gemm_x2.hip
// gemm_f16x2_dyn_kernel) instead of the fp32 matrix cores (1/16 of the fp16 rate).  Every operand -- activations, weights,
// GRADIENTS -- is split at the power of two its own absmax asks for, computed and consumed on the device: no range is
// assumed and nothing synchronises.  forward: out = A W^T + b;  dgrad: dX = dY W (W^T split as the "weight" operand);
// wgrad: dW = dY^T X contracts over the batch's tokens into a handful of output tiles -> split-K over Z chunks of tokens,
// partial products summed in a fixed order (deterministic; the fp32 path's split-K used atomics).
struct X2Train {
  hipStream_t st;
  float* ws;
  const TrainLayout& L;
  int n_cu;
  d3dp_ctx* pc = nullptr;                              // per-kernel profile (d3dp_profile_enable): classes T_LINEAR / T_WGRAD / T_OPERAND
  hipStream_t st_w = nullptr;                          // the weight-gradient products' stream (null: `st`)
  bool tail_blocks = true;                             // d3dp_ctx::train_tail_blocks
  // the dY operand (row form / transposed form) and the weight-gradient product's partial tiles live in one of two sets of
  // regions: the backward pass alternates, so that the operand pass of the next dY need not wait for the product of this one
  size_t o_a = 0, o_at = 0, o_part = 0, set_base = 0, set_used = 0;
  void use_set(int s) {
    set_base = o_a = s ? L.op_a2 : L.op_a; set_used = 0; n_def = 0;
    o_at = s ? L.op_at2 : L.op_at; o_part = s ? L.part2 : L.part;
  }
  // The four weight gradients of a block as ONE launch (gemm_f16x2_tn_kernel's product table): each dY's operand pass takes the
  // next region of the set, wgrad() only records its product, flush_wgrads() launches them and the sum of their partial tiles.
  // On where every Linear of the block has a TN shape (merged_setup); mZ / mTp: split count and padded token count of the merged list.
  bool merged = false;
  int mZ = 1, mTp = 0, n_def = 0;
  D3dpTnProduct def[D3DP_TN_MAX];
  float* def_dw[D3DP_TN_MAX];
  void merged_setup(int T, bool on) {
    const int C = (int)L.C, Hd = (int)L.Hd;
    const int shapes[4][2] = {{C, Hd}, {Hd, C}, {C, C}, {3 * C, C}};     // fc2, fc1, proj, qkv: [N, K] of dW
    int tiles = 0;
    bool ok = on;
    size_t nk_sum = 0;
    for (auto& sh : shapes) { ok = ok && d3dp_tn_applies(sh[0], sh[1]); tiles += ((sh[0] + 255) / 256) * ((sh[1] + 127) / 128); nk_sum += (size_t)sh[0] * sh[1]; }
    const int nk = (T + 31) / 32;
    mZ = std::max(1, std::min(std::min(n_cu / std::max(tiles, 1), 64), nk / 4));
    mTp = mZ * ((nk + mZ - 1) / mZ) * 32;
    merged = ok && (size_t)mTp <= L.Tp_max && (size_t)mZ * nk_sum <= (size_t)(1024 + 64) * 256 * 128;
  }
  // rows the operands of the weight gradient dW[N, K] must have (zero behind T)
  int pad_rows(int T, int N, int K) const {
    int Z, Tp;
    wgrad_split(T, N, K, Z, Tp);
    return merged ? std::max(Tp, mTp) : Tp;
  }
  int flush_wgrads() {
    if (!n_def) return 0;
    hipStream_t sw = st_w ? st_w : st;
    Scope ps(pc, T_WGRAD, sw);
    int r = d3dp_launch_linear_f16x2_tn_many(def, n_def, mTp, mZ, sw);
    if (r) return r;
    D3dpSumTable tb{};
    tb.n = n_def; tb.Z = mZ;
    for (int i = 0; i < n_def; ++i) { tb.part[i] = def[i].out; tb.out[i] = def_dw[i]; tb.n4[i] = (size_t)def[i].N * def[i].K / 4; }
    d3dp_launch_sum_partials_many(tb, sw);
    n_def = 0;
    return 0;
  }
  // absmax / unscale slots: the forward pass owns slots 2 l (activation operand) and 2 l + 1 (weight) of its Linear l = 4 block
  // + {qkv, proj, fc1, fc2} and leaves them -- with the operands themselves: the weights' two forms and every activation's
  // transposed form -- for the backward pass of the same step, whose wgrad / dgrad read the same tensors: 128 of the step's
  // 320 absmax launches are not repeated.  The backward pass allocates its own (the gradients) from kBwdSlot0.
  static constexpr int kBwdSlot0 = 4096;
  static constexpr int kQkvSlot0 = 2048;               // forward range: slot kQkvSlot0 + block = absmax of that block's qkv OUTPUT
  int next = 0;
  unsigned* amax() const { return reinterpret_cast<unsigned*>(ws + L.slots); }
  float* uns() const { return ws + L.slots + 8192; }
  int begin(bool forward_pass) {
    next = forward_pass ? 0 : kBwdSlot0;
    return hipMemsetAsync(ws + L.slots + next, 0, (size_t)kBwdSlot0 * 4, st) == hipSuccess ? 0 : -3;
  }
  int take() { return (next < kBwdSlot0 || next >= 8192) ? -1 : next++; }   // a fresh slot of the backward range for a producer kernel to fill
  int slot_for(const float* src, size_t n) {           // absmax of a tensor into a fresh slot of the backward range
    if (next < kBwdSlot0 || next >= 8192) return -1;
    d3dp_launch_absmax(src, n, amax() + next, st);
    return next++;
  }
  void rows(const float* src, int R, int C, void* dst, int slot) {          // [R][C] -> [R][2 C]
    d3dp_launch_split2_dyn(src, dst, R, C, C, amax() + slot, uns() + slot, st);
  }
  void cols(const float* src, int R, int C, int Rpad, void* dst, int slot) {   // [R][C] -> [C][2 Rpad]
    d3dp_launch_split2_t_dyn(src, dst, R, C, Rpad, amax() + slot, uns() + slot, st);
  }
  // Every weight's absmax, row form and transposed form in three launches at the start of the forward pass (up to 64
  // Linears: depth <= 8; deeper models prepare each weight where it is used).  woff(l): float offset of Linear l's operands
  // inside the w_rows / w_cols regions.
  bool batched = false;
  size_t xoff(int l) const { return (size_t)(l >> 2) * L.x_block + (size_t)(l & 3) * L.C * L.Tp_max; }
  size_t woff(int l) const {
    const size_t CC = L.C * L.C, j = (size_t)(l & 3);
    return (size_t)(l >> 2) * L.w_block + (j == 0 ? 0 : j == 1 ? 3 * CC : j == 2 ? 4 * CC : 4 * CC + L.Hd * L.C);
  }
  int prepare_weights(const d3dp_ctx* c) {
    const int nl = 8 * c->cfg.depth;
    batched = nl <= D3DP_WPREP_MAX;
    if (!batched) return 0;
    D3dpWPrepTable tb{};
    tb.n = nl;
    const int C = (int)L.C, Hd = (int)L.Hd;
    for (int l = 0; l < nl; ++l) {
      const int blk = l >> 2, j = l & 3;
      const BlockDev& w = (blk & 1) ? c->tte[blk >> 1] : c->ste[blk >> 1];
      const void* p = j == 0 ? w.qkv_w : j == 1 ? w.proj_w : j == 2 ? w.fc1_w : w.fc2_w;
      tb.it[l] = D3dpWPrepItem{(const float*)p, j == 0 ? 3 * C : j == 2 ? Hd : C, j == 3 ? Hd : C, 2 * l + 1, 0, woff(l)};
    }
    return d3dp_launch_wprep(tb, ws + L.w_rows, ws + L.w_cols, amax(), uns(), st);
  }
  // out[T, N] = A2 . W2^T (+ bias) with T = 256 q + rem rows.  The persistent kernel works in rounds of n_cu tiles of 256 x 128;
  // the configs[4] batch has T = 16,524 = 64 x 256 + 140, so every forward / dgrad product had ONE row of tiles too many for
  // a whole number of rounds (fc2: 260 tiles on 256 CUs = two rounds for 1.02 rounds of work).  Where the remainder rows would cost
  // a round of their own the kernel takes them as 16 x 64 blocks spread over all workgroups (round 5; gemm_f16x2_dyn_kernel).
  // D3DP_TRAIN_TAIL=split, and contractions that are not a multiple of 16 k-steps, keep round 4's form: from 32 k-steps on a
  // second launch as a split-K product (Z chunks of the contraction: tn Z short work items instead of tn long ones) whose
  // partial sums are added in a fixed order, else the extra round.
  int gemm(const float* A2, const float* W2, const float* bias, const float* ua, const float* uw, float* out, int T, int N,
           int K, unsigned* out_amax = nullptr, int amax_pos = 0) {
    Scope ps(pc, T_LINEAR, st);
    const int tn = (N + 127) / 128, q = T / 256, rem = T - q * 256;
    const int rounds_all = ((q + (rem ? 1 : 0)) * tn + n_cu - 1) / n_cu, rounds_full = (q * tn + n_cu - 1) / n_cu;
    const int nk = K / 32;
    int Z = 1;
    for (int z = 2; z <= 16; ++z)
      if (nk % z == 0) Z = z;
    // (measured: a last round of a few tiles runs its k-steps at 0.75 us -- few CUs active, full clock -- against 1.4 us in a
    //  full round, so for 16 k-steps it costs 12 us, what the second launch and the sum cost too: split from 32 k-steps on)
    // the remainder rows as 16 x 64 blocks at the end of the same launch (gemm_f16x2_dyn_kernel): wherever they would cost a round
    if (tail_blocks && rem && q && rounds_all > rounds_full && nk % 16 == 0)
      return d3dp_launch_linear_f16x2_dyn(A2, W2, bias, ua, uw, out, T, N, K, 1, st, out_amax, amax_pos, 1);
    if (rem == 0 || q == 0 || rounds_all == rounds_full || Z == 1 || nk < 32 || (size_t)Z * rem * N > (size_t)16 * 256 * std::max<size_t>(3 * L.C, L.Hd))
      return d3dp_launch_linear_f16x2_dyn(A2, W2, bias, ua, uw, out, T, N, K, 1, st, out_amax, amax_pos);
    int r = d3dp_launch_linear_f16x2_dyn(A2, W2, bias, ua, uw, out, q * 256, N, K, 1, st, out_amax, amax_pos);
    if (r) return r;
    r = d3dp_launch_linear_f16x2_dyn(A2 + (size_t)q * 256 * K, W2, nullptr, ua, uw, ws + L.part_rem, rem, N, K, Z, st);   // (a row = 2 K fp16 = K floats)
    if (r) return r;
    d3dp_launch_sum_partials_bias(ws + L.part_rem, bias, out + (size_t)q * 256 * N, (size_t)rem * N, N, Z, st, out_amax, amax_pos);
    return 0;
  }
  // out[T, N] = A[T, K] W[N, K]^T + bias     (l: the Linear's index, see the slots above)
  // (a_amax_ready: the kernel that produced A already left its absmax in slot 2 l)
  // (out_amax: optional slot for the OUTPUT's absmax, left by the product's epilogue)
  // (a_prepared: the activation operand of Linear l -- planes at x_cols + xoff(l), zero rows behind T, unscale slot 2 l -- was
  //  written by its producer: gelu_operand below)
  // (ln_w / ln_b / ln_eps: A is the INPUT of the LayerNorm whose output is this Linear's operand, slot 2 l holds that output's
  //  absmax: the operand pass normalises on the fly, d3dp_launch_rowprep_ln -- TN shapes with K <= 512 only, see ln_operand_applies)
  bool ln_operand_applies(int N, int K) const { return d3dp_tn_applies(N, K) && K <= 512; }
  int forward(int l, const float* A, const float* W, const float* bias, float* out, int T, int N, int K, bool a_amax_ready = false,
              unsigned* out_amax = nullptr, int amax_pos = 0, bool a_prepared = false, const float* ln_w = nullptr,
              const float* ln_b = nullptr, float ln_eps = 0.f) {
    const int sa = 2 * l, sw = 2 * l + 1;
    if (l < 0 || sw >= kBwdSlot0) return -1;
    const float* a2 = ws + o_a;
    if (a_prepared) a2 = ws + L.x_cols + xoff(l);
    else {
      Scope ps(pc, T_OPERAND, st);
      if (!a_amax_ready) d3dp_launch_absmax(A, (size_t)T * K, amax() + sa, st);
      // the operand of this product and, in the same pass, what the wgrad of this Linear will want of it: the backward pass
      // then neither recomputes this activation (LayerNorm / GELU outputs) nor reads it again.  Where the TN wgrad kernel applies
      // that is the SAME row form (kept, zero rows behind T), else a second, transposed form.
      const int Tp = pad_rows(T, N, K);
      if ((size_t)Tp > L.Tp_max) return -1;
      int r;
      if (ln_w) {
        if (!ln_operand_applies(N, K) || !a_amax_ready) return -1;
        a2 = ws + L.x_cols + xoff(l);
        r = d3dp_launch_rowprep_ln(A, ln_w, ln_b, ln_eps, ws + L.x_cols + xoff(l), T, Tp, K, amax() + sa, uns() + sa, st);
      } else if (d3dp_tn_applies(N, K)) {
        a2 = ws + L.x_cols + xoff(l);
        r = d3dp_launch_rowprep(A, ws + L.x_cols + xoff(l), nullptr, nullptr, T, Tp, K, amax() + sa, uns() + sa, st);
      } else r = d3dp_launch_dyprep(A, ws + o_a, ws + L.x_cols + xoff(l), nullptr, T, K, Tp, amax() + sa, uns() + sa, st);
      if (r) return r;
    }
    const float* w2 = ws + L.op_w;
    if (batched) w2 = ws + L.w_rows + woff(l);
    else {
      Scope ps(pc, T_OPERAND, st);
      d3dp_launch_absmax(W, (size_t)N * K, amax() + sw, st);
      rows(W, N, K, ws + L.op_w, sw);
    }
    return gemm(a2, w2, bias, uns() + sa, uns() + sw, out, T, N, K, out_amax, amax_pos);
  }
  // whether Linear l (N out features, K in) takes its GELU'd input straight from its producer's output: TN shapes only (the
  // row form is then also the wgrad operand)
  bool gelu_operand_applies(int N, int K) const { return d3dp_tn_applies(N, K); }
  static constexpr int kPmaxSlot0 = 3072;              // forward range: slot kPmaxSlot0 + block = largest positive fc1 output
  int gelu_operand(int l, const float* hpre, int blk, int T, int N, int K) {
    const int Tp = pad_rows(T, N, K);
    if ((size_t)Tp > L.Tp_max) return -1;
    Scope ps(pc, T_OPERAND, st);
    return d3dp_launch_gelu_rowprep(hpre, ws + L.x_cols + xoff(l), T, Tp, K, amax() + kPmaxSlot0 + blk, amax() + 2 * l, uns() + 2 * l, st);
  }
  // dX[T, K] = dY[T, N] W[N, K]   (sdy: the absmax slot of dY, shared with wgrad; l: the forward Linear whose W this is)
  int dgrad(int l, const float* dY, int sdy, const float* W, float* dX, int T, int N, int K, unsigned* out_amax = nullptr) {
    const int sw = 2 * l + 1;
    {
      Scope ps(pc, T_OPERAND, st);
      if (!dy_ready) rows(dY, T, N, ws + o_a, sdy);
      if (!batched) cols(W, N, K, N, ws + L.op_w, sw);   // W^T: [K][2 N]  (N % 32 == 0: the model's widths)
    }
    const float* wt = ws + L.op_w;
    if (batched) wt = ws + L.w_cols + woff(l);         // (prepared by the forward pass of this step)
    return gemm(ws + o_a, wt, nullptr, uns() + sdy, uns() + sw, dX, T, K, N, out_amax);
  }
  // split count / padded token count of the wgrad product dW[N, K] = dY^T X
  void wgrad_split(int T, int N, int K, int& Z, int& Tp) const {
    const int tiles = ((N + 255) / 256) * ((K + 127) / 128);
    const int nk = (T + 31) / 32;
    Z = std::max(1, std::min(std::min(n_cu / tiles, 64), nk / 4));
    const int nkz = (nk + Z - 1) / Z;
    Tp = Z * nkz * 32;
  }
  // everything the backward pass of Linear [N, K] needs from its dY in one pass: row form -> op_a (dgrad), transposed form
  // -> op_at (wgrad), the bias gradient's partial column sums -> bias_part (D3DP_DYPREP_ROWS rows of N floats).  `dy_ready`
  // then tells dgrad / wgrad not to build them again.
  bool dy_ready = false;
  static constexpr int kBiasRows = D3DP_DYPREP_ROWS > D3DP_ROWPREP_ROWS ? D3DP_DYPREP_ROWS : D3DP_ROWPREP_ROWS;   // capacity of bias_part
  // (mask: the DropPath scales still to be applied to dY's rows -- TN shapes only, see mask_in_prep_applies)
  // (gelu_pre: dY is d hidden and the Linear's dY is dY x gelu'(gelu_pre), formed by the operand pass; slot sdy then holds the
  //  absmax of d hidden -- TN shapes the row pass takes only, see gelu_in_prep_applies)
  bool mask_in_prep_applies(int N, int K) const { return d3dp_tn_applies(N, K); }
  bool gelu_in_prep_applies(int N, int K) const { return d3dp_tn_applies(N, K) && N <= 1536; }
  int prep_dy(const float* dY, int sdy, float* bias_part, int* bias_rows, int T, int N, int K, const float* mask = nullptr,
              int axis = 0, int F = 1, int J = 1, const float* gelu_pre = nullptr) {
    const int Tp = pad_rows(T, N, K);
    if ((size_t)Tp > L.Tp_max) return -1;
    if (merged) {                    // the next region of the set: the block's four dY stay until flush_wgrads()
      if (set_used + L.Tp_max * (size_t)N > L.Tp_max * (5 * L.C + L.Hd)) return -1;
      o_a = set_base + set_used;
      set_used += L.Tp_max * (size_t)N;
    }
    dy_ready = true;
    Scope ps(pc, T_OPERAND, st);
    if (d3dp_tn_applies(N, K))       // the row form alone (zero rows behind T): dgrad's operand AND the TN wgrad's
      return d3dp_launch_rowprep(dY, ws + o_a, bias_part, bias_rows, T, Tp, N, amax() + sdy, uns() + sdy, st, mask, axis, F, J,
                                 gelu_pre);
    if (mask || gelu_pre) return -1;
    *bias_rows = D3DP_DYPREP_ROWS;
    return d3dp_launch_dyprep(dY, ws + o_a, ws + o_at, bias_part, T, N, Tp, amax() + sdy, uns() + sdy, st);
  }
  // dW[N, K] = dY[T, N]^T X[T, K]   (X: the activation operand of forward Linear l)
  int wgrad(int l, const float* dY, int sdy, const float* X, float* dW, int T, int N, int K) {
    const int sx = 2 * l;
    int Z, Tp;
    wgrad_split(T, N, K, Z, Tp);
    if ((size_t)Tp > L.Tp_max || (size_t)Z * N * K > (size_t)(1024 + 64) * 256 * 128) return -1;
    (void)X;                                           // X's operand form: left by the forward pass of this step
    if (merged) {                                      // recorded; launched with the block's other three by flush_wgrads()
      if (!dy_ready || n_def >= D3DP_TN_MAX || !d3dp_tn_applies(N, K)) return -1;
      size_t po = 0;
      for (int i = 0; i < n_def; ++i) po += (size_t)mZ * def[i].N * def[i].K;
      def[n_def] = D3dpTnProduct{ws + o_a, ws + L.x_cols + xoff(l), uns() + sdy, uns() + sx, ws + o_part + po, N, K, 0, 0};
      def_dw[n_def++] = dW;
      return 0;
    }
    int r;
    hipStream_t sw = st_w ? st_w : st;
    Scope ps(pc, T_WGRAD, sw);
    if (d3dp_tn_applies(N, K)) {                        // both operands in their row forms [Tp][2 .]: nothing was transposed
      if (!dy_ready) return -1;
      r = d3dp_launch_linear_f16x2_tn(ws + o_a, ws + L.x_cols + xoff(l), uns() + sdy, uns() + sx, ws + o_part, N, K, Tp, Z, sw);
    } else {
      if (!dy_ready) return -1;                         // (the transposed form: left by prep_dy)
      r = d3dp_launch_linear_f16x2_dyn(ws + o_at, ws + L.x_cols + xoff(l), nullptr, uns() + sdy, uns() + sx, ws + o_part, N, K,
                                       Tp, Z, sw);
    }
    if (r) return r;
    d3dp_launch_sum_partials(ws + o_part, dW, (size_t)N * K, Z, sw);
    return 0;
  }
};

}  // namespace

extern "C" {

int d3dp_train_workspace_bytes(const d3dp_ctx* c, int32_t B, size_t* bytes) {
  if (!c || !bytes || B < 1) return fail(D3DP_EINVAL, "d3dp_train_workspace_bytes: bad argument");
  *bytes = train_layout(c->cfg, B).total_floats * 4;
  return D3DP_OK;
}

int d3dp_train_forward(d3dp_ctx* c, const float* x2d, const float* x3d, const int64_t* t, const float* masks, float* out,
                       int32_t B, void* workspace, size_t workspace_bytes, void* stream) {
  if (!c || !x2d || !x3d || !t || !out || !workspace || B < 1) return fail(D3DP_EINVAL, "d3dp_train_forward: bad argument");
  if (!c->train()) return fail(D3DP_ESTATE, "d3dp_train_forward needs a D3DP_MODE_TRAIN context");
  if (!c->weights_set) return fail(D3DP_ESTATE, "weights not set");
  const d3dp_cfg& g = c->cfg;
  // clips longer than 256 frames (reference common/arguments.py:58, main.py:325 train at any `-f`): the split-fp16 attention kernels
  // pass their keys / queries through LDS in chunks (train_attn.hip, round 6); the fp32 cross-check kernels hold whole sequences
  if (g.frames > 256 && !(c->train_x2 && c->train_attn_x2 == 2 && g.channels / g.heads == 64 && g.channels % 32 == 0 && g.hidden % 32 == 0))
    return fail(D3DP_ENOTSUP, "frames=%d > 256: the training step runs such clips on its split-fp16 attention kernels only (head dim 64, "
                              "no D3DP_TRAIN_IMPL=f32 / D3DP_TRAIN_ATTN=f32|x2t cross-check)", g.frames);
  const TrainLayout L = train_layout(g, B);
  if (workspace_bytes < L.total_floats * 4) return fail(D3DP_ESTATE, "train workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const int T = (int)L.T, C = g.channels, F = g.frames, J = g.joints, Hd = g.hidden;
  float *xn = ws + L.xn, *y = ws + L.y, *hid = ws + L.hid;
  X2Train x2{st, ws, L, c->n_cu};
  x2.pc = c->prof ? c : nullptr;
  d3dp_ctx* const pc = x2.pc;                            // (TP: one launch inside a profile scope of its class)
#define TP(cls, call) { Scope ps_(pc, cls, st); LAUNCH_TRY(call); }
  x2.use_set(0);
  x2.tail_blocks = c->train_tail_blocks;
  x2.merged_setup(T, c->train_wgrad_merged);
  const bool use_x2 = c->train_x2 && C % 32 == 0 && Hd % 32 == 0;
  if (use_x2) {
    Scope ps_(pc, T_OTHER, st);
    LAUNCH_TRY(x2.begin(true));
    LAUNCH_TRY(x2.prepare_weights(c));
  }
  auto lin = [&](int l, const float* A, const float* W, const float* bias, float* out, int M, int N, int K, bool a_amax_ready = false,
                 unsigned* out_amax = nullptr, int amax_pos = 0, bool a_prepared = false, const float* ln_w = nullptr,
                 const float* ln_b = nullptr, float ln_eps = 0.f) {
    if (use_x2) return x2.forward(l, A, W, bias, out, M, N, K, a_amax_ready, out_amax, amax_pos, a_prepared, ln_w, ln_b, ln_eps);
    Scope ps_(pc, T_LINEAR, st);
    return lin32(A, W, bias, out, M, N, K, st);
  };
  // the qkv / fc1 operands straight from the INPUT of the LayerNorm in front of them (no fp32 normalised activation is written)
  const bool ln_fused = use_x2 && x2.ln_operand_applies(3 * C, C) && x2.ln_operand_applies(Hd, C);
  // ... and (round 6) written BY the kernel that produces that input, at the scale the LayerNorm's own weights bound (train.hip
  // ln_operand_scale): the operand pass in front of qkv / fc1 disappears (block 0's qkv operand: the embedding kernel is shared
  // with inference and keeps the pass)
  const bool ln_direct = ln_fused && c->train_ln_direct && C % 256 == 0;
  // attention on split-fp16 operands (train_attn.hip): needs the split Linears' device-side scales, head dim 64, <= 256 frames
  const bool attn_x2 = use_x2 && c->train_attn_x2 > 0 && C / g.heads == 64 && F <= 1024;
  const bool attn_x2_s = attn_x2 && c->train_attn_x2 > 1;      // the spatial axis too
  TP(T_OTHER, d3dp_train_time_mlp(t, c->freq, c->t1w, c->t1b, c->t3w, c->t3b, ws + L.dtemb, ws + L.temb, B, C, st));   // (dtemb: free until the backward pass; needs B x 2 C)
  float* slab0 = ws + L.saved0;
  TP(T_LN_FWD, d3dp_launch_embed_ln(0, x2d, x3d, ws + L.temb, c->ew, c->eb, c->spos, c->ste[0].n1w, c->ste[0].n1b,
                                    g.eps_block, slab0 + L.o_xin, xn, 0, B, 1, F, J, C, st));
  // Every Linear's activation operand arrives with its absmax already in slot 2 l, left there by the kernel that produced it
  // (round 4 ran 64 absmax launches per step); the two exceptions are the first block's input, which the inference embedding
  // kernel writes, and the temporal attention's output (the fp32-MFMA kernel is shared with inference).
  auto slot = [&](int l) -> unsigned* { return use_x2 ? x2.amax() + 2 * l : nullptr; };
  if (use_x2) { Scope ps_(pc, T_OTHER, st); d3dp_launch_absmax(xn, (size_t)T * C, slot(0), st); }
  for (int blk = 0; blk < 2 * g.depth; ++blk) {
    const int kind = blk & 1, d = blk >> 1;
    const BlockDev& w = kind ? c->tte[d] : c->ste[d];
    float* S = ws + L.saved0 + (size_t)blk * L.saved_stride;
    const bool ax2 = kind == 1 ? attn_x2 : attn_x2_s;
    unsigned* qkv_amax = ax2 ? x2.amax() + X2Train::kQkvSlot0 + blk : nullptr;
    if (ln_direct && blk > 0) LAUNCH_TRY(lin(4 * blk, nullptr, (const float*)w.qkv_w, w.qkv_b, S + L.o_qkv, T, 3 * C, C, true, qkv_amax, 0, true));
    else if (ln_fused) LAUNCH_TRY(lin(4 * blk, S + L.o_xin, (const float*)w.qkv_w, w.qkv_b, S + L.o_qkv, T, 3 * C, C, true, qkv_amax, 0, false,
                                      w.n1w, w.n1b, g.eps_block));
    else LAUNCH_TRY(lin(4 * blk, xn, (const float*)w.qkv_w, w.qkv_b, S + L.o_qkv, T, 3 * C, C, true, qkv_amax));
    bool att_ready = true;
    // the split-fp16 attention also writes the proj Linear's operand rows (scale: the qkv absmax bounds every output), round 6
    const bool att_direct = ax2 && ln_direct && d3dp_tn_applies(C, C);
    void* att_op = att_direct ? ws + L.x_cols + x2.xoff(4 * blk + 1) : nullptr;
    const int att_tp = att_direct ? x2.pad_rows(T, C, C) : 0;
    float* att_un = att_direct ? x2.uns() + 2 * (4 * blk + 1) : nullptr;
    auto attention = [&]() -> int {                      // (the launcher's own return code)
      if (kind == 0 && ax2)
        return d3dp_train_attn_x2_fwd(S + L.o_qkv, S + L.o_att, ws + L.stats_x2 + (size_t)blk * L.stats_x2_stride, B * F,
                                      spatial_map(F, J), C, g.heads, qkv_amax, slot(4 * blk + 1), st, att_op, T, att_tp, att_un);
      if (kind == 0) return d3dp_launch_attn_rows(0, S + L.o_qkv, S + L.o_att, B * F, spatial_map(F, J), C, g.heads, st, slot(4 * blk + 1));
      if (ax2)
        return d3dp_train_attn_x2_fwd(S + L.o_qkv, S + L.o_att, ws + L.stats_x2 + (size_t)blk * L.stats_x2_stride, B * J,
                                      temporal_map(F, J), C, g.heads, qkv_amax, slot(4 * blk + 1), st, att_op, T, att_tp, att_un);
      if (use_x2 && C / g.heads == 64 && F <= 256) {     // temporal axis on the fp32 matrix cores (bitwise an fp32 fmaf chain per product)
        att_ready = false;
        return d3dp_launch_attn_temporal_f32(0, S + L.o_qkv, S + L.o_att, B * J, temporal_map(F, J), C, g.heads, st);
      }
      return d3dp_launch_attn_rows(0, S + L.o_qkv, S + L.o_att, B * J, temporal_map(F, J), C, g.heads, st, slot(4 * blk + 1));
    };
    TP(kind ? T_ATTN_FWD_T : T_ATTN_FWD_S, attention());
    LAUNCH_TRY(lin(4 * blk + 1, S + L.o_att, (const float*)w.proj_w, w.proj_b, y, T, C, C, att_ready, nullptr, 0, att_direct));
    if (ln_direct)
      TP(T_LN_FWD, d3dp_train_add_mask_ln(S + L.o_xin, y, mask_ptr(masks, g, B, blk, 0), kind, F, J, w.n2w, w.n2b, g.eps_block,
                                          S + L.o_xmid, nullptr, slot(4 * blk + 2), T, C, st, ws + L.x_cols + x2.xoff(4 * blk + 2),
                                          x2.pad_rows(T, Hd, C), x2.uns() + 2 * (4 * blk + 2)))
    else
      TP(T_LN_FWD, d3dp_train_add_mask_ln(S + L.o_xin, y, mask_ptr(masks, g, B, blk, 0), kind, F, J, w.n2w, w.n2b, g.eps_block,
                                          S + L.o_xmid, ln_fused ? nullptr : xn, slot(4 * blk + 2), T, C, st));
    const float* fc1_in = ln_fused ? S + L.o_xmid : xn;
    const float *f1w = ln_fused ? w.n2w : nullptr, *f1b = ln_fused ? w.n2b : nullptr;
    if (use_x2 && x2.gelu_operand_applies(C, Hd)) {
      // the fc2 operand = split(GELU(fc1 output)) in one pass, its scale from the largest positive fc1 output (the fc1
      // epilogue leaves it): no fp32 hidden tensor, no separate operand pass
      if (ln_direct) LAUNCH_TRY(lin(4 * blk + 2, nullptr, (const float*)w.fc1_w, w.fc1_b, S + L.o_hpre, T, Hd, C, true,
                                    x2.amax() + X2Train::kPmaxSlot0 + blk, 1, true));
      else LAUNCH_TRY(lin(4 * blk + 2, fc1_in, (const float*)w.fc1_w, w.fc1_b, S + L.o_hpre, T, Hd, C, true,
                          x2.amax() + X2Train::kPmaxSlot0 + blk, 1, false, f1w, f1b, g.eps_block));
      LAUNCH_TRY(x2.gelu_operand(4 * blk + 3, S + L.o_hpre, blk, T, C, Hd));
      LAUNCH_TRY(lin(4 * blk + 3, nullptr, (const float*)w.fc2_w, w.fc2_b, y, T, C, Hd, true, nullptr, 0, true));
    } else {
      LAUNCH_TRY(lin(4 * blk + 2, fc1_in, (const float*)w.fc1_w, w.fc1_b, S + L.o_hpre, T, Hd, C, true, nullptr, 0, ln_direct, f1w, f1b,
                     g.eps_block));
      TP(T_OPERAND, d3dp_train_gelu_fwd(S + L.o_hpre, hid, (size_t)T * Hd, slot(4 * blk + 3), st));
      LAUNCH_TRY(lin(4 * blk + 3, hid, (const float*)w.fc2_w, w.fc2_b, y, T, C, Hd, true));
    }
    // the block's end in one pass: residual add, the shared norm (+ Temporal_pos_embed after the first spatial block,
    // mixste.py:250) -> the next block's input, that block's norm1 -> its qkv operand (after the last block: the head's
    // LayerNorm -> z)
    const bool last = blk == 2 * g.depth - 1;
    float* x_next = last ? ws + L.x_final : S + L.saved_stride + L.o_xin;
    const float *nw = c->hnw, *nb = c->hnb;
    if (!last) { const BlockDev& wn = kind ? c->ste[d + 1] : c->tte[d]; nw = wn.n1w; nb = wn.n1b; }
    TP(T_LN_FWD, d3dp_train_add_mask_ln2(S + L.o_xmid, y, mask_ptr(masks, g, B, blk, 1), kind, F, J, kind ? c->tnw : c->snw,
                                         kind ? c->tnb : c->snb, g.eps_block, (kind == 0 && d == 0) ? c->tpos : nullptr, nw, nb,
                                         last ? g.eps_head : g.eps_block, S + L.o_xout, x_next,
                                         last ? ws + L.z : (ln_fused ? nullptr : xn), last ? nullptr : slot(4 * (blk + 1)), T, C, st,
                                         (ln_direct && !last) ? ws + L.x_cols + x2.xoff(4 * (blk + 1)) : nullptr,
                                         (ln_direct && !last) ? x2.pad_rows(T, 3 * C, C) : 0,
                                         (ln_direct && !last) ? x2.uns() + 2 * 4 * (blk + 1) : nullptr));
  }
  TP(T_OTHER, d3dp_train_head_linear(ws + L.z, c->hw, c->hb, out, T, C, st));
#undef TP
  HIP_TRY(hipGetLastError());
  return D3DP_OK;
}

// grads: device fp32 buffers with the shapes of the corresponding weights (time_freq ignored); they are ZEROED here
// and then accumulated.  Must be called after d3dp_train_forward with the same inputs, masks and workspace.
int d3dp_train_backward(d3dp_ctx* c, const float* x2d, const float* x3d, const int64_t* t, const float* masks,
                        const float* grad_out, const d3dp_weights* grads, int32_t B, void* workspace,
                        size_t workspace_bytes, void* stream) {
  if (!c || !x2d || !x3d || !t || !grad_out || !grads || !grads->ste || !grads->tte || !workspace || B < 1)
    return fail(D3DP_EINVAL, "d3dp_train_backward: bad argument");
  if (!c->train()) return fail(D3DP_ESTATE, "d3dp_train_backward needs a D3DP_MODE_TRAIN context");
  const d3dp_cfg& g = c->cfg;
  const TrainLayout L = train_layout(g, B);
  if (workspace_bytes < L.total_floats * 4) return fail(D3DP_ESTATE, "train workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const int T = (int)L.T, Tp = (int)L.Tpad, C = g.channels, F = g.frames, J = g.joints, Hd = g.hidden;
  auto G = [](const float* p) { return const_cast<float*>(p); };
  // (one launch for the few gradient buffers that are still ACCUMULATED into -- the time MLP's; everything else is written:
  //  weight matrices by their wgrad product, every other gradient by the fixed-order reduction at the end of this function)
  D3dpZeroTable ztab{};
  auto zero = [&](const float* p, size_t n) -> hipError_t {
    if (ztab.count < D3DP_ZERO_MAX && n < ((size_t)1 << 32)) {
      ztab.p[ztab.count] = G(p); ztab.n[ztab.count] = (unsigned)n; ++ztab.count;
      return hipSuccess;
    }
    return hipMemsetAsync(G(p), 0, n * 4, st);
  };
  const size_t CC = (size_t)C * C;
  HIP_TRY(zero(grads->time1_w, 2 * CC)); HIP_TRY(zero(grads->time1_b, 2 * C));
  HIP_TRY(zero(grads->time3_w, 2 * CC)); HIP_TRY(zero(grads->time3_b, C));
  HIP_TRY(zero(ws + L.zero_bias, 4 * (size_t)C));
  d3dp_ctx* const pc = c->prof ? c : nullptr;
#define TP(cls, call) { Scope ps_(pc, cls, st); LAUNCH_TRY(call); }
  if (ztab.count) TP(T_OTHER, d3dp_train_zero_many(ztab, st));
  const float* zb = ws + L.zero_bias;
  float *xn = ws + L.xn, *hid = ws + L.hid, *dA = ws + L.dA, *dB = ws + L.dB, *dC = ws + L.dC, *dqkv = ws + L.dqkv,
        *dh = ws + L.dh, *At = ws + L.At, *Xt = ws + L.Xt, *Wt = ws + L.Wt;

  X2Train x2{st, ws, L, c->n_cu};
  x2.pc = pc;
  x2.use_set(0);
  x2.tail_blocks = c->train_tail_blocks;
  x2.merged_setup(T, c->train_wgrad_merged);
  const bool use_x2 = c->train_x2 && C % 32 == 0 && Hd % 32 == 0;
  if (use_x2) {
    TP(T_OTHER, x2.begin(false));
    x2.batched = 8 * g.depth <= D3DP_WPREP_MAX;          // (the forward pass of this step left the weight operands in place)
  }
  // second stream for the weight-gradient products (see d3dp_ctx::aux)
  // (under the per-kernel profile everything runs on the caller's stream: a class's time must not contain a wait for CUs that a
  //  product on the second stream holds -- same arithmetic, test_training_step_stream_switches_change_no_bit)
  const bool overlap = use_x2 && c->train_overlap && !c->prof;
  if (overlap && !c->aux) {
    HIP_TRY(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    for (hipEvent_t& e : c->ev_done) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  // Up to two weight-gradient products are in flight on c->aux, each with its own operand / partial-tile set (X2Train::use_set):
  // product k takes set k & 1 and the caller's stream waits for product k - 2 before the operand pass overwrites that set -- not
  // for product k - 1, which goes on beside the dgrad product and the row kernels of this dY.  D3DP_TRAIN_OVERLAP=1: one set,
  // every operand pass waits for the product before it (the first form of this overlap).
  bool pending[2] = {false, false};
  int n_wgrad = 0, cur_set = 0;
  const int n_sets = c->train_overlap_sets;
  auto wait_set = [&](int s) -> int {                    // the caller's stream waits for the product that last used set s
    if (!pending[s]) return 0;
    pending[s] = false;
    return hipStreamWaitEvent(st, c->ev_done[s], 0) == hipSuccess ? 0 : -3;
  };
  auto join = [&]() -> int { int r = wait_set(0); return r ? r : wait_set(1); };
  const bool attn_x2 = use_x2 && c->train_attn_x2 > 0 && C / g.heads == 64 && F <= 1024;  // (as the forward pass of this step)
  const bool attn_x2_s = attn_x2 && c->train_attn_x2 > 1;
  Reducer red{ws + L.red, L.red_floats, 0, st};
  red.items.reserve(24 * (size_t)g.depth + 32);
  const int lnrows = d3dp_train_ln_bwd_blocks(T);        // partial rows one LayerNorm-backward call leaves
  // [dgamma | dbeta] partial rows of a LayerNorm with `calls` backward calls per step (the shared norms: one per depth), and
  // their two destinations; call i writes rows [i lnrows, (i + 1) lnrows)
  auto ln_part = [&](const float* dgamma, const float* dbeta, int calls) -> float* {
    float* p = red.take((size_t)calls * lnrows * 2 * C);
    if (p) {
      red.add(p, G(dgamma), C, (size_t)calls * lnrows, 2 * (size_t)C);
      red.add(p + C, G(dbeta), C, (size_t)calls * lnrows, 2 * (size_t)C);
    }
    return p;
  };
  float* part_sn = ln_part(grads->spatial_norm_w, grads->spatial_norm_b, g.depth);
  float* part_tn = ln_part(grads->temporal_norm_w, grads->temporal_norm_b, g.depth);
  float* part_hn = ln_part(grads->head_norm_w, grads->head_norm_b, 1);
  if (red.rc) return fail(D3DP_ESTATE, "d3dp_train_backward: partial-sum region too small");
  int sdy = -1;                                          // absmax slot of the dY the next wgrad / dgrad pair shares
  // wgrad: dW[N, K] = dY[T, N]^T X[T, K]  (fp32 path: both operands transposed to [*, Tpad], zero padded)
  // (dbias: the Linear's bias gradient = column sums of dY, left as partial rows and summed at the end)
  // (pre_slot >= 0: the kernel that produced dY left its absmax there)
  // (mk / mk_axis: DropPath scales the operand pass still has to apply to dY's rows -- then dY is the UNSCALED gradient)
  // (gelu_pre: dY is d hidden, the Linear's dY is dY x gelu'(gelu_pre) and pre_slot holds the absmax of d hidden: split-fp16 path)
  auto wgrad = [&](int l, const float* dY, int N, const float* X, int K, float* dW, float* dbias, int pre_slot = -1,
                   const float* mk = nullptr, int mk_axis = 0, const float* gelu_pre = nullptr) -> int {
    int r;
    if (use_x2) {
      sdy = pre_slot >= 0 ? pre_slot : x2.slot_for(dY, (size_t)T * N);
      if (sdy < 0) return -1;
      float* bp = red.take((size_t)X2Train::kBiasRows * N);
      int brows = 0;
      if (!bp) return -1;
      if (x2.merged) {
        // one launch per block: l = 4 block + {3, 2, 1, 0} arrive in this order; the set changes with the block
        if ((l & 3) == 3) {
          cur_set = overlap ? (n_wgrad++ % n_sets) : 0;
          if ((r = wait_set(cur_set))) return r;         // the block that last used this set is done with its operands and partial tiles
          x2.use_set(cur_set);
        }
        if ((r = x2.prep_dy(dY, sdy, bp, &brows, T, N, K, mk, mk_axis, F, J, gelu_pre))) return r;
        red.add(bp, dbias, N, brows, N);
        if ((r = x2.wgrad(l, dY, sdy, X, dW, T, N, K))) return r;      // (recorded)
        if ((l & 3) != 0) return 0;
        if (overlap) {
          if (hipEventRecord(c->ev_fork, st) != hipSuccess || hipStreamWaitEvent(c->aux, c->ev_fork, 0) != hipSuccess) return -3;
          x2.st_w = c->aux;
        }
        if ((r = x2.flush_wgrads())) return r;
        if (overlap) {
          if (hipEventRecord(c->ev_done[cur_set], c->aux) != hipSuccess) return -3;
          pending[cur_set] = true;
        }
        return 0;
      }
      const int set = overlap ? (n_wgrad++ % n_sets) : 0;
      if ((r = wait_set(set))) return r;                 // the product that last used this set is done with its operand and partial tiles
      x2.use_set(set);
      if ((r = x2.prep_dy(dY, sdy, bp, &brows, T, N, K, mk, mk_axis, F, J, gelu_pre))) return r;
      red.add(bp, dbias, N, brows, N);
      if (!overlap) return x2.wgrad(l, dY, sdy, X, dW, T, N, K);
      if (hipEventRecord(c->ev_fork, st) != hipSuccess || hipStreamWaitEvent(c->aux, c->ev_fork, 0) != hipSuccess) return -3;
      x2.st_w = c->aux;
      if ((r = x2.wgrad(l, dY, sdy, X, dW, T, N, K))) return r;
      if (hipEventRecord(c->ev_done[set], c->aux) != hipSuccess) return -3;
      pending[set] = true;
      return 0;
    }
    float* bp = red.take((size_t)D3DP_DYPREP_ROWS * N);
    int rows = 0;
    if (!bp || mk || gelu_pre) return -1;
    {
      Scope ps_(pc, T_OPERAND, st);
      if ((r = d3dp_train_colsum(dY, bp, &rows, D3DP_DYPREP_ROWS, T, N, st))) return r;
      red.add(bp, dbias, N, rows, N);
      if ((r = d3dp_train_transpose_pad(dY, At, T, N, Tp, st))) return r;
      if ((r = d3dp_train_transpose_pad(X, Xt, T, K, Tp, st))) return r;
    }
    Scope ps_(pc, T_WGRAD, st);
    // contraction over tokens: split-K, the chunks' partial products added in a fixed order (no float atomics: this path's gradients
    // are bit-reproducible too)
    return d3dp_launch_linear_f32_splitk(At, Xt, dW, N, K, Tp, st, ws + L.part, (size_t)(1024 + 64) * 256 * 128);
  };
  // dgrad: dX[T, K] = dY[T, N] W[N, K]   (W transposed to [K, N])
  auto dgrad = [&](int l, const float* dY, int N, const float* W, int K, float* dX, unsigned* out_amax = nullptr) -> int {
    if (use_x2) return sdy < 0 ? -1 : x2.dgrad(l, dY, sdy, W, dX, T, N, K, out_amax);  // (always right behind the wgrad of the same dY)
    int r;
    { Scope ps_(pc, T_OPERAND, st); if ((r = d3dp_train_transpose_pad(W, Wt, N, K, N, st))) return r; }
    Scope ps_(pc, T_LINEAR, st);
    return lin32(dY, Wt, zb, dX, T, K, N, st);
  };
  // a slot for the absmax the producer of the next dY leaves (split-fp16 path only)
  auto fresh = [&](int& slot) -> unsigned* {
    slot = use_x2 ? x2.take() : -1;
    return slot >= 0 ? x2.amax() + slot : nullptr;
  };
#define D3DP_FRESH(ps, pa)                                                                              \
  int ps = -1;                                                                                          \
  unsigned* pa = fresh(ps);                                                                             \
  if (use_x2 && ps < 0) return fail(D3DP_ESTATE, "d3dp_train_backward: out of operand slots");

  // ---- head: Linear(C, 3) backward, then its LayerNorm and the last block's shared norm in one pass --------------
  const int nblk = 2 * g.depth;
  {
    float* hp = red.take((size_t)512 * (3 * C + 4));
    int rows = 0;
    if (!hp) return fail(D3DP_ESTATE, "d3dp_train_backward: partial-sum region too small");
    TP(T_OTHER, d3dp_train_head_bwd(grad_out, ws + L.z, c->hw, dC, hp, &rows, T, C, st));   // (z: left by the forward pass)
    red.add(hp, G(grads->head_w), 3 * (size_t)C, rows, 3 * (size_t)C + 4);
    red.add(hp + 3 * C, G(grads->head_b), 3, rows, 3 * (size_t)C + 4);
  }
  // dB = d x_out of the last block; its DropPath-scaled form (the fc2 dY) in dC with its absmax in slot ps
  const float* dy = nullptr;                             // the dY of the branch about to be differentiated
  const float* dy_mask = nullptr;                        // ... and the DropPath scales its operand pass still has to apply
  int ps_next = -1;
  // (the DropPath-scaled gradient is not stored where the operand pass can apply the scales itself: only its absmax is)
  const bool mip2 = use_x2 && x2.mask_in_prep_applies(C, Hd), mip1 = use_x2 && x2.mask_in_prep_applies(C, C);
  {
    const int lb = nblk - 1, lk = lb & 1;
    const float* S = ws + L.saved0 + (size_t)lb * L.saved_stride;
    const float* mk = mask_ptr(masks, g, B, lb, 1);
    D3DP_FRESH(ps, pa)
    TP(T_LN_BWD, d3dp_train_ln_bwd(dC, ws + L.x_final, c->hnw, g.eps_head, nullptr, nullptr, S + L.o_xout, lk ? c->tnw : c->snw,
                                 g.eps_block, dB, mk, lk, F, J, (mk && !mip2) ? dC : nullptr, pa, part_hn,
                                 (lk ? part_tn : part_sn) + (size_t)(lb >> 1) * lnrows * 2 * C, T, C, st));
    dy = (mk && !mip2) ? dC : dB;
    dy_mask = mip2 ? mk : nullptr;
    ps_next = ps;
  }
  for (int blk = nblk - 1; blk >= 0; --blk) {
    const int kind = blk & 1, d = blk >> 1;
    const BlockDev& w = kind ? c->tte[d] : c->ste[d];
    const d3dp_block_weights& gw = (kind ? grads->tte : grads->ste)[d];
    float* S = ws + L.saved0 + (size_t)blk * L.saved_stride;
    // here: dB = d x_out(blk), dy = its DropPath-scaled form, absmax in slot ps_next
    // ---- MLP branch ----
    if (!use_x2) TP(T_OPERAND, d3dp_train_gelu_fwd(S + L.o_hpre, hid, (size_t)T * Hd, nullptr, st));   // (x2: wgrad reads the forward pass' operand)
    LAUNCH_TRY(wgrad(4 * blk + 3, dy, C, hid, Hd, G(gw.fc2_w), G(gw.fc2_b), ps_next, dy_mask, kind));
    {
      // d hidden = dy W_fc2; d h_pre = d hidden x gelu'(h_pre).  Split-fp16 path: the product is formed by the operand pass of
      // the fc1 gradients (the only reader of d h_pre), from the absmax of d hidden the fc2 dgrad's epilogue leaves
      const bool gip = use_x2 && c->train_gelu_in_prep && x2.gelu_in_prep_applies(Hd, C);
      D3DP_FRESH(ps, pa)
      LAUNCH_TRY(dgrad(4 * blk + 3, dy, C, (const float*)w.fc2_w, Hd, dh, gip ? pa : nullptr));                         // d hidden
      if (!gip) TP(T_OPERAND, d3dp_train_gelu_bwd(dh, S + L.o_hpre, dh, (size_t)T * Hd, pa, st));             // d h_pre
      if (!use_x2) TP(T_LN_FWD, d3dp_train_ln_pos(S + L.o_xmid, w.n2w, w.n2b, g.eps_block, nullptr, F, J, xn, T, C, st));   // xn2
      LAUNCH_TRY(wgrad(4 * blk + 2, dh, Hd, xn, C, G(gw.fc1_w), G(gw.fc1_b), ps, nullptr, 0, gip ? S + L.o_hpre : nullptr));
      LAUNCH_TRY(dgrad(4 * blk + 2, dh, Hd, (const float*)w.fc1_w, C, dC));                                             // d xn2
    }
    // norm2 backward + the residual: dA = d x_mid; its DropPath-scaled form (the proj dY) over dC
    {
      const float* mk = mask_ptr(masks, g, B, blk, 0);
      float* pn = ln_part(gw.norm2_w, gw.norm2_b, 1);
      D3DP_FRESH(ps, pa)
      if (!pn) return fail(D3DP_ESTATE, "d3dp_train_backward: partial-sum region too small");
      TP(T_LN_BWD, d3dp_train_ln_bwd(dC, S + L.o_xmid, w.n2w, g.eps_block, dB, nullptr, nullptr, nullptr, 0.f, dA, mk, kind, F, J,
                                   (mk && !mip1) ? dC : nullptr, pa, pn, nullptr, T, C, st));
      dy = (mk && !mip1) ? dC : dA;
      // ---- attention branch ----
      LAUNCH_TRY(wgrad(4 * blk + 1, dy, C, S + L.o_att, C, G(gw.proj_w), G(gw.proj_b), ps, mip1 ? mk : nullptr, kind));
    }
    int ps_dqkv = -1;
    if (kind == 1 ? attn_x2 : attn_x2_s) {
      // split-fp16 operands: the proj dgrad leaves d att's absmax, the attention backward that of dqkv
      D3DP_FRESH(pdo, pado)
      D3DP_FRESH(pq, paq)
      LAUNCH_TRY(dgrad(4 * blk + 1, dy, C, (const float*)w.proj_w, C, dB, pado));                                       // d att
      // (the per-kernel profile times the two passes apart; otherwise one call launches both)
      for (int part = pc ? 1 : 0; part <= (pc ? 2 : 0); ++part)
        TP(part == 2 ? (kind ? T_ATTN_BKV_T : T_ATTN_BKV_S) : (kind ? T_ATTN_BQ_T : T_ATTN_BQ_S),
           d3dp_train_attn_x2_bwd(S + L.o_qkv, S + L.o_att, dB, dqkv, ws + L.stats_x2 + (size_t)blk * L.stats_x2_stride,
                                  kind ? B * J : B * F, kind ? temporal_map(F, J) : spatial_map(F, J), C, g.heads,
                                  x2.amax() + X2Train::kQkvSlot0 + blk, pado, paq, st, part));
      ps_dqkv = pq;
    } else {
      LAUNCH_TRY(dgrad(4 * blk + 1, dy, C, (const float*)w.proj_w, C, dB));                                             // d att
      // (fp32 kernels: both passes behind one launcher -- timed together under the pass-Q class)
      if (kind == 0) TP(T_ATTN_BQ_S, d3dp_train_attn_bwd(S + L.o_qkv, S + L.o_att, dB, dqkv, ws + L.stats, B * F, spatial_map(F, J), C, g.heads, st))
      else TP(T_ATTN_BQ_T, d3dp_train_attn_bwd(S + L.o_qkv, S + L.o_att, dB, dqkv, ws + L.stats, B * J, temporal_map(F, J), C, g.heads, st));
    }
    if (!use_x2) TP(T_LN_FWD, d3dp_train_ln_pos(S + L.o_xin, w.n1w, w.n1b, g.eps_block, nullptr, F, J, xn, T, C, st));    // xn1
    LAUNCH_TRY(wgrad(4 * blk, dqkv, 3 * C, xn, C, G(gw.qkv_w), G(gw.qkv_b), ps_dqkv));
    LAUNCH_TRY(dgrad(4 * blk, dqkv, 3 * C, (const float*)w.qkv_w, C, dC));                                          // d xn1
    float* pn1 = ln_part(gw.norm1_w, gw.norm1_b, 1);
    if (!pn1) return fail(D3DP_ESTATE, "d3dp_train_backward: partial-sum region too small");
    if (blk > 0) {
      // norm1 backward + the residual (= d x_in of this block = d of the previous block's shared-norm output), then that
      // shared norm's backward, in one pass: dB = d x_out(blk - 1) and its DropPath-scaled form over dC
      const int pb = blk - 1, pk = pb & 1;
      const float* Sp = ws + L.saved0 + (size_t)pb * L.saved_stride;
      const float* mk = mask_ptr(masks, g, B, pb, 1);
      float* g_out = pb == 0 ? ws + L.y : nullptr;       // d of Temporal_pos_embed's sum (added behind block 0's shared norm); y: a forward
                                                         // temporary -- NOT z, which a second backward over the same forward reads again
      D3DP_FRESH(ps, pa)
      TP(T_LN_BWD, d3dp_train_ln_bwd(dC, S + L.o_xin, w.n1w, g.eps_block, dA, g_out, Sp + L.o_xout, pk ? c->tnw : c->snw, g.eps_block,
                                   dB, mk, pk, F, J, (mk && !mip2) ? dC : nullptr, pa, pn1,
                                   (pk ? part_tn : part_sn) + (size_t)(pb >> 1) * lnrows * 2 * C, T, C, st));
      if (g_out) TP(T_OTHER, d3dp_train_groupsum(g_out, G(grads->temporal_pos), T, C, 1, F, J, 1, st));
      dy = (mk && !mip2) ? dC : dB;
      dy_mask = mip2 ? mk : nullptr;
      ps_next = ps;
    } else {
      TP(T_LN_BWD, d3dp_train_ln_bwd(dC, S + L.o_xin, w.n1w, g.eps_block, dA, nullptr, nullptr, nullptr, 0.f, dB, nullptr, 0, F, J,
                                   nullptr, nullptr, pn1, nullptr, T, C, st));
    }
  }
#undef D3DP_FRESH
  LAUNCH_TRY(join());                                    // every weight gradient is on the caller's stream's timeline again
  // ---- embedding, position and time embeddings (dB = d of the embedded tokens) ------------------------------------
  {
    Scope ps_(pc, T_OTHER, st);
    float* ep = red.take((size_t)D3DP_EMBED_BWD_ROWS * 5 * C);
    float* bp = red.take((size_t)512 * C);
    float* sp = red.take((size_t)kGroupSlices * J * C);
    float* tp = red.take((size_t)kGroupSlices * B * C);
    int rows = 0;
    if (!ep || !bp || !sp || !tp) return fail(D3DP_ESTATE, "d3dp_train_backward: partial-sum region too small");
    LAUNCH_TRY(d3dp_train_embed_bwd(dB, x2d, x3d, ep, T, C, st));
    red.add(ep, G(grads->embed_w), 5 * (size_t)C, D3DP_EMBED_BWD_ROWS, 5 * (size_t)C);
    LAUNCH_TRY(d3dp_train_colsum(dB, bp, &rows, 512, T, C, st));
    red.add(bp, G(grads->embed_b), C, rows, C);
    LAUNCH_TRY(d3dp_train_groupsum(dB, sp, T, C, 0, F, J, kGroupSlices, st));
    red.add(sp, G(grads->spatial_pos), (size_t)J * C, kGroupSlices, (size_t)J * C);
    LAUNCH_TRY(d3dp_train_groupsum(dB, tp, T, C, 2, F, J, kGroupSlices, st));
    red.add(tp, ws + L.dtemb, (size_t)B * C, kGroupSlices, (size_t)B * C);
    red.flush();                                          // (dtemb feeds the time MLP's backward below)
  }
  if (red.rc) return fail(D3DP_EHIP, "d3dp_train_backward: the gradient reduction failed");
  TP(T_OTHER, d3dp_train_time_mlp_bwd(t, c->freq, c->t1w, c->t1b, c->t3w, ws + L.dtemb, G(grads->time1_w),
                                      G(grads->time1_b), G(grads->time3_w), G(grads->time3_b), B, C, st));
#undef TP
  // (profile only) eight empty scopes on the still-busy stream: the time an event pair adds to every launch it brackets --
  // event-to-event durations of 15 - 80 us kernels are not kernel durations without it (bench.py subtracts the average)
  for (int i = 0; pc && i < 8; ++i) { Scope ps_(pc, P_EMPTY, st); }
  HIP_TRY(hipGetLastError());
  return D3DP_OK;
}

}  // extern "C"
