/* d3dp_hip.h -- C ABI of libd3dp_hip.so: the MI355X (gfx950) implementation of the D3DP hot path.
 *
 * The reference (paTRICK-swk/D3DP) is pure PyTorch: it has no FFI, its "plugin boundary" for this path is
 * the Python API D3DP.forward / ddim_sample_flip / q_sample and MixSTE2.forward.  This header is what a
 * Python (ctypes), C++ or any other FFI host binds INSTEAD of the ATen call sites listed per function.
 * d3dp_amd/_lib.py is the ctypes binding shipped here; INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *  - plain C types only; every pointer is a DEVICE pointer to a contiguous buffer unless marked "host";
 *  - `stream` is a hipStream_t passed as void* (0 = the null stream); every call is asynchronous on it, does
 *    not allocate and does not synchronise (exceptions: d3dp_create/d3dp_set_weights/d3dp_destroy and the
 *    d3dp_profile_* calls);
 *  - every function returns 0 on success, a negative D3DP_E* code otherwise; d3dp_last_error() returns a
 *    thread-local message for the last failure;
 *  - a context is bound to the HIP device current at d3dp_create and is not thread-safe (one ctx per rank).
 *
 * Tensor layouts (row-major, last index fastest) follow the reference:
 *   x2d  (B, F, J, 2) fp32      2D keypoints                       common/mixste.py:278
 *   x_t  (B, H, F, J, 3) fp32   noisy 3D hypotheses                common/mixste.py:278
 *   t    (B) int64              diffusion timestep per batch item  common/diffusionpose.py:230
 *   out  (B, H, F, J, 3) fp32   predicted x0                       common/mixste.py:296
 */
#ifndef D3DP_HIP_H
#define D3DP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D3DP_ABI_VERSION 4

/* The library is built with -fvisibility=hidden: the functions declared in this header -- and nothing else -- are its dynamic
 * symbols (tests/test_abi.py compares `nm -D` with this file). */
#if defined(__GNUC__) || defined(__clang__)
#define D3DP_API __attribute__((visibility("default")))
#else
#define D3DP_API
#endif

enum {
  D3DP_OK = 0,
  D3DP_EINVAL = -1,      /* bad argument / unsupported shape */
  D3DP_ENOTSUP = -2,     /* configuration not supported by the kernels (e.g. head dim) */
  D3DP_EHIP = -3,        /* a HIP runtime call failed */
  D3DP_ESTATE = -4,      /* call out of order (weights not set, workspace too small, ...) */
};

/* numerics mode of the denoiser */
enum {
  D3DP_MODE_EXACT = 0,   /* fp32-equivalent: Linears on split-fp16 operands (2 planes, 3 fp16-MFMA passes, fp32
                            accumulate), fp32 everything else: <= 1e-3 mm vs the reference.  Linear inputs must stay
                            below 65504 in magnitude.  (env D3DP_EXACT_IMPL=bf16x3 | f32: six-pass split-bf16 /
                            plain fp32-MFMA Linears, kept as cross-checks)                                        */
  D3DP_MODE_FAST = 1,    /* bf16 activations/weights into v_mfma_f32_16x16x32_bf16, fp32 accumulate/LN/softmax */
  D3DP_MODE_TRAIN = 2,   /* fp32 weights and activations; Linears (forward, dgrad, wgrad) on split-fp16 operands whose scales are
                            found on the device: required by d3dp_train_*; inference also works (fp32-MFMA Linears)          */
};

/* MixSTE2 hyper-parameters -- reference common/diffusionpose.py:123-124, common/mixste.py:142-163 */
typedef struct d3dp_cfg {
  int32_t frames;        /* F: args.number_of_frames                 */
  int32_t joints;        /* J: 17                                    */
  int32_t channels;      /* C: args.cs (embed_dim_ratio)             */
  int32_t depth;         /* args.dep: number of (spatial, temporal) block pairs */
  int32_t heads;         /* 8                                        */
  int32_t hidden;        /* mlp hidden = 2*C (mlp_ratio 2.)          */
  float eps_block;       /* 1e-6: LayerNorm eps of blocks and Spatial/Temporal_norm (mixste.py:163) */
  float eps_head;        /* 1e-5: nn.LayerNorm default in head (mixste.py:208)                     */
  int32_t mode;          /* D3DP_MODE_*                              */
  int32_t chunk_seqs;    /* most (clip,hypothesis) sequences per internal pass; 0 = library default (31 EXACT, 15 FAST).  EXACT
                          * mode sizes its passes (<= this) so that the persistent Linear kernels' last tile rounds are full
                          * (capi.hip plan()); results do not depend on the split, bit for bit.
                          * EXACT mode addresses a pass's Linear outputs with 32-bit byte offsets: chunk_seqs * F * J * 3 *
                          * channels * 4 must stay below 2^32 (169 sequences at F=243, J=17, C=512), else d3dp_denoise fails. */
} d3dp_cfg;

/* One transformer Block's parameters, fp32, reference state_dict names in comments (mixste.py:96-101) */
typedef struct d3dp_block_weights {
  const float* norm1_w; const float* norm1_b;     /* norm1.weight/bias   (C)      */
  const float* qkv_w;   const float* qkv_b;       /* attn.qkv.weight (3C,C)/bias  */
  const float* proj_w;  const float* proj_b;      /* attn.proj.weight (C,C)/bias  */
  const float* norm2_w; const float* norm2_b;     /* norm2.weight/bias            */
  const float* fc1_w;   const float* fc1_b;       /* mlp.fc1.weight (2C,C)/bias   */
  const float* fc2_w;   const float* fc2_b;       /* mlp.fc2.weight (C,2C)/bias   */
} d3dp_block_weights;

/* All MixSTE2 parameters (device fp32, caller-owned; the library keeps its own packed copies). */
typedef struct d3dp_weights {
  const float* spatial_pos;       /* Spatial_pos_embed (1,J,C)                 mixste.py:169 */
  const float* temporal_pos;      /* Temporal_pos_embed (1,F,C)                mixste.py:172 */
  const float* embed_w;           /* Spatial_patch_to_embedding.weight (C,5)   mixste.py:166 */
  const float* embed_b;
  const float* time_freq;         /* (C/2) fp32: exp(arange(C/2) * -ln(1e4)/(C/2-1)), mixste.py:135-136 (host-computed) */
  const float* time1_w;           /* time_mlp.1.weight (2C,C)                  mixste.py:181 */
  const float* time1_b;
  const float* time3_w;           /* time_mlp.3.weight (C,2C)                  mixste.py:183 */
  const float* time3_b;
  const float* spatial_norm_w;    /* Spatial_norm (shared by all depths)       mixste.py:203 */
  const float* spatial_norm_b;
  const float* temporal_norm_w;   /* Temporal_norm                             mixste.py:204 */
  const float* temporal_norm_b;
  const float* head_norm_w;       /* head.0 (LayerNorm)                        mixste.py:208 */
  const float* head_norm_b;
  const float* head_w;            /* head.1.weight (3,C)                       mixste.py:209 */
  const float* head_b;
  const d3dp_block_weights* ste;  /* host array [depth]: STEblocks.i           mixste.py:190 */
  const d3dp_block_weights* tte;  /* host array [depth]: TTEblocks.i           mixste.py:197 */
} d3dp_weights;

typedef struct d3dp_ctx d3dp_ctx;

D3DP_API int d3dp_abi_version(void);
D3DP_API const char* d3dp_last_error(void);

/* Lifetime.  Replaces: MixSTE2.__init__ (mixste.py:142-210) + .cuda() (main.py:243).
 * Shapes (D3DP_ENOTSUP outside them, with the reason in d3dp_last_error):
 *   1 <= frames <= 1024   up to 256 frames the MFMA attention kernels hold a whole sequence; longer clips (`-f 351`,
 *                         common/arguments.py:58) take chunked-key forms of the same kernels (EXACT, TRAIN) or the row kernel (FAST);
 *   1 <= joints <= 256    MixSTE2's num_joints (mixste.py:141; D3DP builds 17): above 32 the spatial axis runs on the
 *                         whole-sequence attention kernels of the temporal axis;
 *   channels in {64, 128, 256, 512} with head dim in {8, 16, 32, 64} and hidden % 64 == 0: every mode, on the matrix-core
 *                         kernels (split-fp16 / bf16 operands) -- `-cs 512`, the width of every published checkpoint
 *                         (README.md:33-39), and its smaller powers of two;
 *   any other width the reference's 8 heads divide (common/arguments.py:49, mixste.py:46-62) with channels <= 1024,
 *                         head dim % 4 == 0 and <= 128, hidden % 4 == 0: D3DP_MODE_EXACT only, on the fp32 implementation
 *                         (fp32-MFMA Linears, fp32 row attention, run-time-width row kernels; d3dp_exact_scales reports
 *                         implementation 2): the same 1e-3 mm tolerance at roughly a fifth of the throughput.  FAST and TRAIN
 *                         contexts exist for the instantiated widths only. */
D3DP_API int d3dp_create(const d3dp_cfg* cfg, d3dp_ctx** out);
D3DP_API int d3dp_destroy(d3dp_ctx* ctx);
/* Replaces: load_state_dict (main.py:257).  Converts/packs weights for cfg.mode (synchronises `stream`).
 * Non-finite weights (a diverged checkpoint) are not an error: like the reference, the library loads them and the outputs
 * are non-finite where the reference's are (d3dp_status reports it); an EXACT context then runs its split-bf16
 * implementation (d3dp_exact_scales: implementation 1), which has fp32's exponent range. */
D3DP_API int d3dp_set_weights(d3dp_ctx* ctx, const d3dp_weights* w, void* stream);
/* D3DP_MODE_TRAIN only: use the caller's fp32 device buffers in place (no packed copy, no launch, no synchronisation), so
 * an optimizer step needs no re-push.  The buffers must stay allocated while the context uses them. */
D3DP_API int d3dp_set_weights_borrowed(d3dp_ctx* ctx, const d3dp_weights* w);

/* EXACT mode's split-fp16 operands (x 2^s = hi + lo, two fp16) hold |x| 2^s < 65504: with the default s = 4 that is
 * |x| < D3DP_SPLIT_RANGE, beyond which hi would be inf and the Linear NaN where the fp32 reference stays finite.  The
 * library therefore PROVES the range of every data-dependent operand from the weights and scales accordingly:
 *  - d3dp_set_weights computes (on the device) an upper bound of the magnitude each split operand can take for ANY
 *    input -- a LayerNorm output is at most sqrt(C-1) |gamma_k| + |beta_k| per channel; a Linear row over it at most
 *    sum_k |w_k| (sqrt(C-1) |gamma_k| + |beta_k|) + |b| (q, k, v and the fc1 pre-activation; |GELU(x)| <= |x|); an attention
 *    output is a convex combination of v -- and gives every block two operand scales: s_kv for q / k / v / the attention
 *    output, s_hidden for the MLP hidden.  Bound below D3DP_SPLIT_RANGE: 2^4.  Above: the largest power of two with
 *    bound x scale < 65504, so no operand can overflow, at an absolute representation error of at most
 *    max(2^-22 |x|, 2^-25 / scale) -- below 2^-40 of the bound, far under the fp32 rounding of the sums it enters.
 *    LayerNorm outputs always use 2^4; if a LayerNorm's own bound reaches the range (|gamma| ~ 180) the context moves to the
 *    six-pass split-bf16 implementation (fp32's exponent range, twice the MFMA work).  No environment variable is involved.
 *  - d3dp_exact_range_bound: the largest of those bounds (0 in the other modes, which have no such limit);
 *  - d3dp_exact_scales: the chosen scales, s_kv[2 depth] and s_hidden[2 depth] (STE blocks 0..depth-1, then TTE; either
 *    pointer may be null), and the implementation in use (0 split-fp16, 1 split-bf16, 2 fp32 MFMA, -1 not an EXACT context);
 *  - d3dp_status: *nonfinite = 1 if, since the last d3dp_status, a d3dp_denoise output held inf / nan (every call ends with a
 *    scan of its output) -- i.e. the INPUT held inf / nan or the fp32 arithmetic itself overflowed.  Synchronises the
 *    device (do not call it during stream capture); resets the flag.
 * (In a variants build -- see d3dp_op_linear_x2 -- ONE operand is outside the proof: with norm2 folded into fc1, D3DP_FOLD_LN=1,
 * the proj Linear hands fc1 the un-normalised residual stream; that epilogue checks every value it splits, exactly, at run
 * time, and reports through d3dp_status.) */
#define D3DP_SPLIT_RANGE 4094.0f
D3DP_API int d3dp_exact_range_bound(const d3dp_ctx* ctx, float* bound);
D3DP_API int d3dp_exact_scales(const d3dp_ctx* ctx, float* s_kv, float* s_hidden, int32_t* implementation);
D3DP_API int d3dp_status(d3dp_ctx* ctx, int32_t* nonfinite);

/* Scratch needed by d3dp_denoise for a (B, H) call. */
D3DP_API int d3dp_workspace_bytes(const d3dp_ctx* ctx, int32_t B, int32_t H, size_t* bytes);

/* One denoiser evaluation.  Replaces: MixSTE2.forward(x_2d, x_3d, t), eval branch (mixste.py:278-298) and,
 * with H = 1, the train branch's forward (mixste.py:215-225). */
D3DP_API int d3dp_denoise(d3dp_ctx* ctx, const float* x2d, const float* x_t, const int64_t* t, float* out, int32_t B,
                 int32_t H, void* workspace, size_t workspace_bytes, void* stream);

/* Flip-TTA pre-step.  Replaces diffusionpose.py:148-153.
 *   xt2[0:B]  = clamp(img, +-1.1 scale) / scale
 *   xt2[B:2B] = x negated, joints permuted (perm[j] = source joint of joint j; device int32[J])
 * img (B,H,F,J,3) -> xt2 (2B,H,F,J,3). */
D3DP_API int d3dp_ddim_pre(const float* img, float* xt2, const int32_t* perm, float scale, int32_t B, int32_t H, int32_t F,
                  int32_t J, void* stream);

/* Flip-TTA post-step + DDIM update.  Replaces diffusionpose.py:158-169 and :244-254.
 *   pred    = (pred2[b] + unflip(pred2[B+b])) / 2
 *   x_start = clamp(pred * scale, +-1.1 scale)                    -> x_start (row b at x_start + b*xs_bstride)
 *   pn      = float((sqrt_recip * img - x_start) / sqrt_recipm1)  (fp64, predict_noise_from_start :129-133)
 *   img_next= x_start*c_xstart + c_noise*pn + sigma*noise         (fp32, unless `last`)
 * `noise` may be NULL only when `last` != 0.  img_next may alias img. */
D3DP_API int d3dp_ddim_post(const float* pred2, const float* img, const float* noise, const int32_t* perm, float scale,
                   double sqrt_recip, double sqrt_recipm1, float c_xstart, float c_noise, float sigma, int32_t last,
                   float* x_start, size_t xs_bstride, float* img_next, int32_t B, int32_t H, int32_t F, int32_t J,
                   void* stream);

/* Train-time forward diffusion.  Replaces prepare_diffusion_concat/q_sample (diffusionpose.py:260-267, 290-306):
 *   out[b] = float(clamp(a[b]*(x0[b]*scale) + s[b]*noise[b], +-1.1 scale) / scale); a, s are device fp64 (B). */
D3DP_API int d3dp_q_sample(const float* x0, const float* noise, const double* sqrt_ac, const double* sqrt_1mac, float scale,
                  float* out, int32_t B, int32_t per_b, void* stream);

/* JPMA: joint-wise reprojection-based multi-hypothesis aggregation (the consumer of the sampler / all-gather output).
 * Replaces main.py:700 (root zeroing, if zero_root), :706-712 (trajectory add + project_to_2d, camera.py:30-60) and
 * the per-joint argmin + gather of loss.py:54-76 / main_3dhp.py:797-835 in ONE pass, without (B,K,H,F,J,*) temporaries.
 *   pred (B,K,H,F,J,3), traj (B,F,1,3), cam (9) = f(2) c(2) k(3) p(2), gt2d (B,F,J,2), gt3d (B,F,J,3) or NULL
 *   agg (B,K,F,J,3)  <- the hypothesis whose reprojection is closest to gt2d (first minimum, like torch.min)
 *   sel (B,K,F,J) int32 or NULL; err_sel / err_min (B,K,F,J) or NULL: |selected - gt3d| and min_h |pred_h - gt3d|
 *   (their means over (B,F,J) are the reference's J_Agg and J_Best errors per step). */
D3DP_API int d3dp_jpma(const float* pred, const float* traj, const float* cam, const float* gt2d, const float* gt3d, float* agg,
              int32_t* sel, float* err_sel, float* err_min, int32_t B, int32_t K, int32_t H, int32_t F, int32_t J,
              int32_t zero_root, void* stream);
/* d3dp_jpma straight on an all-gather result: `gathered` = (R, B, K, H_local, F, J, 3), rank-major, exactly what
 * ncclAllGather leaves when every rank contributes its (B, K, H_local, F, J, 3) stack -- hypothesis h = r H_local + hl.
 * Same outputs and the same selection, bit for bit, as d3dp_jpma on the (B, K, R H_local, F, J, 3) tensor that a
 * permute + copy of `gathered` would produce (158.6 MB per rank and step at configs[3] that are never moved).
 * Replaces: main.py:700-718 after the hypothesis exchange of SURVEY.md 8 E1. */
D3DP_API int d3dp_jpma_gathered(const float* gathered, const float* traj, const float* cam, const float* gt2d, const float* gt3d,
                       float* agg, int32_t* sel, float* err_sel, float* err_min, int32_t R, int32_t B, int32_t K,
                       int32_t H_local, int32_t F, int32_t J, int32_t zero_root, void* stream);
/* d3dp_jpma with the 3DHP evaluation's options and pose outputs (main_3dhp.py:777-835): root_joint = index of the joint
 * written as 0 before use (0 for Human3.6M, 14 for 3DHP, -1 none); linear_projection = camera.py:62-83
 * project_to_2d_linear (f * clamp(X/Z) + c) instead of the distortion model; jbest (B,K,F,J,3) = per joint the
 * hypothesis closest to gt3d (J-Best pose), mean (B,K,F,J,3) = average over hypotheses (P-Agg pose).  Any output may be
 * NULL. */
D3DP_API int d3dp_jpma_ex(const float* pred, const float* traj, const float* cam, const float* gt2d, const float* gt3d, float* agg,
                 int32_t* sel, float* err_sel, float* err_min, float* jbest, float* mean, int32_t B, int32_t K, int32_t H,
                 int32_t F, int32_t J, int32_t root_joint, int32_t linear_projection, void* stream);

/* ---- training step (reference main.py:387-401 around diffusionpose.py:279-287 / mixste.py:215-225) ----------------
 * Context must be D3DP_MODE_TRAIN.  x3d (B,F,J,3) is the diffused pose from d3dp_q_sample, t (B) int64.
 * masks: NULL, or DropPath scales (timm semantics, values 0 or 1/keep) laid out [2*depth blocks (STE0,TTE0,STE1,..)]
 * [2 branches (attention, MLP)][B*max(F,J)] floats; entry s of a spatial block is sample b*F+f, of a temporal block b*J+n.
 * d3dp_train_forward keeps every activation the backward needs in `workspace`; d3dp_train_backward must follow with the
 * same inputs/masks/workspace.  grads: device fp32 buffers shaped like the weights (zeroed, then filled, here).
 * d3dp_train_backward launches a block's weight-gradient product on a second, library-owned stream, forked from and joined
 * back to `stream` with events before it returns (the host is never synchronised; env D3DP_TRAIN_OVERLAP=0: one stream).
 * No gradient is accumulated with float atomics: the same inputs give the same bits.
 * Clip length: any F <= 1024 like inference (reference common/arguments.py:58); beyond 256 frames the step needs head dim 64 and
 * its split-fp16 attention kernels (keys / queries through LDS in chunks) -- the fp32 cross-check implementations
 * (D3DP_TRAIN_IMPL=f32, D3DP_TRAIN_ATTN=f32|x2t) hold whole sequences and answer D3DP_ENOTSUP there. */
D3DP_API int d3dp_train_workspace_bytes(const d3dp_ctx* ctx, int32_t B, size_t* bytes);
D3DP_API int d3dp_train_forward(d3dp_ctx* ctx, const float* x2d, const float* x3d, const int64_t* t, const float* masks, float* out,
                       int32_t B, void* workspace, size_t workspace_bytes, void* stream);
D3DP_API int d3dp_train_backward(d3dp_ctx* ctx, const float* x2d, const float* x3d, const int64_t* t, const float* masks,
                        const float* grad_out, const d3dp_weights* grads, int32_t B, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ---- caller-side rows of SURVEY.md section 8(f) (d3dp_amd/csrc/caller.hip) ------------------------------------------
 * N2 clip chunking: src (n,J,D) -> dst (n_clips,F,J,D), n_clips = d3dp_clip_count(n,F): consecutive clips, a trailing
 * partial clip = the LAST F frames, n < F replicate-padded on the right (reference main.py:267-299,
 * in_the_wild/utils.py:199-240).  dst_flip (optional): the flipped input main.py:646-648 builds (x negated, joints
 * permuted by perm[J], perm[j] = source joint). */
D3DP_API int d3dp_clip_count(int32_t n, int32_t F);
D3DP_API int d3dp_clip_gather(const float* src, float* dst, float* dst_flip, const int32_t* perm, int32_t n, int32_t F, int32_t J,
                     int32_t D, void* stream);
/* de-chunking: pred (n_clips,K,H,F,J,D) -> out (K,H,n,J,D)  (in_the_wild/videopose_diffusion.py:150-164, including
 * its n < F behaviour: the last n frames of the padded clip). */
D3DP_API int d3dp_clip_scatter(const float* pred, float* out, int32_t n, int32_t K, int32_t H, int32_t F, int32_t J, int32_t D,
                      int32_t last_wins, void* stream);   /* last_wins: main_3dhp.py:327-330 (final clip owns the last F frames) */
/* E1 reduced exchange: d3dp_jpma_winners writes this rank's per-joint winner win (B,K,F,J,5) =
 * (2D error, x, y, z, bits of int32 global hypothesis index h_offset + h); after an all-gather over R ranks
 * (rank-major) d3dp_jpma_combine picks the smallest 2D error per joint, lowest rank on ties (= lowest global h, the
 * element torch.min returns in loss.py:67).  n = B*K*F*J. */
D3DP_API int d3dp_jpma_winners(const float* pred, const float* traj, const float* cam, const float* gt2d, float* win,
                      int32_t h_offset, int32_t B, int32_t K, int32_t H, int32_t F, int32_t J, int32_t zero_root,
                      void* stream);
D3DP_API int d3dp_jpma_combine(const float* win, int32_t R, size_t n, float* agg, int32_t* sel, void* stream);
/* N3 training batch assembly from device-resident pools (common/generators.py:12-171 ChunkedGenerator_Seq):
 * pool2d (N,J,2), pool3d (N,J,3) or NULL; table (nb,4) int32 = (first pool frame of the sequence, sequence length,
 * chunk start frame (may be negative / run past the end: edge frames repeat), flip); perm2d/perm3d (J) = left/right
 * swap for flipped items (x negated); zero_root: joint 0 of out3d written as 0 (main.py:365). */
D3DP_API int d3dp_batch_gather(const float* pool2d, const float* pool3d, const int32_t* table, const int32_t* perm2d,
                      const int32_t* perm3d, float* out2d, float* out3d, int32_t nb, int32_t F, int32_t J,
                      int32_t zero_root, void* stream);
/* AdamW (main.py:311: torch.optim.AdamW(lr, weight_decay=0.1)) over all parameter tensors in one launch.
 * chunks: device array of d3dp_adam_chunk (one block each); step = 1-based step count of this update. */
typedef struct d3dp_adam_chunk { float* p; const float* g; float* m; float* v; int32_t n; int32_t pad; } d3dp_adam_chunk;
D3DP_API int d3dp_adamw_step(const void* chunks, int32_t n_chunks, double lr, double beta1, double beta2, double eps,
                    double weight_decay, int64_t step, void* stream);
/* N4 Procrustes-aligned per-joint errors (common/loss.py:190-395 p_mpjpe*): pred (B,KH,F,J,3), target (B,F,J,3) ->
 * err (B,KH,F,J) and (optional) the aligned poses. */
D3DP_API int d3dp_procrustes(const float* pred, const float* target, float* err, float* aligned, int32_t B, int32_t KH, int32_t F,
                    int32_t J, void* stream);

/* ---- single operators (unit parity tests; same kernels the denoiser launches) ---------------------------- */
/* out[M,N] = epi(A[M,K] W[N,K]^T + bias).  epi & 3: 0 bias, 1 bias+GELU(erf), 2 out(fp32) += result.
 * mode EXACT: everything fp32 (fp32 MFMA; the EXACT denoiser itself runs mode 3 below).  mode FAST: A, W bf16 (uint16
 * storage), fp32 accumulate, epi 0 or 1 only, out bf16 unless (epi & 16) (fp32): the persistent streaming kernel the
 * denoiser uses. */
D3DP_API int d3dp_op_linear(int32_t mode, int32_t epi, const void* A, const void* W, const float* bias, void* out, int32_t M,
                   int32_t N, int32_t K, void* stream);
/* Multi-head attention over qkv[T,3C] -> out[T,C]; axis 0 = spatial (sequences of J joints), 1 = temporal
 * (sequences of F frames); tokens ordered (seq_batch, f, n).  impl 0 = fp32-VALU row kernel (any activation type),
 * 1 = matrix-core kernel (head dim 64): bf16 MFMA for bf16 activations (both axes), fp32 MFMA for fp32 activations
 * (temporal axis), 2 = the EXACT-mode kernels: split-fp16 operands on the fp16 matrix cores (both axes; head dim 64).  Inside
 * the denoiser those read the packed rows of its qkv Linear (d3dp_op_linear_x2, epi 4); this entry point takes plain fp32
 * rows and repacks them into a stream-ordered temporary (hipMallocAsync) first. */
D3DP_API int d3dp_op_attention(int32_t act_bf16, int32_t impl, int32_t axis, const void* qkv, void* out, int32_t n_bh,
                      int32_t F, int32_t J, int32_t C, int32_t heads, void* stream);
D3DP_API int d3dp_op_layernorm(int32_t out_bf16, const float* x, const float* w, const float* b, float eps, void* out,
                      int32_t T, int32_t C, void* stream);
/* mode 2 of d3dp_op_linear: split-bf16.  A and W are three bf16 planes each (x = x0 + x1 + x2, made by
 * d3dp_op_split3: dst[0..n) | dst[n..2n) | dst[2n..3n)); epi 0 -> fp32 out, epi 1 -> GELU then three bf16 planes out. */
D3DP_API int d3dp_op_split3(const float* src, void* dst, size_t n, void* stream);
/* The EXACT-mode Linear on split-fp16 operands.  A2 [M][K] and W2 [N][K] hold TWO fp16 per element, y = src * scale =
 * hi + lo with hi = fp16(y), lo = fp16(y - hi), in the line-interleaved layout "h2i": a matrix row is K/32 blocks of 64 fp16
 * (128 bytes), block kb = [hi of columns 32 kb .. 32 kb + 31 | lo of the same 32 columns] -- one 32-deep k-step of one row,
 * both values of every element, is one 128-byte line.  d3dp_op_split2 writes that layout for a flat array of n = rows x K
 * elements: dst[64 b .. 64 b + 31] = hi, dst[64 b + 32 .. 64 b + 63] = lo of src[32 b .. 32 b + 31]; n (and hence every row
 * length) must be a multiple of 32 (D3DP_EINVAL otherwise: a partial block would write past 2 n).
 * A2 must be at scale 16 (the library's default activation scale); W2 at any power-of-two `w_scale` (the denoiser picks it
 * per matrix so that max |w| w_scale lies in [2^13, 2^14)).  epi 0 -> fp32 out [M][N]; epi 1 -> GELU, then the result as an
 * h2i matrix [M][N] at scale 16 (the fc2 operand; N % 32 == 0); epi 4 (N = 3 C, C % 64 == 0; the qkv Linear of the EXACT
 * denoiser) -> packed rows of 12 C bytes: q fp32 [C] | k hi [C] | k lo [C] | v hi [C] | v lo [C] (plain fp16 planes, scale
 * 16), the operand format of the split-fp16 attention kernels; epi 2 -> out[M,N] fp32 is read and written (out += A W^T + b:
 * the residual-adding form of proj and fc2).
 * Constraints (D3DP_EINVAL otherwise): K % 64 == 0 (the k-loop runs two 32-deep k-steps per iteration), N % 4 == 0
 * (N % 32 == 0 for epi 1: whole h2i blocks), N <= 2048, M * N * 4 < 2^32.  Every operand value must stay below 65504 / scale
 * in magnitude (at scale 16: |x| < 4094) -- inside d3dp_denoise the library proves and arranges that (d3dp_exact_scales);
 * a caller of this entry point owns it.
 * The product library holds ONE form of this kernel.  Three more forms (epi | (D << 8), D in {1, 2, 4}: the row-class skewed
 * schedule; epi | 2048: ping-pong wave teams; epi | 4096: 256 x 256 tiles) and the epilogues that fold norm2 into proj / fc1
 * were built, measured no faster and now exist only in a `make -C d3dp_amd/csrc variants` library
 * (lib/variants/libd3dp_variants.so; tests marked `variants`): here those flags, and the environment switches D3DP_X2_SKEW,
 * D3DP_X2_PP, D3DP_X2_WIDE, D3DP_SEQ_PAD, D3DP_FOLD_LN read by d3dp_create, fail with D3DP_ENOTSUP -- they are never ignored.
 * Cross-check switches the product library does honour (read in d3dp_create): D3DP_EXACT_IMPL=bf16x3|f32, D3DP_NO_FOLD=1
 * (other implementations of EXACT mode's Linears / residual adds, same tolerance), D3DP_TRAIN_IMPL=f32. */
D3DP_API int d3dp_op_split2(const float* src, void* dst, size_t n, float scale, void* stream);
D3DP_API int d3dp_op_linear_x2(int32_t epi, const void* A2, const void* W2, const float* bias, float w_scale, void* out, int32_t M,
                      int32_t N, int32_t K, void* stream);
/* fp32 <-> bf16 conversion helper (round-to-nearest-even), n elements */
D3DP_API int d3dp_op_to_bf16(const float* src, void* dst, size_t n, void* stream);

/* ---- per-kernel timing (HIP events on the launch stream) -------------------------------------------------
 * While enabled, every kernel launched by d3dp_denoise, d3dp_train_forward and d3dp_train_backward is bracketed by
 * hipEventRecord on the caller's stream (while enabled the training step keeps its weight-gradient products on that stream too
 * instead of its second one: no class's time contains a wait for compute units another stream holds; same results bit for bit).
 * d3dp_profile_read synchronises the recorded events and returns, per kernel class, the launch count and the
 * summed duration in milliseconds.  Classes are listed by d3dp_profile_class_name: 0-11 the denoiser's (d3dp_denoise),
 * 12-23 the training step's (ABI v4): train_linear (forward and dgrad products), train_wgrad (a block's merged weight-gradient
 * product + the sum of its partial tiles), train_attn_{fwd,bwd_q,bwd_kv}_{spatial,temporal}, train_operand_pass (fp32 rows -> split
 * operands), train_ln_fwd, train_ln_bwd, train_other; 24 event_pair_overhead: event pairs with nothing between them, recorded at
 * the end of d3dp_train_backward -- the time a bracket adds to the launch it times. */
#define D3DP_PROFILE_CLASSES 25
D3DP_API int d3dp_profile_enable(d3dp_ctx* ctx, int32_t on);
D3DP_API int d3dp_profile_read(d3dp_ctx* ctx, int64_t* counts /*host[D3DP_PROFILE_CLASSES]*/,
                      double* total_ms /*host[D3DP_PROFILE_CLASSES]*/);
D3DP_API const char* d3dp_profile_class_name(int32_t cls);

/* ---- test hooks ------------------------------------------------------------------------------------------
 * Exported for tests/ only; no reference call site stands behind them and hosts must not bind them.
 * d3dp_debug_x2_variants: 1 if this library was built with the measured-negative experiment kernels of gemm_x2.hip
 *   (`make variants`: lib/variants/libd3dp_variants.so), 0 for the product library.
 * d3dp_debug_train_linear: the training step's split-fp16 Linear alone, out[M, N] = A[M, K] W[N, K]^T + bias on fp32 device
 *   operands -- absmax, operand split and gemm_f16x2_dyn_kernel exactly as d3dp_train_forward launches them.  tail: 0 = the rows
 *   behind the last whole 256-row tile as one more row of tiles, 1 = as 16 x 64 blocks inside the same launch; amax_out:
 *   optional pre-zeroed device word receiving the output's absmax (amax_pos = 1: its largest positive value).  Allocates its
 *   operand buffers and synchronises `stream`. */
D3DP_API int d3dp_debug_x2_variants(void);
D3DP_API int d3dp_debug_train_linear(const float* A, const float* W, const float* bias, float* out, int32_t M, int32_t N,
                                     int32_t K, int32_t tail, unsigned* amax_out, int32_t amax_pos, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* D3DP_HIP_H */
