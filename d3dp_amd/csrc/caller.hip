// Caller-side rows of SURVEY.md §8(f), as device kernels -- the steps either side of the sampler that the reference
// runs as numpy / torch-CPU loops:
//   N2  clip chunking / de-chunking              main.py:267-299, in_the_wild/utils.py:199-240,
//                                                in_the_wild/videopose_diffusion.py:150-164
//   E1  reduced hypothesis exchange for JPMA     local per-joint winners -> all-gather of 5 floats per joint -> combine
//                                                (loss.py:54-76 selection; ties resolved to the lowest global h like
//                                                torch.min)
//   N3  training batch assembly + AdamW          common/generators.py:12-171 (chunks, edge padding, flip augmentation),
//                                                main.py:311 (AdamW, weight decay 0.1), main.py:364-366 (root zeroing)
//   N4  Procrustes-aligned errors (P-MPJPE)      common/loss.py:190-395 (batched 3x3 SVD alignment)
// All of them are HBM-bound index/elementwise work on tensors that are tiny next to the denoiser's; they exist so the
// data never leaves the device between the dataset pool, the sampler and the metrics.
#include "../../include/d3dp_hip.h"
#include "common.h"
#include "kernels.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// N2: source frame of (clip c, frame f) for a sequence of n frames cut into clips of F frames: clips 0..n/F-1 are
// consecutive, a trailing partial clip is the LAST F frames (main.py:296), a sequence shorter than F is one clip
// replicate-padded on the right (main.py:284-295).
__device__ __forceinline__ int clip_src_frame(int c, int f, int n, int F) {
  if (n <= F) return min(f, n - 1);
  const int full = n / F;
  const int start = (c < full) ? c * F : n - F;
  return start + f;
}

// src [n, J, D] -> dst [n_clips, F, J, D] and (optional) the flipped copy main.py:646-648 builds first:
// x negated, joints permuted (perm[j] = source joint).
__global__ __launch_bounds__(256) void clip_gather_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                          float* __restrict__ dst_flip, const int* __restrict__ perm,
                                                          int n, int n_clips, int F, int J, int D) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)n_clips * F * J;
  if (i >= total) return;
  const int j = (int)(i % J);
  const size_t cf = i / J;
  const int f = (int)(cf % F), c = (int)(cf / F);
  const int sf = clip_src_frame(c, f, n, F);
  const float* s = src + ((size_t)sf * J + j) * D;
  for (int d = 0; d < D; ++d) dst[i * D + d] = s[d];
  if (dst_flip != nullptr) {
    const float* sp = src + ((size_t)sf * J + perm[j]) * D;
    dst_flip[i * D] = -sp[0];
    for (int d = 1; d < D; ++d) dst_flip[i * D + d] = sp[d];
  }
}

// pred [n_clips, K, H, F, J, D] -> out [K, H, n, J, D]  (videopose_diffusion.py:150-164, including its behaviour for
// n < F: the LAST n frames of the single padded clip are taken).
__global__ __launch_bounds__(256) void clip_scatter_kernel(const float* __restrict__ pred, float* __restrict__ out,
                                                           int n, int n_clips, int KH, int F, int JD, int last_wins) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)KH * n * JD;
  if (i >= total) return;
  const int e = (int)(i % JD);
  const size_t r = i / JD;
  const int fr = (int)(r % n), kh = (int)(r / n);
  int c, f;
  const int covered = (n_clips - 1) * F;         // frames owned by the leading full clips
  // last_wins: the final clip overwrites ALL of the last F frames (main_3dhp.py:327-330 pose_post_process) instead of
  // only the frames no earlier clip produced
  const int own = (last_wins && n >= F) ? n - F : covered;
  if (fr < own) { c = fr / F; f = fr % F; }
  else { c = n_clips - 1; f = F - (n - fr); }
  out[i] = pred[(((size_t)c * KH + kh) * F + f) * JD + e];
}

// ---------------------------------------------------------------------------------------------------------------------
// E1: combine per-rank JPMA winners.  win [R][n][5] = (err2d, x, y, z, bits(global h)) -> agg [n][3], sel [n].
__global__ __launch_bounds__(256) void jpma_combine_kernel(const float* __restrict__ win, int R, size_t n,
                                                           float* __restrict__ agg, int* __restrict__ sel) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float best = INFINITY, x = 0.f, y = 0.f, z = 0.f;
  int h = 0;
  for (int r = 0; r < R; ++r) {
    const float* w = win + ((size_t)r * n + i) * 5;
    if (w[0] < best) { best = w[0]; x = w[1]; y = w[2]; z = w[3]; h = __float_as_int(w[4]); }
  }
  agg[i * 3] = x; agg[i * 3 + 1] = y; agg[i * 3 + 2] = z;
  if (sel != nullptr) sel[i] = h;
}

// ---------------------------------------------------------------------------------------------------------------------
// N3: one training batch from the device-resident pose pools.  table [nb][4] = (first frame of the sequence in the
// pool, sequence length, chunk start (may be < 0), flip).  Frames outside the sequence repeat its edge frame
// (generators.py:113-118 np.pad 'edge'); flip negates x and swaps left/right joints (generators.py:120-123,136-140);
// zero_root writes joint 0 of the 3D target as 0 (main.py:365).
__global__ __launch_bounds__(256) void batch_gather_kernel(const float* __restrict__ pool2d,
                                                           const float* __restrict__ pool3d,
                                                           const int* __restrict__ table, const int* __restrict__ perm2d,
                                                           const int* __restrict__ perm3d, float* __restrict__ out2d,
                                                           float* __restrict__ out3d, int nb, int F, int J,
                                                           int zero_root) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)nb * F * J;
  if (i >= total) return;
  const int j = (int)(i % J);
  const size_t bf = i / J;
  const int f = (int)(bf % F), b = (int)(bf / F);
  const int off = table[b * 4], len = table[b * 4 + 1], start = table[b * 4 + 2], flip = table[b * 4 + 3];
  const int fr = min(max(start + f, 0), len - 1);
  const size_t row = (size_t)(off + fr) * J;
  {
    const float* s = pool2d + (row + (flip ? perm2d[j] : j)) * 2;
    out2d[i * 2] = flip ? -s[0] : s[0];
    out2d[i * 2 + 1] = s[1];
  }
  if (pool3d != nullptr) {
    const float* s = pool3d + (row + (flip ? perm3d[j] : j)) * 3;
    const bool z = zero_root && j == 0;
    out3d[i * 3] = z ? 0.f : (flip ? -s[0] : s[0]);
    out3d[i * 3 + 1] = z ? 0.f : s[1];
    out3d[i * 3 + 2] = z ? 0.f : s[2];
  }
}

// AdamW over every parameter tensor in one launch.  Each block takes one chunk of one tensor from the table; the update
// follows torch.optim.AdamW's single-tensor order of operations (decoupled weight decay first; exp_avg by lerp;
// exp_avg_sq by mul + addcmul; denom = sqrt(v)/sqrt(bc2) + eps; p += -step_size * m/denom).
struct AdamChunk { float* p; const float* g; float* m; float* v; int n; int pad; };

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, float decay, float w1, float beta2,
                                          float w2, float bc2_sqrt, float eps, float neg_step_size) {
  p = p * decay;
  m = __fadd_rn(m, __fmul_rn(w1, __fsub_rn(g, m)));
  v = __fmul_rn(v, beta2);
  v = __fadd_rn(v, __fmul_rn(__fmul_rn(w2, g), g));
  const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), eps);
  p = __fadd_rn(p, __fdiv_rn(__fmul_rn(neg_step_size, m), denom));
}

__global__ __launch_bounds__(256) void adamw_kernel(const AdamChunk* __restrict__ chunks, float decay, float w1,
                                                    float beta2, float w2, float bc2_sqrt, float eps,
                                                    float neg_step_size) {
  const AdamChunk c = chunks[blockIdx.x];
  // 16-byte path when the four pointers allow it (torch allocations are 512-byte aligned; chunk offsets are multiples
  // of 32 K elements), scalar tail / fallback otherwise
  const bool vec = ((((uintptr_t)c.p | (uintptr_t)c.g | (uintptr_t)c.m | (uintptr_t)c.v) & 15) == 0);
  const int n4 = vec ? c.n / 4 : 0;
  for (int i = threadIdx.x; i < n4; i += 256) {
    float4 p = reinterpret_cast<float4*>(c.p)[i], m = reinterpret_cast<float4*>(c.m)[i], v = reinterpret_cast<float4*>(c.v)[i];
    const float4 g = reinterpret_cast<const float4*>(c.g)[i];
    adamw_one(p.x, g.x, m.x, v.x, decay, w1, beta2, w2, bc2_sqrt, eps, neg_step_size);
    adamw_one(p.y, g.y, m.y, v.y, decay, w1, beta2, w2, bc2_sqrt, eps, neg_step_size);
    adamw_one(p.z, g.z, m.z, v.z, decay, w1, beta2, w2, bc2_sqrt, eps, neg_step_size);
    adamw_one(p.w, g.w, m.w, v.w, decay, w1, beta2, w2, bc2_sqrt, eps, neg_step_size);
    reinterpret_cast<float4*>(c.p)[i] = p; reinterpret_cast<float4*>(c.m)[i] = m; reinterpret_cast<float4*>(c.v)[i] = v;
  }
  for (int i = n4 * 4 + threadIdx.x; i < c.n; i += 256) {
    float p = c.p[i], m = c.m[i], v = c.v[i];
    adamw_one(p, c.g[i], m, v, decay, w1, beta2, w2, bc2_sqrt, eps, neg_step_size);
    c.p[i] = p; c.m[i] = m; c.v[i] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// N4: Procrustes-aligned per-joint error of one predicted pose against its target (loss.py:190-247 and its
// *_diffusion variants: similarity transform = scale, rotation, translation minimising the squared error).
// One thread per pose; the 3x3 problem is solved in fp64: V and the singular values from a Jacobi eigen-decomposition
// of H^T H, u_i = H v_i / s_i for the two leading pairs, the third pair completed by cross products -- which IS the
// reflection-corrected rotation of loss.py:218-222 (the unique proper rotation agreeing on the two leading pairs) --
// and the corrected trace s0 + s1 + sign(det H) s2.
__device__ void jacobi_eig3(double A[3][3], double V[3][3]) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = A[p][q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < 3; ++k) {          // A <- A J
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = cs * akp - sn * akq; A[k][q] = sn * akp + cs * akq;
        }
        for (int k = 0; k < 3; ++k) {          // A <- J^T A
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = cs * apk - sn * aqk; A[q][k] = sn * apk + cs * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = cs * vkp - sn * vkq; V[k][q] = sn * vkp + cs * vkq;
        }
      }
  }
}

__global__ __launch_bounds__(128) void procrustes_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                         float* __restrict__ err, float* __restrict__ aligned,
                                                         size_t n_pose, int KH, int F, int J) {
  const size_t i = (size_t)blockIdx.x * 128 + threadIdx.x;
  if (i >= n_pose) return;
  // pose i = ((b*KH + kh)*F + f)  ->  target (b*F + f)
  const size_t f = i % F, b = i / ((size_t)KH * F);
  const float* Y = pred + i * J * 3;
  const float* X = tgt + (b * F + f) * J * 3;
  double muX[3] = {0, 0, 0}, muY[3] = {0, 0, 0};
  for (int j = 0; j < J; ++j)
    for (int d = 0; d < 3; ++d) { muX[d] += X[j * 3 + d]; muY[d] += Y[j * 3 + d]; }
  for (int d = 0; d < 3; ++d) { muX[d] /= J; muY[d] /= J; }
  double nX = 0, nY = 0, Hm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int j = 0; j < J; ++j) {
    double x[3], y[3];
    for (int d = 0; d < 3; ++d) { x[d] = X[j * 3 + d] - muX[d]; y[d] = Y[j * 3 + d] - muY[d]; nX += x[d] * x[d]; nY += y[d] * y[d]; }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Hm[r][c] += x[r] * y[c];
  }
  nX = sqrt(nX); nY = sqrt(nY);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Hm[r][c] /= (nX * nY);       // H = X0^T Y0 of the normalised sets
  // H = U S V^T ; eigen-decomposition of H^T H = V S^2 V^T
  double A[3][3], V[3][3];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[r][c] = Hm[0][r] * Hm[0][c] + Hm[1][r] * Hm[1][c] + Hm[2][r] * Hm[2][c];
  jacobi_eig3(A, V);
  int o[3] = {0, 1, 2};                                                                 // sort eigenvalues descending
  if (A[o[0]][o[0]] < A[o[1]][o[1]]) { int t = o[0]; o[0] = o[1]; o[1] = t; }
  if (A[o[1]][o[1]] < A[o[2]][o[2]]) { int t = o[1]; o[1] = o[2]; o[2] = t; }
  if (A[o[0]][o[0]] < A[o[1]][o[1]]) { int t = o[0]; o[0] = o[1]; o[1] = t; }
  double s[3], v[3][3], u[3][3];
  for (int k = 0; k < 3; ++k) { s[k] = sqrt(fmax(A[o[k]][o[k]], 0.0)); for (int r = 0; r < 3; ++r) v[k][r] = V[r][o[k]]; }
  for (int k = 0; k < 2; ++k) {
    for (int r = 0; r < 3; ++r) u[k][r] = Hm[r][0] * v[k][0] + Hm[r][1] * v[k][1] + Hm[r][2] * v[k][2];
    const double nn = sqrt(u[k][0] * u[k][0] + u[k][1] * u[k][1] + u[k][2] * u[k][2]);
    for (int r = 0; r < 3; ++r) u[k][r] /= fmax(nn, 1e-300);
  }
  // re-orthogonalise u1 against u0 (matters only when s1 is tiny), then complete both bases
  { const double d = u[0][0] * u[1][0] + u[0][1] * u[1][1] + u[0][2] * u[1][2];
    for (int r = 0; r < 3; ++r) u[1][r] -= d * u[0][r];
    const double nn = sqrt(u[1][0] * u[1][0] + u[1][1] * u[1][1] + u[1][2] * u[1][2]);
    for (int r = 0; r < 3; ++r) u[1][r] /= fmax(nn, 1e-300); }
  u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1]; u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2]; u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
  v[2][0] = v[0][1] * v[1][2] - v[0][2] * v[1][1]; v[2][1] = v[0][2] * v[1][0] - v[0][0] * v[1][2]; v[2][2] = v[0][0] * v[1][1] - v[0][1] * v[1][0];
  const double detH = Hm[0][0] * (Hm[1][1] * Hm[2][2] - Hm[1][2] * Hm[2][1]) - Hm[0][1] * (Hm[1][0] * Hm[2][2] - Hm[1][2] * Hm[2][0]) +
                      Hm[0][2] * (Hm[1][0] * Hm[2][1] - Hm[1][1] * Hm[2][0]);
  const double tr = s[0] + s[1] + (detH < 0 ? -s[2] : s[2]);
  double R[3][3];                                                                       // R = V U^T
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[r][c] = v[0][r] * u[0][c] + v[1][r] * u[1][c] + v[2][r] * u[2][c];
  const double a = tr * nX / nY;
  double t[3];
  for (int c = 0; c < 3; ++c) t[c] = muX[c] - a * (muY[0] * R[0][c] + muY[1] * R[1][c] + muY[2] * R[2][c]);
  for (int j = 0; j < J; ++j) {
    double e2 = 0;
    for (int c = 0; c < 3; ++c) {
      const double al = a * (Y[j * 3] * R[0][c] + Y[j * 3 + 1] * R[1][c] + Y[j * 3 + 2] * R[2][c]) + t[c];
      if (aligned != nullptr) aligned[(i * J + j) * 3 + c] = (float)al;
      const double d = al - X[j * 3 + c];
      e2 += d * d;
    }
    err[i * J + j] = (float)sqrt(e2);
  }
}

}  // namespace

extern "C" {

int d3dp_clip_count(int32_t n, int32_t F) { return (n <= 0 || F <= 0) ? 0 : (n <= F ? 1 : n / F + (n % F ? 1 : 0)); }

int d3dp_clip_gather(const float* src, float* dst, float* dst_flip, const int32_t* perm, int32_t n, int32_t F, int32_t J,
                     int32_t D, void* stream) {
  if (src == nullptr || dst == nullptr || n <= 0 || F <= 0 || J <= 0 || D <= 0 || (dst_flip != nullptr && perm == nullptr))
    return d3dp_set_error(-1, "d3dp_clip_gather: bad argument");
  const int n_clips = d3dp_clip_count(n, F);
  const size_t total = (size_t)n_clips * F * J;
  hipLaunchKernelGGL(clip_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                     dst, dst_flip, perm, n, n_clips, F, J, D);
  return d3dp_check_launch("clip_gather");
}

int d3dp_clip_scatter(const float* pred, float* out, int32_t n, int32_t K, int32_t H, int32_t F, int32_t J, int32_t D,
                      int32_t last_wins, void* stream) {
  if (pred == nullptr || out == nullptr || n <= 0 || K <= 0 || H <= 0 || F <= 0 || J <= 0 || D <= 0)
    return d3dp_set_error(-1, "d3dp_clip_scatter: bad argument");
  const int n_clips = d3dp_clip_count(n, F);
  const size_t total = (size_t)K * H * n * J * D;
  hipLaunchKernelGGL(clip_scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred,
                     out, n, n_clips, K * H, F, J * D, last_wins);
  return d3dp_check_launch("clip_scatter");
}

int d3dp_jpma_combine(const float* win, int32_t R, size_t n, float* agg, int32_t* sel, void* stream) {
  if (win == nullptr || agg == nullptr || R <= 0 || n == 0) return d3dp_set_error(-1, "d3dp_jpma_combine: bad argument");
  hipLaunchKernelGGL(jpma_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, win, R, n,
                     agg, sel);
  return d3dp_check_launch("jpma_combine");
}

int d3dp_batch_gather(const float* pool2d, const float* pool3d, const int32_t* table, const int32_t* perm2d,
                      const int32_t* perm3d, float* out2d, float* out3d, int32_t nb, int32_t F, int32_t J,
                      int32_t zero_root, void* stream) {
  if (pool2d == nullptr || table == nullptr || perm2d == nullptr || out2d == nullptr || nb <= 0 || F <= 0 || J <= 0 ||
      (pool3d != nullptr && (out3d == nullptr || perm3d == nullptr)))
    return d3dp_set_error(-1, "d3dp_batch_gather: bad argument");
  const size_t total = (size_t)nb * F * J;
  hipLaunchKernelGGL(batch_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     pool2d, pool3d, table, perm2d, perm3d, out2d, out3d, nb, F, J, zero_root);
  return d3dp_check_launch("batch_gather");
}

int d3dp_adamw_step(const void* chunks, int32_t n_chunks, double lr, double beta1, double beta2, double eps,
                    double weight_decay, int64_t step, void* stream) {
  if (chunks == nullptr || n_chunks <= 0 || step <= 0) return d3dp_set_error(-1, "d3dp_adamw_step: bad argument");
  // scalars are formed in double on the host exactly as torch/optim/adamw.py does, then rounded once to fp32
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  const double step_size = lr / bc1;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream,
                     (const AdamChunk*)chunks, (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps, (float)(-step_size));
  return d3dp_check_launch("adamw");
}

int d3dp_procrustes(const float* pred, const float* target, float* err, float* aligned, int32_t B, int32_t KH, int32_t F,
                    int32_t J, void* stream) {
  if (pred == nullptr || target == nullptr || err == nullptr || B <= 0 || KH <= 0 || F <= 0 || J < 3 || J > 64)
    return d3dp_set_error(-1, "d3dp_procrustes: bad argument");
  const size_t n_pose = (size_t)B * KH * F;
  hipLaunchKernelGGL(procrustes_kernel, dim3((unsigned)((n_pose + 127) / 128)), dim3(128), 0, (hipStream_t)stream,
                     pred, target, err, aligned, n_pose, KH, F, J);
  return d3dp_check_launch("procrustes");
}

}  // extern "C"
