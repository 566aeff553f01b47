// Linear layers of the MixSTE2 denoiser:  out[M,N] = epilogue(A[M,K] . W[N,K]^T + bias[N])
// (reference: every nn.Linear on the path -- common/mixste.py:65 qkv, :80 proj, :38-41 fc1/fc2).
//
// W is the nn.Linear weight as stored in the reference state_dict ([out, in], K contiguous), so both
// MFMA operands are read along K and no transpose is ever materialised.
//
//   gemm_bf16_stream_kernel : FAST mode (bottom of this file).  bf16 operands, fp32 accumulate on
//                      v_mfma_f32_16x16x32_bf16; persistent 256x128x64 streaming kernel.  Tiles are DMA'd HBM->LDS with
//                      global_load_lds_dwordx4 (16 B/lane); the LDS image is lane-linear, so the bank swizzle is applied
//                      on the per-lane SOURCE address and again on the ds_read_b128 fragment address (same involution).
//   gemm_f32_kernel  : TRAIN mode (and the EXACT cross-check D3DP_EXACT_IMPL=f32).  fp32 operands on
//                      v_mfma_f32_32x32x2_f32 (bit-for-bit an fp32 fmaf chain over k).  128x128x16 block tile, 2x2 32x32
//                      tiles per wave.
//   gemm_bf16x3_kernel : six-pass split-bf16 Linear (EXACT cross-check D3DP_EXACT_IMPL=bf16x3; the EXACT default is
//                      the three-pass split-fp16 kernel in gemm_x2.hip).
//
// All compute the TRANSPOSED product per wave (weight fragment as the MFMA A operand) so that each
// lane ends up with consecutive output columns of one row -> vector stores, bias as one float4.
//
// Epilogues: EPI_BIAS (out = acc+b), EPI_GELU (out = gelu(acc+b)), EPI_RESID (resid += acc+b, fp32).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128;

template <int EPI, typename OutT>
__device__ __forceinline__ void epilogue4(const float* v, const float* __restrict__ bias, OutT* out,
                                          size_t row_off, int n) {
  const float4 b = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  float r[4] = {v[0] + b.x, v[1] + b.y, v[2] + b.z, v[3] + b.w};
  if constexpr (EPI == EPI_GELU) {
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = gelu_erf(r[i]);
  }
  if constexpr (EPI == EPI_RESID) {
    float* o = reinterpret_cast<float*>(out) + row_off + n;
    float4 x = *reinterpret_cast<float4*>(o);
    x.x += r[0]; x.y += r[1]; x.z += r[2]; x.w += r[3];
    *reinterpret_cast<float4*>(o) = x;
  } else if constexpr (sizeof(OutT) == 4) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + row_off + n) = make_float4(r[0], r[1], r[2], r[3]);
  } else {
    bf16x4 p = {(bf16)r[0], (bf16)r[1], (bf16)r[2], (bf16)r[3]};
    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(out) + row_off + n) = p;
  }
}

// physical 16-B slot of logical slot s in row `row` (8 slots per 128-B row)
__device__ __forceinline__ int swz(int row, int s) { return s ^ ((row >> 1) & 7); }

// ------------------------------------------------------------------------------------------------
// EXACT: fp32 MFMA
// ------------------------------------------------------------------------------------------------
constexpr int XBK = 16;
constexpr int XLD = BM + 4;   // padded leading dimension of the k-major LDS tiles

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* out, int M, int N,
                                                       int K, int n_tiles_n, int nk_per_split) {
  __shared__ float As[2][XBK][XLD];
  __shared__ float Ws[2][XBK][XLD];

  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (L / n_tiles_n) * BM;
  const int n0 = (L % n_tiles_n) * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // global->register staging: 128 rows x 4 float4 per operand tile, 2 float4 per thread each
  const float* pa[2];
  const float* pw[2];
  int lrow[2], lk[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * 256;
    lrow[i] = idx >> 2;
    lk[i] = (idx & 3) * 4;
    pa[i] = A + (size_t)min(m0 + lrow[i], M - 1) * K + lk[i];
    pw[i] = W + (size_t)min(n0 + lrow[i], N - 1) * K + lk[i];
  }
  float4 ra[2], rw[2];
  // (K is any multiple of 4: the float4 slots of the last k-step that lie behind K contribute zeros -- widths that are not a
  //  multiple of 16, round 6)
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool in = kt * XBK + lk[i] < K;
      ra[i] = in ? *reinterpret_cast<const float4*>(pa[i] + kt * XBK) : make_float4(0.f, 0.f, 0.f, 0.f);
      rw[i] = in ? *reinterpret_cast<const float4*>(pw[i] + kt * XBK) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      As[buf][lk[i] + 0][lrow[i]] = ra[i].x; As[buf][lk[i] + 1][lrow[i]] = ra[i].y;
      As[buf][lk[i] + 2][lrow[i]] = ra[i].z; As[buf][lk[i] + 3][lrow[i]] = ra[i].w;
      Ws[buf][lk[i] + 0][lrow[i]] = rw[i].x; Ws[buf][lk[i] + 1][lrow[i]] = rw[i].y;
      Ws[buf][lk[i] + 2][lrow[i]] = rw[i].z; Ws[buf][lk[i] + 3][lrow[i]] = rw[i].w;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // split-K: blockIdx.y owns k-steps [kt0, nk) of the K / XBK total (gridDim.y == 1: everything)
  const int kt0 = blockIdx.y * nk_per_split;
  const int nk = min((K + XBK - 1) / XBK, kt0 + nk_per_split);
  if (kt0 >= nk) return;
  gload(kt0);
  lstore(0);
  __syncthreads();
  const int l31 = lane & 31, lh = lane >> 5;
  int cur = 0;
  for (int kt = kt0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int kk = 0; kk < XBK / 2; ++kk) {
      float wv[2], av[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        wv[i] = Ws[cur][kk * 2 + lh][wc * 64 + i * 32 + l31];
        av[i] = As[cur][kk * 2 + lh][wr * 64 + i * 32 + l31];
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[ni], av[mi], acc[mi][ni], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // lane holds out[m = .. + l31][n = .. + 8q + 4*lh + (0..3)], q = 0..3
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int m = m0 + wr * 64 + mi * 32 + l31;
    if (m >= M) continue;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wc * 64 + ni * 32 + q * 8 + lh * 4;
        if (n >= N) continue;
        float v[4] = {acc[mi][ni][q * 4 + 0], acc[mi][ni][q * 4 + 1], acc[mi][ni][q * 4 + 2], acc[mi][ni][q * 4 + 3]};
        // EPI_PARTIAL: split-K chunk blockIdx.y leaves ITS product as out[blockIdx.y][M][N]; the caller adds the chunks in order
        // (round 6: the split-K form added them with fp32 atomics -- the last float atomics of the library)
        if constexpr (EPI == EPI_PARTIAL) epilogue4<EPI_BIAS, float>(v, nullptr, out + (size_t)blockIdx.y * M * N, (size_t)m * N, n);
        else epilogue4<EPI, float>(v, bias, out, (size_t)m * N, n);
      }
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
int d3dp_launch_linear_f32(int epi, const float* A, const float* W, const float* bias, float* out, int M, int N,
                           int K, hipStream_t st) {
  if (K < 4 || K % 4 != 0 || N % 4 != 0 || M <= 0) return -1;
  const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
  dim3 g(tm * tn), b(256);
  const int nk = (K + XBK - 1) / XBK;
  if (epi == EPI_RESID) hipLaunchKernelGGL((gemm_f32_kernel<EPI_RESID>), g, b, 0, st, A, W, bias, out, M, N, K, tn, nk);
  else if (epi == EPI_GELU) hipLaunchKernelGGL((gemm_f32_kernel<EPI_GELU>), g, b, 0, st, A, W, bias, out, M, N, K, tn, nk);
  else if (epi == EPI_BIAS) hipLaunchKernelGGL((gemm_f32_kernel<EPI_BIAS>), g, b, 0, st, A, W, bias, out, M, N, K, tn, nk);
  else return -1;
  return 0;
}

// split-K form for tall contractions (wgrad: K = tokens): the k-steps are cut into ~ (CUs / output tiles) chunks, every chunk leaves
// its partial product in `part` ([chunks][M][N] floats, part_floats of room) and d3dp_launch_sum_partials adds them in chunk order
// into `out`: a fixed summation order, bit-reproducible (through round 5 the chunks were added with fp32 atomics).
int d3dp_launch_linear_f32_splitk(const float* A, const float* W, float* out, int M, int N, int K, hipStream_t st, float* part,
                                  size_t part_floats) {
  if (K < 4 || K % 4 != 0 || N % 4 != 0 || M <= 0 || !part) return -1;
  const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
  const int nk = (K + XBK - 1) / XBK;
  const int target = 256;                                         // workgroups aimed at: ~1 per CU
  int splits = (target + tm * tn - 1) / (tm * tn);
  if (splits > nk / 8) splits = nk / 8 > 0 ? nk / 8 : 1;
  const int per = (nk + splits - 1) / splits;
  const int chunks = (nk + per - 1) / per;                        // (every chunk owns at least one k-step: every partial is written)
  if ((size_t)chunks * M * N > part_floats) return -1;
  hipLaunchKernelGGL((gemm_f32_kernel<EPI_PARTIAL>), dim3(tm * tn, chunks), dim3(256), 0, st, A, W, nullptr, part, M, N, K, tn, per);
  d3dp_launch_sum_partials(part, out, (size_t)M * N, chunks, st);
  return 0;
}

// ================================================================================================
// EXACT mode on the bf16 matrix cores: split-bf16 ("bf16x3") GEMM.
//   A3 [3][M][K], W3 [3][N][K] bf16 planes (x = x0 + x1 + x2 exactly).  out = sum over the six leading plane pairs
//   (a_i, w_j), i + j <= 2, accumulated in fp32: fp32-class accuracy (better than an fp32 fmaf chain, because each
//   v_mfma_f32_16x16x32_bf16 sums 32 exact products before one rounding).
//   256x128x32 tile, 8 waves (4x2, 64x64 each), 2-stage LDS ring of 72 KiB filled by global_load_lds_dwordx4.
//   Per k-step and wave: 24 ds_read_b128 feed 96 MFMA -> the kernel is bound by the matrix pipe, not by issue slots.
//   Epilogues: EPI_BIAS -> fp32 out; EPI_GELU -> gelu_erf(acc + b) re-split into three bf16 planes (the next
//   Linear's A operand).
// ================================================================================================
namespace {

constexpr int TBM = 256, TBN = 128, TBK = 32;
constexpr int TA_PLANE = TBM * TBK * 2;            // 16 KiB
constexpr int TW_PLANE = TBN * TBK * 2;            //  8 KiB
constexpr int TSTAGE = 3 * TA_PLANE + 3 * TW_PLANE;   // 72 KiB
constexpr int TLDS = 2 * TSTAGE;

// 64-byte rows (4 slots of 16 B): XOR bit 1 of the slot with bit 3 of the row -> conflict-free ds_read_b128 fragments
__device__ __forceinline__ int swz3(int row, int s) { return s ^ (((row >> 3) & 1) << 1); }

template <int EPI>
__global__ __launch_bounds__(512) void gemm_bf16x3_kernel(const bf16* __restrict__ A3, const bf16* __restrict__ W3,
                                                          const float* __restrict__ bias, float* __restrict__ outf,
                                                          bf16* __restrict__ out3, int M, int N, int K, int n_tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (L / n_tiles_n) * TBM;
  const int n0 = (L % n_tiles_n) * TBN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const size_t planeA = (size_t)M * K, planeW = (size_t)N * K;

  // 72 pieces of 1 KiB (16 rows x 64 B) per stage: A plane p rows 16j.. (48 pieces), then W (24 pieces); wave w
  // issues pieces w, w+8, ..., w+64.
  const bf16* src[9];
  int dst[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int q = wave + 8 * i;
    const int r = lane >> 2, ps = lane & 3;
    if (q < 48) {
      const int pl = q >> 4, row = (q & 15) * 16 + r;
      src[i] = A3 + pl * planeA + (size_t)min(m0 + row, M - 1) * K + swz3(row, ps) * 8;
      dst[i] = pl * TA_PLANE + (q & 15) * 1024;
    } else {
      const int q2 = q - 48, pl = q2 >> 3, row = (q2 & 7) * 16 + r;
      src[i] = W3 + pl * planeW + (size_t)min(n0 + row, N - 1) * K + swz3(row, ps) * 8;
      dst[i] = 3 * TA_PLANE + pl * TW_PLANE + (q2 & 7) * 1024;
    }
  }
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * TSTAGE;
#pragma unroll
    for (int i = 0; i < 9; ++i)
      __builtin_amdgcn_global_load_lds(GPTR(src[i] + kt * TBK), LPTR(base + dst[i]), 16, 0, 0);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = K / TBK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int fi = lane & 15, fg = lane >> 4;
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* sa = smem + cur * TSTAGE;
    const char* sw = sa + 3 * TA_PLANE;
    bf16x8 wf[4][3];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int rw = wc * 64 + ni * 16 + fi;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        wf[ni][pl] = *reinterpret_cast<const bf16x8*>(sw + pl * TW_PLANE + rw * 64 + swz3(rw, fg) * 16);
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int ra = wr * 64 + mi * 16 + fi;
      bf16x8 af[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        af[pl] = *reinterpret_cast<const bf16x8*>(sa + pl * TA_PLANE + ra * 64 + swz3(ra, fg) * 16);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        f32x4 c = acc[mi][ni];
        c = D3DP_MFMA_16x16x32_BF16(wf[ni][0], af[2], c);   // smallest terms first
        c = D3DP_MFMA_16x16x32_BF16(wf[ni][1], af[1], c);
        c = D3DP_MFMA_16x16x32_BF16(wf[ni][2], af[0], c);
        c = D3DP_MFMA_16x16x32_BF16(wf[ni][0], af[1], c);
        c = D3DP_MFMA_16x16x32_BF16(wf[ni][1], af[0], c);
        c = D3DP_MFMA_16x16x32_BF16(wf[ni][0], af[0], c);
        acc[mi][ni] = c;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  // lane holds out[m = .. + fi][n = .. + 4*fg .. +3]
  const size_t planeO = (size_t)M * N;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wr * 64 + mi * 16 + fi;
    if (m >= M) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wc * 64 + ni * 16 + fg * 4;
      if (n >= N) continue;
      const float4 b = *reinterpret_cast<const float4*>(bias + n);
      float r[4] = {acc[mi][ni][0] + b.x, acc[mi][ni][1] + b.y, acc[mi][ni][2] + b.z, acc[mi][ni][3] + b.w};
      if constexpr (EPI == EPI_GELU) {
        bf16x4 p0, p1, p2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bf16 a0, a1, a2;
          split3(gelu_erf(r[e]), a0, a1, a2);
          p0[e] = a0; p1[e] = a1; p2[e] = a2;
        }
        bf16* o = out3 + (size_t)m * N + n;
        *reinterpret_cast<bf16x4*>(o) = p0;
        *reinterpret_cast<bf16x4*>(o + planeO) = p1;
        *reinterpret_cast<bf16x4*>(o + 2 * planeO) = p2;
      } else {
        *reinterpret_cast<float4*>(outf + (size_t)m * N + n) = make_float4(r[0], r[1], r[2], r[3]);
      }
    }
  }
}

__global__ void split3_kernel(const float* __restrict__ s, bf16* __restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    bf16 a0, a1, a2;
    split3(s[i], a0, a1, a2);
    d[i] = a0; d[i + n] = a1; d[i + 2 * n] = a2;
  }
}

}  // namespace

// out = epi(A W^T + bias) with A3/W3 split-bf16 planes; EPI_BIAS: fp32 `outf`; EPI_GELU: three bf16 planes `out3`.
int d3dp_launch_linear_bf16x3(int epi, const void* A3, const void* W3, const float* bias, float* outf, void* out3, int M,
                              int N, int K, hipStream_t st) {
  if (K % TBK != 0 || N % 4 != 0 || M <= 0) return -1;
  const int tm = (M + TBM - 1) / TBM, tn = (N + TBN - 1) / TBN;
  static PerDeviceOnce once;
  if (once.get([&](int) {
        return d3dp_lds_opt_in(reinterpret_cast<const void*>(gemm_bf16x3_kernel<EPI_BIAS>), TLDS) < 0 ? -3
               : d3dp_lds_opt_in(reinterpret_cast<const void*>(gemm_bf16x3_kernel<EPI_GELU>), TLDS);
      }) < 0) return -3;
  if (epi == EPI_BIAS)
    hipLaunchKernelGGL((gemm_bf16x3_kernel<EPI_BIAS>), dim3(tm * tn), dim3(512), TLDS, st, (const bf16*)A3, (const bf16*)W3,
                       bias, outf, (bf16*)out3, M, N, K, tn);
  else if (epi == EPI_GELU)
    hipLaunchKernelGGL((gemm_bf16x3_kernel<EPI_GELU>), dim3(tm * tn), dim3(512), TLDS, st, (const bf16*)A3, (const bf16*)W3,
                       bias, outf, (bf16*)out3, M, N, K, tn);
  else return -1;
  return 0;
}

void d3dp_launch_split3(const float* src, void* dst, size_t n, hipStream_t st) {
  const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(split3_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, (bf16*)dst, n);
}

// ================================================================================================
// FAST mode, persistent streaming GEMM ("v2"): the kernel the denoiser launches for every Linear.
//
//   * 256x128 output tile per workgroup pass, BK = 64, 8 compute waves (4 x 2, 64x64 each = 4x4 MFMA tiles)
//     + 4 LOADER waves.  Workgroups are persistent (<= 256, one per CU) and walk tiles L, L+G, L+2G, ...
//   * A and W k-slabs stream through a 3-stage LDS ring (48 KiB per stage) as one continuous sequence of
//     k-steps across tile boundaries: the next tile's first slabs are already in flight while the current
//     tile's epilogue stores drain -- no per-tile prologue.
//   * Only the loader waves issue global_load_lds and wait on vmcnt (counted: one k-step stays in flight
//     across every barrier).  Compute waves never wait on vmcnt, so their epilogue stores (which also count
//     in vmcnt on gfx950) cannot stall the operand pipeline.  One raw s_barrier per k-step.
//   * bias sits in LDS (read with ds_read, lgkmcnt), so the epilogue issues no vector loads at all.
// ================================================================================================
namespace {

constexpr int SBM = 256, SBN = 128, SBK = 64;
constexpr int SA_BYTES = SBM * SBK * 2;              // 32 KiB
constexpr int SW_BYTES = SBN * SBK * 2;              // 16 KiB
constexpr int SSTAGE = SA_BYTES + SW_BYTES;          // 48 KiB
constexpr int SNSTAGE = 3;
constexpr int SBIAS_MAX = 2048;                      // floats of bias kept in LDS
constexpr int SLDS_BYTES = SNSTAGE * SSTAGE + SBIAS_MAX * 4;

// GELU for the FAST path's bf16 outputs: erf by an odd minimax polynomial on a clamped argument
// (|error| <= 2.4e-4 absolute in GELU -- an eighth of the bf16 output resolution at |x| ~ 1), 11 full-rate VALU
// operations that pair into packed instructions, against ~40 for the correctly-rounded erff the EXACT path uses (and
// 10 + a quarter-rate v_rcp_f32 for Abramowitz-Stegun 7.1.27, which this replaced: fc1 100 -> 89 us back to back).
// The streaming kernel is bound by instruction issue slots, not by the matrix pipe alone, so epilogue VALU count is
// first-order.
__device__ __forceinline__ float gelu_fast(float x) {
  // erf(u / sqrt 2) ~ u P(u^2) on |u| <= 3.8 (odd minimax polynomial of degree 13, |error| <= 1.3e-4; beyond 3.8 the
  // clamp leaves 1 - erf <= 1.5e-4): max |GELU error| 2.4e-4 over all x, no transcendental, and every operation is an
  // FMA or multiply that the compiler pairs into v_pk_fma_f32 / v_pk_mul_f32 across neighbouring elements.
  const float u = fminf(fmaxf(x, -3.8f), 3.8f);
  const float t = u * u;
  float p = fmaf(t, 7.331557583256654e-08f, -4.5449246499629226e-06f);
  p = fmaf(t, p, 0.0001213696159538813f);
  p = fmaf(t, p, -0.0018630953272804618f);
  p = fmaf(t, p, 0.018633270636200905f);
  p = fmaf(t, p, -0.13143958151340485f);
  p = fmaf(t, p, 0.7973535060882568f);
  const float e = u * p;
  const float h = 0.5f * x;
  return fmaf(h, e, h);                     // 0.5 x (1 + erf(x / sqrt 2))
}

// Column permutation inside a wave's 64-wide output strip.  The activation fragment is the MFMA A operand, so a lane
// (fi, fg) holds C[row 4 fg + r][column fi] of each 16x16 tile; LDS row ni*16 + i of the wave's W strip carries output
// column 4 i + ni, which makes the lane's four tiles ni four CONSECUTIVE columns: one 8-byte bf16 store per (mi, r) in
// which the 16 lanes of a group cover one complete 128-byte line of ONE row (as in gemm_x2.hip, where the row-per-lane
// store pattern this replaces was measured at 11-17 us of store tail per 256x128 tile).  The loader applies the same
// map when it picks the W row for an LDS row, so the MFMA-side reads stay in natural (conflict-free) order.
__device__ __forceinline__ int sperm(int q) {   // q = ni*16 + i in [0,64)
  return (q & 15) * 4 + (q >> 4);
}

// MI = 16-row MFMA blocks per compute wave: MI = 4 -> 8 compute waves (4 x 2, 64x64 each, two per SIMD);
// MI = 8 -> 4 compute waves (2 x 2, 128x64 each, ONE per SIMD next to one loader wave: no matrix-pipe / issue
// contention between compute waves, 0.375 instead of 0.5 LDS fragment reads per MFMA).
// WIDE is a name tag only (same code): the qkv Linear (N = 3K) is instantiated as its own kernel symbol so that
// rocprofv3 --stats reports it separately from the proj Linear, which shares EPI / OutT / NK with it.
template <int EPI, typename OutT, int NK, int MI, int WIDE = 0>
__global__ __launch_bounds__(MI == 4 ? 768 : 512) void gemm_bf16_stream_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                               const float* __restrict__ bias, OutT* __restrict__ out,
                                                               int M, int N, int tiles_n, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sbias = reinterpret_cast<float*>(smem + SNSTAGE * SSTAGE);
  constexpr int K = NK * SBK;

  const int G = gridDim.x;
  const int L = xcd_remap(blockIdx.x, G);
  const int n_my = (total_tiles - L + G - 1) / G;      // tiles L, L+G, ...
  const int gtot = n_my * NK;

  constexpr int NCW = (SBM / (MI * 16)) * 2;          // compute waves
  constexpr int NU = 4 * MI;                          // pending 4-column units (mi, r) per wave and tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  for (int i = tid; i < N; i += (NCW + 4) * 64) sbias[i] = bias[i];
  __syncthreads();

  if (wave >= NCW) {
    // ------------------------------------------------------------------ loader waves
    const int lw = wave - NCW;
    const int lrow = lane >> 3, lslot = lane & 7;
    auto issue = [&](int g) {
      const int ti = g / NK, ks = g - ti * NK;
      const int t = L + ti * G;
      const int m0 = (t / tiles_n) * SBM, n0 = (t % tiles_n) * SBN;
      char* stage = smem + (g % SNSTAGE) * SSTAGE;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = lw * 64 + i * 8 + lrow;
        const int gr = min(m0 + row, M - 1);
        const bf16* src = A + (size_t)gr * K + ks * SBK + swz(row, lslot) * 8;
        __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(stage + (lw * 8 + i) * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = lw * 32 + i * 8 + lrow;                    // LDS row of the W slab
        const int wrow = (row & 64) + sperm(row & 63);              // output column it carries
        const int gr = min(n0 + wrow, N - 1);
        const bf16* src = W + (size_t)gr * K + ks * SBK + swz(row, lslot) * 8;
        __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(stage + SA_BYTES + (lw * 4 + i) * 1024), 16, 0, 0);
      }
    };
    if (gtot > 0) issue(0);
    if (gtot > 1) issue(1);
    for (int g = 0; g < gtot; ++g) {
      // 12 glds per loader wave per k-step (8 A + 4 W pieces): loads(g) landed, loads(g+1) stay in flight
      if (g + 1 < gtot) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      if (g + 2 < gtot) issue(g + 2);
    }
    return;
  }

  // -------------------------------------------------------------------- compute waves
  const int wr = wave >> 1, wc = wave & 1;
  const int fi = lane & 15, fg = lane >> 4;
  f32x4 acc[MI][4];
  // Finished tile waiting to be stored: 16 units (mi, r) of 4 packed bf16 (acc + bias) = 4 consecutive columns of one
  // row.  Units are drained behind the MFMAs of the k-steps of the NEXT tile, so output traffic is a steady trickle
  // instead of a per-tile burst in which every CU of the (phase-locked) persistent grid hits the HBM write path at once.
  // (GELU form: the parked pre-activation is fp16, not bf16 -- 11 significand bits, so the rounding in front of the
  //  nonlinearity is an eighth of the bf16 rounding behind it instead of a second rounding of the same size)
  using pend_t = typename std::conditional<EPI == EPI_GELU, f16x4, bf16x4>::type;
  using pelt_t = typename std::conditional<EPI == EPI_GELU, f16, bf16>::type;
  pend_t pend[NU];
  int pm0 = 0, pn0 = 0;
  bool have_pend = false;

  // store the unit at the head of the pending queue (unit index u -> row block mi = u>>2, row r = u&3), then
  // rotate the queue by one so the head index stays compile-time constant (no runtime-indexed register arrays)
  auto drain_one = [&](int u) {
    const int m = pm0 + wr * (MI * 16) + (u >> 2) * 16 + 4 * fg + (u & 3);
    const int n = pn0 + wc * 64 + 4 * fi;
    bf16x4 v;
    if constexpr (EPI == EPI_GELU) {
      const f16x4 pre = pend[0];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (bf16)gelu_fast((float)pre[e]);
      // keep the activation OUT of the store's bounds branch (LLVM would sink it there, behind the MFMA block)
      asm volatile("" : "+v"(v));
    } else {
      v = pend[0];
    }
    if (m < M && n < N) {
      if constexpr (sizeof(OutT) == 2) {
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(out) + (size_t)m * N + n) = v;
      } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)m * N + n) =
            make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
      }
    }
#pragma unroll
    for (int q = 0; q < NU - 1; ++q) pend[q] = pend[q + 1];
  };
  constexpr int UPS = (NU + NK - 1) / NK;                 // units drained per draining k-step
  constexpr int DEVERY = NK / NU > 0 ? NK / NU : 1;       // drain every DEVERY-th k-step (K > 1024 only)
  // one k-step: 16 ds_read_b128 + 32 MFMA (64 x 64 x 64 per wave)
  auto kstep = [&](int g) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const char* sa = smem + (g % SNSTAGE) * SSTAGE;
    const char* sw = sa + SA_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[MI], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rw = wc * 64 + i * 16 + fi;
        wf[i] = *reinterpret_cast<const bf16x8*>(sw + rw * 128 + swz(rw, kk * 4 + fg) * 16);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int ra = wr * (MI * 16) + i * 16 + fi;
        af[i] = *reinterpret_cast<const bf16x8*>(sa + ra * 128 + swz(ra, kk * 4 + fg) * 16);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = D3DP_MFMA_16x16x32_BF16(af[mi], wf[ni], acc[mi][ni]);
    }
  };
  __builtin_amdgcn_s_setprio(1);
  for (int ti = 0; ti < n_my; ++ti) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int g0 = ti * NK;
    // The drain of the previous tile sits in the SAME basic block as the MFMAs of the draining k-step (no runtime
    // branch around it), so the scheduler can put its VALU work between MFMAs: both compute waves of a SIMD leave
    // each barrier together and a trailing VALU block would leave the matrix pipe idle.
    if (!have_pend) {
#pragma unroll 1
      for (int ks = 0; ks < NK; ++ks) kstep(g0 + ks);
    } else if constexpr (DEVERY > 1) {
      int unit = 0;                                   // one row block every DEVERY k-steps (uniform scalar branch)
#pragma unroll 1
      for (int ks = 0; ks < NK; ++ks) {
        kstep(g0 + ks);
        if (ks % DEVERY == 0 && unit < NU) { drain_one(unit); unit += 1; }
      }
    } else if constexpr (NK * UPS == NU) {
      int unit = 0;                                   // exactly UPS units behind every k-step, no branch
#pragma unroll 1
      for (int ks = 0; ks < NK; ++ks) {
        kstep(g0 + ks);
#pragma unroll
        for (int r = 0; r < UPS; ++r) drain_one(unit++);
      }
    } else {
      int unit = 0;
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        kstep(g0 + ks);
#pragma unroll
        for (int r = 0; r < UPS; ++r)
          if (unit < NU) { drain_one(unit); ++unit; }
      }
    }
    const int t = L + ti * G;
    pm0 = (t / tiles_n) * SBM;
    pn0 = (t % tiles_n) * SBN;
    {
      const float4 b = *reinterpret_cast<const float4*>(sbias + pn0 + wc * 64 + 4 * fi);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          pend[mi * 4 + r] = (pend_t){(pelt_t)(acc[mi][0][r] + b.x), (pelt_t)(acc[mi][1][r] + b.y),
                                      (pelt_t)(acc[mi][2][r] + b.z), (pelt_t)(acc[mi][3][r] + b.w)};
    }
    have_pend = true;
  }
  if (have_pend) {
#pragma unroll
    for (int u = 0; u < NU; ++u) drain_one(u);
  }
}

template <int EPI, typename OutT, int NK, int MI>
int launch_stream_nk_mi(const void* A, const void* W, const float* bias, void* out, int M, int N, hipStream_t st) {
  const int tm = (M + SBM - 1) / SBM, tn = (N + SBN - 1) / SBN;
  const int total = tm * tn;
  void (*kern)(const bf16*, const bf16*, const float*, OutT*, int, int, int, int) =
      gemm_bf16_stream_kernel<EPI, OutT, NK, MI, 0>;
  if constexpr (EPI == EPI_BIAS && sizeof(OutT) == 2 && NK == 8) {
    if (N == 3 * NK * SBK) kern = gemm_bf16_stream_kernel<EPI, OutT, NK, MI, 1>;
  }
  static PerDeviceOnce once[2];                       // per kernel variant: LDS opt-in, then the device's CU count
  const int which = (kern == gemm_bf16_stream_kernel<EPI, OutT, NK, MI, 0>) ? 0 : 1;
  const int n_cu = once[which].get([&](int dev) {
    return d3dp_lds_opt_in(reinterpret_cast<const void*>(kern), SLDS_BYTES) < 0 ? -3 : d3dp_cu_count(dev);
  });
  if (n_cu < 0) return -3;
  const int grid = total < n_cu ? total : n_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(MI == 4 ? 768 : 512), SLDS_BYTES, st, (const bf16*)A, (const bf16*)W, bias,
                     (OutT*)out, M, N, tn, total);
  return 0;
}

template <int EPI, typename OutT, int NK>
int launch_stream_nk(const void* A, const void* W, const float* bias, void* out, int M, int N, hipStream_t st) {
  // MI = 8 (four compute waves of 128x64, one per SIMD) was measured: slower on qkv (a lone wave cannot hide its own
  // ds_read latency), within noise elsewhere -- not instantiated.
  return launch_stream_nk_mi<EPI, OutT, NK, 4>(A, W, bias, out, M, N, st);
}

template <int EPI, typename OutT>
int launch_stream(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, hipStream_t st) {
  switch (K / SBK) {
    case 1: return launch_stream_nk<EPI, OutT, 1>(A, W, bias, out, M, N, st);
    case 2: return launch_stream_nk<EPI, OutT, 2>(A, W, bias, out, M, N, st);
    case 4: return launch_stream_nk<EPI, OutT, 4>(A, W, bias, out, M, N, st);
    case 8: return launch_stream_nk<EPI, OutT, 8>(A, W, bias, out, M, N, st);
    case 16: return launch_stream_nk<EPI, OutT, 16>(A, W, bias, out, M, N, st);
    default: return -1;
  }
}

}  // namespace

// out[M,N] = epi(A W^T + bias); epi in {EPI_BIAS, EPI_GELU}; out bf16 or fp32.
int d3dp_launch_linear_bf16_stream(int epi, int out_f32, const void* A, const void* W, const float* bias, void* out,
                                   int M, int N, int K, hipStream_t st) {
  if (K % SBK != 0 || N % 4 != 0 || N > SBIAS_MAX || M <= 0) return -1;
  if (epi == EPI_BIAS && out_f32) return launch_stream<EPI_BIAS, float>(A, W, bias, out, M, N, K, st);
  if (epi == EPI_BIAS && !out_f32) return launch_stream<EPI_BIAS, bf16>(A, W, bias, out, M, N, K, st);
  if (epi == EPI_GELU && !out_f32) return launch_stream<EPI_GELU, bf16>(A, W, bias, out, M, N, K, st);
  return -1;
}
