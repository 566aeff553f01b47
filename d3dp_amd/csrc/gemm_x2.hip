// EXACT-mode Linear layers on the fp16 matrix cores:  out[M,N] = epi((A . W^T) + bias), fp32-class accuracy.
// (reference: every nn.Linear on the path -- common/mixste.py:65 qkv, :80 proj, :38-41 fc1/fc2 -- which the reference
//  evaluates in fp32.)
//
// Operands are "f16x2" pairs (common.h): x ~ hi + lo_s 2^-11, two fp16 planes each, A2 [2][M][K], W2 [2][N][K]
// (W additionally multiplied by a per-matrix power of two before splitting so its hi plane sits in the middle of the
// fp16 range; `w_unscale` undoes it).  Per output element three v_mfma_f32_16x16x32_f16 passes,
//     hh += Wh.Ah           xx += Wl.Ah + Wh.Al           out = (hh + xx 2^-11) w_unscale + bias,
// with the cross terms in their OWN fp32 accumulator: its rounding noise is scaled by 2^-11 on the way out, so only
// the K/32 accumulations of the hh chain round at full magnitude (the six-pass split-bf16 kernel this replaces rounded
// 6 K/32 times into one accumulator).  The lo.lo pass is below fp32 resolution and is dropped
// (tools/err_budget_split.py; tests/test_hip_parity.py::test_linear_split_f16_is_fp32_class).
//
// Structure: 128x128x32 block tile, 4 waves (2x2, 64x64 each = 4x4 MFMA tiles, 48 MFMA + 16 ds_read_b128 per k-step),
// two workgroups per CU (2 x 72 KiB LDS, 2 x 4 waves at <= 256 registers): one wave of each workgroup per SIMD, so one
// workgroup's tile-end work (GELU + re-split, 64 KiB of stores, next prologue) runs beside the other's MFMA stream.
// Operand slabs go HBM/L2 -> LDS with global_load_lds_dwordx4 into a 2-stage ring (lane-linear 1 KiB pieces of 16 rows
// x 64 B; bank swizzle on the per-lane SOURCE address and again on the fragment read).
// The wave computes the TRANSPOSED product (weight fragment = MFMA A operand) and the loader permutes W rows inside
// each 64-column strip, so a lane ends up with 16 CONSECUTIVE output columns of one token row: 64-byte runs per lane,
// 256-byte runs per row and store instruction group.
//   EPI_BIAS -> fp32 out (feeds attention / the residual-adding row kernels)
//   EPI_GELU -> gelu_erf(.) re-split into two fp16 planes (the fc2 operand)
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int XBM = 128, XBN = 128, XBK = 32;
constexpr int XA_PLANE = XBM * XBK * 2;              // 8 KiB
constexpr int XW_PLANE = XBN * XBK * 2;              // 8 KiB
constexpr int XSTAGE = 2 * XA_PLANE + 2 * XW_PLANE;  // 32 KiB
constexpr int XNSTAGE = 2;
constexpr int XLDS = XNSTAGE * XSTAGE + XBN * 4;      // 64.5 KiB: two workgroups per CU

// 64-byte rows (4 slots of 16 B): XOR bit 1 of the slot with bit 3 of the row -> conflict-free ds_read_b128 fragments
__device__ __forceinline__ int swz64(int row, int s) { return s ^ (((row >> 3) & 1) << 1); }

// W row carried by LDS row q of a 64-column strip: MFMA tile ni = q>>4, operand row i = q&15 -> output column
// (i>>2)*16 + ni*4 + (i&3): lane group fg = i>>2 then owns columns fg*16 .. fg*16+15 in (ni, r) order.
__device__ __forceinline__ int colperm(int q) {
  const int ni = q >> 4, i = q & 15;
  return (i >> 2) * 16 + ni * 4 + (i & 3);
}

// Tile epilogue shared by both kernels.  A lane holds, for each of its 4 row blocks, out[m = mbase + mi*16 + fi]
// [n = n0 + nl + ni*4 + r]: 16 CONSECUTIVE columns (the loader's W-row permutation), i.e. 64-byte runs per lane.
template <int EPI>
__device__ __forceinline__ void x2_epilogue(const f32x4 (&hh)[4][4], const f32x4 (&xx)[4][4], const float* sbias_tile,
                                            float w_unscale, float* __restrict__ outf, f16* __restrict__ out2, int M, int N,
                                            int mbase, int n0, int nl, int fi) {
  if (n0 + nl >= N) return;
  const size_t planeO = (size_t)M * N;
  float bz[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 b = *reinterpret_cast<const float4*>(sbias_tile + nl + q * 4);
    bz[q * 4] = b.x; bz[q * 4 + 1] = b.y; bz[q * 4 + 2] = b.z; bz[q * 4 + 3] = b.w;
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = mbase + mi * 16 + fi;
    float v[16];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        v[ni * 4 + r] = fmaf(fmaf(xx[mi][ni][r], kLoInv, hh[mi][ni][r]), w_unscale, bz[ni * 4 + r]);
    if constexpr (EPI == EPI_GELU) {
      f16x8 ph[2], pl[2];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        f16 a, b;
        split2h(gelu_erf(v[e]), a, b);
        ph[e >> 3][e & 7] = a; pl[e >> 3][e & 7] = b;
      }
      if (m < M) {
        f16* o = out2 + (size_t)m * N + n0 + nl;
        *reinterpret_cast<f16x8*>(o) = ph[0];
        *reinterpret_cast<f16x8*>(o + 8) = ph[1];
        *reinterpret_cast<f16x8*>(o + planeO) = pl[0];
        *reinterpret_cast<f16x8*>(o + planeO + 8) = pl[1];
      }
    } else {
      if (m < M) {
        float* o = outf + (size_t)m * N + n0 + nl;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(o + q * 4) = make_float4(v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]);
      }
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f16x2_kernel(const f16* __restrict__ A2, const f16* __restrict__ W2,
                                                            const float* __restrict__ bias, float w_unscale,
                                                            float* __restrict__ outf, f16* __restrict__ out2, int M,
                                                            int N, int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sbias = reinterpret_cast<float*>(smem + XNSTAGE * XSTAGE);
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (L / tiles_n) * XBM;
  const int n0 = (L % tiles_n) * XBN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const size_t planeA = (size_t)M * K, planeW = (size_t)N * K;

  for (int i = tid; i < XBN; i += 256) sbias[i] = (n0 + i < N) ? bias[n0 + i] : 0.f;

  // 32 pieces of 1 KiB (16 rows x 64 B) per stage: {A, W} x plane x 8 row groups; wave w issues row groups w and w+4
  // of every (operand, plane).
  const f16* src[8];
  int dst[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = lane >> 2, ps = lane & 3;
    const int pl = (i >> 1) & 1, rg = wave + 4 * (i & 1), row = rg * 16 + r;
    if (i < 4) {
      src[i] = A2 + pl * planeA + (size_t)min(m0 + row, M - 1) * K + swz64(row, ps) * 8;
      dst[i] = pl * XA_PLANE + rg * 1024;
    } else {
      const int wrow = (row & 64) + colperm(row & 63);
      src[i] = W2 + pl * planeW + (size_t)min(n0 + wrow, N - 1) * K + swz64(row, ps) * 8;
      dst[i] = 2 * XA_PLANE + pl * XW_PLANE + rg * 1024;
    }
  }
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * XSTAGE;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds(GPTR(src[i] + kt * XBK), LPTR(base + dst[i]), 16, 0, 0);
  };

  f32x4 hh[4][4], xx[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { hh[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; xx[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  const int nk = K / XBK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int fi = lane & 15, fg = lane >> 4;
  // per-lane fragment offsets inside a stage (the swizzle depends on the lane only: rows advance in multiples of 16)
  const int offA = (wr * 64 + fi) * 64 + swz64(fi, fg) * 16;
  const int offW = 2 * XA_PLANE + (wc * 64 + fi) * 64 + swz64(fi, fg) * 16;
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* sb = smem + cur * XSTAGE;
    f16x8 wf[4][2];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        wf[ni][pl] = *reinterpret_cast<const f16x8*>(sb + offW + pl * XW_PLANE + ni * 1024);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const f16x8 ah = *reinterpret_cast<const f16x8*>(sb + offA + mi * 1024);
      const f16x8 al = *reinterpret_cast<const f16x8*>(sb + offA + XA_PLANE + mi * 1024);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) xx[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni][1], ah, xx[mi][ni], 0, 0, 0);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) xx[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni][0], al, xx[mi][ni], 0, 0, 0);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) hh[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni][0], ah, hh[mi][ni], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  x2_epilogue<EPI>(hh, xx, sbias, w_unscale, outf, out2, M, N, m0 + wr * 64, n0, wc * 64 + (lane >> 4) * 16, lane & 15);
}

// ------------------------------------------------------------------------------------------------------------------
// Persistent streaming form: 256x128x32 tiles, 8 waves (4 x 2, 64x64 each), one workgroup per CU walking tiles
// L, L+G, ...; operand slabs stream through a 3-stage LDS ring (48 KiB per stage: two A planes of 256 rows, two W
// planes of 128 rows) as ONE continuous sequence of k-steps across tile boundaries, so a tile's first slabs are in
// flight while the previous tile's epilogue runs.  Every wave loads (6 one-KiB pieces per k-step) and computes
// (48 MFMA + 16 ds_read_b128 per k-step); one raw s_barrier per k-step; counted vmcnt keeps one k-step in flight
// across every barrier.
// Epilogue stores and vmcnt: stores count in vmcnt too, and loads/stores retire out of order with respect to each
// other, so the tile epilogue first drains its (long landed) loads with vmcnt(0), then fires the stores and does not
// wait for them; the two k-steps that follow need no wait (their slabs landed before the stores), and from the third
// on `vmcnt(6)` is exact again -- it can only be satisfied once every load older than the newest six has retired,
// whatever the stores do -- by which time the stores have had two k-steps (~1.5 us) to drain.
// ------------------------------------------------------------------------------------------------------------------
constexpr int SBM2 = 256;
constexpr int SA_PLANE = SBM2 * XBK * 2;             // 16 KiB
constexpr int SSTAGE2 = 2 * SA_PLANE + 2 * XW_PLANE;  // 48 KiB
constexpr int SNST = 3;
constexpr int SBIAS2 = 2048;
constexpr int SLDS2 = SNST * SSTAGE2 + SBIAS2 * 4;   // 152 KiB

template <int EPI>
__global__ __launch_bounds__(512) void gemm_f16x2_stream_kernel(const f16* __restrict__ A2, const f16* __restrict__ W2,
                                                                const float* __restrict__ bias, float w_unscale,
                                                                float* __restrict__ outf, f16* __restrict__ out2, int M,
                                                                int N, int K, int tiles_n, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sbias = reinterpret_cast<float*>(smem + SNST * SSTAGE2);
  const int G = gridDim.x;
  const int L = xcd_remap(blockIdx.x, G);
  const int n_my = (total_tiles - L + G - 1) / G;      // tiles L, L+G, ...
  const int NK = K / XBK;
  const int gtot = n_my * NK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const size_t planeA = (size_t)M * K, planeW = (size_t)N * K;

  for (int i = tid; i < N; i += 512) sbias[i] = bias[i];

  // 48 pieces of 1 KiB (16 rows x 64 B) per stage: A plane p row group j (32 pieces), W plane p row group j (16);
  // wave w issues A row groups 2w, 2w+1 of both planes and W row group w of both planes.
  const int lr = lane >> 2, lps = lane & 3;
  int lti = 0, lks = 0;                                // (tile, k-step) of the NEXT slab to issue
  auto issue = [&](int slot) {
    const int t = L + lti * G;
    const int m0 = (t / tiles_n) * SBM2, n0 = (t % tiles_n) * XBN;
    char* base = smem + slot * SSTAGE2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rg = wave * 2 + i, row = rg * 16 + lr;
      const f16* src = A2 + (size_t)min(m0 + row, M - 1) * K + lks * XBK + swz64(row, lps) * 8;
      __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(base + rg * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GPTR(src + planeA), LPTR(base + SA_PLANE + rg * 1024), 16, 0, 0);
    }
    {
      const int row = wave * 16 + lr;
      const int wrow = (row & 64) + colperm(row & 63);
      const f16* src = W2 + (size_t)min(n0 + wrow, N - 1) * K + lks * XBK + swz64(row, lps) * 8;
      __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(base + 2 * SA_PLANE + wave * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GPTR(src + planeW), LPTR(base + 2 * SA_PLANE + XW_PLANE + wave * 1024), 16, 0, 0);
    }
    if (++lks == NK) { lks = 0; ++lti; }
  };

  const int fi = lane & 15, fg = lane >> 4;
  const int offA = (wr * 64 + fi) * 64 + swz64(fi, fg) * 16;
  const int offW = 2 * SA_PLANE + (wc * 64 + fi) * 64 + swz64(fi, fg) * 16;
  f32x4 hh[4][4], xx[4][4];

  if (gtot > 0) issue(0);
  if (gtot > 1) issue(1);
  __syncthreads();                                     // bias visible (plain LDS stores: lgkmcnt only)
  int g = 0, slot = 0;
  for (int ti = 0; ti < n_my; ++ti) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { hh[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; xx[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
    for (int ks = 0; ks < NK; ++ks, ++g) {
      if (ti == 0 || ks >= 2) {                        // (the two k-steps after an epilogue were drained before its stores)
        if (g + 1 < gtot) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (g + 2 < gtot) issue(slot == 0 ? 2 : slot - 1);           // slot of k-step g+2 == slot of g-1: free since this barrier
      const char* sb = smem + slot * SSTAGE2;
      f16x8 wf[4][2];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          wf[ni][pl] = *reinterpret_cast<const f16x8*>(sb + offW + pl * XW_PLANE + ni * 1024);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const f16x8 ah = *reinterpret_cast<const f16x8*>(sb + offA + mi * 1024);
        const f16x8 al = *reinterpret_cast<const f16x8*>(sb + offA + SA_PLANE + mi * 1024);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) xx[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni][1], ah, xx[mi][ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) xx[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni][0], al, xx[mi][ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) hh[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni][0], ah, hh[mi][ni], 0, 0, 0);
      }
      slot = (slot == 2) ? 0 : slot + 1;
    }
    const int t = L + ti * G;
    const int pm0 = (t / tiles_n) * SBM2, pn0 = (t % tiles_n) * XBN;
    if (ti + 1 < n_my) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile's first two slabs: landed long ago
    x2_epilogue<EPI>(hh, xx, sbias + pn0, w_unscale, outf, out2, M, N, pm0 + wr * 64, pn0, wc * 64 + fg * 16, fi);
  }
}

// dst[0][i] = hi, dst[1][i] = lo_s of src[i] * scale
__global__ void split2h_kernel(const float* __restrict__ s, f16* __restrict__ d, size_t n, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    f16 a, b;
    split2h(s[i] * scale, a, b);
    d[i] = a; d[i + n] = b;
  }
}

// out[0] = max |src[i]| (as the bit pattern of a non-negative float: integer max == float max); out pre-zeroed
__global__ void absmax_kernel(const float* __restrict__ s, size_t n, unsigned* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float m = 0.f;
  for (; i < n; i += stride) m = fmaxf(m, fabsf(s[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

}  // namespace

// out = epi((A W^T) w_unscale + bias) with A2/W2 split-fp16 planes; EPI_BIAS: fp32 `outf`; EPI_GELU: two fp16 planes `out2`.
int d3dp_launch_linear_f16x2_stream(int epi, const void* A2, const void* W2, const float* bias, float w_unscale,
                                    float* outf, void* out2, int M, int N, int K, hipStream_t st) {
  if (K % XBK != 0 || N % 16 != 0 || N > SBIAS2 || M <= 0) return -1;
  const int tm = (M + SBM2 - 1) / SBM2, tn = (N + XBN - 1) / XBN;
  static bool attr_set = false;
  static int n_cu = 0;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16x2_stream_kernel<EPI_BIAS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, SLDS2) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16x2_stream_kernel<EPI_GELU>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, SLDS2) != hipSuccess) return -3;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -3;
    n_cu = prop.multiProcessorCount;
    attr_set = true;
  }
  const int total = tm * tn, grid = total < n_cu ? total : n_cu;
  if (epi == EPI_BIAS)
    hipLaunchKernelGGL((gemm_f16x2_stream_kernel<EPI_BIAS>), dim3(grid), dim3(512), SLDS2, st, (const f16*)A2,
                       (const f16*)W2, bias, w_unscale, outf, (f16*)out2, M, N, K, tn, total);
  else if (epi == EPI_GELU)
    hipLaunchKernelGGL((gemm_f16x2_stream_kernel<EPI_GELU>), dim3(grid), dim3(512), SLDS2, st, (const f16*)A2,
                       (const f16*)W2, bias, w_unscale, outf, (f16*)out2, M, N, K, tn, total);
  else return -1;
  return 0;
}

int d3dp_launch_linear_f16x2(int epi, const void* A2, const void* W2, const float* bias, float w_unscale, float* outf,
                             void* out2, int M, int N, int K, hipStream_t st) {
  if (K % XBK != 0 || N % 16 != 0 || M <= 0) return -1;
  const int tm = (M + XBM - 1) / XBM, tn = (N + XBN - 1) / XBN;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16x2_kernel<EPI_BIAS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, XLDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16x2_kernel<EPI_GELU>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, XLDS) != hipSuccess) return -3;
    attr_set = true;
  }
  if (epi == EPI_BIAS)
    hipLaunchKernelGGL((gemm_f16x2_kernel<EPI_BIAS>), dim3(tm * tn), dim3(256), XLDS, st, (const f16*)A2, (const f16*)W2,
                       bias, w_unscale, outf, (f16*)out2, M, N, K, tn);
  else if (epi == EPI_GELU)
    hipLaunchKernelGGL((gemm_f16x2_kernel<EPI_GELU>), dim3(tm * tn), dim3(256), XLDS, st, (const f16*)A2, (const f16*)W2,
                       bias, w_unscale, outf, (f16*)out2, M, N, K, tn);
  else return -1;
  return 0;
}

void d3dp_launch_split2(const float* src, void* dst, size_t n, float scale, hipStream_t st) {
  const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(split2h_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, (f16*)dst, n, scale);
}

void d3dp_launch_absmax(const float* src, size_t n, unsigned* out, hipStream_t st) {
  const unsigned blocks = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, n, out);
}
