// EXACT-mode Linear layers on the fp16 matrix cores:  out[M,N] = epi((A . W^T) + bias), fp32-class accuracy.
// (reference: every nn.Linear on the path -- common/mixste.py:65 qkv, :80 proj, :38-41 fc1/fc2 -- which the reference
//  evaluates in fp32.)
//
// Operands are "f16x2" pairs (common.h): x 2^s = hi + lo, two fp16 values per element, A2 [M][K] (s = 4, fixed) and
// W2 [N][K] (s per matrix: max |w| 2^s in [2^13, 2^14)) in the LINE-INTERLEAVED layout "h2i" (common.h): a row is K/32
// blocks of 128 bytes, block kb = [hi of columns 32 kb .. 32 kb + 31 | lo of the same columns], so ONE k-step of ONE row --
// both planes -- is ONE 128-byte line.  `unscale` = 2^-(s_a + s_w) undoes both scales.  Per output element
// three v_mfma_f32_16x16x32_f16 passes into one fp32 accumulator,
//     acc += Wl.Ah + Wh.Al + Wh.Ah            out = acc unscale + bias,
// the lo.lo pass being below fp32 resolution (tools/err_budget_split.py;
// tests/test_hip_parity.py::test_linear_split_f16_is_fp32_class: mean error below torch's own fp32 matmul).
//
// Structure (the FAST kernel's, gemm.hip): persistent, 256x128 output tile per workgroup pass, BK = 32, 8 compute waves
// (4 x 2, 64x64 each = 4x4 MFMA tiles: 48 MFMA + 16 ds_read_b128 per k-step) + 4 LOADER waves.  Workgroups (<= one per
// CU) walk tiles L, L+G, ...; the A and W slabs (hi | lo of 32 columns per row: 48 KiB per k-step) stream through a 3-stage LDS ring
// as one continuous sequence of k-steps across tile boundaries.  Only the loader waves issue global_load_lds
// (1 KiB pieces of 8 rows x 128 B = 8 whole lines, 12 per wave and k-step; round 2 had two separate planes and pieces of
// 16 rows x 64 B = 16 half lines: the kernel is bound by the line-request rate of the CU's vector memory path,
// profiles/r03_gemm_l2_prefetch.md, and the interleaved layout halves the requests for the same bytes) and wait on
// vmcnt (counted: one k-step stays in flight across
// every barrier); measured on the two earlier structures of this kernel in which every wave loaded AND computed
// (profiles/r02_gemm_x2_structures_pmc.md): half of all wave cycles were issue stalls with the matrix pipe a third
// busy -- an LDS-DMA costs 100-185 issue cycles beside ds_reads -- so the compute waves must not carry them, and the
// 168-register budget of a 12-wave workgroup is why there is ONE accumulator set (lo planes unscaled) instead of a
// separate accumulator for the cross terms.  One raw s_barrier per k-step.  Compute waves never wait on vmcnt, so
// their epilogue stores (which also count in vmcnt) cannot stall the operand pipeline; the loaders keep prefetching
// the next tile's slabs while the epilogue runs.
// Output stores: the activation fragment is the MFMA A operand, so a lane holds C[row 4 fg + r][column fi] of each 16x16
// tile, and the loader permutes the W rows of a 64-column strip (LDS row ni*16 + i carries column 4 i + ni) so that the
// lane's four tiles ni are four CONSECUTIVE columns: one 16-byte store per (mi, r) in which the 16 lanes of a group cover
// 256 contiguous bytes of ONE row -- every store instruction writes whole 128-byte lines.  (First version: transposed
// product, 16 columns per lane, each store instruction touching 64 different lines with 16 bytes: the tile-end store
// tail cost 11-17 us per 256x128 tile against 18-22 us for its 16 k-steps, fitted over the four Linear shapes.)
//   EPI_BIAS     -> fp32 out (feeds the residual-adding row kernels)
//   EPI_GELU     -> GELU (rational erf, common.h) re-split into the h2i rows of the fc2 operand
//   EPI_QKV_PACK -> q fp32, k and v as fp16 planes: the packed rows the split-fp16 attention kernels read (attention.hip)
//   EPI_RESID    -> x += A W^T + b in place on the fp32 residual stream (proj, fc2: mixste.py:113-115)
//   EPI_RESID_LN -> the same, and the sum leaves a second time as the NEXT Linear's split-fp16 operand (un-normalised) together
//                   with (mean, M2) of each 64-column slice of each row: the statistics of the LayerNorm that follows
//   EPI_GELU_LN  -> EPI_GELU of a Linear with that LayerNorm folded in: LN(x) W^T + b = rstd (x W'^T - mean c1) + c2 with
//                   W' = W diag(gamma), c1 = W' 1, c2 = W beta + b -- norm2 + fc1 (mixste.py:115) without a row kernel in between
// Plane outputs leave as ONE 16-byte store per lane too (neighbouring lanes swap halves, store_planes_paired).
// Timing probes of this kernel (loads / stores / MFMAs / barriers compiled out one at a time) and what they say about
// the clock the chip sustains under this instruction mix: profiles/r02_gemm_probes.md, profiles/r03_gemm_probes.md.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.h"
#include "kernels.h"

// Timing probes (results INVALID), compiled in only with -DD3DP_X2_PROBE=bits for tools/gemm_bench.py A/B builds; the
// product library is built without it.  1: loaders issue no loads; 2: fragments read from one fixed stage (hoisted out of
// the k-loop); 4: no output stores (the epilogue sits behind a condition that is false at run time, so the accumulators
// and with them the MFMAs stay live -- compiling the epilogue out removes every MFMA); 8: no MFMAs (fragment reads kept
// live); 16: odd workgroups start half a tile late (D3DP_X2_STAGGER s_sleep(127) periods).
#ifndef D3DP_X2_PROBE
#define D3DP_X2_PROBE 0
#endif
// -DD3DP_X2_VARIANTS=1 (tools/build_variant.sh variants ...; `make variants`): also build the MEASURED-NEGATIVE forms of this
// Linear -- the ping-pong, wide (256 x 256) and row-class skewed kernels and the norm2-folding epilogues (EPI_RESID_LN /
// EPI_GELU_LN), DESIGN.md section 7 -- behind their D3DP_X2_PP / D3DP_X2_WIDE / D3DP_X2_SKEW / D3DP_FOLD_LN switches.  The
// product library is built without them: asking it for one fails loudly (d3dp_create: D3DP_ENOTSUP).
#ifndef D3DP_X2_VARIANTS
#define D3DP_X2_VARIANTS 0
#endif

// Output stores carry the nontemporal hint: the tile round's 4 MiB of output per XCD is not read again on that XCD
// and otherwise pushes the weight matrix out of the 4 MiB L2 between rounds (FETCH_SIZE measurement in DESIGN.md).
#if D3DP_X2_PROBE & 32                                  // 32: no k-step barriers (with 1|4: the loop without its synchronisation)
#define X2_BARRIER() asm volatile("" ::: "memory")
#else
#define X2_BARRIER() asm volatile("s_barrier" ::: "memory")
#endif
// GELU of the fc1 epilogue two elements at a time on v_pk_fma_f32 (the epilogue runs with no MFMA in flight): -45 ms
// of the 1387 ms fc1 class per step
#ifndef D3DP_X2_PKGELU
#define D3DP_X2_PKGELU 1
#endif
#ifndef D3DP_X2_LAG
#define D3DP_X2_LAG 1
#endif
// cache-policy bits of the loaders' LDS-DMA (aux: 1 = sc0, 2 = nt, 16 = sc1), for A/B builds
#ifndef D3DP_X2_AAUX
#define D3DP_X2_AAUX 0
#endif
#ifndef D3DP_X2_WAUX
#define D3DP_X2_WAUX 0
#endif
#ifndef D3DP_NT_OUT
#define D3DP_NT_OUT 1
#endif
// skewed kernel: VALU per MFMA requested from the scheduler (sched_group_barrier) inside a row block's region; 0 = leave
// the order inside a region to the compiler
#ifndef D3DP_X2_SKEW_SGB
#define D3DP_X2_SKEW_SGB 0
#endif
// where the leaving rows' VALU work and stores go inside a k-step: 1 = in ONE packet right behind the k-step's fragment
// reads, in front of its first MFMA -- the window in which every wave of the workgroup waits for LDS after the barrier
// anyway (measured: interleaved with the MFMAs the same work cost as much as an exposed epilogue, gpurun c2); 0 = beside
// the MFMAs of the four row blocks
#ifndef D3DP_X2_SKEW_TOP
#define D3DP_X2_SKEW_TOP 1
#endif
// timing probes of the skewed kernel (results INVALID): 1 = the leaving rows' VALU work and stores sit behind a condition
// that is false at run time (the schedule alone); 2 = no accumulator shift at a park; 4 = the VALU work runs, the stores do
// not (kept live through an empty asm); 8 = the stores run on the raw parked values, no epilogue arithmetic
#ifndef D3DP_X2_SKEW_PROBE
#define D3DP_X2_SKEW_PROBE 0
#endif
#if D3DP_NT_OUT
#define OUT_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define OUT_STORE(ptr, val) (*(ptr) = (val))
#endif

namespace {

constexpr int XBM = 256, XBN = 128, XBK = 32;
constexpr int XA_BYTES = XBM * 128;                  // 32 KiB: 256 rows x (64 B hi | 64 B lo)
constexpr int XW_BYTES = XBN * 128;                  // 16 KiB
constexpr int XSTAGE = XA_BYTES + XW_BYTES;          // 48 KiB
constexpr int XNSTAGE = 3;
constexpr int XBIAS_MAX = 2048;                      // floats of bias kept in LDS
constexpr int XROWSTAT = XNSTAGE * XSTAGE + XBIAS_MAX * 4;   // EPI_GELU_LN: 2 x [256 rows][mean, rstd] (tile t in buffer t & 1)
constexpr int XLDS = XROWSTAT + 2 * XBM * 8;             // 156 KiB
constexpr int XNCW = 8;                              // compute waves (4 x 2); waves 8..11 are loaders

// LDS image of a slab: 128-byte rows = 8 slots of 16 B (q = 4 plane + k-group of 8 columns); slot q of row r lives at
// physical slot q ^ ((r >> 1) & 7).  A fragment read (lane: row fi, k-group fg, one plane) is then conflict-free: the four
// 16-lane groups of ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32) each cover 8 rows of one
// k-group and 8 rows of the next, and (row & 1) 8 + (q ^ (row >> 1 & 7)) takes 16 different values on them.
__device__ __forceinline__ int swz128(int row, int q) { return q ^ ((row >> 1) & 7); }

// W row carried by LDS row q of a 64-column strip: MFMA tile ni = q>>4, operand row i = q&15 -> output column 4 i + ni
__device__ __forceinline__ int colperm(int q) { return (q & 15) * 4 + (q >> 4); }

// Two fp16 planes of a 4-column group per lane -> ONE 16-byte store per lane: lanes 2j / 2j+1 hold neighbouring column
// groups of the same row; the even lane collects both hi halves (8 columns of the hi plane), the odd lane both lo halves.
// `dst`: where this lane's 16 bytes go (even lanes: own columns in the hi plane; odd lanes: the even neighbour's columns
// in the lo plane).
__device__ __forceinline__ void store_planes_paired(char* dst, f16x4 ph, f16x4 pl, bool odd, bool live) {
  const uint2 h = __builtin_bit_cast(uint2, ph), l = __builtin_bit_cast(uint2, pl);
  const uint2 send = odd ? h : l;
  uint2 recv;                                          // quad_perm [1,0,3,2]: the value of lane ^ 1
  recv.x = __builtin_amdgcn_update_dpp(0, (int)send.x, 0xB1, 0xF, 0xF, false);
  recv.y = __builtin_amdgcn_update_dpp(0, (int)send.y, 0xB1, 0xF, 0xF, false);
  using u32x4 = unsigned __attribute__((ext_vector_type(4)));
  const u32x4 out = odd ? (u32x4){recv.x, recv.y, l.x, l.y} : (u32x4){h.x, h.y, recv.x, recv.y};
  if (live) OUT_STORE(reinterpret_cast<u32x4*>(dst), out);
}

// TAG 1: the qkv Linear feeding the split-fp16 attention kernels -- packed output rows (see the epilogue); as its own
// kernel symbol rocprofv3 --stats also reports it separately from the proj Linear, which shares EPI with it.
template <int EPI, int TAG>
__global__ __launch_bounds__(768) void gemm_f16x2_kernel(const f16* __restrict__ A2, const f16* __restrict__ W2,
                                                         const float* __restrict__ bias, float unscale, float oscale,
                                                         float* __restrict__ outf, f16* __restrict__ out2,
                                                         float* __restrict__ aux, unsigned* __restrict__ flag, int M,
                                                         int N, int K, int tiles_n_arg, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sbias = reinterpret_cast<float*>(smem + XNSTAGE * XSTAGE);
  const int G = gridDim.x;
  const int L = xcd_remap(blockIdx.x, G);
  const int n_my = (total_tiles - L + G - 1) / G;      // tiles L, L+G, ...
  const int NK = K / XBK;
  const int gtot = n_my * NK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // tiles_n_arg < 0: the tiles from the LAST row of tiles to the first (X2_TILES_LAST_TO_FIRST: proj, whose consumer -- the
  // norm2 row kernel -- then starts on the rows written last)
  const int tiles_n = tiles_n_arg < 0 ? -tiles_n_arg : tiles_n_arg;
  const int t_flip = tiles_n_arg < 0 ? total_tiles - 1 : 0, t_sign = tiles_n_arg < 0 ? -1 : 1;
  auto tile_of = [&](int ti_) { return t_flip + t_sign * (L + ti_ * G); };

  for (int i = tid; i < (EPI == EPI_GELU_LN ? 2 * N : N); i += (XNCW + 4) * 64) sbias[i] = bias[i];   // (GELU_LN: c2 | c1)
  __syncthreads();

  if (wave >= XNCW) {
    // ------------------------------------------------------------------ loader waves
    const int lw = wave - XNCW;
    const int lr = lane >> 3, lq = lane & 7;           // row within an 8-row piece, physical 16-byte slot
#ifdef D3DP_X2_LPRIO
    __builtin_amdgcn_s_setprio(D3DP_X2_LPRIO);
#endif
    int ti = 0, ks = 0, slot = 0;                      // (tile, k-step, ring slot) of the next slab to issue
    const f16* pa[8];                                  // this lane's source slots of the current tile (k-step 0)
    const f16* pw[4];
    auto issue = [&]() {
      if (ks == 0) {                                   // new tile: row pointers once per tile, not per k-step
        const int t = tile_of(ti);
        const int m0 = (t / tiles_n) * XBM, n0 = (t % tiles_n) * XBN;
#pragma unroll
        for (int i = 0; i < 8; ++i) {                  // A pieces 8 lw .. 8 lw + 7 (8 rows each)
          const int row = (lw * 8 + i) * 8 + lr;
          pa[i] = A2 + (size_t)min(m0 + row, M - 1) * (2 * K) + swz128(row, lq) * 8;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                  // W pieces 4 lw .. 4 lw + 3
          const int row = (lw * 4 + i) * 8 + lr;                        // LDS row of the W slab
          const int wrow = (row & 64) + colperm(row & 63);              // output column it carries
          pw[i] = W2 + (size_t)min(n0 + wrow, N - 1) * (2 * K) + swz128(row, lq) * 8;
        }
      }
      if constexpr (EPI == EPI_GELU_LN) {
        // the tile's 256 (mean, rstd) pairs: 2 KiB = two pieces, by loader wave 0, the OLDEST operations of this k-step
        // (every counted wait below then covers them); buffer ti & 1 -- the epilogue of tile ti - 1 may still be reading
        if (ks == 0 && lw == 0) {
          const int t = tile_of(ti);
          const float* rs = aux + (size_t)(t / tiles_n) * XBM * 2 + lane * 4;     // (aux has 256 rows of slack: no clamp)
          char* dst = smem + XROWSTAT + (ti & 1) * (XBM * 8);
          __builtin_amdgcn_global_load_lds(GPTR(rs), LPTR(dst), 16, 0, 0);
          __builtin_amdgcn_global_load_lds(GPTR(rs + 256), LPTR(dst + 1024), 16, 0, 0);
        }
      }
      char* base = smem + slot * XSTAGE;
      const int ko = ks * (2 * XBK);                   // one k-step of a row = 64 fp16 (hi | lo)
#if D3DP_X2_PROBE & 1
      if (ko >= 0) { if (++ks == NK) { ks = 0; ++ti; } slot = (slot == XNSTAGE - 1) ? 0 : slot + 1; return; }
#endif
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds(GPTR(pa[i] + ko), LPTR(base + (lw * 8 + i) * 1024), 16, 0, D3DP_X2_AAUX);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds(GPTR(pw[i] + ko), LPTR(base + XA_BYTES + (lw * 4 + i) * 1024), 16, 0, D3DP_X2_WAUX);
      if (++ks == NK) { ks = 0; ++ti; }
      slot = (slot == XNSTAGE - 1) ? 0 : slot + 1;
    };
    if (gtot > 0) issue();
    if (gtot > 1) issue();
    for (int g = 0; g < gtot; ++g) {
      // 12 glds per loader wave per k-step: loads(g) landed, loads(g+1) stay in flight
      if (g + 1 < gtot) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      X2_BARRIER();
      if (g + 2 < gtot) issue();                       // into the slot of k-step g-1: every wave has passed barrier g
    }
    return;
  }

  // -------------------------------------------------------------------- compute waves
  const int wr = wave >> 1, wc = wave & 1;
  const int fi = lane & 15, fg = lane >> 4;
  // per-lane fragment offsets inside a stage (the swizzle depends on the lane only: rows advance in multiples of 16);
  // the lo plane's slot is the hi plane's ^ 4, i.e. byte offset ^ 64
  const int offA = (wr * 64 + fi) * 128 + swz128(fi, fg) * 16, offAl = offA ^ 64;
  const int offW = XA_BYTES + (wc * 64 + fi) * 128 + swz128(fi, fg) * 16, offWl = offW ^ 64;
  f32x4 acc[4][4];
  __builtin_amdgcn_s_setprio(1);
#if D3DP_X2_PROBE & 16
  if (L & 1)
    for (int i = 0; i < D3DP_X2_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
  int slot = 0;
  for (int ti = 0; ti < n_my; ++ti) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#if D3DP_X2_LAG
    // The last two products of row block 3 (al.wh, ah.wh: 8 MFMAs) are issued AFTER the next k-step's barrier, behind
    // that step's first fragment reads: they cover the LDS latency that otherwise idles the matrix pipe after every
    // barrier release (all waves of the workgroup read at once).  Their operands stay in 24 registers across the
    // barrier; the k-loop is unrolled by two so that the two fragment sets swap roles without register copies.
    f16x8 wfa[4][2], wfb[4][2], taa[2], tab[2];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) wfb[ni][0] = wfb[ni][1] = (f16x8){};
    tab[0] = tab[1] = (f16x8){};
    auto kstep = [&](f16x8 (&wf)[4][2], f16x8 (&ta)[2], const f16x8 (&pw)[4][2], const f16x8 (&pa)[2]) {
      __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): this wave has read everything it wanted from the old slot
      X2_BARRIER();
      __builtin_amdgcn_sched_barrier(0);
      const char* sb = smem + slot * XSTAGE;
      slot = (slot == XNSTAGE - 1) ? 0 : slot + 1;
      f16x8 ah[2], al[2];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          wf[ni][pl] = *reinterpret_cast<const f16x8*>(sb + (pl ? offWl : offW) + ni * 2048);
      ah[0] = *reinterpret_cast<const f16x8*>(sb + offA);
      al[0] = *reinterpret_cast<const f16x8*>(sb + offAl);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[3][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa[1], pw[ni][0], acc[3][ni], 0, 0, 0);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[3][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa[0], pw[ni][0], acc[3][ni], 0, 0, 0);
#if D3DP_X2_LAG >= 2                                   // probe: all twelve MFMAs of row block 3 lag (16 more registers)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[3][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa[0], pw[ni][1], acc[3][ni], 0, 0, 0);
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < 3; ++mi) {
        const int b = mi & 1;
        if (mi < 2) {                                  // next row block's fragments while this one multiplies
          ah[b ^ 1] = *reinterpret_cast<const f16x8*>(sb + offA + (mi + 1) * 2048);
          al[b ^ 1] = *reinterpret_cast<const f16x8*>(sb + offAl + (mi + 1) * 2048);
        } else {
          ta[0] = *reinterpret_cast<const f16x8*>(sb + offA + 3 * 2048);
          ta[1] = *reinterpret_cast<const f16x8*>(sb + offAl + 3 * 2048);
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b], wf[ni][1], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[b], wf[ni][0], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b], wf[ni][0], acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#if D3DP_X2_LAG < 2
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[3][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ta[0], wf[ni][1], acc[3][ni], 0, 0, 0);
#endif
    };
#pragma unroll 1
    for (int ks = 0; ks < NK; ks += 2) {               // NK is even (K % 64 == 0, checked by the launcher)
      kstep(wfa, taa, wfb, tab);
      kstep(wfb, tab, wfa, taa);
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[3][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tab[1], wfb[ni][0], acc[3][ni], 0, 0, 0);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[3][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tab[0], wfb[ni][0], acc[3][ni], 0, 0, 0);
#if D3DP_X2_LAG >= 2
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[3][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tab[0], wfb[ni][1], acc[3][ni], 0, 0, 0);
#endif
#else
#pragma unroll 1
    for (int ks = 0; ks < NK; ++ks) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      X2_BARRIER();
#if D3DP_X2_PROBE & 2
      const char* sb = smem;                           // (one fixed stage: reads hoisted out of the k-loop by the compiler)
#else
      const char* sb = smem + slot * XSTAGE;
#endif
      slot = (slot == XNSTAGE - 1) ? 0 : slot + 1;
      f16x8 wf[4][2];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          wf[ni][pl] = *reinterpret_cast<const f16x8*>(sb + (pl ? offWl : offW) + ni * 2048);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const f16x8 ah = *reinterpret_cast<const f16x8*>(sb + offA + mi * 2048);
        const f16x8 al = *reinterpret_cast<const f16x8*>(sb + offAl + mi * 2048);
        // small terms first; the three products of one output tile are 4 MFMAs apart (no back-to-back dependency)
#if D3DP_X2_PROBE & 8
        asm volatile("" :: "v"(ah), "v"(al));
        if (mi == 3)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) asm volatile("" :: "v"(wf[ni][0]), "v"(wf[ni][1]));
#else
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wf[ni][1], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wf[ni][0], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wf[ni][0], acc[mi][ni], 0, 0, 0);
#endif
      }
    }
#endif
    // ---- tile epilogue: lane holds out[m = pm0 + mi*16 + r][n = nb + ni], pm0 = tile row + wr*64 + 4 fg, nb = tile column
    // + wc*64 + 4 fi.  Every store address is  uniform base + 32-bit lane offset  (the launcher refuses outputs of 4 GiB or
    // more), advanced row by row: per-row 64-bit address arithmetic was most of the epilogue's VALU work, and it runs
    // with the matrix pipes idle.
    const int t = tile_of(ti);
    const int pm0 = (t / tiles_n) * XBM + wr * 64 + 4 * fg, nb = (t % tiles_n) * XBN + wc * 64 + 4 * fi;
    if (nb < N && (!(D3DP_X2_PROBE & 4) || unscale == -12345.f)) {
      const float4 bz = *reinterpret_cast<const float4*>(sbias + nb);
      const bool odd = fi & 1;
      unsigned off, pitch;
      bool planes;                                     // split the values and store fp16 planes (else fp32)
      char* base = reinterpret_cast<char*>(outf);
      const int c0 = nb & ~7;                          // h2i rows: 4 N bytes per row; the lane pair's 8 columns start at c0:
      const unsigned offp = (unsigned)pm0 * (N * 4) + (c0 >> 5) * 128 + (c0 & 31) * 2 + (odd ? 64 : 0);   // even lane -> hi slot, odd -> lo
      if constexpr (EPI == EPI_GELU || EPI == EPI_GELU_LN) {   // the fc2 operand
        base = reinterpret_cast<char*>(out2);
        pitch = N * 4; planes = true;
        off = offp - (unsigned)pm0 * pitch;
      } else if constexpr (TAG == 1) {
        // packed qkv row (12 C bytes, C = N / 3): q fp32 | k hi | k lo | v hi | v lo (fp16 planes x 16) -- the
        // K / V operand images of the split-fp16 attention kernels, which copy them into LDS without touching them
        const int C = N / 3, region = nb / C, cn = nb - region * C;      // a wave's 64 columns lie in one region
        pitch = N * 4; planes = region != 0;
        off = planes ? region * 4 * C + cn * 2 + (odd ? 2 * C - 8 : 0) : cn * 4;
      } else {
        pitch = N * 4; planes = false;
        off = nb * 4;
      }
      off += (unsigned)pm0 * pitch;
      const int rows = M - pm0;                        // row k = mi*16 + r of this lane exists iff k < rows
      [[maybe_unused]] float4 c1z = {};
      [[maybe_unused]] const float* srow = nullptr;
      if constexpr (EPI == EPI_GELU_LN) {
        c1z = *reinterpret_cast<const float4*>(sbias + N + nb);
        srow = reinterpret_cast<const float*>(smem + XROWSTAT + (ti & 1) * (XBM * 8)) + (wr * 64 + 4 * fg) * 2;
      }
      auto value = [&](int mi, int r, int e) {
        const float bze = e == 0 ? bz.x : e == 1 ? bz.y : e == 2 ? bz.z : bz.w;
        if constexpr (EPI == EPI_GELU_LN) {            // rstd (x W'^T - mean c1) + c2
          const float2 st = *reinterpret_cast<const float2*>(srow + (mi * 16 + r) * 2);
          const float c1e = e == 0 ? c1z.x : e == 1 ? c1z.y : e == 2 ? c1z.z : c1z.w;
          return fmaf(st.y, fmaf(acc[mi][e][r], unscale, -(st.x * c1e)), bze);
        } else {
          return fmaf(acc[mi][e][r], unscale, bze);
        }
      };
      auto store_rows = [&](auto planes_c, auto checked_c) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k = mi * 16 + r;
            const bool live = !decltype(checked_c)::value || k < rows;
            char* dst = base + (off + (unsigned)k * pitch);
            if constexpr (decltype(planes_c)::value) {
              f16x4 ph, pl;
              float v[4] = {value(mi, r, 0), value(mi, r, 1), value(mi, r, 2), value(mi, r, 3)};
              if constexpr (EPI == EPI_GELU || EPI == EPI_GELU_LN) {
#if D3DP_X2_PKGELU
                const f32x2 g0 = gelu_erf_rational2((f32x2){v[0], v[1]}), g1 = gelu_erf_rational2((f32x2){v[2], v[3]});
                v[0] = g0.x; v[1] = g0.y; v[2] = g1.x; v[3] = g1.y;
#else
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf_rational(v[e]);
#endif
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                f16 h, l;
                split2h_scaled(v[e] * oscale, h, l);       // (oscale: the consumer's operand scale, 2^4 unless capi.hip lowered it)
                ph[e] = h; pl[e] = l;
              }
              store_planes_paired(dst, ph, pl, odd, live);
            } else if constexpr (EPI != EPI_RESID && EPI != EPI_RESID_LN) {
              if (live) OUT_STORE(reinterpret_cast<f32x4*>(dst), ((f32x4){value(mi, r, 0), value(mi, r, 1), value(mi, r, 2), value(mi, r, 3)}));
            }
          }
        if constexpr (EPI == EPI_RESID && !decltype(planes_c)::value) {
          // x += A W^T + b in place (the residual stream): all sixteen reads of the tile in flight before the first add
          f32x4 res[4][4];
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int k = mi * 16 + r;
              const bool live = !decltype(checked_c)::value || k < rows;
              res[mi][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
              if (live) res[mi][r] = *reinterpret_cast<const f32x4*>(base + (off + (unsigned)k * pitch));
            }
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int k = mi * 16 + r;
              const bool live = !decltype(checked_c)::value || k < rows;
              const f32x4 v = {res[mi][r][0] + value(mi, r, 0), res[mi][r][1] + value(mi, r, 1),
                               res[mi][r][2] + value(mi, r, 2), res[mi][r][3] + value(mi, r, 3)};
              if (live) *reinterpret_cast<f32x4*>(base + (off + (unsigned)k * pitch)) = v;   // (re-read by the next row kernel: no nt hint)
            }
        }
        if constexpr (EPI == EPI_RESID_LN && !decltype(planes_c)::value) {
          // x += A W^T + b in place, in two halves of eight rows per lane (the sums stay in registers for what follows, and
          // sixteen reads + sixteen sums + the accumulators would not fit the 168 registers of a 12-wave workgroup); each
          // sum leaves a second time as the next Linear's operand (un-normalised, x 16, h2i), and the 16 lanes of a row
          // group, which hold a row's 64 values, reduce (mean, M2) of this wave's slice of every row: two passes in registers,
          // butterfly over the DPP row (xor 1, xor 2, half mirror, mirror: every lane ends with the sum)
          auto rowsum16 = [](float x) {
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false));
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, false));
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, false));
            return x;
          };
          const int S = (N + 63) >> 6, slice = nb >> 6;
          float* sdst = aux + ((size_t)pm0 * S + slice) * 2;
          bool over = false;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            f32x4 res[2][4];
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int k = (half * 2 + m2) * 16 + r;
                const bool live = !decltype(checked_c)::value || k < rows;
                res[m2][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (live) res[m2][r] = *reinterpret_cast<const f32x4*>(base + (off + (unsigned)k * pitch));
              }
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int mi = half * 2 + m2, k = mi * 16 + r;
                const bool live = !decltype(checked_c)::value || k < rows;
                const f32x4 v = {res[m2][r][0] + value(mi, r, 0), res[m2][r][1] + value(mi, r, 1),
                                 res[m2][r][2] + value(mi, r, 2), res[m2][r][3] + value(mi, r, 3)};
                if (live) *reinterpret_cast<f32x4*>(base + (off + (unsigned)k * pitch)) = v;
                f16x4 ph, pl;
#pragma unroll
                for (int e = 0; e < 4; ++e) { f16 h, l; split2h_scaled(v[e] * oscale, h, l); ph[e] = h; pl[e] = l; }
                store_planes_paired(reinterpret_cast<char*>(out2) + (offp + (unsigned)k * (N * 4)), ph, pl, odd, live);
                // exact range check of the un-normalised operand (its magnitude has no useful bound from the weights alone)
                over |= !(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) * oscale < 65504.0f);
                const float mean = rowsum16((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 64.0f);
                const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                const float q2 = rowsum16(fmaf(d0, d0, d1 * d1) + fmaf(d2, d2, d3 * d3));
                if (live && fi == 0) *reinterpret_cast<float2*>(sdst + (size_t)k * S * 2) = make_float2(mean, q2);
              }
          }
          if (__builtin_expect(__any(over), 0) && lane == 0) atomicOr(flag, 2u);   // d3dp_status: operand left the split range
        }
      };
      using T_ = std::true_type; using F_ = std::false_type;
      constexpr bool kPlanesOnly = EPI == EPI_GELU || EPI == EPI_GELU_LN;
      if (rows >= 64) {                                // (all but the last row of tiles)
        if (planes) { if constexpr (kPlanesOnly || TAG == 1) store_rows(T_{}, F_{}); }
        else { if constexpr (!kPlanesOnly) store_rows(F_{}, F_{}); }
      } else {
        if (planes) { if constexpr (kPlanesOnly || TAG == 1) store_rows(T_{}, T_{}); }
        else { if constexpr (!kPlanesOnly) store_rows(F_{}, T_{}); }
      }
    }
  }
}

#if D3DP_X2_VARIANTS
// The tile epilogue of gemm_f16x2_kernel as a function (the ping-pong kernel below calls it from two places; the kernel above
// keeps its own inline copy so that building without the ping-pong path reproduces its code exactly).  `acc`: this wave's
// 64 x 64 block of tile `t`; `ti`: the workgroup's running tile index (EPI_GELU_LN's row-statistics buffer parity).
// (x2_epilogue_at: the same for a 64 x 64 block at an explicit position -- lane rows pm0 + mi*16 + r, lane columns nb..nb+3;
// `rs_row`: the block's first lane row inside the tile, EPI_GELU_LN's row statistics only.  `sbias` may be LDS or global.)
template <int EPI, int TAG>
__device__ __forceinline__ void x2_epilogue_at(f32x4 (&acc)[4][4], const int pm0, const int nb, int ti, int M, int N, int rs_row,
                                               int fi, int lane, float unscale, float oscale, float* __restrict__ outf,
                                               f16* __restrict__ out2, float* __restrict__ aux, unsigned* __restrict__ flag,
                                               const float* sbias, char* smem, const f32x4* bias_regs = nullptr) {
  // ---- tile epilogue: lane holds out[m = pm0 + mi*16 + r][n = nb + ni], pm0 = tile row + wr*64 + 4 fg, nb = tile column
  // + wc*64 + 4 fi.  Every store address is  uniform base + 32-bit lane offset  (the launcher refuses outputs of 4 GiB or
  // more), advanced row by row: per-row 64-bit address arithmetic was most of the epilogue's VALU work, and it runs
  // with the matrix pipes idle.
  if (nb < N && (!(D3DP_X2_PROBE & 4) || unscale == -12345.f)) {
    const float4 bz = bias_regs ? make_float4((*bias_regs)[0], (*bias_regs)[1], (*bias_regs)[2], (*bias_regs)[3])
                                : *reinterpret_cast<const float4*>(sbias + nb);
    const bool odd = fi & 1;
    unsigned off, pitch;
    bool planes;                                     // split the values and store fp16 planes (else fp32)
    char* base = reinterpret_cast<char*>(outf);
    const int c0 = nb & ~7;                          // h2i rows: 4 N bytes per row; the lane pair's 8 columns start at c0:
    const unsigned offp = (unsigned)pm0 * (N * 4) + (c0 >> 5) * 128 + (c0 & 31) * 2 + (odd ? 64 : 0);   // even lane -> hi slot, odd -> lo
    if constexpr (EPI == EPI_GELU || EPI == EPI_GELU_LN) {   // the fc2 operand
      base = reinterpret_cast<char*>(out2);
      pitch = N * 4; planes = true;
      off = offp - (unsigned)pm0 * pitch;
    } else if constexpr (TAG == 1) {
      // packed qkv row (12 C bytes, C = N / 3): q fp32 | k hi | k lo | v hi | v lo (fp16 planes x 16) -- the
      // K / V operand images of the split-fp16 attention kernels, which copy them into LDS without touching them
      const int C = N / 3, region = nb / C, cn = nb - region * C;      // a wave's 64 columns lie in one region
      pitch = N * 4; planes = region != 0;
      off = planes ? region * 4 * C + cn * 2 + (odd ? 2 * C - 8 : 0) : cn * 4;
    } else {
      pitch = N * 4; planes = false;
      off = nb * 4;
    }
    off += (unsigned)pm0 * pitch;
    const int rows = M - pm0;                        // row k = mi*16 + r of this lane exists iff k < rows
    [[maybe_unused]] float4 c1z = {};
    [[maybe_unused]] const float* srow = nullptr;
    if constexpr (EPI == EPI_GELU_LN) {
      c1z = *reinterpret_cast<const float4*>(sbias + N + nb);
      srow = reinterpret_cast<const float*>(smem + XROWSTAT + (ti & 1) * (XBM * 8)) + rs_row * 2;
    }
    auto value = [&](int mi, int r, int e) {
      const float bze = e == 0 ? bz.x : e == 1 ? bz.y : e == 2 ? bz.z : bz.w;
      if constexpr (EPI == EPI_GELU_LN) {            // rstd (x W'^T - mean c1) + c2
        const float2 st = *reinterpret_cast<const float2*>(srow + (mi * 16 + r) * 2);
        const float c1e = e == 0 ? c1z.x : e == 1 ? c1z.y : e == 2 ? c1z.z : c1z.w;
        return fmaf(st.y, fmaf(acc[mi][e][r], unscale, -(st.x * c1e)), bze);
      } else {
        return fmaf(acc[mi][e][r], unscale, bze);
      }
    };
    auto store_rows = [&](auto planes_c, auto checked_c) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = mi * 16 + r;
          const bool live = !decltype(checked_c)::value || k < rows;
          char* dst = base + (off + (unsigned)k * pitch);
          if constexpr (decltype(planes_c)::value) {
            f16x4 ph, pl;
            float v[4] = {value(mi, r, 0), value(mi, r, 1), value(mi, r, 2), value(mi, r, 3)};
            if constexpr (EPI == EPI_GELU || EPI == EPI_GELU_LN) {
#if D3DP_X2_PKGELU
              const f32x2 g0 = gelu_erf_rational2((f32x2){v[0], v[1]}), g1 = gelu_erf_rational2((f32x2){v[2], v[3]});
              v[0] = g0.x; v[1] = g0.y; v[2] = g1.x; v[3] = g1.y;
#else
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = gelu_erf_rational(v[e]);
#endif
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              f16 h, l;
              split2h_scaled(v[e] * oscale, h, l);       // (oscale: the consumer's operand scale, 2^4 unless capi.hip lowered it)
              ph[e] = h; pl[e] = l;
            }
            store_planes_paired(dst, ph, pl, odd, live);
          } else if constexpr (EPI != EPI_RESID && EPI != EPI_RESID_LN) {
            if (live) OUT_STORE(reinterpret_cast<f32x4*>(dst), ((f32x4){value(mi, r, 0), value(mi, r, 1), value(mi, r, 2), value(mi, r, 3)}));
          }
        }
      if constexpr (EPI == EPI_RESID && !decltype(planes_c)::value) {
        // x += A W^T + b in place (the residual stream): all sixteen reads of the tile in flight before the first add
        f32x4 res[4][4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k = mi * 16 + r;
            const bool live = !decltype(checked_c)::value || k < rows;
            res[mi][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (live) res[mi][r] = *reinterpret_cast<const f32x4*>(base + (off + (unsigned)k * pitch));
          }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k = mi * 16 + r;
            const bool live = !decltype(checked_c)::value || k < rows;
            const f32x4 v = {res[mi][r][0] + value(mi, r, 0), res[mi][r][1] + value(mi, r, 1),
                             res[mi][r][2] + value(mi, r, 2), res[mi][r][3] + value(mi, r, 3)};
            if (live) *reinterpret_cast<f32x4*>(base + (off + (unsigned)k * pitch)) = v;   // (re-read by the next row kernel: no nt hint)
          }
      }
      if constexpr (EPI == EPI_RESID_LN && !decltype(planes_c)::value) {
        // x += A W^T + b in place, in two halves of eight rows per lane (the sums stay in registers for what follows, and
        // sixteen reads + sixteen sums + the accumulators would not fit the 168 registers of a 12-wave workgroup); each
        // sum leaves a second time as the next Linear's operand (un-normalised, x 16, h2i), and the 16 lanes of a row
        // group, which hold a row's 64 values, reduce (mean, M2) of this wave's slice of every row: two passes in registers,
        // butterfly over the DPP row (xor 1, xor 2, half mirror, mirror: every lane ends with the sum)
        auto rowsum16 = [](float x) {
          x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));
          x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false));
          x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, false));
          x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, false));
          return x;
        };
        const int S = (N + 63) >> 6, slice = nb >> 6;
        float* sdst = aux + ((size_t)pm0 * S + slice) * 2;
        bool over = false;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          f32x4 res[2][4];
#pragma unroll
          for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int k = (half * 2 + m2) * 16 + r;
              const bool live = !decltype(checked_c)::value || k < rows;
              res[m2][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
              if (live) res[m2][r] = *reinterpret_cast<const f32x4*>(base + (off + (unsigned)k * pitch));
            }
#pragma unroll
          for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int mi = half * 2 + m2, k = mi * 16 + r;
              const bool live = !decltype(checked_c)::value || k < rows;
              const f32x4 v = {res[m2][r][0] + value(mi, r, 0), res[m2][r][1] + value(mi, r, 1),
                               res[m2][r][2] + value(mi, r, 2), res[m2][r][3] + value(mi, r, 3)};
              if (live) *reinterpret_cast<f32x4*>(base + (off + (unsigned)k * pitch)) = v;
              f16x4 ph, pl;
#pragma unroll
              for (int e = 0; e < 4; ++e) { f16 h, l; split2h_scaled(v[e] * oscale, h, l); ph[e] = h; pl[e] = l; }
              store_planes_paired(reinterpret_cast<char*>(out2) + (offp + (unsigned)k * (N * 4)), ph, pl, odd, live);
              // exact range check of the un-normalised operand (its magnitude has no useful bound from the weights alone)
              over |= !(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) * oscale < 65504.0f);
              const float mean = rowsum16((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 64.0f);
              const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
              const float q2 = rowsum16(fmaf(d0, d0, d1 * d1) + fmaf(d2, d2, d3 * d3));
              if (live && fi == 0) *reinterpret_cast<float2*>(sdst + (size_t)k * S * 2) = make_float2(mean, q2);
            }
        }
        if (__builtin_expect(__any(over), 0) && lane == 0) atomicOr(flag, 2u);   // d3dp_status: operand left the split range
      }
    };
    using T_ = std::true_type; using F_ = std::false_type;
    constexpr bool kPlanesOnly = EPI == EPI_GELU || EPI == EPI_GELU_LN;
    if (rows >= 64) {                                // (all but the last row of tiles)
      if (planes) { if constexpr (kPlanesOnly || TAG == 1) store_rows(T_{}, F_{}); }
      else { if constexpr (!kPlanesOnly) store_rows(F_{}, F_{}); }
    } else {
      if (planes) { if constexpr (kPlanesOnly || TAG == 1) store_rows(T_{}, T_{}); }
      else { if constexpr (!kPlanesOnly) store_rows(F_{}, T_{}); }
    }
  }
}

template <int EPI, int TAG>
__device__ __forceinline__ void x2_tile_epilogue(f32x4 (&acc)[4][4], int t, int ti, int tiles_n, int M, int N, int wr, int wc,
                                                 int fi, int fg, int lane, float unscale, float oscale, float* __restrict__ outf,
                                                 f16* __restrict__ out2, float* __restrict__ aux, unsigned* __restrict__ flag,
                                                 const float* sbias, char* smem) {
  x2_epilogue_at<EPI, TAG>(acc, (t / tiles_n) * XBM + wr * 64 + 4 * fg, (t % tiles_n) * XBN + wc * 64 + 4 * fi, ti, M, N,
                           wr * 64 + 4 * fg, fi, lane, unscale, oscale, outf, out2, aux, flag, sbias, smem);
}

// ---------------------------------------------------------------------------------------------------------------------
// PING-PONG form of the same Linear: the two compute waves of a SIMD work half a k-step apart.
//
// Measured on the kernel above with its loads AND its epilogue compiled out (gpurun c7, profiles/r04_gemm_probes.md): 463 us for
// the qkv shape at M = 128,960 = 1.22 us per k-step, of which the matrix pipe needs 96 MFMAs x 16 = 1536 cycles (0.85 us at the 1.8 GHz
// the chip holds here).  All
// eight compute waves pass the k-step's barrier together, all read their fragments from LDS together (the pipe idles), then
// both waves of every SIMD push their 48 MFMAs through the one pipe together: the pipe is never fed during the read phase.
// Here the compute waves form two teams -- X = waves 0..3, Y = waves 4..7: one wave of each per SIMD (waves go to SIMDs round
// robin) -- and a k-step has TWO barriers, A and B:
//     phase a (after A_g):  X reads ALL its fragments of slab g           | Y multiplies slab g - 1 (48 MFMAs, the pipe to itself)
//     phase b (after B_g):  X multiplies slab g                           | Y reads all its fragments of slab g
// so while one team waits for LDS the other owns the matrix pipe.  A wave holds one k-step's fragments (16 x 16 bytes per lane)
// instead of streaming the A fragments behind the MFMAs: nothing reads slab g after phase b(g), the ring and the loaders'
// schedule (slab g + 2 issued behind A_g, landed by A_{g+2}) are those of the kernel above; the lagged MFMAs are gone (they
// existed to cover the read phase).  A tile's epilogue falls into the team's read phase at the start of its next tile, beside
// the other team's MFMAs (team Y: after the MFMAs of its last k-step, in phase a).
template <int EPI, int TAG>
__global__ __launch_bounds__(768) void gemm_f16x2_pp_kernel(const f16* __restrict__ A2, const f16* __restrict__ W2,
                                                            const float* __restrict__ bias, float unscale, float oscale,
                                                            float* __restrict__ outf, f16* __restrict__ out2,
                                                            float* __restrict__ aux, unsigned* __restrict__ flag, int M,
                                                            int N, int K, int tiles_n, int total_tiles) {
  static_assert(EPI == EPI_BIAS || EPI == EPI_GELU || EPI == EPI_RESID, "the LayerNorm-folding epilogues keep the kernel above");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sbias = reinterpret_cast<float*>(smem + XNSTAGE * XSTAGE);
  const int G = gridDim.x;
  const int L = xcd_remap(blockIdx.x, G);
  const int n_my = (total_tiles - L + G - 1) / G;      // tiles L, L+G, ...
  const int NK = K / XBK;
  const int gtot = n_my * NK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  for (int i = tid; i < N; i += (XNCW + 4) * 64) sbias[i] = bias[i];
  __syncthreads();
  if (gtot == 0) return;

  if (wave >= XNCW) {
    // ------------------------------------------------------------------ loader waves (as above, two barriers per k-step)
    const int lw = wave - XNCW;
    const int lr = lane >> 3, lq = lane & 7;
    int ti = 0, ks = 0, slot = 0;
    const f16* pa[8];
    const f16* pw[4];
    auto issue = [&]() {
      if (ks == 0) {
        const int t = L + ti * G;
        const int m0 = (t / tiles_n) * XBM, n0 = (t % tiles_n) * XBN;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = (lw * 8 + i) * 8 + lr;
          pa[i] = A2 + (size_t)min(m0 + row, M - 1) * (2 * K) + swz128(row, lq) * 8;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = (lw * 4 + i) * 8 + lr;
          const int wrow = (row & 64) + colperm(row & 63);
          pw[i] = W2 + (size_t)min(n0 + wrow, N - 1) * (2 * K) + swz128(row, lq) * 8;
        }
      }
      char* base = smem + slot * XSTAGE;
      const int ko = ks * (2 * XBK);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds(GPTR(pa[i] + ko), LPTR(base + (lw * 8 + i) * 1024), 16, 0, D3DP_X2_AAUX);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds(GPTR(pw[i] + ko), LPTR(base + XA_BYTES + (lw * 4 + i) * 1024), 16, 0, D3DP_X2_WAUX);
      if (++ks == NK) { ks = 0; ++ti; }
      slot = (slot == XNSTAGE - 1) ? 0 : slot + 1;
    };
    issue();
    if (gtot > 1) issue();
    for (int g = 0; g < gtot; ++g) {
      if (g + 1 < gtot) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      X2_BARRIER();                                    // A_g: slab g has landed; nobody reads slab g - 1 any more
      if (g + 2 < gtot) issue();                       // slab g + 2 into the slot of slab g - 1
      X2_BARRIER();                                    // B_g
    }
    return;
  }

  // -------------------------------------------------------------------- compute waves
  const int team = wave >> 2;                          // 0 = X, 1 = Y
  const int wr = wave >> 1, wc = wave & 1;
  const int fi = lane & 15, fg = lane >> 4;
  const int offA = (wr * 64 + fi) * 128 + swz128(fi, fg) * 16, offAl = offA ^ 64;
  const int offW = XA_BYTES + (wc * 64 + fi) * 128 + swz128(fi, fg) * 16, offWl = offW ^ 64;
  f32x4 acc[4][4];
  f16x8 wf[4][2], ah[4], al[4];                        // one k-step's fragments
  __builtin_amdgcn_s_setprio(1);
  int slot = 0;
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  auto read_all = [&]() {
    const char* sb = smem + slot * XSTAGE;
    slot = (slot == XNSTAGE - 1) ? 0 : slot + 1;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        wf[ni][pl] = *reinterpret_cast<const f16x8*>(sb + (pl ? offWl : offW) + ni * 2048);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      ah[mi] = *reinterpret_cast<const f16x8*>(sb + offA + mi * 2048);
      al[mi] = *reinterpret_cast<const f16x8*>(sb + offAl + mi * 2048);
    }
  };
  auto mfma_all = [&]() {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      // small terms first; the three products of one output tile are four MFMAs apart (no back-to-back dependency)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], wf[ni][1], acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mi], wf[ni][0], acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], wf[ni][0], acc[mi][ni], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto epilogue = [&](int ti) {
    x2_tile_epilogue<EPI, TAG>(acc, L + ti * G, ti, tiles_n, M, N, wr, wc, fi, fg, lane, unscale, oscale, outf, out2, aux, flag,
                               sbias, smem);
  };

  // (one pass more than there are k-steps: the last tile's epilogue runs where every other tile's does, so the epilogue is
  //  instantiated once per team -- two more copies of it cost registers the kernel does not have)
  if (team == 0) {
    int ks = 0, ti = 0;
#pragma unroll 1
    for (int g = 0; g <= gtot; ++g) {
      if (g < gtot) X2_BARRIER();                      // A_g
      if (ks == 0) {
        if (g > 0) epilogue(ti - 1);                   // the finished tile leaves beside team Y's MFMAs
        zero_acc();
      }
      if (g == gtot) break;
      read_all();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      X2_BARRIER();                                    // B_g
      mfma_all();
      if (++ks == NK) { ks = 0; ++ti; }
    }
  } else {
    int ks = 0, ti = 0;
#pragma unroll 1
    for (int g = 0; g <= gtot; ++g) {
      if (g < gtot) X2_BARRIER();                      // A_g
      if (g > 0) mfma_all();                           // slab g - 1
      if (ks == 0) {
        if (g > 0) epilogue(ti - 1);
        zero_acc();
      }
      if (g == gtot) break;
      X2_BARRIER();                                    // B_g
      read_all();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (++ks == NK) { ks = 0; ++ti; }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// WIDE form of the same Linear: 256 x 256 tile, eight waves of 256 registers, no loader waves.
//
// An experiment (profiles/r04_gemm_probes.md section 4): per MFMA a 64 x 128 register tile per wave (128 accumulator registers,
// 64 for the eight W fragments it keeps for the whole k-step) reads 0.75 x the LDS bytes of the kernel at the top of this file,
// receives 0.67 x the LDS-DMA bytes, fetches A half as often and passes half as many barriers.  That needs the 256 registers
// of a two-waves-per-SIMD workgroup, so nothing is left for loader waves: every wave issues its own share of the k-step's
// LDS-DMA (4 A pieces + 4 W pieces of 8 rows) in inline assembly and waits for it with counted vmcnt; the epilogue's
// stores pass through the same counter and are counted with it (x2_epilogue_at issues exactly 16 stores per 64 x 64
// block of a full tile).  MEASURED: ties with the kernel at the top within 3 % as built, without loads, without epilogue
// and without both -- the Linear's time is its MFMA count at the clock the power budget allows, not its LDS traffic.
// Kept behind D3DP_X2_WIDE=1 (and epi | 4096 of d3dp_op_linear_x2), off.
//   LDS: A ring 3 x 32 KiB at 0, W ring 2 x 32 KiB at 96 KiB = 160 KiB; the bias is read from global memory.
//   per k-step g:  wait own A(g), W(g) | barrier | issue W(g+1) -> W slot of g-1, A(g+2) -> A slot of g-1 | 96 MFMAs
// Requires N % 256 == 0 and K % 32 == 0.  The products meet every accumulator in the order of the plain kernel
// (k ascending; ah.wl, al.wh, ah.wh): results are bit-identical to it.
constexpr int WBN = 256;
constexpr int WA_STAGES = 3, WW_STAGES = 2;
constexpr int WW_BYTES = WBN * 128;                  // 32 KiB
constexpr int WW_BASE = WA_STAGES * XA_BYTES;        // 96 KiB
constexpr int WLDS = WW_BASE + WW_STAGES * WW_BYTES; // 160 KiB

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"          // (the expected "clobber list contains reserved registers" note for m0)
// one LDS-DMA wave-instruction in the  uniform base + 32-bit lane offset  form: lane l's 16 bytes at base + off -> LDS
// bytes [lds + 16 l, +16).  Inline assembly for the reasons given at lds_dma16 in attention.hip: the builtin makes the
// compiler turn every vector-memory wait of the kernel into vmcnt(0).
__device__ __forceinline__ void wide_dma16(unsigned off, const char* base, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop
// 16-byte global load the compiler does not track (it would wait for it with vmcnt(0) counted without the LDS-DMA operations
// around it); the caller waits with wide_wait_vmcnt and passes the registers through wide_settle.
__device__ __forceinline__ f32x4 wide_gload16(const float* p) {
  f32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
  return r;
}
__device__ __forceinline__ void wide_settle(f32x4& r) { asm volatile("" : "+v"(r)); }
template <int V>
__device__ __forceinline__ void wide_wait_vmcnt() { __builtin_amdgcn_s_waitcnt((V & 15) | (7 << 4) | (15 << 8) | ((V >> 4) << 14)); }

template <int EPI, int TAG>
__global__ __launch_bounds__(512) void gemm_f16x2_wide_kernel(const f16* __restrict__ A2, const f16* __restrict__ W2,
                                                              const float* __restrict__ bias, float unscale, float oscale,
                                                              float* __restrict__ outf, f16* __restrict__ out2,
                                                              float* __restrict__ aux, unsigned* __restrict__ flag, int M,
                                                              int N, int K, int tiles_n, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int G = gridDim.x;
  const int L = xcd_remap(blockIdx.x, G);
  const int n_my = (total_tiles - L + G - 1) / G;      // tiles L, L+G, ...
  const int NK = K / XBK;
  const int gtot = n_my * NK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int fi = lane & 15, fg = lane >> 4;
  const int lr = lane >> 3, lq = lane & 7;             // loader role: row within an 8-row piece, physical 16-byte slot
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)LPTR(smem);
  const unsigned rowbytes = (unsigned)K * 4;           // one operand row: K x (hi | lo) fp16

  // ---- this wave's share of the loads: pieces 4 wave .. 4 wave + 3 of the A slab and of the W slab
  unsigned voffW[4], voffA[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + lr;                            // LDS row of the W slab
    const int wrow = (row & ~63) + colperm(row & 63);                   // output column (inside the tile) it carries
    voffW[i] = (unsigned)wrow * rowbytes + swz128(row, lq) * 16;        // (N % 256 == 0: no clamp)
  }
  int tiA = 0, ksA = 0, slotA = 0, tiW = 0, ksW = 0, slotW = 0;         // (tile, k-step, ring slot) of the next slab to issue
  const char* baseA = nullptr;
  const char* baseW = nullptr;
  auto issueA = [&]() {
    if (ksA == 0) {
      const int t = L + tiA * G;
      const int m0 = (t / tiles_n) * XBM;
      baseA = reinterpret_cast<const char*>(A2) + (size_t)m0 * rowbytes;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + lr;
        voffA[i] = (unsigned)min(row, M - 1 - m0) * rowbytes + swz128(row, lq) * 16;
      }
    }
    const char* b = baseA + ksA * (4 * XBK);           // one k-step of a row = 64 fp16 = 128 B
    const unsigned dst = lds0 + slotA * XA_BYTES + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (!(D3DP_X2_PROBE & 1) || unscale == -12345.f) wide_dma16(voffA[i], b, dst + i * 1024);
    if (++ksA == NK) { ksA = 0; ++tiA; }
    slotA = (slotA == WA_STAGES - 1) ? 0 : slotA + 1;
  };
  auto issueW = [&]() {
    if (ksW == 0) {
      const int t = L + tiW * G;
      baseW = reinterpret_cast<const char*>(W2) + (size_t)((t % tiles_n) * WBN) * rowbytes;
    }
    const char* b = baseW + ksW * (4 * XBK);
    const unsigned dst = lds0 + WW_BASE + slotW * WW_BYTES + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (!(D3DP_X2_PROBE & 1) || unscale == -12345.f) wide_dma16(voffW[i], b, dst + i * 1024);
    if (++ksW == NK) { ksW = 0; ++tiW; }
    slotW ^= 1;
  };

  // per-lane fragment offsets (the swizzle depends on the lane only: rows advance in multiples of 16; lo plane = offset ^ 64)
  const int offA = (wr * 64 + fi) * 128 + swz128(fi, fg) * 16, offAl = offA ^ 64;
  const int offW = WW_BASE + (wc * 128 + fi) * 128 + swz128(fi, fg) * 16, offWl = offW ^ 64;
  f32x4 acc[2][4][4];                                  // [column half][mi][ni]: two 64 x 64 blocks as the epilogue wants them

  if (gtot > 0) { issueA(); issueW(); }
  if (gtot > 1) issueA();
  int g = 0, cslotA = 0, cslotW = 0;
  bool after_full_tile = false;                        // the previous k-step ended with the 32 stores of a full tile
#pragma unroll 1
  for (int ti = 0; ti < n_my; ++ti) {
    const int t = L + ti * G;
    const int m0 = (t / tiles_n) * XBM, nb0 = (t % tiles_n) * WBN + wc * 128 + 4 * fi;   // (nb0: this lane's first column)
    f32x4 bz[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[h][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ks = 0; ks < NK; ++ks, ++g) {
      // own loads of slab g landed; younger operations that may stay in flight: A(g+1) (4) and, right behind a full
      // tile's epilogue, its 32 stores.  (Behind a partial tile the number of stores issued is not known: the strict
      // count is always safe -- the stores are the youngest operations.)
      if (g + 1 >= gtot) wide_wait_vmcnt<0>();
      else if (after_full_tile) wide_wait_vmcnt<36>();
      else wide_wait_vmcnt<4>();
      after_full_tile = false;
      __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): this wave has read everything it wanted from the old slots
      X2_BARRIER();
      if (g + 1 < gtot) issueW();                      // into the slots of k-step g-1: every wave has passed barrier g
      if (g + 2 < gtot) issueA();
      if (ks == NK - 1) {                              // the tile's bias, a k-step ahead of its use (youngest operations)
        bz[0] = wide_gload16(bias + nb0);
        bz[1] = wide_gload16(bias + nb0 + 64);
      }
      __builtin_amdgcn_sched_barrier(0);
      const char* sa = smem + cslotA * XA_BYTES;
      const char* sw = smem + cslotW * WW_BYTES;
      cslotA = (cslotA == WA_STAGES - 1) ? 0 : cslotA + 1;
      cslotW ^= 1;
      f16x8 wf[8][2], ah[2], al[2];
      ah[0] = *reinterpret_cast<const f16x8*>(sa + offA);
      al[0] = *reinterpret_cast<const f16x8*>(sa + offAl);
#pragma unroll
      for (int nj = 0; nj < 8; ++nj) {
        wf[nj][1] = *reinterpret_cast<const f16x8*>(sw + offWl + nj * 2048);
        wf[nj][0] = *reinterpret_cast<const f16x8*>(sw + offW + nj * 2048);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int b = mi & 1;
        if (mi < 3) {                                  // next row block's fragments while this one multiplies
          ah[b ^ 1] = *reinterpret_cast<const f16x8*>(sa + offA + (mi + 1) * 2048);
          al[b ^ 1] = *reinterpret_cast<const f16x8*>(sa + offAl + (mi + 1) * 2048);
        }
        if (mi == 0) {
          // column-major through the first row block: its first MFMAs need 4 of the 18 fragment reads, not 10
#pragma unroll
          for (int nj = 0; nj < 8; ++nj) {
            f32x4& c = acc[nj >> 2][0][nj & 3];
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0], wf[nj][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[0], wf[nj][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0], wf[nj][0], c, 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int nj = 0; nj < 8; ++nj)
            acc[nj >> 2][mi][nj & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b], wf[nj][1], acc[nj >> 2][mi][nj & 3], 0, 0, 0);
#pragma unroll
          for (int nj = 0; nj < 8; ++nj)
            acc[nj >> 2][mi][nj & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[b], wf[nj][0], acc[nj >> 2][mi][nj & 3], 0, 0, 0);
#pragma unroll
          for (int nj = 0; nj < 8; ++nj)
            acc[nj >> 2][mi][nj & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b], wf[nj][0], acc[nj >> 2][mi][nj & 3], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // everything this wave has in flight was issued a k-step ago or earlier: the loads of the next two k-steps and the bias
    wide_wait_vmcnt<0>();
    wide_settle(bz[0]);
    wide_settle(bz[1]);
#pragma unroll
    for (int h = 0; h < 2; ++h)
      x2_epilogue_at<EPI, TAG>(acc[h], m0 + wr * 64 + 4 * fg, nb0 + h * 64, ti, M, N, 0, fi, lane, unscale, oscale, outf, out2,
                               aux, flag, bias, smem, &bz[h]);
    after_full_tile = m0 + XBM <= M;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same Linear with a ROW-CLASS SKEWED schedule: a tile's epilogue runs UNDER the k-loop of the next tile.
//
// In the kernel above all eight compute waves reach a tile's epilogue together (one s_barrier per k-step keeps them in
// lock-step), so its VALU work and stores -- 12 % of the qkv Linear, 27 % of fc1 with its GELU (profiles/r03_gemm_probes.md)
// -- run with the matrix pipes idle; parking a finished 256 x 128 tile beside the next one's accumulators would take 64 more
// registers than a 12-wave workgroup has.  Here the four 16-row blocks of a wave's 64 rows (row class c = 0..3) end their
// tiles at DIFFERENT k-steps: class c switches to its next tile when ks == c D.  Sums over k commute, so class c simply
// runs the k-steps of a tile in the rotated order c D, ..., NK - 1, 0, ..., c D - 1; all classes still consume the SAME W
// slab in every k-step (consecutive tiles of a workgroup lie in one 128-column strip), and only the A rows of class c
// belong to another tile for a while -- a matter of which rows the loader waves fetch.  At most ONE class is between tiles
// at any time: its 16 finished values per lane are PARKED (16 registers) and leave over the next D k-steps, 4 / D output
// rows per k-step, their VALU work and stores issued between that k-step's 48 MFMAs.
//   registers  Which accumulator block parks must not be a run-time choice (selecting acc[c] dynamically costs the register
//              allocator ~40 registers of copies, and unrolling a whole tile round spills as well): the block that parks is
//              always accumulator block 0.  After it parks, the blocks shift down (acc[p] <- acc[p + 1], acc[3] <- 0: 40
//              v_mov per D k-steps) and the LOADERS rotate the LDS image to match: with rot = number of parks so far, LDS
//              row slot p of every 64-row group holds the rows of class (p + rot) & 3.  The compute code is static.
//   schedule   workgroup L owns strip L % tiles_n and the row-tile range [lo, hi) of its row group L / tiles_n (Q = G /
//              tiles_n row groups; G - Q tiles_n workgroups idle).  Round ti = NK k-steps; class c works on tile ti (ks >= c D)
//              or ti - 1.  (hi - lo) full rounds and a flush of 3 D k-steps: in the first c D steps class c has no tile yet
//              (its sums are discarded), in the flush the classes that are done multiply rows nobody stores -- 1.5 D k-steps
//              of matrix work lost per launch and workgroup, against one exposed epilogue per tile.
//   loaders    A piece i of loader wave lw lands in LDS rows lw 64 + 8 i .. + 7 (row slot i >> 1) as before; it FETCHES the
//              rows of class ((i >> 1) + rot) & 3, from tile ti or ti - 1 as that class stands; pointers re-derived at the
//              four park steps of a round.  W pieces: one strip for the whole launch.  Ring, barriers, vmcnt: unchanged.
//   numerics   the rotation changes the ORDER of a row's fp32 partial sums with its row class: capi.hip pads every sequence
//              to a multiple of 64 rows (d3dp_ctx::seq_pitch), which makes the class a function of the token's index in its
//              sequence -- results stay bit-identical across batch compositions, pass splits and ranks.
// Built for the two Linears whose epilogue is pure register work: EPI_BIAS / TAG 1 (qkv, packed rows) and EPI_GELU (fc1).
// proj / fc2 (x += ..., their epilogue waits on loads of the residual rows) keep the kernel above.
template <int EPI, int TAG, int D>
__global__ __launch_bounds__(768) void gemm_f16x2_skew_kernel(const f16* __restrict__ A2, const f16* __restrict__ W2,
                                                              const float* __restrict__ bias, float unscale, float oscale,
                                                              float* __restrict__ outf, f16* __restrict__ out2, int M, int N,
                                                              int K, int tiles_n, int tm, int Q) {
  static_assert(D == 1 || D == 2 || D == 4, "a parked class leaves in D k-steps, 4 / D rows per k-step");
  static_assert(EPI == EPI_GELU || (EPI == EPI_BIAS && TAG == 1), "epilogues without loads only");
  constexpr int RPK = 4 / D;                           // output rows (of the parked class) per k-step
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sbias = reinterpret_cast<float*>(smem + XNSTAGE * XSTAGE);
  const int G = gridDim.x;
  const int L = xcd_remap(blockIdx.x, G);
  const int strip = L % tiles_n, rg = L / tiles_n;
  const int lo = rg < Q ? (int)((long)rg * tm / Q) : 0, hi = rg < Q ? (int)((long)(rg + 1) * tm / Q) : 0;
  const int n_tiles = hi - lo;
  const int NK = K / XBK;                              // >= 4 D (launcher)
  const int gtot = n_tiles > 0 ? n_tiles * NK + 3 * D : 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  for (int i = tid; i < N; i += (XNCW + 4) * 64) sbias[i] = bias[i];
  __syncthreads();
  if (gtot == 0) return;

  if (wave >= XNCW) {
    // ------------------------------------------------------------------ loader waves
    const int lw = wave - XNCW;
    const int lr = lane >> 3, lq = lane & 7;
    const f16* pa[8];
    const f16* pw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (lw * 4 + i) * 8 + lr;
      const int wrow = (row & 64) + colperm(row & 63);
      pw[i] = W2 + (size_t)min(strip * XBN + wrow, N - 1) * (2 * K) + swz128(row, lq) * 8;
    }
    int ti = 0, ks = 0, slot = 0;                      // (round, k-step, ring slot) of the next slab to issue
    auto issue = [&]() {
      if ((ks & (D - 1)) == 0 && ks < 4 * D) {         // a park step: the LDS image rotates and one class changes tile
        const int rot = (ks / D + 1) & 3;              // parks so far, mod 4, once this step's park is done
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int cls = ((i >> 1) + rot) & 3;        // the class whose rows row slot i >> 1 holds from this k-step on
          int j = ks >= cls * D ? ti : ti - 1;         // that class has switched to tile ti in this round, or has not yet
          j = min(max(j, 0), n_tiles - 1);             // (before its first tile / after its last: any valid rows)
          const int lds_row = (lw * 8 + i) * 8 + lr;   // where the piece lands (the swizzle goes by the LDS row)
          const int row = lw * 64 + cls * 16 + (i & 1) * 8 + lr;
          pa[i] = A2 + (size_t)min((lo + j) * XBM + row, M - 1) * (2 * K) + swz128(lds_row, lq) * 8;
        }
      }
      char* base = smem + slot * XSTAGE;
      const int ko = ks * (2 * XBK);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds(GPTR(pa[i] + ko), LPTR(base + (lw * 8 + i) * 1024), 16, 0, D3DP_X2_AAUX);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds(GPTR(pw[i] + ko), LPTR(base + XA_BYTES + (lw * 4 + i) * 1024), 16, 0, D3DP_X2_WAUX);
      if (++ks == NK) { ks = 0; ++ti; }
      slot = (slot == XNSTAGE - 1) ? 0 : slot + 1;
    };
    issue();
    if (gtot > 1) issue();
    for (int g = 0; g < gtot; ++g) {
      if (g + 1 < gtot) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      X2_BARRIER();
      if (g + 2 < gtot) issue();
    }
    return;
  }

  // -------------------------------------------------------------------- compute waves
  const int wr = wave >> 1, wc = wave & 1;
  const int fi = lane & 15, fg = lane >> 4;
  const int offA = (wr * 64 + fi) * 128 + swz128(fi, fg) * 16, offAl = offA ^ 64;
  const int offW = XA_BYTES + (wc * 64 + fi) * 128 + swz128(fi, fg) * 16, offWl = offW ^ 64;
  __builtin_amdgcn_s_setprio(1);

  // what does not change over the launch: this lane's four output columns nb .. nb + 3 of the workgroup's strip.  (A wave's
  // 64 columns lie in one region of the packed qkv row: `planes` is wave-uniform.)
  const int nbw = strip * XBN + wc * 64;               // wave-uniform
  const int nb = nbw + 4 * fi;
  const bool cols_live = nb < N;
  const bool odd = fi & 1;
  const unsigned pitch = (unsigned)N * 4;              // bytes per output row in every form (fp32 [N], h2i [2 N] fp16, packed 12 C)
  bool planes;
  unsigned coff;
  char* base;
  {
    const int c0 = nb & ~7;
    if constexpr (EPI == EPI_GELU) {                   // the fc2 operand, h2i: even lane -> hi slot of the pair's 8 columns, odd -> lo
      base = reinterpret_cast<char*>(out2);
      planes = true;
      coff = (c0 >> 5) * 128 + (c0 & 31) * 2 + (odd ? 64 : 0);
    } else {                                           // packed qkv row: q fp32 | k hi | k lo | v hi | v lo
      base = reinterpret_cast<char*>(outf);
      const int C = N / 3, region = nbw / C, cn = nb - region * C;
      planes = region != 0;
      coff = planes ? region * 4 * C + cn * 2 + (odd ? 2 * C - 8 : 0) : cn * 4;
    }
  }
  const int row0 = wr * 64 + 4 * fg;                   // this lane's row (r = 0) of row class 0 inside a tile

  f32x4 acc[4][4];                                     // acc[p]: the class whose rows LDS row slot p holds ((p + rot) & 3)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 park[4];                                       // the class between tiles: park[ni][r]
#pragma unroll
  for (int j = 0; j < 4; ++j) park[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned joff = 0;                                   // byte offset of row r = 0 of the parked class (+ coff)
  int jrows = 0;                                       // its rows r < jrows exist (<= 0: nothing to store)
  [[maybe_unused]] float4 jbz = {};                    // (D3DP_X2_SKEW_TOP) the lane's four biases, re-read at every park

  // value e of output row r of the parked class, through the epilogue's arithmetic (the lane's four biases are re-read
  // from LDS in every k-step that needs them -- one ds_read_b128 -- instead of living in registers across the k-loop)
  const float* bias4 = sbias + min(nb, N - 4);
  auto value = [&](int r, int e, const float4& bz) {
    return fmaf(park[e][r], unscale, e == 0 ? bz.x : e == 1 ? bz.y : e == 2 ? bz.z : bz.w);
  };
  // one output row of the parked class: 4 values per lane -> one 16-byte store per lane
  auto store_row = [&](int r, float (&v)[4]) {
    const bool live = cols_live && r < jrows;
    char* dst = base + (joff + (unsigned)r * pitch);
#if D3DP_X2_SKEW_PROBE & 8
    if (live) OUT_STORE(reinterpret_cast<f32x4*>(dst), ((f32x4){park[0][r & 3], park[1][r & 3], park[2][r & 3], park[3][r & 3]}));
    return;
#endif
    if (planes) {
      f16x4 ph, pl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f16 h, l;
        split2h_scaled(v[e] * oscale, h, l);
        ph[e] = h; pl[e] = l;
      }
#if D3DP_X2_SKEW_PROBE & 4
      asm volatile("" :: "v"(ph), "v"(pl));
#else
      store_planes_paired(dst, ph, pl, odd, live);
#endif
    } else {
#if D3DP_X2_SKEW_PROBE & 4
      asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
#else
      if (live) OUT_STORE(reinterpret_cast<f32x4*>(dst), ((f32x4){v[0], v[1], v[2], v[3]}));
#endif
    }
  };
  // the class in accumulator block 0 changes tile: park it, shift the blocks down, start its next tile from zero in block 3
  // (the loaders rotate the LDS image by one row slot at the same k-step)
  auto park_and_shift = [&](int cls, int tile) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      park[j] = acc[0][j];
#if !(D3DP_X2_SKEW_PROBE & 2)
      acc[0][j] = acc[1][j]; acc[1][j] = acc[2][j]; acc[2][j] = acc[3][j];
      acc[3][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#endif
    }
    const int m_row = (lo + tile) * XBM + row0 + cls * 16;      // tile < 0 (no tile finished yet): nothing is stored
    jrows = tile >= 0 ? M - m_row : 0;
    joff = (unsigned)m_row * pitch + coff;
#if D3DP_X2_SKEW_TOP
    jbz = *reinterpret_cast<const float4*>(bias4);
#endif
  };

  int slot = 0;
  // one k-step; JOB: RPK rows of the parked class leave beside its MFMAs (a compile-time flag: a run-time branch inside the
  // k-step would cut its MFMAs and the epilogue's VALU into separate scheduling regions); r0 = first of those rows
  auto kstep = [&](auto job_c, int r0) {
    constexpr bool JOB = decltype(job_c)::value;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    X2_BARRIER();
    __builtin_amdgcn_sched_barrier(0);
    const char* sb = smem + slot * XSTAGE;
    slot = (slot == XNSTAGE - 1) ? 0 : slot + 1;
    f16x8 wf[4][2], ah[2], al[2];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        wf[ni][pl] = *reinterpret_cast<const f16x8*>(sb + (pl ? offWl : offW) + ni * 2048);
    ah[0] = *reinterpret_cast<const f16x8*>(sb + offA);
    al[0] = *reinterpret_cast<const f16x8*>(sb + offAl);
    [[maybe_unused]] float4 bz = {};
#if !D3DP_X2_SKEW_TOP
    if constexpr (JOB) bz = *reinterpret_cast<const float4*>(bias4);
#endif
    __builtin_amdgcn_sched_barrier(0);
#if D3DP_X2_SKEW_TOP
    if constexpr (JOB && !(D3DP_X2_SKEW_PROBE & 1)) {
      // the leaving rows of this k-step, whole, while the fragment reads above are in flight (nothing here depends on them)
#pragma unroll
      for (int rr = 0; rr < RPK; ++rr) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#if D3DP_X2_SKEW_PROBE & 8
          v[e] = 0.f;
#else
          v[e] = value(r0 + rr, e, jbz);
          if constexpr (EPI == EPI_GELU) v[e] = gelu_erf_rational(v[e]);
#endif
        }
        store_row(r0 + rr, v);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    // (D3DP_X2_SKEW_TOP == 0) value e of a leaving row is computed beside accumulator block e's twelve MFMAs (GELU: ~20 VALU
    // per value), the row is split and stored beside the last block's
    [[maybe_unused]] float ev[RPK][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int b = mi & 1;
      if (mi < 3) {                                    // next block's fragments while this one multiplies
        ah[b ^ 1] = *reinterpret_cast<const f16x8*>(sb + offA + (mi + 1) * 2048);
        al[b ^ 1] = *reinterpret_cast<const f16x8*>(sb + offAl + (mi + 1) * 2048);
      }
      if constexpr (JOB && !(D3DP_X2_SKEW_PROBE & 1) && !D3DP_X2_SKEW_TOP) {
#pragma unroll
        for (int rr = 0; rr < RPK; ++rr) {
          float x = value(r0 + rr, mi, bz);
          if constexpr (EPI == EPI_GELU) x = gelu_erf_rational(x);
          ev[rr][mi] = x;
        }
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b], wf[ni][1], acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[b], wf[ni][0], acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b], wf[ni][0], acc[mi][ni], 0, 0, 0);
      if constexpr (JOB && !(D3DP_X2_SKEW_PROBE & 1) && !D3DP_X2_SKEW_TOP) {
        if (mi == 3) {
#pragma unroll
          for (int rr = 0; rr < RPK; ++rr) store_row(r0 + rr, ev[rr]);
        }
      }
      if constexpr (JOB && (D3DP_X2_SKEW_PROBE & 1)) {
        if (mi == 3 && unscale == -12345.f) {          // (never true: keeps the parked values, and with them the MFMAs, live)
#pragma unroll
          for (int rr = 0; rr < RPK; ++rr) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = value(r0 + rr, e, bz);
            store_row(r0 + rr, v);
          }
        }
      }
#if D3DP_X2_SKEW_SGB
      // ask the scheduler for an even mix: one MFMA, then up to D3DP_X2_SKEW_SGB VALU, twelve times over
      if constexpr (JOB) {
#pragma unroll
        for (int q = 0; q < 12; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, D3DP_X2_SKEW_SGB, 0);
        }
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const int rest = NK - 4 * D;                         // k-steps of a round in which no class is between tiles
#pragma unroll 1
  for (int ti = 0; ti <= n_tiles; ++ti) {
    const int ngrp = ti < n_tiles ? 4 : 3;             // (the flush: classes 0..2 leave in 3 D k-steps; class 3 after the loop)
#pragma unroll 1
    for (int grp = 0; grp < ngrp; ++grp) {
      park_and_shift(grp, ti - 1);                     // class grp has just finished tile ti - 1
#pragma unroll
      for (int s = 0; s < D; ++s) kstep(std::true_type{}, s * RPK);   // (the row index must be a constant: park[e][r])
    }
    if (ti < n_tiles) {
#pragma unroll 1
      for (int s = 0; s < rest; ++s) kstep(std::false_type{}, 0);
    }
  }
  // the last class (3) of the last tile sits in accumulator block 0 by now: nothing left to hide it under
  {
    park_and_shift(3, n_tiles - 1);
    const float4 bz = *reinterpret_cast<const float4*>(bias4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = value(r, e, bz);
        if constexpr (EPI == EPI_GELU) v[e] = gelu_erf_rational(v[e]);
      }
      store_row(r, v);
    }
  }
}

#endif  // D3DP_X2_VARIANTS

// ---------------------------------------------------------------------------------------------------------------------
// The training step's Linear (forward, dgrad and the split-K wgrad of SURVEY.md row A13) on the same three-pass split-fp16
// scheme: out_z[M, N] = (A2 . W2^T over the k-steps of chunk z) x dynA[0] x dynW[0] (+ bias), fp32 out.
//   * operand scales are DEVICE values: gradients have no range known on the host, so every operand is split at the power
//     of two its own absmax asks for (split2h_dyn_kernel / split2h_t_dyn_kernel below write 1 / scale next to the planes) and
//     this kernel reads the two factors -- no host synchronisation anywhere in the step;
//   * Z chunks of the contraction (k-steps z NKz .. (z + 1) NKz - 1 of rows of 2 Kfull fp16) go to Z x tiles work items:
//     a weight gradient in its transposed-operand form (shapes the TN kernel below does not take) contracts over the 16,524
//     tokens of a batch into at most 24 output tiles, far too few for 256 CUs unsplit; the partial results are summed in a
//     fixed order by sum_partials_kernel (deterministic, unlike the fp32-atomic split-K of the fp32-MFMA path this replaces);
//   * rem_m0 (optional; Z = 1 launches): the rows behind the last whole 256-row tile as 16 x 64 blocks -- see the compute waves.
// Structure: the lock-step kernel above without the lagged MFMAs and with the plain fp32 epilogue (8 + 4 waves, 256 x 128 x 32
// tiles, 3-stage LDS-DMA ring, one barrier per k-step).
//   * amax_out (optional; Z = 1 launches): the output's absmax (amax_pos: its largest positive value), one atomicMax per
//     workgroup -- the operand scale of whatever split-fp16 kernel consumes the output next (the training attention kernels:
//     q / k / v and dO; the fc2 operand GELU(fc1 output): gelu_rowprep_kernel), without an absmax pass.
__global__ __launch_bounds__(768) void gemm_f16x2_dyn_kernel(const f16* __restrict__ A2, const f16* __restrict__ W2,
                                                             const float* __restrict__ bias, const float* __restrict__ dynA,
                                                             const float* __restrict__ dynW, float* __restrict__ out, int M,
                                                             int N, int Kfull, int NKz, int tiles_n, int tiles_per_z,
                                                             int total_items, unsigned* __restrict__ amax_out, int amax_pos,
                                                             int rem_m0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int G = gridDim.x;
  const int L = xcd_remap(blockIdx.x, G);
  const int n_my = (total_items - L + G - 1) / G;      // work items L, L+G, ...: item = z tiles_per_z + tile
  const int gtot = n_my * NKz;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (gtot <= 0) return;
  // remainder rows (rem_m0 > 0, Z = 1): blocks of 16 rows x 64 columns, block b to workgroup b mod G -- see the end of the kernel
  const int rem_cg = (N + 63) >> 6;
  const int rem_blocks = rem_m0 > 0 ? ((M - rem_m0 + 15) >> 4) * rem_cg : 0;

  if (wave >= XNCW) {
    const int lw = wave - XNCW;
    const int lr = lane >> 3, lq = lane & 7;
    int ti = 0, ks = 0, slot = 0;
    const f16* pa[8];
    const f16* pw[4];
    auto issue = [&]() {
      if (ks == 0) {
        const int item = L + ti * G;
        const int z = item / tiles_per_z, t = item - z * tiles_per_z;
        const int m0 = (t / tiles_n) * XBM, n0 = (t % tiles_n) * XBN;
        const size_t k0 = (size_t)z * NKz * (2 * XBK);   // first k-step of the chunk, in fp16 elements of an h2i row
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = (lw * 8 + i) * 8 + lr;
          pa[i] = A2 + (size_t)min(m0 + row, M - 1) * (2 * Kfull) + k0 + swz128(row, lq) * 8;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = (lw * 4 + i) * 8 + lr;
          const int wrow = (row & 64) + colperm(row & 63);
          pw[i] = W2 + (size_t)min(n0 + wrow, N - 1) * (2 * Kfull) + k0 + swz128(row, lq) * 8;
        }
      }
      char* base = smem + slot * XSTAGE;
      const int ko = ks * (2 * XBK);
#pragma unroll
      for (int i = 0; i < 8; ++i) __builtin_amdgcn_global_load_lds(GPTR(pa[i] + ko), LPTR(base + (lw * 8 + i) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds(GPTR(pw[i] + ko), LPTR(base + XA_BYTES + (lw * 4 + i) * 1024), 16, 0, 0);
      if (++ks == NKz) { ks = 0; ++ti; }
      slot = (slot == XNSTAGE - 1) ? 0 : slot + 1;
    };
    issue();
    if (gtot > 1) issue();
    // (the compute waves' remainder blocks, two barriers each: FIRST, while these two stages are in flight -- see below)
    for (int b = L; b < rem_blocks; b += G) { X2_BARRIER(); X2_BARRIER(); }
    for (int g = 0; g < gtot; ++g) {
      if (g + 1 < gtot) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      X2_BARRIER();
      if (g + 2 < gtot) issue();
    }
    if (amax_out) X2_BARRIER();                        // (the compute waves' absmax hand-over below)
    return;
  }

  const int wr = wave >> 1, wc = wave & 1;
  const int fi = lane & 15, fg = lane >> 4;
  const int offA = (wr * 64 + fi) * 128 + swz128(fi, fg) * 16, offAl = offA ^ 64;
  const int offW = XA_BYTES + (wc * 64 + fi) * 128 + swz128(fi, fg) * 16, offWl = offW ^ 64;
  const float unscale = dynA[0] * dynW[0];             // 1 / (scale of A x scale of W): powers of two, exact
  __builtin_amdgcn_s_setprio(1);
  float am = 0.f;
  int slot = 0;
  // The remainder rows.  The configs[4] batch has M = 16,524 = 64 x 256 + 140 rows: as a 65th row of 256 x 128 tiles the last 140
  // rows cost every forward / dgrad product a whole extra round of the persistent loop with 4 - 12 of the 256 CUs busy (12 us
  // of a 35 - 70 us product; round 4 sent them to a second, split-K launch plus a sum kernel where the contraction was long
  // enough -- no cheaper).  Here the tiles cover the first rem_m0 = 256 q rows only -- a whole number of rounds when q x tiles_n is
  // a multiple of the CU count, as 64 x {4, 8, 12} is -- and the rows behind them are cut into 16 x 64 blocks, one per workgroup
  // (140 rows x 512 .. 1536 columns: 72 .. 216 blocks, every CU busy once more for a few microseconds): the eight compute waves
  // split the contraction (NKz / 8 k-steps each, fragments straight from global memory -- 64 KiB per block, no LDS staging),
  // exchange their partial accumulators through LDS and wave 0 adds them in wave order (deterministic), applies scale and bias
  // and stores.  The blocks come FIRST: the loader waves have just requested the first two stages of the first tile and the
  // compute waves would wait for them anyway -- the exchange uses the third stage of the ring, which the loaders fill only
  // after the first k-step's barrier.  Same MFMA triple per k-step as the tiles, same fp32 accumulation: the only difference to a
  // tile's result is the grouping of the k-steps into eight partial sums.
  if (rem_blocks > 0) {
    const int ksw = NKz / XNCW;                        // k-steps per wave (launcher: NKz % 16 == 0, so an even number)
    f32x4* xch = reinterpret_cast<f32x4*>(smem + (XNSTAGE - 1) * XSTAGE);
#pragma unroll 1
    for (int b = L; b < rem_blocks; b += G) {
      const int m0 = rem_m0 + (b / rem_cg) * 16, nb0 = (b % rem_cg) * 64;
      const size_t kofs = (size_t)wave * ksw * (2 * XBK) + fg * 8;
      const f16* ap = A2 + (size_t)min(m0 + fi, M - 1) * (2 * Kfull) + kofs;
      const f16* wp[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) wp[ni] = W2 + (size_t)min(nb0 + 4 * fi + ni, N - 1) * (2 * Kfull) + kofs;
      f32x4 pacc[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) pacc[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int ks = 0; ks < ksw; ks += 2) {
        f16x8 ah[2], al[2], wh[2][4], wl[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int o = (ks + u) * (2 * XBK);
          ah[u] = *reinterpret_cast<const f16x8*>(ap + o);
          al[u] = *reinterpret_cast<const f16x8*>(ap + o + XBK);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            wh[u][ni] = *reinterpret_cast<const f16x8*>(wp[ni] + o);
            wl[u][ni] = *reinterpret_cast<const f16x8*>(wp[ni] + o + XBK);
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) pacc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], wl[u][ni], pacc[ni], 0, 0, 0);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) pacc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[u], wh[u][ni], pacc[ni], 0, 0, 0);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) pacc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], wh[u][ni], pacc[ni], 0, 0, 0);
        }
      }
      X2_BARRIER();                                    // nobody reads the previous block's partial sums any more
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) xch[(wave * 4 + ni) * 64 + lane] = pacc[ni];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      X2_BARRIER();
      if (wave == 0) {
        f32x4 sum[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          sum[ni] = xch[ni * 64 + lane];
#pragma unroll
          for (int w = 1; w < XNCW; ++w) sum[ni] += xch[(w * 4 + ni) * 64 + lane];
        }
        const int nb = nb0 + 4 * fi;
        if (nb < N) {
          float4 bz = {0.f, 0.f, 0.f, 0.f};
          if (bias != nullptr) bz = *reinterpret_cast<const float4*>(bias + nb);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = m0 + 4 * fg + r;
            if (m < M) {
              const f32x4 v = {fmaf(sum[0][r], unscale, bz.x), fmaf(sum[1][r], unscale, bz.y), fmaf(sum[2][r], unscale, bz.z),
                               fmaf(sum[3][r], unscale, bz.w)};
              *reinterpret_cast<f32x4*>(out + (size_t)m * N + nb) = v;
              am = amax_pos ? fmaxf(am, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])))
                            : fmaxf(am, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
            }
          }
        }
      }
    }
  }
#pragma unroll 1
  for (int ti = 0; ti < n_my; ++ti) {
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ks = 0; ks < NKz; ++ks) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      X2_BARRIER();
      const char* sb = smem + slot * XSTAGE;
      slot = (slot == XNSTAGE - 1) ? 0 : slot + 1;
      f16x8 wf[4][2];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          wf[ni][pl] = *reinterpret_cast<const f16x8*>(sb + (pl ? offWl : offW) + ni * 2048);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const f16x8 ah = *reinterpret_cast<const f16x8*>(sb + offA + mi * 2048);
        const f16x8 al = *reinterpret_cast<const f16x8*>(sb + offAl + mi * 2048);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wf[ni][1], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wf[ni][0], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wf[ni][0], acc[mi][ni], 0, 0, 0);
      }
    }
    // epilogue: lane holds out[pm0 + mi*16 + r][nb + ni]; one 16-byte store per (mi, r)
    const int item = L + ti * G;
    const int z = item / tiles_per_z, t = item - z * tiles_per_z;
    const int pm0 = (t / tiles_n) * XBM + wr * 64 + 4 * fg, nb = (t % tiles_n) * XBN + wc * 64 + 4 * fi;
    if (nb < N) {                                      // (N % 4 == 0: a lane's four columns exist together)
      float4 bz = {0.f, 0.f, 0.f, 0.f};
      if (bias != nullptr) bz = *reinterpret_cast<const float4*>(bias + nb);
      float* o = out + (size_t)z * M * N + nb;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = pm0 + mi * 16 + r;
          if (m < M) {
            const f32x4 v = {fmaf(acc[mi][0][r], unscale, bz.x), fmaf(acc[mi][1][r], unscale, bz.y),
                             fmaf(acc[mi][2][r], unscale, bz.z), fmaf(acc[mi][3][r], unscale, bz.w)};
            *reinterpret_cast<f32x4*>(o + (size_t)m * N) = v;
            // (amax_pos: the largest POSITIVE value instead of the largest magnitude -- what bounds GELU of this output)
            am = amax_pos ? fmaxf(am, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])))
                          : fmaxf(am, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
          }
        }
    }
  }
  if (amax_out) {                                      // (uniform: a kernel argument)
    float* part = reinterpret_cast<float*>(smem + XNSTAGE * XSTAGE);   // behind the ring, which slower waves may still be reading
    am = wave_max(am);
    if (lane == 0) part[wave] = am;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    X2_BARRIER();
    if (wave == 0 && lane == 0) {
      float m = part[0];
#pragma unroll
      for (int w = 1; w < XNCW; ++w) m = fmaxf(m, part[w]);
      atomicMax(amax_out, __float_as_uint(m));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The training step's weight gradient  dW[N, K] = dY[T, N]^T X[T, K]  straight from the ROW forms of its two operands (the "TN"
// form): out_z[n][k] = sum over the tokens t of chunk z of A2[t][n] W2[t][k], x dynA[0] x dynW[0].
// Round 4 fed wgrad the TRANSPOSED forms [features][2 Tp] through the kernel above, which every forward Linear's input and
// every dY had to be written out in a second time by a tile transpose (dyprep_kernel: 9.8 GB and 3.5 ms of the configs[4] step,
// its scattered 128-byte writes at 2.8 TB/s).  Here the contraction runs over ROWS of the operands: a k-step is 32 token rows,
// the A slab 32 rows x 256 features (1 KiB per row = 8 h2i blocks, ONE LDS-DMA wave-instruction), the W slab 32 rows x 128
// features (512 B per row, two rows per instruction) -- the same 48 KiB per stage, 12 pieces per loader wave and k-step as the
// kernel above, and the same 8 + 4 waves, 3-stage ring and barrier structure.  The fragments an MFMA wants -- 8 consecutive
// TOKENS of one feature -- are columns of the slab: ds_read_b64_tr_b16 (the transposed fragment read of attention.hip's V
// image: 4 token rows x 16 features per 16 lanes) delivers them; lane group g holds the tokens {4 g + j} and {16 + 4 g + j} of
// the k-step for BOTH operands, so the k order of the two fragments agrees.  LDS swizzle: 16-byte slot s of token row r sits at
// s ^ ((r & 7) << 1): the eight rows a transposed read touches per cycle land in eight different 32-byte bank groups (the rows
// are 1 KiB / 512 B apart: unswizzled they would all hit the same one).
// Rows T .. Tp - 1 of both operands must exist and be ZERO (dyprep_kernel writes them).  Output: a lane holds
// out[n = .. + 4 g + r][k = .. + 16 ni + i] -- 4-byte stores, 64 contiguous bytes per 16 lanes; the partial tiles are small (N K
// floats per chunk) and summed by sum_partials_kernel.
// Several products in one launch (D3dpTnTable, kernels.h): the four weight gradients of a block -- fc2, fc1, proj, qkv: 8 + 16 + 16 +
// 24 = 64 output tiles at configs[4] -- contract over the SAME token rows, so their tiles form one list and the split count is
// chosen for the list: Z = 4 chunks of 130 k-steps x 64 tiles = 256 equal work items, one per CU.  Launched one by one each
// product had to be cut into Z = 10 .. 32 chunks to fill the chip by itself: 32 MB of partial tiles written and read again per
// product (one 128 KiB tile per CU whatever the shape) and a pipeline ramp and a store phase per 17 .. 52 k-steps -- together about
// as long as the products' matrix work.  Merged, a block writes 32 MB of partial tiles ONCE and ramps once per 130 k-steps.
__device__ __forceinline__ int tn_find(const D3dpTnTable& tb, int t) {
  int i = 0;
#pragma unroll
  for (int j = 1; j < D3DP_TN_MAX; ++j)
    if (j < tb.n && t >= tb.p[j].tile0) i = j;
  return i;
}
__global__ __launch_bounds__(768) void gemm_f16x2_tn_kernel(D3dpTnTable tb, int NKz, int tiles_per_z, int total_items) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int G = gridDim.x;
  const int L = xcd_remap(blockIdx.x, G);
  const int n_my = (total_items - L + G - 1) / G;      // work items L, L+G, ...: item = z tiles_per_z + tile (of the merged list)
  const int gtot = n_my * NKz;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (gtot <= 0) return;

  if (wave >= XNCW) {
    // ---------------------------------------------------------------- loader waves: 8 A pieces + 4 W pieces per k-step each
    const int lw = wave - XNCW;
    int ti = 0, ks = 0, slot = 0;
    const f16* pa;                                     // this lane's source of A piece 0 of the current item (k-step 0)
    const f16* pw;
    size_t a_row = 0, w_row = 0;                       // fp16 elements per operand row
    auto issue = [&]() {
      if (ks == 0) {
        const int item = L + ti * G;
        const int z = item / tiles_per_z, tt = item - z * tiles_per_z;
        const D3dpTnProduct& P = tb.p[tn_find(tb, tt)];
        const int t = tt - P.tile0;
        const int n0 = (t / P.tiles_k) * XBM, k0 = (t % P.tiles_k) * XBN;
        const size_t t0 = (size_t)z * NKz * XBK;       // first token row of the chunk
        a_row = (size_t)2 * P.N; w_row = (size_t)2 * P.K;
        // A piece i = token row lw 8 + i of the k-step (1 KiB: features n0 .. n0 + 255); lane l fetches logical slot l ^ swz
        // (N % 256 == 0 and K % 128 == 0: the launcher sends other shapes to the kernel above)
        pa = (const f16*)P.A2 + (t0 + lw * 8) * a_row + (size_t)n0 * 2;
        pw = (const f16*)P.W2 + (t0 + lw * 8) * w_row + (size_t)k0 * 2;
      }
      char* base = smem + slot * XSTAGE;
      const size_t ro = (size_t)ks * XBK;              // token rows into the chunk
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = lw * 8 + i;                      // token row of the k-step; r & 7 == i
        __builtin_amdgcn_global_load_lds(GPTR(pa + (ro + i) * a_row + ((lane ^ (i << 1)) << 3)), LPTR(base + r * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = lw * 8 + 2 * i + (lane >> 5);    // two token rows per piece: lanes 0-31 / 32-63
        const int l32 = lane & 31;
        __builtin_amdgcn_global_load_lds(GPTR(pw + (ro + 2 * i + (lane >> 5)) * w_row + ((l32 ^ ((r & 7) << 1)) << 3)),
                                         LPTR(base + XA_BYTES + (lw * 8 + 2 * i) * 512), 16, 0, 0);
      }
      if (++ks == NKz) { ks = 0; ++ti; }
      slot = (slot == XNSTAGE - 1) ? 0 : slot + 1;
    };
    issue();
    if (gtot > 1) issue();
    for (int g = 0; g < gtot; ++g) {
      if (g + 1 < gtot) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      X2_BARRIER();
      if (g + 2 < gtot) issue();
    }
    return;
  }

  // -------------------------------------------------------------------- compute waves: 64 (n) x 64 (k) of the 256 x 128 tile
  const int wr = wave >> 1, wc = wave & 1;
  const int fi = lane & 15, fg = lane >> 4;
  const int j = fi >> 2, qd = fi & 3, row = 4 * fg + j;        // transposed read: lanes 4 j .. 4 j + 3 point at token row 4 fg + j
  // slot of the 16 features (mi / ni) x plane inside a token row: block b = wr 2 + (mi >> 1) (A) / wc 2 + (ni >> 1) (W),
  // slot = b 8 + plane 4 + (mi & 1) 2 + (qd >> 1); the swizzle of rows `row` and `row + 16` is the same (r & 7)
  auto a_off = [&](int mi, int pl) { return row * 1024 + (((((wr * 2 + (mi >> 1)) << 3) | (pl << 2) | ((mi & 1) << 1) | (qd >> 1)) ^ ((row & 7) << 1)) << 4) + ((qd & 1) << 3); };
  auto w_off = [&](int ni, int pl) { return XA_BYTES + row * 512 + (((((wc * 2 + (ni >> 1)) << 3) | (pl << 2) | ((ni & 1) << 1) | (qd >> 1)) ^ ((row & 7) << 1)) << 4) + ((qd & 1) << 3); };
  typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
  typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
  auto tr8 = [&](const char* p, int second) {         // 8 tokens of one feature: rows 4 fg + {0..3} and 16 + 4 fg + {0..3}
    const v4bf a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) v4bf*)(p));
    const v4bf b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) v4bf*)(p + second));
    return __builtin_bit_cast(f16x8, (v8bf){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
  };
  int offA[4][2], offW[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) { offA[m][pl] = a_off(m, pl); offW[m][pl] = w_off(m, pl); }
  __builtin_amdgcn_s_setprio(1);
  int slot = 0;
#pragma unroll 1
  for (int ti = 0; ti < n_my; ++ti) {
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ks = 0; ks < NKz; ++ks) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      X2_BARRIER();
      const char* sb = smem + slot * XSTAGE;
      slot = (slot == XNSTAGE - 1) ? 0 : slot + 1;
      f16x8 wf[4][2];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) wf[ni][pl] = tr8(sb + offW[ni][pl], 16 * 512);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const f16x8 ah = tr8(sb + offA[mi][0], 16 * 1024);
        const f16x8 al = tr8(sb + offA[mi][1], 16 * 1024);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wf[ni][1], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wf[ni][0], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wf[ni][0], acc[mi][ni], 0, 0, 0);
      }
    }
    // epilogue: lane holds out[nn = n0 + wr 64 + mi 16 + 4 fg + r][kk = k0 + wc 64 + ni 16 + fi]
    const int item = L + ti * G;
    const int z = item / tiles_per_z, tt = item - z * tiles_per_z;
    const D3dpTnProduct& P = tb.p[tn_find(tb, tt)];
    const int t = tt - P.tile0, K = P.K;
    const int n0 = (t / P.tiles_k) * XBM, k0 = (t % P.tiles_k) * XBN;
    const float unscale = P.dynA[0] * P.dynW[0];
    float* o = P.out + (size_t)z * P.N * K + (size_t)(n0 + wr * 64 + 4 * fg) * K + k0 + wc * 64 + fi;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) o[(size_t)(mi * 16 + r) * K + ni * 16] = acc[mi][ni][r] * unscale;
  }
}

// scale of a split operand from its absmax (bit pattern of a non-negative float, absmax_kernel): the power of two that
// puts the largest magnitude in [2^13, 2^14) (as capi.hip does for the inference weights); 1 for an all-zero tensor
__device__ __forceinline__ float dyn_scale(unsigned amax_bits) {
  const float m = __uint_as_float(amax_bits);
  if (!(m > 0.f) || !(m < INFINITY)) return 1.0f;
  int e;
  frexpf(m, &e);                                       // m = f 2^e, f in [0.5, 1)
  return ldexpf(1.0f, 14 - e);
}

// src [R][C] fp32 -> dst [R][2 Cpad] h2i (columns C .. Cpad - 1 zero), scale from the tensor's absmax; unscale[0] = 1 / scale.
// A thread owns 8 consecutive columns (C % 8 == 0, Cpad % 32 == 0): two 16-byte loads, one 16-byte store per plane.
__global__ void split2h_dyn_kernel(const float* __restrict__ s, f16* __restrict__ d, int R, int C, int Cpad,
                                   const unsigned* __restrict__ amax, float* __restrict__ unscale) {
  const float sc = dyn_scale(amax[0]);
  if (blockIdx.x == 0 && threadIdx.x == 0) unscale[0] = 1.0f / sc;
  const int G8 = Cpad / 8;
  const size_t n = (size_t)R * G8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / G8;
    const int c = (int)(i - r * G8) * 8;
    f16x8 hi = {}, lo = {};
    if (c < C) {
      const float4 a = *reinterpret_cast<const float4*>(s + r * C + c), b = *reinterpret_cast<const float4*>(s + r * C + c + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) { f16 h, l; split2h_scaled(v[e] * sc, h, l); hi[e] = h; lo[e] = l; }
    }
    f16* row = d + r * 2 * (size_t)Cpad + h2i_col(c);
    *reinterpret_cast<f16x8*>(row) = hi;
    *reinterpret_cast<f16x8*>(row + kH2iLo) = lo;
  }
}

// src [R][C] fp32 -> dst [C][2 Rpad] h2i: the TRANSPOSE as a split operand (rows R .. Rpad - 1 zero).  32 x 32 tiles through LDS;
// one tile = one 128-byte h2i block (hi of 32 source rows | lo of the same) of 32 destination rows.
__global__ __launch_bounds__(256) void split2h_t_dyn_kernel(const float* __restrict__ s, f16* __restrict__ d, int R, int C,
                                                            int Rpad, const unsigned* __restrict__ amax,
                                                            float* __restrict__ unscale) {
  __shared__ float tile[32][33];
  const float sc = dyn_scale(amax[0]);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) unscale[0] = 1.0f / sc;
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    tile[k][tx] = (r < R && c < C) ? s[(size_t)r * C + c] * sc : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k;                              // destination row; tx = source row within the block
    if (c < C) {
      f16 a, b;
      split2h_scaled(tile[tx][k], a, b);
      f16* blk = d + (size_t)c * 2 * Rpad + (size_t)blockIdx.x * 64;
      blk[tx] = a;
      blk[32 + tx] = b;
    }
  }
}

// One pass over an fp32 matrix src [R][C] for everything the training step needs from it as a split-fp16 operand: the row
// form [R][2 C] (a forward Linear's / dgrad's operand), the transposed form [C][2 Rpad] (wgrad's operand; rows R .. Rpad - 1
// zero) and, for a gradient dY, its column sums (the bias gradient) -- as three kernels (split2h_dyn, split2h_t_dyn, colsum)
// the same tensor was read three times.  blockIdx.y = a strip of 32 columns; a workgroup takes tiles of DY_ROWS rows x 32
// columns (one tile per workgroup at the configs[4] size: every workgroup resident at once): SIX 16-byte loads per thread in
// flight (round 4: one load per thread on 512 workgroups -- two per CU -- which left the kernel latency-bound at 2.7 TB/s),
// and every store instruction writes WHOLE 128-byte lines of both outputs: the row form as one 16-byte store per lane
// (neighbouring lanes swap their hi / lo halves by DPP: even lane -> 8 columns of hi, odd lane -> lo), the transposed form
// through LDS with lanes 0-3 of an 8-lane group writing the hi half and lanes 4-7 the lo half of ONE line (8 source rows of
// one destination row and plane per lane).  Column sums: this workgroup's rows into row blockIdx.x of `colpart` (gridDim.x
// rows of C floats, summed in a fixed order by d3dp_train_reduce_many: no float atomics).  C % 32 == 0, Rpad % 32 == 0.
constexpr int DY_SUB = 6, DY_ROWS = 32 * DY_SUB;      // 192 rows per tile
__global__ __launch_bounds__(256) void dyprep_kernel(const float* __restrict__ s, f16* __restrict__ drow, f16* __restrict__ dcol,
                                                     float* __restrict__ colpart, int R, int C, int Rpad,
                                                     const unsigned* __restrict__ amax, float* __restrict__ unscale) {
  __shared__ float tile[DY_ROWS][33];
  const float sc = dyn_scale(amax[0]);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) unscale[0] = 1.0f / sc;
  const int c0 = blockIdx.y * 32;
  const int tr = threadIdx.x >> 3, tj = threadIdx.x & 7, tq = tj * 4;   // phase 1: source row u 32 + tr, columns tq .. tq + 3
  const int nblk = Rpad / 32;                           // 32-row blocks of the transposed form
  const bool odd = tj & 1;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int rt = blockIdx.x; rt * DY_SUB < nblk; rt += gridDim.x) {
    float4 v[DY_SUB];
#pragma unroll
    for (int u = 0; u < DY_SUB; ++u) {
      const int r = rt * DY_ROWS + u * 32 + tr;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < R) v[u] = *reinterpret_cast<const float4*>(s + (size_t)r * C + c0 + tq);
    }
    __syncthreads();                                    // (the previous tile's transposed reads are done)
#pragma unroll
    for (int u = 0; u < DY_SUB; ++u) {
      const int r = rt * DY_ROWS + u * 32 + tr;
      acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
      const float vs[4] = {v[u].x * sc, v[u].y * sc, v[u].z * sc, v[u].w * sc};
      float* trow = &tile[u * 32 + tr][tq];
      trow[0] = vs[0]; trow[1] = vs[1]; trow[2] = vs[2]; trow[3] = vs[3];
      // row form: the 128-byte h2i block of row r, columns c0 .. c0 + 31 -- lanes 2 j / 2 j + 1 hold columns 8 j .. 8 j + 7:
      // the even lane stores both hi halves, the odd lane both lo halves (quad_perm [1,0,3,2]: the value of lane ^ 1)
      f16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) { f16 h, l; split2h_scaled(vs[e], h, l); hi[e] = h; lo[e] = l; }
      const uint2 hh = __builtin_bit_cast(uint2, hi), ll = __builtin_bit_cast(uint2, lo);
      const uint2 send = odd ? hh : ll;
      uint2 recv;
      recv.x = __builtin_amdgcn_update_dpp(0, (int)send.x, 0xB1, 0xF, 0xF, false);
      recv.y = __builtin_amdgcn_update_dpp(0, (int)send.y, 0xB1, 0xF, 0xF, false);
      using u32x4 = unsigned __attribute__((ext_vector_type(4)));
      const u32x4 o16 = odd ? (u32x4){recv.x, recv.y, ll.x, ll.y} : (u32x4){hh.x, hh.y, recv.x, recv.y};
      if (r < R) {
        f16* blk = drow + (size_t)r * 2 * C + (size_t)blockIdx.y * 64 + (tq & ~7) + (odd ? 32 : 0);
        *reinterpret_cast<u32x4*>(blk) = o16;
      }
    }
    __syncthreads();
    // transposed form: destination row c0 + tr; lanes tj 0-3 write the hi half, tj 4-7 the lo half of the line of block u
    {
      const int g8 = tj & 3;
      const bool lo_half = tj >= 4;
#pragma unroll
      for (int u = 0; u < DY_SUB; ++u) {
        if (rt * DY_SUB + u < nblk) {
          f16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = tile[u * 32 + g8 * 8 + e][tr];
            const f16 h = (f16)x;
            o[e] = lo_half ? (f16)(x - (float)h) : h;
          }
          f16* blk = dcol + (size_t)(c0 + tr) * 2 * Rpad + (size_t)(rt * DY_SUB + u) * 64 + (lo_half ? 32 : 0) + g8 * 8;
          *reinterpret_cast<f16x8*>(blk) = o;
        }
      }
    }
  }
  if (colpart) {
    __syncthreads();
    float* cs = &tile[0][0];                             // [32][33] of the tile buffer
    cs[tr * 33 + tq] = acc.x; cs[tr * 33 + tq + 1] = acc.y; cs[tr * 33 + tq + 2] = acc.z; cs[tr * 33 + tq + 3] = acc.w;
    __syncthreads();
    if (threadIdx.x < 32) {
      float t = 0.f;
#pragma unroll 8
      for (int j = 0; j < 32; ++j) t += cs[j * 33 + threadIdx.x];
      colpart[(size_t)blockIdx.x * C + c0 + threadIdx.x] = t;
    }
  }
}

// The ROW form alone, as a streaming pass over whole rows (the TN weight-gradient kernel above reads both of its operands in
// this form, so nothing is transposed): src [R][C] fp32 -> drow [Rpad][2 C] h2i (rows R .. Rpad - 1 ZERO: the TN kernel
// contracts over them) and, for a gradient, its column sums as gridDim.x partial rows of C floats (fixed-order reduction by
// d3dp_train_reduce_many).  A thread owns 8 consecutive columns (two 16-byte loads, one 16-byte store per plane) of the rows
// r = 4 blockIdx.x + (tid >> 6), + 4 gridDim.x, ...: reads and writes are contiguous in memory -- unlike the tile transpose of
// dyprep_kernel, whose 128-byte pieces land 66 KB apart.  C % 32 == 0, C <= 1536 (P = ceil(C / 512) column passes per row).
// mask (optional): per-sample DropPath scales -- row r is multiplied by mask[sample(r)] first (sample = r / J on the spatial
// axis, (r / (F J)) J + r % J on the temporal one): the backward pass then never stores the scaled gradient, only its absmax.
// GELU (gelu_pre [R][C]: the fc1 output of the forward pass): src is d hidden and the operand is d h_pre = src x gelu'(gelu_pre),
// formed here instead of by a pass of its own (gelu_bwd_kernel: a read of both tensors and a write of the product that only this
// kernel read again).  `amax` then holds the absmax of SRC (left by the fc2 dgrad's epilogue) and the operand scale follows
// from the bound |gelu'| <= 1.13: at most one binade below the scale of the true absmax whenever the largest |src| sits where
// gelu' >= 0.57, and never an overflow -- the two fp16 planes keep 22 bits below whatever power of two is chosen.
template <int P, bool GELU>
__global__ __launch_bounds__(256) void rowprep_kernel(const float* __restrict__ s, f16* __restrict__ drow, float* __restrict__ colpart,
                                                      int R, int Rpad, int C, const unsigned* __restrict__ amax,
                                                      float* __restrict__ unscale, const float* __restrict__ mask, int axis, int F,
                                                      int J, const float* __restrict__ gelu_pre) {
  __shared__ float cs[4][P * 512];
  const float sc = GELU ? dyn_scale(__float_as_uint(__uint_as_float(amax[0]) * kGeluGradMax)) : dyn_scale(amax[0]);
  if (blockIdx.x == 0 && threadIdx.x == 0) unscale[0] = 1.0f / sc;
  const int l = threadIdx.x & 63, rl = threadIdx.x >> 6;
  float acc[P][8];
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[p][e] = 0.f;
  constexpr int U = P == 1 ? 4 : 2;                    // rows per wave and iteration: 8 loads of 16 bytes in flight per thread
  const int rstep = gridDim.x * 4;
  for (int r0 = blockIdx.x * 4 + rl; r0 < Rpad; r0 += rstep * U) {
    float4 a[U][P], b[U][P], ga[GELU ? U : 1][P], gb[GELU ? U : 1][P];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int r = r0 + u * rstep, c = p * 512 + l * 8;
        a[u][p] = make_float4(0.f, 0.f, 0.f, 0.f); b[u][p] = a[u][p];
        if constexpr (GELU) { ga[u][p] = a[u][p]; gb[u][p] = a[u][p]; }
        if (r < R && c < C) {
          a[u][p] = *reinterpret_cast<const float4*>(s + (size_t)r * C + c);
          b[u][p] = *reinterpret_cast<const float4*>(s + (size_t)r * C + c + 4);
          if constexpr (GELU) {
            ga[u][p] = *reinterpret_cast<const float4*>(gelu_pre + (size_t)r * C + c);
            gb[u][p] = *reinterpret_cast<const float4*>(gelu_pre + (size_t)r * C + c + 4);
          }
        }
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int r = r0 + u * rstep, c = p * 512 + l * 8;
        if (r < Rpad && c < C) {
          float mk = 1.0f;
          if (mask && r < R) mk = mask[axis == 0 ? r / J : (r / (F * J)) * J + r % J];
          float v[8] = {a[u][p].x * mk, a[u][p].y * mk, a[u][p].z * mk, a[u][p].w * mk,
                        b[u][p].x * mk, b[u][p].y * mk, b[u][p].z * mk, b[u][p].w * mk};
          if constexpr (GELU) {
            const float g[8] = {ga[u][p].x, ga[u][p].y, ga[u][p].z, ga[u][p].w, gb[u][p].x, gb[u][p].y, gb[u][p].z, gb[u][p].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= gelu_grad_rational(g[e]);
          }
          f16x8 hi, lo;
#pragma unroll
          for (int e = 0; e < 8; ++e) { acc[p][e] += v[e]; f16 h, lw; split2h_scaled(v[e] * sc, h, lw); hi[e] = h; lo[e] = lw; }
          f16* row = drow + (size_t)r * 2 * C + h2i_col(c);
          *reinterpret_cast<f16x8*>(row) = hi;
          *reinterpret_cast<f16x8*>(row + kH2iLo) = lo;
        }
      }
  }
  if (colpart) {
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int e = 0; e < 8; ++e) cs[rl][p * 512 + l * 8 + e] = acc[p][e];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
      colpart[(size_t)blockIdx.x * C + c] = ((cs[0][c] + cs[1][c]) + cs[2][c]) + cs[3][c];
  }
}

// The row form of a LayerNorm's OUTPUT from its INPUT: drow [Rpad][2 C] = split(LN(src [R][C]; w, b, eps)) (rows R .. Rpad - 1
// zero), the scale from `amax` -- which the kernel that produced src left there as the absmax of this very LayerNorm output,
// computed on the fly and not stored.  The forward pass of the training step then never writes the fp32 normalised activation
// (its only reader was this operand pass).  One wave per row, a lane owns 8 consecutive columns (C <= 512, C % 8 == 0).
__global__ __launch_bounds__(256) void rowprep_ln_kernel(const float* __restrict__ s, const float* __restrict__ w,
                                                         const float* __restrict__ b, float eps, f16* __restrict__ drow, int R,
                                                         int Rpad, int C, const unsigned* __restrict__ amax,
                                                         float* __restrict__ unscale) {
  const float sc = dyn_scale(amax[0]);
  if (blockIdx.x == 0 && threadIdx.x == 0) unscale[0] = 1.0f / sc;
  const int l = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = l * 8;
  const bool on = c < C;
  float wl[8], bl[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { wl[e] = on ? w[c + e] : 0.f; bl[e] = on ? b[c + e] : 0.f; }
  const int rstep = gridDim.x * 4;
  const float invC = 1.0f / (float)C;
  constexpr int U = 4;                                   // rows per wave and iteration: their loads and reductions interleave
  for (int r0 = blockIdx.x * 4 + rl; r0 < Rpad; r0 += U * rstep) {
    float v[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + u * rstep;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), bb = a;
      if (r < R && on) {
        a = *reinterpret_cast<const float4*>(s + (size_t)r * C + c);
        bb = *reinterpret_cast<const float4*>(s + (size_t)r * C + c + 4);
      }
      v[u][0] = a.x; v[u][1] = a.y; v[u][2] = a.z; v[u][3] = a.w; v[u][4] = bb.x; v[u][5] = bb.y; v[u][6] = bb.z; v[u][7] = bb.w;
    }
    float mean[U], rstd[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float sm = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) sm += v[u][e];
      mean[u] = sm;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int u = 0; u < U; ++u) mean[u] += __shfl_xor(mean[u], o, 64);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      mean[u] *= invC;
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = on ? v[u][e] - mean[u] : 0.f; q = fmaf(d, d, q); }
      rstd[u] = q;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int u = 0; u < U; ++u) rstd[u] += __shfl_xor(rstd[u], o, 64);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + u * rstep;
      if (r < Rpad && on) {
        const float rs = 1.0f / sqrtf(rstd[u] * invC + eps);
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float y = r < R ? fmaf((v[u][e] - mean[u]) * rs, wl[e], bl[e]) : 0.f;
          f16 h, lw;
          split2h_scaled(y * sc, h, lw);
          hi[e] = h; lo[e] = lw;
        }
        f16* row = drow + (size_t)r * 2 * C + h2i_col(c);
        *reinterpret_cast<f16x8*>(row) = hi;
        *reinterpret_cast<f16x8*>(row + kH2iLo) = lo;
      }
    }
  }
}

// The fc2 operand of the training step straight from the fc1 output:  drow [Rpad][2 C] (h2i, rows R .. Rpad - 1 zero) =
// split(GELU(src)) -- without the fp32 hidden tensor in between (round 4: gelu_fwd wrote it, the operand pass read it again).
// The operand scale needs max |GELU(x)| BEFORE the first element is written; it follows from the largest positive x alone,
// which the fc1 Linear's epilogue leaves in `pmax` (gemm_f16x2_dyn_kernel, amax_pos): GELU is increasing on x > -0.75 and
// |GELU(x)| <= 0.17 for x < 0, so max |GELU| = max(GELU(pmax), at most 0.17) -- exact whenever GELU(pmax) >= 0.17, and the
// same power of two otherwise unless every activation is tiny.  Writes amax_out[0] (the bits of that bound) and unscale[0].
__global__ __launch_bounds__(256) void gelu_rowprep_kernel(const float* __restrict__ s, f16* __restrict__ drow, int R, int Rpad,
                                                           int C, const unsigned* __restrict__ pmax, unsigned* __restrict__ amax_out,
                                                           float* __restrict__ unscale) {
  const float pm = __uint_as_float(pmax[0]);
  const float bound = fmaxf(gelu_erf_rational(pm), 0.17004f);
  const float sc = dyn_scale(__float_as_uint(bound));
  if (blockIdx.x == 0 && threadIdx.x == 0) { unscale[0] = 1.0f / sc; amax_out[0] = __float_as_uint(bound); }
  const int G8 = C / 8;
  const size_t n = (size_t)Rpad * G8, stride = (size_t)gridDim.x * 256;
  for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 2 * stride) {
    float4 a[2], b[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const size_t i = i0 + u * stride;
      const size_t r = i / G8;
      const int c = (int)(i - r * G8) * 8;
      a[u] = make_float4(0.f, 0.f, 0.f, 0.f); b[u] = a[u];
      if (i < n && r < (size_t)R) {
        a[u] = *reinterpret_cast<const float4*>(s + r * C + c);
        b[u] = *reinterpret_cast<const float4*>(s + r * C + c + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const size_t i = i0 + u * stride;
      if (i < n) {
        const size_t r = i / G8;
        const int c = (int)(i - r * G8) * 8;
        const float v[8] = {a[u].x, a[u].y, a[u].z, a[u].w, b[u].x, b[u].y, b[u].z, b[u].w};
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float g = r < (size_t)R ? gelu_erf_rational(v[e]) : 0.f;
          f16 h, l;
          split2h_scaled(g * sc, h, l);
          hi[e] = h; lo[e] = l;
        }
        f16* row = drow + r * 2 * (size_t)C + h2i_col(c);
        *reinterpret_cast<f16x8*>(row) = hi;
        *reinterpret_cast<f16x8*>(row + kH2iLo) = lo;
      }
    }
  }
}

// ---- the training step's weight operands, all Linears of the model in three launches ------------------------------------
// (per-weight launches of absmax / split2h_dyn / split2h_t_dyn: 192 launches of 5 - 6 us for 100 MB of weights per step)
// blockIdx.y = the weight; a workgroup row strides over that weight only.
__global__ __launch_bounds__(256) void wprep_absmax_kernel(D3dpWPrepTable tb, unsigned* __restrict__ amax) {
  __shared__ float part[4];
  const D3dpWPrepItem it = tb.it[blockIdx.y];
  const size_t n4 = (size_t)it.N * it.K / 4, stride = (size_t)gridDim.x * blockDim.x;
  const float4* s4 = reinterpret_cast<const float4*>(it.w);
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = s4[i];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(amax + it.slot, __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))));
}
// rows form [N][2 K] (the forward operand) of every weight: the loop of split2h_dyn_kernel
__global__ __launch_bounds__(256) void wprep_rows_kernel(D3dpWPrepTable tb, f16* __restrict__ base, const unsigned* __restrict__ amax,
                                                         float* __restrict__ unscale) {
  const D3dpWPrepItem it = tb.it[blockIdx.y];
  const float sc = dyn_scale(amax[it.slot]);
  if (blockIdx.x == 0 && threadIdx.x == 0) unscale[it.slot] = 1.0f / sc;
  f16* d = base + 2 * it.off;
  const int G8 = it.K / 8;
  const size_t n = (size_t)it.N * G8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / G8;
    const int c = (int)(i - r * G8) * 8;
    const float4 a = *reinterpret_cast<const float4*>(it.w + r * it.K + c), b = *reinterpret_cast<const float4*>(it.w + r * it.K + c + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) { f16 h, l; split2h_scaled(v[e] * sc, h, l); hi[e] = h; lo[e] = l; }
    f16* row = d + r * 2 * (size_t)it.K + h2i_col(c);
    *reinterpret_cast<f16x8*>(row) = hi;
    *reinterpret_cast<f16x8*>(row + kH2iLo) = lo;
  }
}
// transposed form [K][2 N] (the dgrad operand) of every weight: the tiles of split2h_t_dyn_kernel, strided over by a
// workgroup row (N % 32 == 0, K % 32 == 0)
__global__ __launch_bounds__(256) void wprep_cols_kernel(D3dpWPrepTable tb, f16* __restrict__ base, const unsigned* __restrict__ amax) {
  __shared__ float tile[32][33];
  const D3dpWPrepItem it = tb.it[blockIdx.y];
  const float sc = dyn_scale(amax[it.slot]);
  f16* d = base + 2 * it.off;
  const int tr = it.N / 32, tc = it.K / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int tI = blockIdx.x; tI < tr * tc; tI += gridDim.x) {
    const int r0 = (tI / tc) * 32, c0 = (tI % tc) * 32;
    __syncthreads();
    for (int k = ty; k < 32; k += 8) tile[k][tx] = it.w[(size_t)(r0 + k) * it.K + c0 + tx] * sc;
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
      f16 a, b;
      split2h_scaled(tile[tx][k], a, b);
      f16* blk = d + (size_t)(c0 + k) * 2 * it.N + (size_t)(r0 / 32) * 64;
      blk[tx] = a;
      blk[32 + tx] = b;
    }
  }
}

// out[i] = sum_z part[z n + i] in the order z = 0, 1, ... (the deterministic end of a split-K product)
// A thread owns four consecutive outputs (n % 4 == 0) and keeps eight partials in flight: the partial tiles of a weight gradient
// are 32 MB (one 256 x 128 tile per CU) whatever its shape, and the one-load-at-a-time form of this kernel read them at 1.9 TB/s.
__global__ __launch_bounds__(256) void sum_partials_kernel(const float4* __restrict__ part, float4* __restrict__ out, size_t n4, int Z) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4* p = part + i;
    float4 a = p[0];
    int z = 1;
    for (; z + 8 <= Z; z += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(z + u) * n4];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    for (; z < Z; ++z) {
      const float4 v = p[(size_t)z * n4];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    out[i] = a;
  }
}

// the same for up to D3DP_TN_MAX products at once (blockIdx.y = the product; Z partials each, as d3dp_launch_linear_f16x2_tn_many leaves them)
__global__ __launch_bounds__(256) void sum_partials_many_kernel(D3dpSumTable tb) {
  const float4* part = reinterpret_cast<const float4*>(tb.part[blockIdx.y]);
  float4* out = reinterpret_cast<float4*>(tb.out[blockIdx.y]);
  const size_t n4 = tb.n4[blockIdx.y];
  const int Z = tb.Z;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4* p = part + i;
    float4 a = p[0];
    int z = 1;
    for (; z + 4 <= Z; z += 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = p[(size_t)(z + u) * n4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    for (; z < Z; ++z) {
      const float4 v = p[(size_t)z * n4];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    out[i] = a;
  }
}

// out[r][c] = sum_z part[z][r][c] + bias[c]   (the split-K remainder rows of a training Linear; bias may be null)
// (amax: optional absmax slot of these rows, as gemm_f16x2_dyn_kernel's)
__global__ __launch_bounds__(256) void sum_partials_bias_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                                float* __restrict__ out, size_t n, int N, int Z,
                                                                unsigned* __restrict__ amax, int amax_pos) {
  __shared__ float pm[4];
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float a = part[i];
    for (int z = 1; z < Z; ++z) a += part[(size_t)z * n + i];
    a = bias ? a + bias[i % N] : a;
    out[i] = a;
    m = fmaxf(m, amax_pos ? a : fabsf(a));
  }
  if (amax) {
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) pm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(amax, __float_as_uint(fmaxf(fmaxf(pm[0], pm[1]), fmaxf(pm[2], pm[3]))));
  }
}

// src[i] * scale -> h2i layout (common.h): blocks of 32 elements, dst[64 b .. 64 b + 31] = hi, dst[64 b + 32 .. 64 b + 63] = lo
// of elements 32 b .. 32 b + 31 (n % 32 == 0; any row length that is a multiple of 32)
__global__ void split2h_kernel(const float* __restrict__ s, f16* __restrict__ d, size_t n, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    f16 a, b;
    split2h_scaled(s[i] * scale, a, b);
    const size_t o = (i >> 5) * 64 + (i & 31);
    d[o] = a; d[o + 32] = b;
  }
}

// out[0] = max |src[i]| (as the bit pattern of a non-negative float: integer max == float max); out pre-zeroed.
// One atomic per WORKGROUP (the training step calls this three hundred times per step: with one atomic per wave the 4,096
// same-address atomics of a 34 MB tensor took 56 us, five times the read itself -- profiles/r04_train_step_kernel_stats.md)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ s, size_t n, unsigned* __restrict__ out) {
  __shared__ float part[4];
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float m = 0.f;
  const size_t n4 = n / 4;
  const float4* s4 = reinterpret_cast<const float4*>(s);   // (operands are 16-byte aligned: whole tensors)
  auto amax4 = [](const float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); };
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {         // four independent 16-byte loads in flight per thread
    const float4 v0 = s4[i], v1 = s4[i + stride], v2 = s4[i + 2 * stride], v3 = s4[i + 3 * stride];
    m = fmaxf(m, fmaxf(fmaxf(amax4(v0), amax4(v1)), fmaxf(amax4(v2), amax4(v3))));
  }
  for (; i < n4; i += stride) m = fmaxf(m, amax4(s4[i]));
  for (size_t j = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) m = fmaxf(m, fabsf(s[j]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))));
}

// see kernels.h (d3dp_launch_rowbound): one wave per row of W; fp32 sums of non-negative terms, rounded UP by a margin
// that covers their accumulated rounding (K <= 2048 terms: relative error < 2^-12)
__global__ void rowbound_kernel(const float* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ bias, int N, int K, float sq, unsigned* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (n >= N) return;
  float a = 0.f, inmax = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float in = fmaf(sq, fabsf(gamma[k]), fabsf(beta[k]));
    inmax = fmaxf(inmax, in);
    a = fmaf(fabsf(W[(size_t)n * K + k]), in, a);
  }
  a = wave_sum(a);
  inmax = wave_max(inmax);
  if (lane == 0) {
    atomicMax(out, __float_as_uint((a + fabsf(bias[n])) * 1.001f));
    atomicMax(out + 1, __float_as_uint(inmax * 1.001f));
  }
}

#if D3DP_X2_VARIANTS
// rowstat[m] = (mean, 1 / sqrt(var + eps)) of row m from its S slices of 64 (mean_i, M2_i): mean = avg of means,
// M2 = sum M2_i + 64 sum (mean_i - mean)^2 (the exact pairwise update for equal counts), var = M2 / (64 S)
__global__ void ln_combine_kernel(const float* __restrict__ sl, float* __restrict__ rowstat, int M, int S, float eps) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float2* p = reinterpret_cast<const float2*>(sl) + (size_t)m * S;
  float mean = 0.f;
  for (int i = 0; i < S; ++i) mean += p[i].x;
  mean *= 1.0f / (float)S;
  float m2 = 0.f;
  for (int i = 0; i < S; ++i) { const float d = p[i].x - mean; m2 += fmaf(64.0f * d, d, p[i].y); }
  const float rstd = 1.0f / sqrtf(m2 * (1.0f / (64.0f * (float)S)) + eps);
  reinterpret_cast<float2*>(rowstat)[m] = make_float2(mean, rstd);
}

// one wave per output row n: Wp[n][k] = W[n][k] gamma[k]; c12[n] = sum_k W[n][k] beta[k] + bias[n]; c12[N + n] = sum_k Wp[n][k]
// (sums in fp64: they stand in for fp32 dot products of the reference, and are computed once per weight load)
__global__ void fold_ln_kernel(const float* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ bias, float* __restrict__ Wp, float* __restrict__ c12, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (n >= N) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = lane; k < K; k += 64) {
    const float w = W[(size_t)n * K + k], wp = w * gamma[k];
    Wp[(size_t)n * K + k] = wp;
    s1 += (double)wp;
    s2 += (double)w * (double)beta[k];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  if (lane == 0) { c12[n] = (float)(s2 + (double)bias[n]); c12[N + n] = (float)s1; }
}

#endif  // D3DP_X2_VARIANTS
__global__ void nonfinite_flag_kernel(const float* __restrict__ x, size_t n, unsigned* __restrict__ flag) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (; i < n; i += stride) bad |= !(fabsf(x[i]) <= 3.0e38f);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

}  // namespace

// out = epi((A W^T) unscale + bias) with A2/W2 split-fp16 operands in the h2i layout; EPI_BIAS: fp32 `outf`; EPI_GELU: `out2` in the h2i layout [M][N] (the fc2 operand);
// EPI_QKV_PACK (N = 3 C, C % 64 == 0): `outf` rows of 12 C bytes = q fp32 | k hi | k lo | v hi | v lo (fp16 x 16);
// EPI_RESID: `outf` [M, N] fp32 is read and written (outf += ...).
// `unscale` = 1 / (scale of the A planes * scale of the W planes).
// K must be a multiple of 64: the k-loop is unrolled by two k-steps of 32 (the lagged products alternate between two
// register sets), and the loader / compute waves count barriers per k-step.
// skew_d: 0 = the plain schedule; 1, 2, 4 = the row-class skewed kernel with its parked class leaving in that many k-steps
// (EPI_QKV_PACK and EPI_GELU only; falls back to the plain kernel where the schedule does not apply: see d3dp_x2_skew_applies)
bool d3dp_x2_variants_built() { return D3DP_X2_VARIANTS != 0; }

bool d3dp_x2_skew_applies(int epi, int M, int N, int K, int skew_d, int n_cu) {
  if (!D3DP_X2_VARIANTS) return false;
  if (skew_d != 1 && skew_d != 2 && skew_d != 4) return false;
  if (epi != EPI_QKV_PACK && epi != EPI_GELU) return false;
  const int tn = (N + XBN - 1) / XBN, tm = (M + XBM - 1) / XBM;
  if (K % XBK != 0 || K / XBK < 4 * skew_d) return false;   // four classes, D k-steps apart, inside one tile round
  if (N % XBN != 0) return false;                        // whole strips (the denoiser's 1536 and 1024)
  (void)tm;
  return n_cu / tn >= 1;                                 // (any M: with fewer row tiles than row groups, fewer groups work --
                                                         //  the schedule, and with it the summation order, must not depend on M)
}

int d3dp_launch_linear_f16x2(int epi, const void* A2, const void* W2, const float* bias, float unscale, float oscale,
                             float* outf, void* out2, float* aux, unsigned* flag, int M, int N, int K, hipStream_t st,
                             int skew_d, int pingpong) {
  const bool reverse = (pingpong & X2_TILES_LAST_TO_FIRST) != 0;   // (the plain kernel only)
  pingpong &= 0xff;
  if (K % (2 * XBK) != 0 || N % 4 != 0 || N > XBIAS_MAX || M <= 0) return -1;
  if ((size_t)M * N * 4 >= ((size_t)1 << 32)) return -1;   // 32-bit byte offsets in the epilogue
  if (epi != EPI_BIAS && epi != EPI_GELU && epi != EPI_QKV_PACK && epi != EPI_RESID && epi != EPI_RESID_LN &&
      epi != EPI_GELU_LN) return -1;
  if ((epi == EPI_RESID_LN || epi == EPI_GELU_LN) && (!aux || !out2)) return -1;
  if (epi == EPI_RESID_LN && !flag) return -1;
  if (epi == EPI_RESID_LN && N % 64 != 0) return -1;    // whole 64-column slices: every compute wave's columns exist
  if (epi == EPI_GELU_LN && 2 * N > XBIAS_MAX) return -1;   // c2 | c1 in the bias area
  if (epi == EPI_QKV_PACK && (N % 3 != 0 || (N / 3) % 64 != 0)) return -1;
  // h2i output rows are whole 32-column blocks of [hi | lo]: with N % 32 != 0 the lo half of the last block would land in
  // the next row; EPI_GELU_LN keeps two row-statistics buffers, which a k-loop shorter than the 3-stage
  // ring lets the loaders overwrite while the previous tile's epilogue reads them
  if ((epi == EPI_GELU || epi == EPI_GELU_LN || epi == EPI_RESID_LN) && N % 32 != 0) return -1;
  if (epi == EPI_GELU_LN && K / XBK < XNSTAGE) return -1;
  const int tm = (M + XBM - 1) / XBM, tn = (N + XBN - 1) / XBN;
  using KernT = void (*)(const f16*, const f16*, const float*, float, float, float*, f16*, float*, unsigned*, int, int, int, int, int);
#if D3DP_X2_VARIANTS
  constexpr int NKERN = 6;
  static const KernT kerns[NKERN] = {gemm_f16x2_kernel<EPI_BIAS, 0>, gemm_f16x2_kernel<EPI_BIAS, 1>,
                                     gemm_f16x2_kernel<EPI_GELU, 0>, gemm_f16x2_kernel<EPI_RESID, 0>,
                                     gemm_f16x2_kernel<EPI_RESID_LN, 0>, gemm_f16x2_kernel<EPI_GELU_LN, 0>};
#else
  // the product library: the four epilogues the denoiser runs; everything else is a -DD3DP_X2_VARIANTS=1 build
  if (epi == EPI_RESID_LN || epi == EPI_GELU_LN || skew_d != 0 || pingpong != 0) return -2;
  constexpr int NKERN = 4;
  static const KernT kerns[NKERN] = {gemm_f16x2_kernel<EPI_BIAS, 0>, gemm_f16x2_kernel<EPI_BIAS, 1>,
                                     gemm_f16x2_kernel<EPI_GELU, 0>, gemm_f16x2_kernel<EPI_RESID, 0>};
#endif
  // per DEVICE: the 156 KiB dynamic-LDS opt-in of every instantiation and the CU count (one process may drive several
  // devices: nn.DataParallel callers)
  static PerDeviceOnce once;
  const int cus = once.get([&](int dev) {
    for (int k = 0; k < NKERN; ++k)
      if (d3dp_lds_opt_in(reinterpret_cast<const void*>(kerns[k]), XLDS) < 0) return -3;
    return d3dp_cu_count(dev);
  });
  if (cus < 0) return -3;
  const int total = tm * tn, grid = total < cus ? total : cus;
#if D3DP_X2_VARIANTS
  if (d3dp_x2_skew_applies(epi, M, N, K, skew_d, cus)) {
    using SkewT = void (*)(const f16*, const f16*, const float*, float, float, float*, f16*, int, int, int, int, int, int);
    static const SkewT skews[6] = {gemm_f16x2_skew_kernel<EPI_BIAS, 1, 1>, gemm_f16x2_skew_kernel<EPI_BIAS, 1, 2>,
                                   gemm_f16x2_skew_kernel<EPI_BIAS, 1, 4>, gemm_f16x2_skew_kernel<EPI_GELU, 0, 1>,
                                   gemm_f16x2_skew_kernel<EPI_GELU, 0, 2>, gemm_f16x2_skew_kernel<EPI_GELU, 0, 4>};
    static PerDeviceOnce once_skew;
    if (once_skew.get([&](int) {
          for (int k = 0; k < 6; ++k)
            if (d3dp_lds_opt_in(reinterpret_cast<const void*>(skews[k]), XLDS) < 0) return -3;
          return 1;
        }) < 0) return -3;
    const int Q = cus / tn < tm ? cus / tn : tm;         // row groups: Q tn workgroups work, the others idle
    const SkewT kern = skews[(epi == EPI_GELU ? 3 : 0) + (skew_d == 1 ? 0 : skew_d == 2 ? 1 : 2)];
    hipLaunchKernelGGL(kern, dim3(cus), dim3((XNCW + 4) * 64), XLDS, st, (const f16*)A2, (const f16*)W2, bias, unscale, oscale,
                       outf, (f16*)out2, M, N, K, tn, tm, Q);
    return 0;
  }
  if (pingpong == 2 && N % WBN == 0 && (epi == EPI_BIAS || epi == EPI_QKV_PACK || epi == EPI_GELU || epi == EPI_RESID)) {
    static const KernT wides[4] = {gemm_f16x2_wide_kernel<EPI_BIAS, 0>, gemm_f16x2_wide_kernel<EPI_BIAS, 1>,
                                   gemm_f16x2_wide_kernel<EPI_GELU, 0>, gemm_f16x2_wide_kernel<EPI_RESID, 0>};
    static PerDeviceOnce once_wide;
    if (once_wide.get([&](int) {
          for (int k = 0; k < 4; ++k)
            if (d3dp_lds_opt_in(reinterpret_cast<const void*>(wides[k]), WLDS) < 0) return -3;
          return 1;
        }) < 0) return -3;
    const int tw = N / WBN, totw = tm * tw;
    const KernT kern = wides[epi == EPI_GELU ? 2 : epi == EPI_RESID ? 3 : epi == EPI_QKV_PACK ? 1 : 0];
    hipLaunchKernelGGL(kern, dim3(totw < cus ? totw : cus), dim3(512), WLDS, st, (const f16*)A2, (const f16*)W2, bias, unscale,
                       oscale, outf, (f16*)out2, aux, flag, M, N, K, tw, totw);
    return 0;
  }
  if (pingpong == 1 && (epi == EPI_BIAS || epi == EPI_QKV_PACK || epi == EPI_GELU || epi == EPI_RESID)) {
    static const KernT pps[4] = {gemm_f16x2_pp_kernel<EPI_BIAS, 0>, gemm_f16x2_pp_kernel<EPI_BIAS, 1>,
                                 gemm_f16x2_pp_kernel<EPI_GELU, 0>, gemm_f16x2_pp_kernel<EPI_RESID, 0>};
    static PerDeviceOnce once_pp;
    if (once_pp.get([&](int) {
          for (int k = 0; k < 4; ++k)
            if (d3dp_lds_opt_in(reinterpret_cast<const void*>(pps[k]), XLDS) < 0) return -3;
          return 1;
        }) < 0) return -3;
    const KernT kern = pps[epi == EPI_GELU ? 2 : epi == EPI_RESID ? 3 : epi == EPI_QKV_PACK ? 1 : 0];
    hipLaunchKernelGGL(kern, dim3(grid), dim3((XNCW + 4) * 64), XLDS, st, (const f16*)A2, (const f16*)W2, bias, unscale, oscale,
                       outf, (f16*)out2, aux, flag, M, N, K, tn, total);
    return 0;
  }
  const KernT kern = kerns[epi == EPI_GELU ? 2 : epi == EPI_RESID ? 3 : epi == EPI_QKV_PACK ? 1 : epi == EPI_RESID_LN ? 4
                           : epi == EPI_GELU_LN ? 5 : 0];
#else
  const KernT kern = kerns[epi == EPI_GELU ? 2 : epi == EPI_RESID ? 3 : epi == EPI_QKV_PACK ? 1 : 0];
#endif
  hipLaunchKernelGGL(kern, dim3(grid), dim3((XNCW + 4) * 64), XLDS, st, (const f16*)A2, (const f16*)W2, bias, unscale, oscale, outf,
                     (f16*)out2, aux, flag, M, N, K, reverse ? -tn : tn, total);
  return 0;
}

void d3dp_launch_split2(const float* src, void* dst, size_t n, float scale, hipStream_t st) {
  const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(split2h_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, (f16*)dst, n, scale);
}

void d3dp_launch_rowbound(const float* W, const float* gamma, const float* beta, const float* bias, int N, int K,
                          unsigned* out, hipStream_t st) {
  hipLaunchKernelGGL(rowbound_kernel, dim3((N + 3) / 4), dim3(256), 0, st, W, gamma, beta, bias, N, K,
                     sqrtf((float)(K - 1)) * 1.0001f, out);
}

void d3dp_launch_absmax(const float* src, size_t n, unsigned* out, hipStream_t st) {
  const unsigned blocks = (unsigned)((n / 4 + 255) / 256 < 512 ? (n / 4 + 255) / 256 : 512);
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, n, out);
}

void d3dp_launch_nonfinite_flag(const float* x, size_t n, unsigned* flag, hipStream_t st) {
  const unsigned blocks = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(nonfinite_flag_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, x, n, flag);
}

// (norm2 folded into proj / fc1: D3DP_X2_VARIANTS builds only; capi.hip never reaches these without one)
void d3dp_launch_ln_combine(const float* slices, float* rowstat, int M, int C, float eps, hipStream_t st) {
#if D3DP_X2_VARIANTS
  hipLaunchKernelGGL(ln_combine_kernel, dim3((M + 255) / 256), dim3(256), 0, st, slices, rowstat, M, (C + 63) / 64, eps);
#endif
}

void d3dp_launch_fold_ln(const float* W, const float* gamma, const float* beta, const float* bias, float* Wp, float* c12,
                         int N, int K, hipStream_t st) {
#if D3DP_X2_VARIANTS
  hipLaunchKernelGGL(fold_ln_kernel, dim3((N + 3) / 4), dim3(256), 0, st, W, gamma, beta, bias, Wp, c12, N, K);
#endif
}

// ---- training-step launchers (gemm_f16x2_dyn_kernel and its operand kernels) ---------------------------------------
// out_z[M, N] = A2 (chunk z) . W2 (chunk z)^T x dynA x dynW (+ bias): Kfull = Z NKz 32 columns per operand row
int d3dp_launch_linear_f16x2_dyn(const void* A2, const void* W2, const float* bias, const float* dynA, const float* dynW,
                                 float* out, int M, int N, int Kfull, int Z, hipStream_t st, unsigned* amax_out, int amax_pos,
                                 int rem_blocks) {
  if (M <= 0 || N % 4 != 0 || Z < 1 || Kfull % (XBK * Z) != 0 || (amax_out && Z != 1)) return -1;
  const int NKz = Kfull / XBK / Z;
  // rem_blocks: the rows behind the last whole 256-row tile as 16 x 64 blocks at the end of the kernel (see there)
  const bool rem = rem_blocks && Z == 1 && M > XBM && M % XBM != 0 && NKz % (2 * XNCW) == 0;
  const int tm = rem ? M / XBM : (M + XBM - 1) / XBM, tn = (N + XBN - 1) / XBN;
  static PerDeviceOnce once;
  const int cus = once.get([&](int dev) {
    return d3dp_lds_opt_in(reinterpret_cast<const void*>(gemm_f16x2_dyn_kernel), XNSTAGE * XSTAGE + 64) < 0 ? -3 : d3dp_cu_count(dev);
  });
  if (cus < 0) return -3;
  const int items = tm * tn * Z, grid = items < cus ? items : cus;
  hipLaunchKernelGGL(gemm_f16x2_dyn_kernel, dim3(grid), dim3((XNCW + 4) * 64), XNSTAGE * XSTAGE + 64, st, (const f16*)A2,
                     (const f16*)W2, bias, dynA, dynW, out, M, N, Kfull, NKz, tn, tm * tn, items, amax_out, amax_pos,
                     rem ? tm * XBM : 0);
  return 0;
}

void d3dp_launch_split2_dyn(const float* src, void* dst, int R, int C, int Cpad, const unsigned* amax, float* unscale,
                            hipStream_t st) {
  const size_t n = (size_t)R * (Cpad / 8);
  const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(split2h_dyn_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, (f16*)dst, R, C, Cpad, amax, unscale);
}

void d3dp_launch_split2_t_dyn(const float* src, void* dst, int R, int C, int Rpad, const unsigned* amax, float* unscale,
                              hipStream_t st) {
  hipLaunchKernelGGL(split2h_t_dyn_kernel, dim3(Rpad / 32, (C + 31) / 32), dim3(256), 0, st, src, (f16*)dst, R, C, Rpad, amax,
                     unscale);
}

int d3dp_launch_dyprep(const float* src, void* drow, void* dcol, float* colpart, int R, int C, int Rpad, const unsigned* amax,
                       float* unscale, hipStream_t st) {
  if (C % 32 != 0 || Rpad % 32 != 0 || Rpad < R) return -1;
  // one 192-row tile per workgroup where that needs at most D3DP_DYPREP_ROWS of them (configs[4]: 87 x C / 32 workgroups, all
  // resident at once); `colpart` always receives D3DP_DYPREP_ROWS rows (workgroup rows without a tile write zeros)
  hipLaunchKernelGGL(dyprep_kernel, dim3(D3DP_DYPREP_ROWS, C / 32), dim3(256), 0, st, src, (f16*)drow, (f16*)dcol, colpart, R, C, Rpad, amax, unscale);
  return 0;
}

// the row form alone (+ column sums): see rowprep_kernel.  *rows = partial rows written to colpart (<= D3DP_ROWPREP_ROWS)
int d3dp_launch_rowprep_ln(const float* src, const float* w, const float* b, float eps, void* drow, int R, int Rpad, int C,
                           const unsigned* amax, float* unscale, hipStream_t st) {
  if (C % 32 != 0 || C > 512 || Rpad < R || !w || !b) return -1;
  int g = (Rpad + 3) / 4;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(rowprep_ln_kernel, dim3(g), dim3(256), 0, st, src, w, b, eps, (f16*)drow, R, Rpad, C, amax, unscale);
  return 0;
}

int d3dp_launch_rowprep(const float* src, void* drow, float* colpart, int* rows, int R, int Rpad, int C, const unsigned* amax,
                        float* unscale, hipStream_t st, const float* mask, int axis, int F, int J, const float* gelu_pre) {
  if (C % 32 != 0 || Rpad < R || C > 1536) return -1;      // (P = ceil(C / 512) column passes; a lane whose 8 columns lie behind C idles)
  int g = (Rpad + 3) / 4;
  const int cap = colpart ? D3DP_ROWPREP_ROWS : 1024;   // (every workgroup leaves one partial row of column sums)
  if (g > cap) g = cap;
  if (rows) *rows = g;
  const int P = (C + 511) / 512;
  f16* d = (f16*)drow;
  if (mask && (J < 1 || F < 1)) return -1;
#define D3DP_ROWPREP(PP, GG) hipLaunchKernelGGL((rowprep_kernel<PP, GG>), dim3(g), dim3(256), 0, st, src, d, colpart, R, Rpad, C, amax, unscale, mask, axis, F, J, gelu_pre)
  if (gelu_pre) {
    if (P == 1) D3DP_ROWPREP(1, true); else if (P == 2) D3DP_ROWPREP(2, true); else D3DP_ROWPREP(3, true);
  } else {
    if (P == 1) D3DP_ROWPREP(1, false); else if (P == 2) D3DP_ROWPREP(2, false); else D3DP_ROWPREP(3, false);
  }
#undef D3DP_ROWPREP
  return 0;
}

int d3dp_launch_gelu_rowprep(const float* src, void* drow, int R, int Rpad, int C, const unsigned* pmax, unsigned* amax_out,
                             float* unscale, hipStream_t st) {
  if (C % 32 != 0 || Rpad < R) return -1;
  const size_t n = (size_t)Rpad * (C / 8);
  const unsigned blocks = (unsigned)((n + 511) / 512 < 2048 ? (n + 511) / 512 : 2048);
  hipLaunchKernelGGL(gelu_rowprep_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, (f16*)drow, R, Rpad, C, pmax, amax_out, unscale);
  return 0;
}

// out_z[N, K] = sum over the token rows of chunk z of A2[t][n] W2[t][k], x dynA x dynW: Tp = Z NKz 32 rows per operand (rows
// beyond the real ones zero), N % 256 == 0, K % 128 == 0 (d3dp_tn_applies)
bool d3dp_tn_applies(int N, int K) { return N % XBM == 0 && K % XBN == 0; }
// n products over the same Tp token rows in one launch: out_p,z[N_p, K_p] (z = 0 .. Z - 1, N_p K_p floats apart)
int d3dp_launch_linear_f16x2_tn_many(const D3dpTnProduct* prods, int n, int Tp, int Z, hipStream_t st) {
  if (n < 1 || n > D3DP_TN_MAX || Z < 1 || Tp % (XBK * Z) != 0) return -1;
  D3dpTnTable tb{};
  tb.n = n;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    tb.p[i] = prods[i];
    if (!d3dp_tn_applies(tb.p[i].N, tb.p[i].K)) return -1;
    tb.p[i].tiles_k = tb.p[i].K / XBN;
    tb.p[i].tile0 = tiles;
    tiles += (tb.p[i].N / XBM) * tb.p[i].tiles_k;
  }
  const int NKz = Tp / XBK / Z;
  static PerDeviceOnce once;
  const int cus = once.get([&](int dev) {
    return d3dp_lds_opt_in(reinterpret_cast<const void*>(gemm_f16x2_tn_kernel), XNSTAGE * XSTAGE) < 0 ? -3 : d3dp_cu_count(dev);
  });
  if (cus < 0) return -3;
  const int items = tiles * Z, grid = items < cus ? items : cus;
  hipLaunchKernelGGL(gemm_f16x2_tn_kernel, dim3(grid), dim3((XNCW + 4) * 64), XNSTAGE * XSTAGE, st, tb, NKz, tiles, items);
  return 0;
}
int d3dp_launch_linear_f16x2_tn(const void* A2, const void* W2, const float* dynA, const float* dynW, float* out, int N, int K,
                                int Tp, int Z, hipStream_t st) {
  const D3dpTnProduct one{A2, W2, dynA, dynW, out, N, K, 0, 0};
  return d3dp_launch_linear_f16x2_tn_many(&one, 1, Tp, Z, st);
}

int d3dp_launch_wprep(const D3dpWPrepTable& tb, void* rows_base, void* cols_base, unsigned* amax, float* unscale, hipStream_t st) {
  if (tb.n < 1 || tb.n > D3DP_WPREP_MAX) return -1;
  for (int i = 0; i < tb.n; ++i)
    if (tb.it[i].N % 32 != 0 || tb.it[i].K % 32 != 0) return -1;
  hipLaunchKernelGGL(wprep_absmax_kernel, dim3(16, tb.n), dim3(256), 0, st, tb, amax);
  hipLaunchKernelGGL(wprep_rows_kernel, dim3(32, tb.n), dim3(256), 0, st, tb, (f16*)rows_base, (const unsigned*)amax, unscale);
  hipLaunchKernelGGL(wprep_cols_kernel, dim3(32, tb.n), dim3(256), 0, st, tb, (f16*)cols_base, (const unsigned*)amax);
  return 0;
}

void d3dp_launch_sum_partials_bias(const float* part, const float* bias, float* out, size_t n, int N, int Z, hipStream_t st,
                                   unsigned* amax, int amax_pos) {
  const unsigned cap = amax ? 256 : 2048;              // (with an absmax: one atomic per workgroup)
  const unsigned blocks = (unsigned)((n + 255) / 256 < cap ? (n + 255) / 256 : cap);
  hipLaunchKernelGGL(sum_partials_bias_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, part, bias, out, n, N, Z, amax, amax_pos);
}
void d3dp_launch_sum_partials_many(const D3dpSumTable& tb, hipStream_t st) {
  size_t most = 0;
  for (int i = 0; i < tb.n; ++i) most = tb.n4[i] > most ? tb.n4[i] : most;
  const unsigned blocks = (unsigned)((most + 255) / 256 < 1024 ? (most + 255) / 256 : 1024);
  hipLaunchKernelGGL(sum_partials_many_kernel, dim3(blocks ? blocks : 1, tb.n), dim3(256), 0, st, tb);
}
void d3dp_launch_sum_partials(const float* part, float* out, size_t n, int Z, hipStream_t st) {
  const size_t n4 = n / 4;                               // (n = N K of a Linear: a multiple of 4)
  const unsigned blocks = (unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, reinterpret_cast<const float4*>(part),
                     reinterpret_cast<float4*>(out), n4, Z);
}
