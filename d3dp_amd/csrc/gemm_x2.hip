// EXACT-mode Linear layers on the fp16 matrix cores:  out[M,N] = epi((A . W^T) + bias), fp32-class accuracy.
// (reference: every nn.Linear on the path -- common/mixste.py:65 qkv, :80 proj, :38-41 fc1/fc2 -- which the reference
//  evaluates in fp32.)
//
// Operands are "f16x2" pairs (common.h): x 2^s = hi + lo, two fp16 planes each, A2 [2][M][K] (s = 4, fixed) and
// W2 [2][N][K] (s per matrix: max |w| 2^s in [2^13, 2^14)); `unscale` = 2^-(s_a + s_w) undoes both.  Per output element
// three v_mfma_f32_16x16x32_f16 passes into one fp32 accumulator,
//     acc += Wl.Ah + Wh.Al + Wh.Ah            out = acc unscale + bias,
// the lo.lo pass being below fp32 resolution (tools/err_budget_split.py;
// tests/test_hip_parity.py::test_linear_split_f16_is_fp32_class: mean error below torch's own fp32 matmul).
//
// Structure (the FAST kernel's, gemm.hip): persistent, 256x128 output tile per workgroup pass, BK = 32, 8 compute waves
// (4 x 2, 64x64 each = 4x4 MFMA tiles: 48 MFMA + 16 ds_read_b128 per k-step) + 4 LOADER waves.  Workgroups (<= one per
// CU) walk tiles L, L+G, ...; the A and W slabs (two planes each: 48 KiB per k-step) stream through a 3-stage LDS ring
// as one continuous sequence of k-steps across tile boundaries.  Only the loader waves issue global_load_lds
// (1 KiB pieces of 16 rows x 64 B, 12 per wave and k-step) and wait on vmcnt (counted: one k-step stays in flight across
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
//   EPI_GELU     -> GELU (rational erf, common.h) re-split into two fp16 planes (the fc2 operand)
//   EPI_QKV_PACK -> q fp32, k and v as fp16 planes: the packed rows the split-fp16 attention kernels read (attention.hip)
//   EPI_RESID    -> x += A W^T + b in place on the fp32 residual stream (proj, fc2: mixste.py:113-115)
// Plane outputs leave as ONE 16-byte store per lane too (neighbouring lanes swap halves, store_planes_paired).
// Timing probes of this kernel (loads / stores / MFMAs / barriers compiled out one at a time) and what they say about
// the clock the chip sustains under this instruction mix: profiles/r02_gemm_probes.md.
#include <cstdlib>
#include <cstring>
#include <mutex>
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
#ifndef D3DP_X2_SBEND
#define D3DP_X2_SBEND 0
#endif
// L2 prefetch distance in k-steps (even; 0 = off): the compute waves touch, D k-steps ahead of the loaders, every
// 128-byte line of the A and W slabs (and, PFR, of the residual tile the EPI_RESID epilogue will read) with a 4-byte
// LDS-DMA into a junk area -- no register destination, nothing ever waits for it.  The LDS ring bounds the bytes a CU
// has in flight at two 48 KiB slabs (1.6 us at the rate it consumes them), less than the latency of a line that comes
// from HBM or the memory-side cache; with the touch ahead of it the loaders' own LDS-DMA finds the line in the XCD's L2.
#ifndef D3DP_X2_PFD
#define D3DP_X2_PFD 0
#endif
#ifndef D3DP_X2_PFW
#define D3DP_X2_PFW 1
#endif
#ifndef D3DP_X2_PFR
#define D3DP_X2_PFR 1
#endif
#ifndef D3DP_NT_OUT
#define D3DP_NT_OUT 1
#endif
#if D3DP_NT_OUT
#define OUT_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define OUT_STORE(ptr, val) (*(ptr) = (val))
#endif

namespace {

constexpr int XBM = 256, XBN = 128, XBK = 32;
constexpr int XA_PLANE = XBM * XBK * 2;              // 16 KiB
constexpr int XW_PLANE = XBN * XBK * 2;              //  8 KiB
constexpr int XSTAGE = 2 * XA_PLANE + 2 * XW_PLANE;  // 48 KiB
constexpr int XNSTAGE = 3;
constexpr int XBIAS_MAX = 2048;                      // floats of bias kept in LDS
constexpr int XJUNK = XNSTAGE * XSTAGE + XBIAS_MAX * 4;  // 256 bytes nobody reads: destination of the L2 prefetch touches
constexpr int XLDS = XJUNK + 256;                        // 152.25 KiB
constexpr int XNCW = 8;                              // compute waves (4 x 2); waves 8..11 are loaders

// 64-byte rows (4 slots of 16 B): XOR bit 1 of the slot with bit 3 of the row -> conflict-free ds_read_b128 fragments
__device__ __forceinline__ int swz64(int row, int s) { return s ^ (((row >> 3) & 1) << 1); }

// W row carried by LDS row q of a 64-column strip: MFMA tile ni = q>>4, operand row i = q&15 -> output column 4 i + ni
__device__ __forceinline__ int colperm(int q) { return (q & 15) * 4 + (q >> 4); }

// ---- SH = 1: the same kernel on v_mfma_f32_32x32x16_f16 (half the MFMA instructions and half the operand-register
// reads per FLOP; the guide's micro-benchmark ceiling for fp16 is 2178 TFLOP/s on 32x32 against 1955 on 16x16).  Each
// compute wave owns 32 rows x all 128 columns of the tile (4 accumulator tiles of 32x32 = the same 64 registers): a lane
// holds C[row (reg&3) + 8 (reg>>2) + 4 (lane>>5)][column lane&31] of each tile, and the loader permutes the W rows of the
// 128-column slab (LDS row ni*32 + i carries column 4 i + ni), so the lane's four tiles are again four CONSECUTIVE
// columns: the epilogue's 16-byte stores are those of the 16x16 form, with 32 lanes covering 512 contiguous bytes of a row.
// A 32-row fragment is read by the four 16-lane groups of ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the
// same + 32): rows r, r+4, r+8, ... share a bank group, so the 16-byte slot is XORed with h(r>>2), h(q) = (q ^ q>>1) & 3,
// which is injective on {0,3,5,6} and on {1,2,4,7} -- the row quads each lane group touches: conflict-free.
__device__ __forceinline__ int h32(int row) { const int q = (row & 31) >> 2; return (q ^ (q >> 1)) & 3; }
__device__ __forceinline__ int swz32(int row, int s) { return s ^ h32(row); }
__device__ __forceinline__ int colperm32(int q) { return (q & 31) * 4 + (q >> 5); }
typedef float f32x16_ __attribute__((ext_vector_type(16)));

// One 4-byte LDS-DMA per lane from base + voff into the junk area at LDS byte address `junk` (wave-uniform): brings the
// lane's 128-byte line into the L2.  M0 is saved and restored inside the statement (the compiler does not model it).
__device__ __forceinline__ void touch_line(const void* base, unsigned voff, unsigned junk) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(junk) : "memory");
}

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
template <int EPI, int TAG, int SH>
__global__ __launch_bounds__(768) void gemm_f16x2_kernel(const f16* __restrict__ A2, const f16* __restrict__ W2,
                                                         const float* __restrict__ bias, float unscale,
                                                         float* __restrict__ outf, f16* __restrict__ out2, int M, int N,
                                                         int K, int tiles_n, int total_tiles) {
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

  if (wave >= XNCW) {
    // ------------------------------------------------------------------ loader waves
    const int lw = wave - XNCW;
    const int lr = lane >> 2, lps = lane & 3;
#ifdef D3DP_X2_LPRIO
    __builtin_amdgcn_s_setprio(D3DP_X2_LPRIO);
#endif
    const size_t planeA = (size_t)M * K, planeW = (size_t)N * K;
    int ti = 0, ks = 0, slot = 0;                      // (tile, k-step, ring slot) of the next slab to issue
    const f16* pa[4];                                  // this lane's source rows of the current tile (k = 0, hi plane)
    const f16* pw[2];
    auto issue = [&]() {
      if (ks == 0) {                                   // new tile: row pointers once per tile, not per k-step
        const int t = L + ti * G;
        const int m0 = (t / tiles_n) * XBM, n0 = (t % tiles_n) * XBN;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                  // A row groups 4 lw .. 4 lw + 3 (16 rows each)
          const int row = (lw * 4 + i) * 16 + lr;
          pa[i] = A2 + (size_t)min(m0 + row, M - 1) * K + (SH ? swz32(row, lps) : swz64(row, lps)) * 8;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {                  // W row groups 2 lw, 2 lw + 1
          const int row = (lw * 2 + i) * 16 + lr;                       // LDS row of the W slab
          const int wrow = SH ? colperm32(row) : (row & 64) + colperm(row & 63);   // output column it carries
          pw[i] = W2 + (size_t)min(n0 + wrow, N - 1) * K + (SH ? swz32(row, lps) : swz64(row, lps)) * 8;
        }
      }
      char* base = smem + slot * XSTAGE;
      const int ko = ks * XBK;
#if D3DP_X2_PROBE & 1
      if (ko >= 0) { if (++ks == NK) { ks = 0; ++ti; } slot = (slot == XNSTAGE - 1) ? 0 : slot + 1; return; }
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rg = lw * 4 + i;
        __builtin_amdgcn_global_load_lds(GPTR(pa[i] + ko), LPTR(base + rg * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GPTR(pa[i] + ko + planeA), LPTR(base + XA_PLANE + rg * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rg = lw * 2 + i;
        __builtin_amdgcn_global_load_lds(GPTR(pw[i] + ko), LPTR(base + 2 * XA_PLANE + rg * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GPTR(pw[i] + ko + planeW), LPTR(base + 2 * XA_PLANE + XW_PLANE + rg * 1024), 16, 0, 0);
      }
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
  const int fi = SH ? (lane & 31) : (lane & 15), fg = SH ? (lane >> 5) : (lane >> 4);
  // per-lane fragment offsets inside a stage (the swizzle depends on the lane only: rows advance in multiples of 16 / 32)
  const int offA = SH ? (wave * 32 + fi) * 64 + swz32(fi, fg) * 16 : (wr * 64 + fi) * 64 + swz64(fi, fg) * 16;
  const int offW = 2 * XA_PLANE + (SH ? fi * 64 + swz32(fi, fg) * 16 : (wc * 64 + fi) * 64 + swz64(fi, fg) * 16);
  const size_t planeO = (size_t)M * N;
  f32x4 acc[4][4];                                     // SH = 0: [row block mi][column tile ni][r]
  f32x16_ acc32[4];                                    // SH = 1: [column tile ni][reg]
  __builtin_amdgcn_s_setprio(1);
#if D3DP_X2_PROBE & 16
  if (L & 1)
    for (int i = 0; i < D3DP_X2_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
  int slot = 0;
  for (int ti = 0; ti < n_my; ++ti) {
   if constexpr (SH == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
    // One k-step = two k-halves of 16 (slots 2 s + fg of the 64-byte rows: the second half's offset is the first's ^ 32).
    // Per half and column tile three products into one accumulator, same-accumulator MFMAs four instructions apart.  The
    // last product of a k-step (ah . wh of its second half: 4 MFMAs) is issued after the NEXT k-step's barrier, behind that
    // step's first fragment reads, where it covers the LDS latency every wave meets at once; its operands (20 registers)
    // alternate between two sets as the loop is unrolled by two.
    f16x8 lwa[4], lwb[4], laa, lab;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) lwb[ni] = (f16x8){};
    lab = (f16x8){};
    const int offA1 = offA ^ 32, offW1 = offW ^ 32;
    auto kstep32 = [&](f16x8 (&lw)[4], f16x8& la, const f16x8 (&pw)[4], const f16x8& pa) {
      __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0)
      X2_BARRIER();
      __builtin_amdgcn_sched_barrier(0);
      const char* sb = smem + slot * XSTAGE;
      slot = (slot == XNSTAGE - 1) ? 0 : slot + 1;
      f16x8 wh[4], wl[4], wl1[4];
      const f16x8 ah = *reinterpret_cast<const f16x8*>(sb + offA);
      const f16x8 al = *reinterpret_cast<const f16x8*>(sb + offA + XA_PLANE);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        wl[ni] = *reinterpret_cast<const f16x8*>(sb + offW + XW_PLANE + ni * 2048);
        wh[ni] = *reinterpret_cast<const f16x8*>(sb + offW + ni * 2048);
      }
      __builtin_amdgcn_sched_barrier(0);
#if !(D3DP_X2_PROBE & 8)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc32[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa, pw[ni], acc32[ni], 0, 0, 0);
#endif
      __builtin_amdgcn_sched_barrier(0);
      la = *reinterpret_cast<const f16x8*>(sb + offA1);
      const f16x8 al1 = *reinterpret_cast<const f16x8*>(sb + offA1 + XA_PLANE);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        wl1[ni] = *reinterpret_cast<const f16x8*>(sb + offW1 + XW_PLANE + ni * 2048);
        lw[ni] = *reinterpret_cast<const f16x8*>(sb + offW1 + ni * 2048);
      }
#if D3DP_X2_PROBE & 8
      asm volatile("" :: "v"(ah), "v"(al), "v"(la), "v"(al1));
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) asm volatile("" :: "v"(wh[ni]), "v"(wl[ni]), "v"(wl1[ni]), "v"(lw[ni]));
#else
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc32[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[ni], acc32[ni], 0, 0, 0);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc32[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[ni], acc32[ni], 0, 0, 0);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc32[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[ni], acc32[ni], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc32[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(la, wl1[ni], acc32[ni], 0, 0, 0);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc32[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, lw[ni], acc32[ni], 0, 0, 0);
#endif
#if D3DP_X2_SBEND
      __builtin_amdgcn_sched_barrier(0);               // (A/B: keep the k-step's products above the next barrier)
#endif
    };
#pragma unroll 1
    for (int ks = 0; ks < NK; ks += 2) {               // NK is even (K % 64 == 0, checked by the launcher)
      kstep32(lwa, laa, lwb, lab);
      kstep32(lwb, lab, lwa, laa);
    }
#if !(D3DP_X2_PROBE & 8)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc32[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(lab, lwb[ni], acc32[ni], 0, 0, 0);
#endif
   } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#if D3DP_X2_PFD
    // ---- L2 prefetch touches (see D3DP_X2_PFD): byte offsets of this lane's lines in the current and the next tile
    const int L512 = wave * 64 + lane;
    unsigned pfa0, pfw0, pfda = 0, pfdw = 0, pfr = 0;   // (pfd*: next tile's offset minus this tile's, modulo 2^32)
    int pf_lim = NK;                                   // prefetch only k-steps below this (no next tile: none beyond NK)
    {
      const int t0 = L + ti * G, m0 = (t0 / tiles_n) * XBM, n0 = (t0 % tiles_n) * XBN;
      const int prow = L512 & 255, ppl = L512 >> 8, wrow_ = L512 & 127, wpl = (L512 >> 7) & 1;
      pfa0 = (unsigned)((ppl * M + min(m0 + prow, M - 1)) * K) * 2u;
      pfw0 = (unsigned)((wpl * N + min(n0 + wrow_, N - 1)) * K) * 2u;
      if (ti + 1 < n_my) {
        const int t1 = t0 + G, m1 = (t1 / tiles_n) * XBM, n1 = (t1 % tiles_n) * XBN;
        pfda = (unsigned)((ppl * M + min(m1 + prow, M - 1)) * K) * 2u - pfa0 - (unsigned)K * 2u;
        pfdw = (unsigned)((wpl * N + min(n1 + wrow_, N - 1)) * K) * 2u - pfw0 - (unsigned)K * 2u;
        pf_lim = 2 * NK;
      }
      if constexpr (EPI == EPI_RESID)   // residual tile: 256 rows x 512 bytes = 1024 lines, two per lane (L512, L512 + 512)
        pfr = (unsigned)(min(m0 + (L512 >> 2), M - 1) * N + min(n0 + (L512 & 3) * 32, N - 4)) * 4u;
    }
    const unsigned junk = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<size_t>(smem + XJUNK));
#define X2_PREFETCH(ks)                                                                                       \
    do {             /* after the barrier of even k-step ks: the lines of k-steps ks+D, ks+D+1 (next tile past NK) */ \
      const int kk_ = (ks) + D3DP_X2_PFD;                                                                      \
      if (kk_ < pf_lim) {                                                                                      \
        const unsigned wrap_ = kk_ >= NK ? 1u : 0u;                                                            \
        touch_line(A2, pfa0 + wrap_ * pfda + (unsigned)kk_ * (XBK * 2), junk);                                 \
        if (D3DP_X2_PFW && wave < 4) touch_line(W2, pfw0 + wrap_ * pfdw + (unsigned)kk_ * (XBK * 2), junk);     \
      }                                                                                                        \
      if constexpr (EPI == EPI_RESID && D3DP_X2_PFR)                                                           \
        if ((ks) == NK - 8 || (ks) == NK - 6) touch_line(outf, pfr + ((ks) == NK - 6 ? 128u * N * 4u : 0u), junk); \
    } while (0)
#endif
#if D3DP_X2_LAG
    // The last two products of row block 3 (al.wh, ah.wh: 8 MFMAs) are issued AFTER the next k-step's barrier, behind
    // that step's first fragment reads: they cover the LDS latency that otherwise idles the matrix pipe after every
    // barrier release (all waves of the workgroup read at once).  Their operands stay in 24 registers across the
    // barrier; the k-loop is unrolled by two so that the two fragment sets swap roles without register copies.
    f16x8 wfa[4][2], wfb[4][2], taa[2], tab[2];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) wfb[ni][0] = wfb[ni][1] = (f16x8){};
    tab[0] = tab[1] = (f16x8){};
    auto kstep = [&](f16x8 (&wf)[4][2], f16x8 (&ta)[2], const f16x8 (&pw)[4][2], const f16x8 (&pa)[2], int pfks) {
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
          wf[ni][pl] = *reinterpret_cast<const f16x8*>(sb + offW + pl * XW_PLANE + ni * 1024);
      ah[0] = *reinterpret_cast<const f16x8*>(sb + offA);
      al[0] = *reinterpret_cast<const f16x8*>(sb + offA + XA_PLANE);
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
#if D3DP_X2_PFD
      if (pfks >= 0) X2_PREFETCH(pfks);
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int mi = 0; mi < 3; ++mi) {
        const int b = mi & 1;
        if (mi < 2) {                                  // next row block's fragments while this one multiplies
          ah[b ^ 1] = *reinterpret_cast<const f16x8*>(sb + offA + (mi + 1) * 1024);
          al[b ^ 1] = *reinterpret_cast<const f16x8*>(sb + offA + XA_PLANE + (mi + 1) * 1024);
        } else {
          ta[0] = *reinterpret_cast<const f16x8*>(sb + offA + 3 * 1024);
          ta[1] = *reinterpret_cast<const f16x8*>(sb + offA + XA_PLANE + 3 * 1024);
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
      kstep(wfa, taa, wfb, tab, ks);
      kstep(wfb, tab, wfa, taa, -1);
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
          wf[ni][pl] = *reinterpret_cast<const f16x8*>(sb + offW + pl * XW_PLANE + ni * 1024);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const f16x8 ah = *reinterpret_cast<const f16x8*>(sb + offA + mi * 1024);
        const f16x8 al = *reinterpret_cast<const f16x8*>(sb + offA + XA_PLANE + mi * 1024);
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
   }
    // ---- tile epilogue: lane holds out[m = pm0 + rowk(q)][n = nb + ni] for its 16 row slots q.  SH = 0: q = mi*4 + r,
    // rowk = mi*16 + r, pm0 = tile row + wr*64 + 4 fg, nb = tile column + wc*64 + 4 fi; SH = 1: q = reg, rowk = (reg&3) +
    // 8 (reg>>2), pm0 = tile row + wave*32 + 4 fg, nb = tile column + 4 fi.  Every store address is  uniform base + 32-bit
    // lane offset  (the launcher refuses outputs of 4 GiB or more), advanced row by row: per-row 64-bit address arithmetic
    // was most of the epilogue's VALU work, and it runs with the matrix pipes idle.
    const int t = L + ti * G;
    const int pm0 = (t / tiles_n) * XBM + (SH ? wave * 32 : wr * 64) + 4 * fg;
    const int nb = (t % tiles_n) * XBN + (SH ? 0 : wc * 64) + 4 * fi;
    if (nb < N && (!(D3DP_X2_PROBE & 4) || unscale == -12345.f)) {
      const float4 bz = *reinterpret_cast<const float4*>(sbias + nb);
      const bool odd = fi & 1;
      unsigned off, pitch;
      bool planes;                                     // split the values and store fp16 planes (else fp32)
      char* base = reinterpret_cast<char*>(outf);
      if constexpr (EPI == EPI_GELU) {
        base = reinterpret_cast<char*>(out2);
        pitch = N * 2; planes = true;
        off = nb * 2 + (odd ? (unsigned)planeO * 2 - 8 : 0);
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
      const int rows = M - pm0;                        // row slot q of this lane exists iff rowk(q) < rows
      auto value = [&](int mi, int r, int e) {
        const float a = SH ? acc32[e][mi * 4 + r] : acc[mi][e][r];
        return fmaf(a, unscale, e == 0 ? bz.x : e == 1 ? bz.y : e == 2 ? bz.z : bz.w);
      };
      auto rowk = [&](int mi, int r) { return SH ? r + 8 * mi : mi * 16 + r; };
      auto store_rows = [&](auto planes_c, auto checked_c) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k = rowk(mi, r);
            const bool live = !decltype(checked_c)::value || k < rows;
            char* dst = base + (off + (unsigned)k * pitch);
            if constexpr (decltype(planes_c)::value) {
              f16x4 ph, pl;
              float v[4] = {value(mi, r, 0), value(mi, r, 1), value(mi, r, 2), value(mi, r, 3)};
              if constexpr (EPI == EPI_GELU) {
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
                split2h(v[e], h, l);
                ph[e] = h; pl[e] = l;
              }
              store_planes_paired(dst, ph, pl, odd, live);
            } else if constexpr (EPI != EPI_RESID) {
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
              const int k = rowk(mi, r);
              const bool live = !decltype(checked_c)::value || k < rows;
              res[mi][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
              if (live) res[mi][r] = *reinterpret_cast<const f32x4*>(base + (off + (unsigned)k * pitch));
            }
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int k = rowk(mi, r);
              const bool live = !decltype(checked_c)::value || k < rows;
              const f32x4 v = {res[mi][r][0] + value(mi, r, 0), res[mi][r][1] + value(mi, r, 1),
                               res[mi][r][2] + value(mi, r, 2), res[mi][r][3] + value(mi, r, 3)};
              if (live) *reinterpret_cast<f32x4*>(base + (off + (unsigned)k * pitch)) = v;   // (re-read by the next row kernel: no nt hint)
            }
        }
      };
      using T_ = std::true_type; using F_ = std::false_type;
      if (rows >= (SH ? 32 : 64)) {                    // (all but the last row of tiles)
        if (planes) { if constexpr (EPI == EPI_GELU || TAG == 1) store_rows(T_{}, F_{}); }
        else { if constexpr (EPI != EPI_GELU) store_rows(F_{}, F_{}); }
      } else {
        if (planes) { if constexpr (EPI == EPI_GELU || TAG == 1) store_rows(T_{}, T_{}); }
        else { if constexpr (EPI != EPI_GELU) store_rows(F_{}, T_{}); }
      }
    }
  }
#if D3DP_X2_PFD
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no touch may still be writing LDS when the workgroup's LDS is released
#endif
}

// dst[0][i] = hi, dst[1][i] = lo of src[i] * scale
__global__ void split2h_kernel(const float* __restrict__ s, f16* __restrict__ d, size_t n, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    f16 a, b;
    split2h_scaled(s[i] * scale, a, b);
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

// out = epi((A W^T) unscale + bias) with A2/W2 split-fp16 planes; EPI_BIAS: fp32 `outf`; EPI_GELU: two fp16 planes `out2`;
// EPI_QKV_PACK (N = 3 C, C % 64 == 0): `outf` rows of 12 C bytes = q fp32 | k hi | k lo | v hi | v lo (fp16 x 16);
// EPI_RESID: `outf` [M, N] fp32 is read and written (outf += ...).
// `unscale` = 1 / (scale of the A planes * scale of the W planes).
// K must be a multiple of 64: the k-loop is unrolled by two k-steps of 32 (the lagged products alternate between two
// register sets), and the loader / compute waves count barriers per k-step.
// env D3DP_X2_SHAPE=32 selects the v_mfma_f32_32x32x16_f16 form of the kernel (measured 11 % slower on the whole step,
// profiles/r03_gemm_mfma_shape.md; kept as a tested cross-check).
int d3dp_launch_linear_f16x2(int epi, const void* A2, const void* W2, const float* bias, float unscale, float* outf,
                             void* out2, int M, int N, int K, hipStream_t st) {
  if (K % (2 * XBK) != 0 || N % 4 != 0 || N > XBIAS_MAX || M <= 0) return -1;
  if ((size_t)M * N * 4 >= ((size_t)1 << 32)) return -1;   // 32-bit byte offsets in the epilogue
  if (epi != EPI_BIAS && epi != EPI_GELU && epi != EPI_QKV_PACK && epi != EPI_RESID) return -1;
  if (epi == EPI_QKV_PACK && (N % 3 != 0 || (N / 3) % 64 != 0)) return -1;
  if (epi == EPI_GELU && N % 8 != 0) return -1;        // plane stores are paired across two 4-column groups
  const int tm = (M + XBM - 1) / XBM, tn = (N + XBN - 1) / XBN;
  using KernT = void (*)(const f16*, const f16*, const float*, float, float*, f16*, int, int, int, int, int);
  static const KernT kerns[2][4] = {
      {gemm_f16x2_kernel<EPI_BIAS, 0, 0>, gemm_f16x2_kernel<EPI_BIAS, 1, 0>, gemm_f16x2_kernel<EPI_GELU, 0, 0>,
       gemm_f16x2_kernel<EPI_RESID, 0, 0>},
      {gemm_f16x2_kernel<EPI_BIAS, 0, 1>, gemm_f16x2_kernel<EPI_BIAS, 1, 1>, gemm_f16x2_kernel<EPI_GELU, 0, 1>,
       gemm_f16x2_kernel<EPI_RESID, 0, 1>}};
  // per DEVICE: the 152 KiB dynamic-LDS opt-in of every instantiation and the CU count (one process may drive several
  // devices: nn.DataParallel callers)
  constexpr int kMaxDev = 64;
  static std::mutex mu;
  static int n_cu[kMaxDev] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return -3;
  int cus;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (n_cu[dev] == 0) {
      for (int sh = 0; sh < 2; ++sh)
        for (int k = 0; k < 4; ++k)
          if (hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[sh][k]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  XLDS) != hipSuccess) return -3;
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount < 1) return -3;
      n_cu[dev] = prop.multiProcessorCount;
    }
    cus = n_cu[dev];
  }
  const char* she = getenv("D3DP_X2_SHAPE");           // read per launch: tests flip it inside one process
  const int shape = (she && !strcmp(she, "32")) ? 1 : 0;
  const int total = tm * tn, grid = total < cus ? total : cus;
  const KernT kern = kerns[shape][epi == EPI_GELU ? 2 : epi == EPI_RESID ? 3 : epi == EPI_QKV_PACK ? 1 : 0];
  hipLaunchKernelGGL(kern, dim3(grid), dim3((XNCW + 4) * 64), XLDS, st, (const f16*)A2, (const f16*)W2, bias, unscale, outf,
                     (f16*)out2, M, N, K, tn, total);
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
