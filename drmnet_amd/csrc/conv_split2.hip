// Split-precision fused GroupNorm+SiLU+conv implicit GEMM, second-generation pipeline (see conv_split.hip for the
// arithmetic: fp16 hi/lo operands, 3 x v_mfma_f32_32x32x16_f16 per product slab, fp32 accumulate).
//
// The first-generation kernel measured weight-stream bound: every 768-cycle tap step needed a 16-KB weight tile through
// registers with only one tile in flight, and skipping the weight reloads doubled its speed.  This kernel
//   * owns 256 output pixels (16x16, or 8x16x2 / 8x8x4 / 4x8x8 / 4x4x16 images) x 128 channels per 512-thread
//     workgroup (8 wave64, one workgroup per CU): the weight bytes per FLOP are halved;
//   * streams the weight tiles global -> LDS with LDS-DMA (global_load_lds_dwordx4, no VGPR staging) into an R-slot
//     ring of TPS-tap groups, retired with counted s_waitcnt vmcnt(N) + raw s_barrier so the prefetch spans barriers
//     (cdna_hip_programming.md "Pipelining across barriers"); the packed weight layout is already the LDS image;
//   * keeps the activation path of v1: halo tile read once per 32-channel chunk, GroupNorm affine + SiLU + fp16 hi/lo
//     split applied once in registers, parked in LDS, re-read by the nine taps at shifted offsets;
//   * is PERSISTENT: a workgroup walks several output tiles in an XCD-contiguous range; the weight ring and the activation
//     prefetch run straight across the tile boundary (no prologue bubble per tile).  The epilogue itself is not hidden:
//     vmcnt retires stores in order, so the counted waits of the next tile also wait for this tile's stores;
//   * accumulates the GroupNorm statistics of its OUTPUT in the epilogue (fp32 partials per lane -> fp64 LDS atomics ->
//     one global fp64 atomic per (image, channel, moment) per tile).
// All vector-memory operations a wave issues are unconditional (clamped addresses, zero-filled afterwards) so the
// vmcnt bookkeeping below is exact and identical for every wave.
// Arithmetic (TERMS): 3 = the three-product fp16 split above; 2 = hi*hi on the f16 MFMA + both cross terms in one block-scaled fp8 MFMA
// ("f16mx", the GroupNorm-fed 3x3 convs); 1 = plain fp16 operands; 4 = plain bf16 operands (v_mfma_f32_32x32x16_bf16: BASELINE configs[2] as
// written); 0 = exact fp32 (v_mfma_f32_32x32x2_f32) -- all on this one pipeline.
// Split-K (deep, small maps): every split writes its own slab; SK instantiations finish the tile inside the launch (last-arriving workgroup).
#include <algorithm>
#include <atomic>
#include <utility>
#include <vector>
#include <cstdio>
#include <string>

#include "common.h"
#include "profiler.h"

// Diagnostic build only (tools/stamp_probe.sh, -DDRM_S2_STAMP, a separate .so): one workgroup records an s_memtime timeline of
// its waves (cdna_hip_programming.md 7 "In-kernel stamps") into the 7.5 KiB of LDS the main 3x3 variant leaves free; it leaves the
// kernel through a buffer nothing else reads.  The product library compiles every S2_STAMP to nothing.
#ifdef DRM_S2_STAMP
#define S2_STAMP(id)                                                                               \
  do {                                                                                             \
    if (stamp_on && stamp_n < 120) {                                                               \
      unsigned long long t_;                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                           \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
      __builtin_amdgcn_sched_barrier(0);                                                           \
      if (lane == 0) {                                                                             \
        stamp_lds[(wave * 120 + stamp_n) * 2] = (unsigned)(id);                                    \
        stamp_lds[(wave * 120 + stamp_n) * 2 + 1] = (unsigned)t_;                                  \
      }                                                                                            \
      ++stamp_n;                                                                                   \
    }                                                                                              \
  } while (0)
#else
#define S2_STAMP(id) do { } while (0)
#endif

namespace drm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

// TERMS == 2 ("f16mx"): a*b ~ ah*bh (two v_mfma_f32_32x32x16_f16 per 32-channel chunk, as in every fp16 mode) + (al*bh + ah*bl) in ONE
// v_mfma_scale_f32_32x32x64_f8f6f4: its K = 64 is two 32-element scale blocks, block 0 = [al8 | bh8], block 1 = [ah8 | bl8] over the chunk's 32
// channels, operands OCP e4m3 with a power-of-two (E8M0) factor per block.  128 matrix-pipe cycles per (tap, chunk, 32x32 block) instead of 192.
// The cross terms are 2^-11 of the product and carry the 2^-4 rounding of an e4m3 factor: ~1e-5 per conv, ~4e-5 through a network (the
// exact split is ~1e-6) -- inside the 1e-4 bar the hot path is held to, outside the 2e-5 the f16x3 mode is tested at.
// Operand layout measured with tools/mx_probe.hip: lane (r, h) supplies 32 bytes; bytes 0-15 of BOTH lane halves form scale block 0
// (factor taken from lanes 0-31), bytes 16-31 form block 1 (factor from lanes 32-63).
// Ranges: the staged activation is clamped to +-MX_A_LIM (GroupNorm + SiLU outputs: nothing gets near), weights are pre-scaled to max |w| in
// [2^13, 2^14) by the packer; the factors below put every fp8 operand inside +-448 (v_cvt_scalef32_pk_fp8_f32 makes NaN above that).
constexpr int MX_EA = 11;                        // staged activations: |v| <= 448 * 2^(EA-8) = 3584
constexpr float MX_A_LIM = 3584.0f;
constexpr float MX_AH_DIV = 8.0f;                // ah8 = v / 2^(EA-8)
constexpr float MX_AL_DIV = 1.0f / 256.0f;       // al8 = (v - ah) / 2^(EA-19)   (|v - ah| <= 2^(EA-11))
constexpr int MX_SA_AL = 127 + MX_EA - 19, MX_SA_AH = 127 + MX_EA - 8;  // E8M0 factors of A's two blocks
// (weights: bh8 = bh / 2^6 (|bh| < 2^14), bl8 = bl * 2^6 (|bl| <= 4), packed by pack_conv_weight_mx_kernel)
constexpr int MX_SB_BH = 127 + 6, MX_SB_BL = 127 - 6;

template <int TAPS, int TH, int TW, int WM, int WN, int MT, int NT, int R, int TPS, int TERMS = 3>
struct S2Cfg {
  static constexpr int KC = 32;
  static constexpr int NW = WM * WN;  // waves
  static constexpr int NTHR = NW * 64;
  static constexpr int BM = WM * MT * 32;
  static constexpr int BN = WN * NT * 32;
  static constexpr int TN = BM / (TH * TW);
  static constexpr int HALO = (TAPS == 9) ? 1 : 0;
  static constexpr int HT = TH + 2 * HALO, WT = TW + 2 * HALO;
  static constexpr int HPI = HT * WT;  // halo pixels per image that are loaded
  // LDS image of the halo tile.  With 16-pixel-wide tiles the 32 rows of an MFMA tile are a 4 x 8 pixel patch and the
  // halo rows are stored 24 pixels apart: the four 16-lane groups of a ds_read_b128 then hit 64 distinct banks for every
  // tap offset (stride = 8 mod 16 pixels; the natural 18-pixel stride with 2 x 16 patches measured 35 % conflict cycles).
  static constexpr bool SUB48 = (TW % 8 == 0) && (TH % 4 == 0);
  static constexpr int WTP_TRY = (TAPS == 9 && TW == 16) ? 24 : WT;
  static constexpr int RING_ST_F4 = R * TPS * 8 * (WN * NT * 32) + TN * (WN * NT * 32);
  static constexpr int A_COPIES = (TAPS == 1) ? 2 : 1;  // 1x1: a new activation tile every step, double-buffered
  static constexpr bool PAD_FITS = (A_COPIES * 8 * TN * HT * WTP_TRY + RING_ST_F4) * 16 <= 160 * 1024;
  static constexpr int WTP = PAD_FITS ? WTP_TRY : WT;  // stored row stride (pixels)
  static constexpr int HPIP = HT * WTP;                // stored pixels per image
  static constexpr int HP = TN * HPIP;
  // Plane stride of the [hi|lo][slab][lane half] planes, padded to 2 (mod 8) pixels: the staging ds_write_b128s of a lane group
  // go to the four octet planes of two neighbouring pixels, and with an unpadded stride that is a multiple of 128 bytes
  // (18 x 24 pixels x 16 B) the four planes fell on the same banks (SQ_LDS_BANK_CONFLICT: 17 % of the LDS cycles of a launch).  The
  // fragment reads of one ds_read_b128 lane group stay inside one plane, so their conflict-free pattern is unchanged.
  static constexpr int HPS_TRY = HP + ((2 - HP % 8) + 8) % 8;
  static constexpr int TPI = NTHR / TN;  // loader threads per image
  static constexpr int OCT = 4;
  static constexpr int A_SLOTS = (HPI * OCT + TPI - 1) / TPI;
  static constexpr bool HPS_FITS = ((TAPS == 1 ? 2 : 1) * 8 * HPS_TRY + R * TPS * 8 * (WN * NT * 32) + TN * (WN * NT * 32)) * 16 <= 160 * 1024;
  static constexpr int HPS = HPS_FITS ? HPS_TRY : HP;
  static constexpr int A1_F4 = 8 * HPS;                      // one activation tile image
  static constexpr int A_F4 = A_COPIES * A1_F4;  // 1x1: double-buffered (a new tile every step)
  static constexpr int B_F4 = 8 * BN;                            // LDS image of one tap's weight tile (hi and lo planes)
  static constexpr bool ONE = (TERMS == 1 || TERMS == 4);       // single-product modes: fp16 / bf16 operands, the hi plane only
  static constexpr int B_DMA_F4 = (ONE ? 4 : 8) * BN;            // what is fetched: the single-product modes need the hi plane only
  static constexpr int B_PER = (B_DMA_F4 + NTHR - 1) / NTHR;      // LDS-DMA instructions per (issuing) wave per weight tile
  static constexpr int NG = TAPS / TPS;      // pipeline steps ("groups" of TPS taps) per 32-channel chunk
  static constexpr int G_PER = TPS * B_PER;  // LDS-DMA instructions per wave per group
  static constexpr int G_F4 = TPS * B_F4;    // float4 per group in the ring
  static constexpr int ST_F4 = TN * BN;      // statistics fold area: [TN][BN] x (sum, sumsq) doubles == one float4 each
  static constexpr int LDS_F4 = A_F4 + R * G_F4 + ST_F4;
  static constexpr int A_CNT = 2 * A_SLOTS + 4;  // ordinary VGPR loads per thread per chunk (activations + GroupNorm scale/shift)
  // GEMM row of the tile (0 .. BM-1) -> (image in tile, pixel row, pixel column)
  static __device__ __forceinline__ void rowmap(int row, int& img, int& py, int& px) {
    if constexpr (SUB48) {
      constexpr int SPI = (TH / 4) * (TW / 8);  // 32-row patches per image
      const int q = row >> 5, rr = row & 31;
      img = q / SPI;
      const int qq = q % SPI;
      py = (qq / (TW / 8)) * 4 + (rr >> 3);
      px = (qq % (TW / 8)) * 8 + (rr & 7);
    } else {
      img = row / (TH * TW);
      py = (row / TW) % TH;
      px = row % TW;
    }
  }
  static_assert(TAPS % TPS == 0, "taps per step must divide the taps");
  static_assert(B_DMA_F4 % 64 == 0 && B_PER >= 1, "whole 1-KiB LDS-DMA instructions; a wave issues B_PER of them or none");
  static_assert(TERMS == 4 || TERMS == 3 || TERMS == 2 || TERMS == 1 || TERMS == 0,
                "3 = fp16 hi/lo split (fp32 accuracy), 2 = fp16 hi*hi + both cross terms in one block-scaled fp8 MFMA, 1 = plain fp16 operands, 4 = plain bf16 operands, 0 = fp32 operands (exact fp32 MFMA)");
  static_assert(ONE || B_DMA_F4 % NTHR == 0, "split mode: every wave owns the same number of distinct 1-KiB pieces");
  static_assert(TH * TW * TN == BM && TPI % OCT == 0 && TPI >= OCT, "tile / loader mapping");
  static_assert(R >= 2, "ring needs >= 2 slots");
};

// v * sigmoid(v) with v_rcp_f32 (1 ulp) in place of the IEEE division sequence (11 VALU instructions per value inside the
// activation staging, which runs on the same SIMDs as the MFMAs); the hi/lo fp16 split that follows keeps 22 bits anyway.
__device__ __forceinline__ float silu2(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ void split2(float v, _Float16& hi, _Float16& lo) {
  const float c = __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
  hi = (_Float16)c;
  lo = (_Float16)fmaf((float)hi, -1.0f, c);  // = c - hi exactly, one v_fma_mix_f32
}
// Two fp32 -> one register of two fp16 (round to nearest even: v_cvt_pk_f16_f32), and c - hi for the fp16 in the low / high half of such a
// register in ONE instruction (v_fma_mix_f32 c * 1.0 + (-hi), the half read in place: exact).  The split of a staged value used to cost
// v_cvt_f16_f32 + v_cvt_f32_f16 + v_sub_f32 per value on top of the packing; the staging VALU work is ~12 % of a 3x3 launch (measured by removing it).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, f16x2)); }
template <int HALF>
__device__ __forceinline__ float sub_packed_f16(float c, unsigned pk) {
  float r;
  if constexpr (HALF == 0) asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(c), "v"(pk));
  else asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(c), "v"(pk));
  return r;
}
// eight values -> hi (four registers of fp16 pairs) and the fp32 residuals c - hi; the values are clamped to [-lim_n, lim_p] first (both 0 on a
// padding pixel: conv zero padding applies after norm + activation)
__device__ __forceinline__ void split8(const float (&v)[8], float lim_n, float lim_p, float (&c8)[8], unsigned (&hi)[4], float (&l8)[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) c8[k] = __builtin_amdgcn_fmed3f(v[k], lim_n, lim_p);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    hi[k] = cvt_pk_f16(c8[2 * k], c8[2 * k + 1]);
    l8[2 * k] = sub_packed_f16<0>(c8[2 * k], hi[k]);
    l8[2 * k + 1] = sub_packed_f16<1>(c8[2 * k + 1], hi[k]);
  }
}
union F4H8b {
  float4 f4;
  f16x8 h8;
  bf16x8 b8;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One LDS-DMA instruction: 64 lanes x 16 B from per-lane global addresses to LDS bytes [lds_dst, lds_dst + 1 KiB).
// Issued from inline asm on purpose: hipcc then does not know an LDS write is pending and does not drain vmcnt(0)
// before every ds_read of the loop (it does when the __builtin_amdgcn_global_load_lds form is used next to LDS reads of
// the same array); completion is tracked by the counted waits below.  M0 carries the LDS base and is restored
// (cdna_hip_programming.md 5.7).
// Same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: the per-instruction address arithmetic is scalar, no
// VALU instruction goes into the MFMA stream for it.
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// Activation loads are issued from inline asm as well: a compiler-visible load makes hipcc put s_waitcnt vmcnt(0) in front
// of its first use, which also drains every LDS-DMA weight group in flight (measured: ~1.2 us stall per 32-channel chunk).
// The destinations are tied to the counted wait that precedes their first use (tie_regs) -- cdna_hip_programming.md 5.7.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gload16x2(f32x4& d0, f32x4& d1, const void* p) {
  asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(d0), "=&v"(d1) : "v"(p) : "memory");
}
__device__ __forceinline__ void tie_regs(f32x4& x0, f32x4& x1) { asm volatile("" : "+v"(x0), "+v"(x1)::"memory"); }
// Agent-scope accesses of the fused split-K hand-off (sc1: coherent across the 8 XCD-private L2s by themselves, the way relaxed agent-scope atomics
// are -- LLVM AMDGPU memory model, gfx942 rows "load / store atomic monotonic agent").  The slabs move only through these, so the hand-off needs
// ordering (s_waitcnt vmcnt(0) before the ticket) but no L2 write-back / invalidate: MI355X_MICROARCH.md "Valid forms", first row of its table
// (one lane of each storing workgroup adds to ONE counter after every storing wave's vmcnt(0) wait and a barrier; the workgroup whose add came
// last loads after a barrier that lane joins; every store and every load of the bytes is a 16-byte global sc1 access).  Checked on this shape
// of hand-off by tools/probes/sc1_handoff_probe.hip: 0 stale values in 2 x 400 launches with the splits of a tile on one XCD and on different
// XCDs (plain accesses without the fences: 358 M stale values across XCDs), 9 / 15 us per launch against 33 / 57 us with the fences.
// (s_nop 1: a store of more than 8 bytes reads its data registers after issue -- the wait states the compiler inserts before it overwrites them
//  are invisible to it inside asm; without them the next quad transpose corrupted the slab)
// -DDRM_SK_FENCED=1: the conservative form of the same hand-off, kept compilable for A/Bs and as the switch to throw should a stale slab
// ever show -- plain slab stores / loads around an agent-scope release fence (one lane, before the ticket) and acquire fences (every wave of the
// finishing workgroup, after it): the LLVM memory model's documented release / acquire pair instead of per-access sc1 coherence.  19 vs 7 us per
// hand-off at the 64x64 level of the batch-1 step (profiles/r04_sc1_handoff_probe.txt); tools/skab.sh compares the two builds.
#ifdef DRM_SK_FENCED
__device__ __forceinline__ void gstore16_agent(void* p, const f32x4& v) { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void gload16_agent(f32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(d) : "v"(p) : "memory"); }
#else
__device__ __forceinline__ void gstore16_agent(void* p, const f32x4& v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void gload16_agent(f32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(d) : "v"(p) : "memory"); }
#endif
__device__ __forceinline__ void tie_reg(f32x4& x) { asm volatile("" : "+v"(x)::"memory"); }

// 4 x 4 transpose across the four lanes of a quad: in, lane q holds x_k = M[q][k]; out, lane q holds x_k = M[k][q].
// Two butterfly stages (lane bit 0 with register pairs (0,1), (2,3); lane bit 1 with pairs (0,2), (1,3)); each moves one register
// per pair through a DPP quad permute and selects by lane parity.
template <int CTRL>
__device__ __forceinline__ float quad_perm(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ void quad_transpose(float& x0, float& x1, float& x2, float& x3, int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
  float t;
  t = quad_perm<0xB1>(b0 ? x0 : x1);  // [1,0,3,2]
  if (b0) x0 = t; else x1 = t;
  t = quad_perm<0xB1>(b0 ? x2 : x3);
  if (b0) x2 = t; else x3 = t;
  t = quad_perm<0x4E>(b1 ? x0 : x2);  // [2,3,0,1]
  if (b1) x0 = t; else x2 = t;
  t = quad_perm<0x4E>(b1 ? x1 : x3);
  if (b1) x1 = t; else x3 = t;
}

template <int... I, typename F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}

struct TilePos {
  int n0, ty0, tx0, co0;
  long long wofs;  // float4 offset of the tile's weights inside one tap/chunk block: its Cout columns (+ its image's own weight set)
};

// RAG: the map is not a whole number of tiles: edge tiles are masked (loads are bounds-checked in every build; the ragged build also
// masks the epilogue's stores, residual reads and statistics per pixel).  A separate instantiation, so the shipped shapes' code is untouched.
// SK: the instantiation split-K launches use (128-row x 32-channel tiles only): it carries the fused finish of the tile -- a separate
// instantiation because its slab loads cost the narrow kernels 45 registers and one of their three waves per SIMD.
// POOL: the launch also writes the 2x2 average pool of its output and that tensor's statistics (ConvArgs::pool_out / pool_stat): the Downsample
// that follows the last ResBlock of a level (openaimodel.py:154-160) re-read a map this epilogue has in registers (avgpool2_kernel: nine launches
// and 2.4 GB of HBM traffic per batch-32 step).  A lane's 16 accumulator rows of a block are a 4 x 4 pixel patch: four pooled pixels, no exchange.
template <int TAPS, int TH, int TW, int WM, int WN, int MT, int NT, int R, int TPS, int TERMS = 3, bool RAG = false, bool SK = false, bool POOL = false>
__global__ __launch_bounds__(WM* WN * 64, 2) void conv_split2_kernel(ConvArgs a) {
  using C = S2Cfg<TAPS, TH, TW, WM, WN, MT, NT, R, TPS, TERMS>;
  static_assert(!POOL || (C::SUB48 && C::TN == 1 && !RAG && !SK), "pooled output: one-image tiles of 4 x 8 patches");
  extern __shared__ float4 lds[];
  float4* As = lds;                                                   // [hl 2][s 2][h 2][HP]  16-byte entries
  float4* Bs = lds + C::A_F4;                                         // R x TPS x [hl 2][s 2][h 2][BN]
  double* lst = reinterpret_cast<double*>(lds + C::A_F4 + R * C::G_F4);  // [TN][BN][2]
  [[maybe_unused]] double* lst2 = lst + C::TN * C::BN * 2;               // POOL: the pooled tensor's statistics, same shape

#ifndef DRM_NO_KERNARG_TOUCH
  {
    // The descriptor is six 64-byte lines of kernel arguments, and under this kernel's SGPR pressure the compiler fetches (and re-fetches) its
    // fields one s_load round trip at a time all along the prologue: on a sparse launch every first touch of a line is a serial miss.  One
    // dword of every line is requested here, all at once, so the later loads hit the scalar cache.
    typedef const __attribute__((address_space(4))) int* kptr_t;
    kptr_t ka = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    int touch = 0;
#pragma unroll
    for (int o = 0; o < (int)sizeof(ConvArgs); o += 64) touch ^= __builtin_nontemporal_load(ka + o / 4);
    asm volatile("" ::"s"(touch));
  }
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  // ---- persistent tile walk.  Logical tile ids are dealt to the 8 XCDs in contiguous ranges (blocks b and b+8 share an
  //      XCD and its L2); inside a range consecutive ids are the Cout tiles of one pixel tile, then spatial neighbours.
  const int tiles_x = RAG ? (a.W + TW - 1) / TW : a.W / TW, tiles_y = RAG ? (a.H + TH - 1) / TH : a.H / TH;
  const int n_tiles = a.Cout / C::BN;
  const int total = ((a.N + C::TN - 1) / C::TN) * tiles_y * tiles_x * n_tiles;
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int q8 = total >> 3, r8 = total & 7;
  const int x_start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int x_count = q8 + (xcd < r8 ? 1 : 0);
  const int J = ((int)gridDim.x - xcd + 7) >> 3;  // workgroups living on this XCD
  auto decode = [&](int logical) {
    TilePos t;
    const int n_tile = logical % n_tiles;
    int m_tile = logical / n_tiles;
    const int tx = m_tile % tiles_x;
    m_tile /= tiles_x;
    const int ty = m_tile % tiles_y;
    t.n0 = (m_tile / tiles_y) * C::TN;
    t.ty0 = ty * TH;
    t.tx0 = tx * TW;
    t.co0 = n_tile * C::BN;
    t.wofs = t.co0 + (long long)t.n0 * a.w_img_stride_f4;  // per-image weights (attention GEMMs): one image per tile
    return t;
  };
  int k_tile = jx;  // index inside this XCD's range
  if (k_tile >= x_count) return;
  TilePos cur = decode(x_start + k_tile);
#ifdef DRM_S2_STAMP
  unsigned* stamp_lds = reinterpret_cast<unsigned*>(lds + C::LDS_F4);  // [NW][120][2]
  int stamp_n = 0;
  int stamp_tiles = 0;
  bool stamp_on = false;
  if (a.stamp_out && (int)blockIdx.x == (a.stamp_block & 0xFFFF))
    for (int k = tid; k < C::NW * 240; k += C::NTHR) stamp_lds[k] = 0;
  stamp_on = a.stamp_out && (int)blockIdx.x == (a.stamp_block & 0xFFFF) && a.stamp_tile0 == 0;  // (first tile recorded: the prologue too)
  S2_STAMP(60);  // kernel entry
#endif

  const int Ctot = a.C0 + a.C1;
  // split-K (deep, small maps: too few output tiles to fill the chip, long K): blockIdx.y owns a contiguous range of the
  // 32-channel chunks and writes its partial result to its own slab; a fixed-order reduction sums the slabs (deterministic)
  const int nch_all = Ctot / C::KC;
  const int ks = a.ksplit > 1 ? a.ksplit : 1;
  const int split = blockIdx.y;
  const int chunk0 = split * nch_all / ks;
  const int nchunks = (split + 1) * nch_all / ks - chunk0;  // chunks of THIS workgroup (indices below are local)
  const int NGT = nchunks * C::NG;  // weight groups (pipeline steps) per tile
  // split-K: this split's partial slab.  Fused form (a.split_ws): the slabs live in their own workspace and a.out is the real output; two-launch
  // form: a.out is the workspace and splitk_reduce_kernel follows.
  constexpr bool FUSE = SK;
  static_assert(!SK || (WM == 4 && WN == 1 && MT == 1 && NT == 1 && !RAG), "split-K launches use the 128-row x 32-channel tiles (conv_split_ksplit)");
  const bool fused = FUSE && ks > 1 && a.split_ws != nullptr;
  float* const out_s = (fused ? a.split_ws : a.out) + (size_t)split * a.split_stride;  // (split_stride = 0 without split-K)
  const bool has_gn = a.gn_scale != nullptr;

  // ---- activation loader (ordinary loads, always issued: addresses are clamped, invalid slots zeroed at store time)
  const int l_img = tid / C::TPI;
  const int l_tid = tid % C::TPI;
  const int l_o = tid % C::OCT;
  f32x4 areg[C::A_SLOTS][2];
  f32x4 sc[2], sh[2];
  unsigned avalid = 0;

  auto load_A_piece = [&](const TilePos& tp, int chunk, int piece) {
    const int c = (chunk0 + chunk) * C::KC;
    const float* src;
    int Cs, coff, up;
    if (c < a.C0) {
      src = a.src0; Cs = a.ld0 ? a.ld0 : a.C0; coff = c; up = a.up0;
    } else {
      src = a.src1; Cs = a.C1; coff = c - a.C0; up = 0;
    }
    const int Hs = up ? (a.H >> 1) : a.H, Ws = up ? (a.W >> 1) : a.W;
    const int l_n = tp.n0 + l_img;
    const int l_nc = l_n < a.N ? l_n : a.N - 1;
    if (piece == 0) avalid = 0;
#pragma unroll
    for (int j = 0; j < C::A_SLOTS; ++j) {
      if (j != piece) continue;
      const int lidx = l_tid + C::TPI * j;
      const int hpl = lidx / C::OCT;
      const int hy = hpl / C::WT, hx = hpl % C::WT;
      const int y = tp.ty0 + hy - C::HALO, x = tp.tx0 + hx - C::HALO;
      const bool ok = (lidx < C::HPI * C::OCT) && (l_n < a.N) && (y >= 0) && (y < a.H) && (x >= 0) && (x < a.W);
      const int yc = min(max(y, 0), a.H - 1), xc = min(max(x, 0), a.W - 1);
      const int ys = up ? (yc >> 1) : yc, xs = up ? (xc >> 1) : xc;
      const size_t pix = ((size_t)l_nc * Hs + ys) * Ws + xs;
      gload16x2(areg[j][0], areg[j][1], src + pix * Cs + coff + 8 * l_o);
      if (ok) avalid |= 1u << j;
    }
    if (piece == C::A_SLOTS) {  // always 4 loads (a harmless re-read of the input when there is no GroupNorm) so every chunk issues A_CNT loads
      const float* gs = has_gn ? a.gn_scale + (size_t)l_nc * Ctot + c + 8 * l_o : a.src0;
      const float* gb = has_gn ? a.gn_shift + (size_t)l_nc * Ctot + c + 8 * l_o : a.src0;
      gload16x2(sc[0], sc[1], gs);
      gload16x2(sh[0], sh[1], gb);
    }
  };
  auto load_A = [&](const TilePos& tp, int chunk) {
#pragma unroll
    for (int p = 0; p <= C::A_SLOTS; ++p) load_A_piece(tp, chunk, p);
  };
  // the loads of load_A have landed (a counted wait came first): make every later use of their registers depend on this point
  auto tie_A = [&]() {
#pragma unroll
    for (int j = 0; j < C::A_SLOTS; ++j) tie_regs(areg[j][0], areg[j][1]);
    tie_regs(sc[0], sc[1]);
    tie_regs(sh[0], sh[1]);
  };
  // Staging of a loaded slot in two halves: transform (GroupNorm affine + SiLU + hi/lo split, VALU only, IN PLACE in the slot's eight
  // registers: areg[j][0] <- eight hi halfs or the first four fp32 values, areg[j][1] <- the lo halfs or the other four) and write (two
  // ds_write_b128).  At a chunk end of the 3x3 loop the transform runs BEFORE the barrier that frees the activation tile: the wave of a SIMD
  // that finishes its MFMAs first transforms under its partner's MFMAs instead of idling at that barrier, and the later one then has the
  // SIMD's VALU to itself (after the barrier both used to share it) -- only the LDS writes are left between the two chunk-end barriers.
  auto transform_A_slot = [&](int j) {
    const int lidx = l_tid + C::TPI * j;
    if (lidx < C::HPI * C::OCT) {
      float v[8] = {areg[j][0].x, areg[j][0].y, areg[j][0].z, areg[j][0].w, areg[j][1].x, areg[j][1].y, areg[j][1].z, areg[j][1].w};
      if (has_gn) {
        const float s8[8] = {sc[0].x, sc[0].y, sc[0].z, sc[0].w, sc[1].x, sc[1].y, sc[1].z, sc[1].w};
        const float b8[8] = {sh[0].x, sh[0].y, sh[0].z, sh[0].w, sh[1].x, sh[1].y, sh[1].z, sh[1].w};
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fmaf(v[k], s8[k], b8[k]);  // (one v_fma_f32 per value; the unfused form compiled to packed mul + packed add, twice the issue time)
      }
      if (a.silu) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = silu2(v[k]);
      }
      const bool ok = (avalid >> j) & 1u;  // conv zero padding applies after norm + activation
      if constexpr (TERMS == 0) {
        // exact-fp32 mode: the same LDS image with fp32 entries -- plane g holds channels 4g .. 4g+3 of the chunk (this thread's octet = planes
        // 2 * l_o and 2 * l_o + 1), the same bytes per element as the hi + lo halves
        areg[j][0] = ok ? f32x4{v[0], v[1], v[2], v[3]} : f32x4{0.f, 0.f, 0.f, 0.f};
        areg[j][1] = ok ? f32x4{v[4], v[5], v[6], v[7]} : f32x4{0.f, 0.f, 0.f, 0.f};
      } else if constexpr (TERMS == 2) {
        // eight fp16 hi halfs + the two fp8 images of the octet: al8 = e4m3((v - hi) * 2^(19-EA)), ah8 = e4m3(v * 2^(8-EA))
        // The staged value itself is clamped to +-MX_A_LIM = 3584 (v_cvt_scalef32_pk_fp8_f32 makes NaN above 448 * 2^(EA-8); no GroupNorm + SiLU
        // output gets near: |GN| <= sqrt(group size) * |gamma| + |beta|).  Clamping only the two fp8 images and leaving the fp16 hi operand on
        // the +-65504 clamp of the other split modes was built and measured: two more v_med3 per value, 782 vs 792 steps/s on
        // one box (-1.3 %) for a range no activation of these networks reaches -- not adopted; the limit is stated in include/drmnet_hip.h
        // (DRM_PREC_F16MX) and tests/test_gpu_f16mx.py drives an input beyond it (finite, saturated).
        float c8[8], l8[8];
        unsigned hi[4];
        split8(v, ok ? -MX_A_LIM : 0.f, ok ? MX_A_LIM : 0.f, c8, hi, l8);
        s16x2 q[4];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          s16x2 z = {0, 0};
          z = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, l8[4 * k], l8[4 * k + 1], MX_AL_DIV, false);
          q[k] = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, l8[4 * k + 2], l8[4 * k + 3], MX_AL_DIV, true);
          s16x2 y = {0, 0};
          y = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(y, c8[4 * k], c8[4 * k + 1], MX_AH_DIV, false);
          q[2 + k] = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(y, c8[4 * k + 2], c8[4 * k + 3], MX_AH_DIV, true);
        }
        areg[j][0] = f32x4{__builtin_bit_cast(float, hi[0]), __builtin_bit_cast(float, hi[1]), __builtin_bit_cast(float, hi[2]), __builtin_bit_cast(float, hi[3])};
        areg[j][1] = f32x4{__builtin_bit_cast(float, q[0]), __builtin_bit_cast(float, q[1]), __builtin_bit_cast(float, q[2]), __builtin_bit_cast(float, q[3])};
      } else if constexpr (TERMS == 4) {
        F4H8b hi;  // bf16 operands (round to nearest even: v_cvt_pk_bf16_f32); no range clamp needed, bf16 has fp32's exponent
#pragma unroll
        for (int k = 0; k < 8; ++k) hi.b8[k] = (__bf16)(ok ? v[k] : 0.f);
        areg[j][0] = f32x4{hi.f4.x, hi.f4.y, hi.f4.z, hi.f4.w};
      } else {
        float c8[8], l8[8];
        unsigned hi[4];
        split8(v, ok ? -65504.0f : 0.f, ok ? 65504.0f : 0.f, c8, hi, l8);
        areg[j][0] = f32x4{__builtin_bit_cast(float, hi[0]), __builtin_bit_cast(float, hi[1]), __builtin_bit_cast(float, hi[2]), __builtin_bit_cast(float, hi[3])};
        if constexpr (TERMS == 3)
          areg[j][1] = f32x4{__builtin_bit_cast(float, cvt_pk_f16(l8[0], l8[1])), __builtin_bit_cast(float, cvt_pk_f16(l8[2], l8[3])),
                             __builtin_bit_cast(float, cvt_pk_f16(l8[4], l8[5])), __builtin_bit_cast(float, cvt_pk_f16(l8[6], l8[7]))};
      }
    }
  };
  auto write_A_slot = [&](float4* Ad, int j) {
    const int lidx = l_tid + C::TPI * j;
    if (lidx < C::HPI * C::OCT) {
      const int hpl = lidx / C::OCT;
      const int pixel = l_img * C::HPIP + (hpl / C::WT) * C::WTP + (hpl % C::WT);
      const float4 w0 = make_float4(areg[j][0].x, areg[j][0].y, areg[j][0].z, areg[j][0].w);
      const float4 w1 = make_float4(areg[j][1].x, areg[j][1].y, areg[j][1].z, areg[j][1].w);
      if constexpr (TERMS == 0) {
        Ad[(2 * l_o) * C::HPS + pixel] = w0;
        Ad[(2 * l_o + 1) * C::HPS + pixel] = w1;
      } else if constexpr (TERMS == 2) {
        // hi plane as in every fp16 mode; planes 4 + g (al8) and 6 + g (ah8) hold the 16 channels of group g = octet / 2, one byte each:
        // this octet owns 8 bytes of either
        Ad[l_o * C::HPS + pixel] = w0;
        reinterpret_cast<float2*>(Ad + (4 + (l_o >> 1)) * C::HPS + pixel)[l_o & 1] = make_float2(w1.x, w1.y);
        reinterpret_cast<float2*>(Ad + (6 + (l_o >> 1)) * C::HPS + pixel)[l_o & 1] = make_float2(w1.z, w1.w);
      } else {
        Ad[l_o * C::HPS + pixel] = w0;
        if (TERMS == 3) Ad[(4 + l_o) * C::HPS + pixel] = w1;
      }
    }
  };
  auto transform_A = [&]() {
    tie_A();
#pragma unroll
    for (int j = 0; j < C::A_SLOTS; ++j) transform_A_slot(j);
  };
  auto write_A = [&](float4* Ad) {
#pragma unroll
    for (int j = 0; j < C::A_SLOTS; ++j) write_A_slot(Ad, j);
  };
  auto store_A = [&](float4* Ad) {
    transform_A();
    write_A(Ad);
  };
  // ---- weight groups: LDS-DMA, G_PER x 1 KiB per wave per group; LDS image == packed global layout.
  //      `gseq` counts groups since kernel start (ring slot = gseq % R); (g_in_tile, co0) say which weights.
  const unsigned lds_bs = __builtin_amdgcn_readfirstlane(
      (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(Bs));  // LDS byte offset of the ring
  // per-lane byte offset of the j-th 1-KiB piece inside a weight tile (the tile's base is wave-uniform: scalar registers)
  unsigned dma_voff[C::B_PER];
#pragma unroll
  for (int j = 0; j < C::B_PER; ++j) {
    const int idx = (wave * 64 + C::NTHR * j) % C::B_DMA_F4 + lane;
    dma_voff[j] = (unsigned)((idx / C::BN) * a.Cout + idx % C::BN) * 16u;
  }
  // k-th LDS-DMA instruction (0 .. G_PER-1) of a group
  auto issue_G1 = [&](int gseq, int g_in_tile, long long co0, int k) {
    const int slot = gseq % R;
    const int chunk = g_in_tile / C::NG, g = g_in_tile - chunk * C::NG;
    const int u = k / C::B_PER, j = k % C::B_PER;
    const int tap = g * TPS + u;
    const float4* wp = reinterpret_cast<const float4*>(a.w) + (((size_t)tap * nch_all + chunk0 + chunk) * 8) * a.Cout + co0;
    // wave-uniform float4 index inside the tile.  Single-product mode on narrow tiles has fewer 1-KiB pieces than waves: the
    // upper waves re-fetch a piece (same bytes to the same LDS address) so that EVERY wave issues the same number of
    // vector-memory operations -- the counted waits rely on that.
    const int base = (wave * 64 + C::NTHR * j) % C::B_DMA_F4;
    unsigned vo = dma_voff[0];
#pragma unroll
    for (int jj = 1; jj < C::B_PER; ++jj) vo = (j == jj) ? dma_voff[jj] : vo;
    glds16s(wp, vo, lds_bs + (unsigned)(slot * C::G_F4 + u * C::B_F4 + base) * 16u);
  };
  auto issue_G = [&](int gseq, int g_in_tile, long long co0) {
    const int slot = gseq % R;
    const int chunk = g_in_tile / C::NG, g = g_in_tile - chunk * C::NG;
#pragma unroll
    for (int u = 0; u < TPS; ++u) {
      const int tap = g * TPS + u;
      const float4* wp = reinterpret_cast<const float4*>(a.w) + (((size_t)tap * nch_all + chunk0 + chunk) * 8) * a.Cout + co0;
#pragma unroll
      for (int j = 0; j < C::B_PER; ++j) {
        const int base = (wave * 64 + C::NTHR * j) % C::B_DMA_F4;  // (see issue_G1)
        glds16s(wp, dma_voff[j], lds_bs + (unsigned)(slot * C::G_F4 + u * C::B_F4 + base) * 16u);
      }
    }
  };

  int a_base[MT], b_base[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = (wm * MT + i) * 32 + r;
    int img, py, px;
    C::rowmap(row, img, py, px);
    a_base[i] = img * C::HPIP + py * C::WTP + px;
  }
#pragma unroll
  for (int c = 0; c < NT; ++c) b_base[c] = (wn * NT + c) * 32 + r;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][c][e] = 0.f;

  const float inv_scale = a.w_inv_scale ? *a.w_inv_scale : 1.0f;
  constexpr int PPI = TH * TW;  // GEMM rows per image inside the tile: 16 .. 256
  const bool st = a.stat_out != nullptr;
  if (st) {
    for (int k = tid; k < C::TN * C::BN * 2; k += C::NTHR) lst[k] = 0.0;
  }
  if constexpr (POOL) {
    for (int k = tid; k < C::TN * C::BN * 2; k += C::NTHR) lst2[k] = 0.0;
  }

  // ---- prologue: R-1 weight groups in flight (they may already belong to the next tile when a tile has < R-1 groups),
  //      first activation tile staged
  int gseq = 0;  // groups issued so far
  {
    int k2 = k_tile;
    TilePos tp = cur;
    int gi = 0;
#pragma unroll
    for (int G = 0; G < R - 1; ++G) {
      if (gi >= NGT) {  // spill into the following tile (or wrap on the last one: harmless duplicate)
        gi = 0;
        if (k2 + J < x_count) {
          k2 += J;
          tp = decode(x_start + k2);
        }
      }
      issue_G(gseq++, gi++, tp.wofs);
    }
  }
  S2_STAMP(61);  // first weight groups requested
  if (a.gnf.mom0) {
    // Sparse launches (batch-1 steps): the GroupNorm tables this launch stages through are finalised here, by every workgroup (identical values),
    // instead of by a launch of their own -- 111 launches of ~5 us per batch-1 DRMNet step; the weight groups above are already in flight.
    // The raw activation loads of the first chunk go out first (they do not depend on the tables).  Scratch: the activation tile, not yet
    // written.  One image and room in the tile: the first chunk takes its (scale, shift) from an LDS copy of the tables -- no wait for the
    // global tables to be written and read back; later chunks read the global tables.
    float* scr = reinterpret_cast<float*>(As);
    const bool tab_in_lds = a.N == 1 && (80 + 2 * Ctot) * (int)sizeof(float) <= C::A_F4 * 16;
#pragma unroll
    for (int p = 0; p < C::A_SLOTS; ++p) load_A_piece(cur, 0, p);
    for (int n = 0; n < a.N; ++n) gn_finalize_image(a.gnf, n, tid, C::NTHR, scr, tab_in_lds ? scr + 80 : nullptr);
    S2_STAMP(62);  // GroupNorm tables finalised
    // (gn_finalize_image's barriers order LDS only: a wave's table stores are complete after ITS vmcnt(0) wait, everyone's after the barrier
    //  that follows it -- before any lane reads a table entry another wave wrote)
    wait_vmcnt<0>();
    if (!tab_in_lds) {
      gn_lds_barrier();
      load_A_piece(cur, 0, C::A_SLOTS);
    } else {  // (no table loads for this chunk: the wait below is vmcnt(0), not a counted one)
      const float4* t4 = reinterpret_cast<const float4*>(scr + 80 + chunk0 * C::KC + 8 * l_o);
      const float4 s0 = t4[0], s1 = t4[1], b0 = t4[Ctot / 4], b1 = t4[Ctot / 4 + 1];
      sc[0] = f32x4{s0.x, s0.y, s0.z, s0.w}; sc[1] = f32x4{s1.x, s1.y, s1.z, s1.w};
      sh[0] = f32x4{b0.x, b0.y, b0.z, b0.w}; sh[1] = f32x4{b1.x, b1.y, b1.z, b1.w};
      gn_lds_barrier();  // every lane has its table entries before the tile is written over them
    }
  } else {
    load_A(cur, 0);
  }
  wait_vmcnt<0>();
  S2_STAMP(63);  // first activation tile (and the first weight groups) landed
  store_A(As);
  if (TAPS == 1) {  // 1x1: the activations of step 1 are requested a full step ahead
    if (nchunks > 1) load_A(cur, 1);
    else if (k_tile + J < x_count) load_A(decode(x_start + k_tile + J), 0);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  S2_STAMP(64);  // prologue done

  // Per step (group g of chunk c of the current tile; g is a compile-time constant):
  //   (1) issue the DMA of the group R-1 steps ahead into the slot consumed one step ago (it may belong to the NEXT tile),
  //   (2) at group A_G request the next chunk's activations -- or the first chunk of the next tile,
  //   (3) MFMAs of the TPS taps of the group,
  //   (4) counted wait: everything up to the next group landed; allowed in flight = the R-2 younger groups
  //       (+ the activation loads while they are younger than the next group), then ONE barrier per TPS taps.
  constexpr int A_G = (C::NG >= 3) ? C::NG - 2 : 0;
  // The two waves that share a SIMD (w and w + NW/2 of an 8-wave workgroup) run the two halves of every step in opposite
  // order: one issues DMA / loads / staging VALU work while the other owns the MFMA pipe, then they swap.  Measured with
  // the phase timers of the experiment build: with every wave doing the same phase at the same time the MFMA pipe sat
  // idle 55 % of the main loop (17 % in DMA issue alone: the CU's texture-address path serialises the 1-KiB DMAs).
  const bool y_first = (C::NW == 8) && (wave >= C::NW / 2);
  // 3x3: static priority for the second-dispatched wave half, which otherwise loses every issue arbitration by age and arrives last at every
  // barrier (A/B on one box: 3x3 family -0.6 %; the HBM-bound 1x1 form gets 5.7 % slower with it, so it stays without)
  if (TAPS == 9 && C::NW == 8 && wave >= C::NW / 2) __builtin_amdgcn_s_setprio(1);
  constexpr int BASE = C::G_PER * (R - 2);
  int step = 0;  // steps executed so far (== gseq - (R-1))
  while (true) {
    const bool has_next = k_tile + J < x_count;
    const TilePos nxt = has_next ? decode(x_start + k_tile + J) : cur;
#ifdef DRM_S2_STAMP
    stamp_on = a.stamp_out && (int)blockIdx.x == (a.stamp_block & 0xFFFF) && stamp_tiles >= a.stamp_tile0;
    ++stamp_tiles;
    S2_STAMP(1);  // tile start
#endif
    // One tap (or 32-channel slab pair) of MFMAs: LDS fragment reads + 3 MFMAs per 32x32x16 product
    auto mma_tap = [&](const float4* Ab, const float4* Bc, int tapoff, auto&& hook) {
      if constexpr (TERMS == 0) {
        // exact fp32: v_mfma_f32_32x32x2_f32 (one fp32 per lane and operand; lane half h supplies k = h).  Lane half h reads k-group 2j + h,
        // so one b128 read per operand feeds four MFMAs; 16 MFMAs of 64 cycles per (tap, block): the matrix pipe has 5.3x the work of the
        // split form per staged byte, under the same weight ring / staging / barrier schedule.
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int gi = 2 * j + h;
          float4 af[MT], bf[NT];
#pragma unroll
          for (int i = 0; i < MT; ++i) af[i] = Ab[gi * C::HPS + a_base[i] + tapoff];
#pragma unroll
          for (int c = 0; c < NT; ++c) bf[c] = Bc[gi * C::BN + b_base[c]];
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int c = 0; c < NT; ++c) {
              acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[c].x, acc[i][c], 0, 0, 0);
              acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[c].y, acc[i][c], 0, 0, 0);
              acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[c].z, acc[i][c], 0, 0, 0);
              acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[c].w, acc[i][c], 0, 0, 0);
            }
          if (j == 0 || j == 2) {
            __builtin_amdgcn_sched_barrier(0);
            hook(j >> 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        return;
      }
      if constexpr (TERMS == 2) {
        // both cross terms of the chunk: lane half h reads the al8 / ah8 (bh8 / bl8) planes of channel group h
        {
          i32x8 am[MT], bm[NT];
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const float4 p0 = Ab[(4 + h) * C::HPS + a_base[i] + tapoff], p1 = Ab[(6 + h) * C::HPS + a_base[i] + tapoff];
            am[i] = i32x8{__builtin_bit_cast(int, p0.x), __builtin_bit_cast(int, p0.y), __builtin_bit_cast(int, p0.z), __builtin_bit_cast(int, p0.w),
                          __builtin_bit_cast(int, p1.x), __builtin_bit_cast(int, p1.y), __builtin_bit_cast(int, p1.z), __builtin_bit_cast(int, p1.w)};
          }
#pragma unroll
          for (int c = 0; c < NT; ++c) {
            const float4 p0 = Bc[(4 + h) * C::BN + b_base[c]], p1 = Bc[(6 + h) * C::BN + b_base[c]];
            bm[c] = i32x8{__builtin_bit_cast(int, p0.x), __builtin_bit_cast(int, p0.y), __builtin_bit_cast(int, p0.z), __builtin_bit_cast(int, p0.w),
                          __builtin_bit_cast(int, p1.x), __builtin_bit_cast(int, p1.y), __builtin_bit_cast(int, p1.z), __builtin_bit_cast(int, p1.w)};
          }
          const int sa = h ? MX_SA_AH : MX_SA_AL, sb = h ? MX_SB_BL : MX_SB_BH;
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[i][c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(am[i], bm[c], acc[i][c], 0, 0, 0, sa, 0, sb);
        }
        __builtin_amdgcn_sched_barrier(0);
        hook(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int seg = s2 * 2 + h;
          F4H8b ah[MT], bh[NT];
#pragma unroll
          for (int i = 0; i < MT; ++i) ah[i].f4 = Ab[seg * C::HPS + a_base[i] + tapoff];
#pragma unroll
          for (int c = 0; c < NT; ++c) bh[c].f4 = Bc[seg * C::BN + b_base[c]];
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i].h8, bh[c].h8, acc[i][c], 0, 0, 0);
          if (s2 == 0) {
            __builtin_amdgcn_sched_barrier(0);
            hook(1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        return;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int seg = s2 * 2 + h;
        F4H8b ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          ah[i].f4 = Ab[seg * C::HPS + a_base[i] + tapoff];
          if (TERMS == 3) al[i].f4 = Ab[(4 + seg) * C::HPS + a_base[i] + tapoff];
        }
#pragma unroll
        for (int c = 0; c < NT; ++c) {
          bh[c].f4 = Bc[seg * C::BN + b_base[c]];
          if (TERMS == 3) bl[c].f4 = Bc[(4 + seg) * C::BN + b_base[c]];
        }
        // term-major order: the three products of one accumulator block are MT * NT instructions apart (per block the order of the
        // additions is unchanged: lo*hi, hi*lo, hi*hi)
        static_for(std::make_integer_sequence<int, 3>{}, [&](auto tc) {
          constexpr int t = decltype(tc)::value;
          constexpr int T0 = (TERMS == 3) ? 0 : 2;  // single-product mode: hi*hi only
          if constexpr (t >= T0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int c = 0; c < NT; ++c) {
                if constexpr (t == 0) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i].h8, bh[c].h8, acc[i][c], 0, 0, 0);
                if constexpr (t == 1) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i].h8, bl[c].h8, acc[i][c], 0, 0, 0);
                if constexpr (t == 2 && TERMS != 4) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i].h8, bh[c].h8, acc[i][c], 0, 0, 0);
                if constexpr (t == 2 && TERMS == 4) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i].b8, bh[c].b8, acc[i][c], 0, 0, 0);
              }
            if constexpr (t == T0) {
              // memory-side instructions of this step go out here, a few at a time, behind MFMAs that keep the pipe busy
              // while the (blocking) vector-memory issue waits for the CU's address path
              __builtin_amdgcn_sched_barrier(0);
              hook(s2);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        });
      }
    };
    // group R-1 steps ahead of tile-local group index gi0: inside this tile, else the matching group of the next tile
    // (else a harmless re-read)
    auto issue_ahead = [&](int gi0) {
      int gi = gi0 + (R - 1);
      long long co0 = cur.wofs;
      if (gi >= NGT) {
        gi -= NGT;
        if (gi >= NGT) gi %= NGT;
        co0 = nxt.wofs;
      }
      issue_G(gseq, gi, co0);
      ++gseq;
    };
    if constexpr (TAPS == 1) {
      // 1x1 pipeline: one 32-channel chunk per step, activation tiles double-buffered in LDS.  areg always holds the
      // NEXT step's activations (requested one step ago).  "Side work" of a step = stage them into the other LDS buffer,
      // request the step after next, issue the weight DMA R-1 steps ahead; the activation request goes out BEFORE the
      // DMA so that waiting for it (in-order retirement) never waits for a younger weight group.
      for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool n1 = (chunk + 1 < nchunks) || has_next;                      // a next step exists
        const bool n2 = (chunk + 2 < nchunks) || (has_next && nchunks >= 2);    // ... and one after it (persistent mode needs nchunks >= 2)
        auto side = [&]() {
          S2_STAMP(40);  // side work starts
          if (n1) {
            if (step == 0) wait_vmcnt<0>();       // first step: the request is the youngest operation
            else wait_vmcnt<C::G_PER>();          // younger: the weight group issued right after it
            S2_STAMP(41);  // next step's activations landed
            store_A(As + ((step + 1) & 1) * C::A1_F4);  // the buffer read one step ago: every wave passed that barrier
            S2_STAMP(42);  // ... and are staged
          }
          if (n2) {
            if (chunk + 2 < nchunks) load_A(cur, chunk + 2);
            else load_A(nxt, chunk + 2 - nchunks);
          }
          issue_ahead(chunk);
        };
        S2_STAMP(2);
        if (!y_first) side();
        mma_tap(As + (step & 1) * C::A1_F4, Bs + (step % R) * C::G_F4, 0, [](int) {});
        S2_STAMP(3);
        if (y_first) side();
        ++step;
        // (ring of 2: the group the next step reads was issued in THIS step, after the activation request: nothing may stay in flight)
        if (n2 && R > 2) wait_vmcnt<BASE + C::A_CNT>();
        else wait_vmcnt<BASE>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        S2_STAMP(7);
        __builtin_amdgcn_s_barrier();
        S2_STAMP(8);
      }
    } else {
    const int nch_run = nchunks;
    for (int chunk = 0; chunk < nch_run; ++chunk) {
      const bool more = chunk + 1 < nch_run;
      const bool a_next = more || has_next;  // activations to stage at the end of this chunk
      static_for(std::make_integer_sequence<int, C::NG>{}, [&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr bool last_g = (g == C::NG - 1);
        // group R-1 steps ahead: inside this tile, else the matching group of the next tile (else a harmless re-read)
        int d_gi = chunk * C::NG + g + (R - 1);
        long long d_co0 = cur.wofs;
        if (d_gi >= NGT) {
          d_gi -= NGT;
          if (d_gi >= NGT) d_gi %= NGT;
          d_co0 = nxt.wofs;
        }
        const int d_seq = gseq++;
        const bool a_req = (g == A_G) && a_next;
        // Hook points of a step: 2 per tap (one per 16-channel slab).  The DMA instructions go to the first hook points,
        // the activation request (in pieces) to the later ones, so the request stays younger than the whole group.
        constexpr int HPN = 2 * TPS;
        constexpr int DMA_HP = (HPN >= 4) ? HPN / 2 : HPN;  // hook points that carry DMA
        constexpr int A_HP0 = (HPN >= 4) ? HPN / 2 : HPN - 1;  // first hook point of the activation request
        constexpr int A_PIECES = C::A_SLOTS + 1;
        auto compute = [&]() {
          const float4* Bg = Bs + (step % R) * C::G_F4;
          const float4* Ac = As;
#pragma unroll
          for (int u = 0; u < TPS; ++u) {
            const int tap = g * TPS + u;
            mma_tap(Ac, Bg + u * C::B_F4, (tap / 3) * C::WTP + (tap % 3), [&](int s2) {
              const int hp = u * 2 + s2;
              if (hp < DMA_HP) {
#pragma unroll
                for (int k = hp * C::G_PER / DMA_HP; k < (hp + 1) * C::G_PER / DMA_HP; ++k) issue_G1(d_seq, d_gi, d_co0, k);
              }
              if (hp >= A_HP0 && a_req) {
                const int q = hp - A_HP0, nq = HPN - A_HP0;
#pragma unroll
                for (int pc = q * A_PIECES / nq; pc < (q + 1) * A_PIECES / nq; ++pc) {
                  if (more) load_A_piece(cur, chunk + 1, pc);
                  else load_A_piece(nxt, 0, pc);
                }
              }
            });
          }
        };
        S2_STAMP(2);  // step start
        compute();
        S2_STAMP(3);  // MFMAs + interleaved DMA / activation requests issued
        ++step;
        constexpr bool a_younger = (g >= A_G) && (g - A_G <= R - 2);
        if (last_g) {
#ifdef DRM_EXP_NOSTAGE  // timing experiment (wrong numbers): the chunk-end staging phase removed -- the bound of what hiding it can return
          if (a_next) {
            wait_vmcnt<C::G_PER * (C::NG - 1 - A_G)>();
            tie_A();
#if DRM_EXP_NOSTAGE >= 2
            transform_A();
            tie_A();  // (keeps the transform alive)
#endif
            wait_vmcnt<BASE>();
            __builtin_amdgcn_s_barrier();
          } else
#endif
          if (a_next) {
            wait_vmcnt<C::G_PER * (C::NG - 1 - A_G)>();  // the request of group A_G; younger: the weight groups issued after it
            S2_STAMP(5);  // activation loads landed
            transform_A();  // (registers only: ahead of the barrier, under the partner wave's MFMAs)
            __builtin_amdgcn_sched_barrier(0);  // (the compiler may not sink register-only work below the barrier)
            __builtin_amdgcn_s_barrier();  // every wave finished reading the old activation tile
            S2_STAMP(4);  // barrier 1 of the chunk end passed
            write_A(As);
            S2_STAMP(6);  // staged (GroupNorm affine + SiLU + split + LDS writes)
            wait_vmcnt<BASE>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            S2_STAMP(7);  // next weight group landed
            __builtin_amdgcn_s_barrier();
            S2_STAMP(8);  // barrier 2 passed
          } else {
            wait_vmcnt<BASE>();
            S2_STAMP(7);
            __builtin_amdgcn_s_barrier();
            S2_STAMP(8);
          }
        } else {
          // the activation loads were issued right after the group of step (A_G)+R-1: they are younger than the next
          // group while g - A_G <= R-2
          if (a_younger && a_next) wait_vmcnt<BASE + C::A_CNT>();
          else wait_vmcnt<BASE>();
          S2_STAMP(7);
          __builtin_amdgcn_s_barrier();
          S2_STAMP(8);
        }
      });
    }

    }

    S2_STAMP(9);  // epilogue start
    [[maybe_unused]] bool fused_last = false;  // fused split-K: this workgroup arrived last at its tile and ran the full epilogue
#ifdef DRM_S2_STAMP
    // timing experiments of the diagnostic build (tools/epi_cost.sh): stamp_block bit 16 = no output stores, bit 17 = no epilogue at all
    const bool skip_epilogue = a.stamp_block & 0x20000;
    if (skip_epilogue) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            t += acc[i][c][e];
            acc[i][c][e] = 0.f;
          }
      if (t == 12345.678f) out_s[0] = t;  // keeps the accumulators alive
    }
    if (!skip_epilogue)
#endif
    // ---- epilogue of the current tile.  Rows of a 32x32 accumulator tile held by this lane: 8g + 4h + k (g, k = 0..3);
    // the four k rows are four consecutive pixels of one image row (TW % 4 == 0), so addresses are formed once per
    // (i, g).  Loads are issued unconditionally on clamped addresses (batched ahead of the math); stores are predicated
    // and fire-and-forget: they drain while the next tile's main loop runs (they are OLDER than every vector-memory operation
    // the next tile counts, so their number does not enter the counted waits).
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      size_t pixb[4];
      int nimg[4];
      bool okg[4];
      [[maybe_unused]] unsigned okp = 0;  // RAG: validity of the 16 pixels (bit 4g + k) this lane holds of the block row group
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = (wm * MT + i) * 32 + 8 * g + 4 * h;
        int img, py, px;
        C::rowmap(row, img, py, px);
        const int n = cur.n0 + img;
        okg[g] = n < a.N;
        nimg[g] = okg[g] ? n : a.N - 1;
        if constexpr (RAG) {
          const int y = cur.ty0 + py, x = cur.tx0 + px;
          okg[g] = okg[g] && y < a.H && x < a.W;  // (the group's first pixel; its other three follow in okp)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (n < a.N && y < a.H && x + k < a.W) okp |= 1u << (4 * g + k);
          pixb[g] = ((size_t)nimg[g] * a.H + min(y, a.H - 1)) * a.W + min(x, a.W - 1);
        } else {
          pixb[g] = ((size_t)nimg[g] * a.H + (cur.ty0 + py)) * a.W + (cur.tx0 + px);
        }
      }
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        const int col = (wn * NT + c) * 32 + r;
        const int co = cur.co0 + col;
        if constexpr (RAG) {
          // masked edge tiles: plain per-element form (residual read, store and statistics under the pixel's own predicate)
          const bool first = ks == 1;
          const float bias = (a.bias && first) ? a.bias[co] : 0.f;
          float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int g = e >> 2, k = e & 3;
            const bool ok = (okp >> e) & 1u;
            float v = acc[i][c][e] * ((a.w_inv_img ? a.w_inv_img[nimg[g]] : inv_scale) * (a.in_inv ? a.in_inv[nimg[g]] : 1.0f)) + bias;
            acc[i][c][e] = 0.f;
            if (!ok) continue;
            const size_t pix = pixb[g] + k;
            if (a.emb && first) v += a.emb[(size_t)nimg[g] * a.emb_stride + co];
            if (a.res && first) v += a.res[pix * a.Cout + co];
            if (a.out_nchw) {
              const size_t hw = (size_t)a.H * a.W;
              if (co < a.cout_valid) a.out[((size_t)nimg[g] * a.cout_valid + co) * hw + (pix - (size_t)nimg[g] * hw)] = v;
            } else {
              out_s[pix * a.Cout + co] = v;
            }
            if (e < 8) {
              s0 += v;
              q0 += v * v;
            } else {
              s1 += v;
              q1 += v * v;
            }
          }
          if (st) {
            const int row0 = (wm * MT + i) * 32;
            if (PPI >= 32) {
              s0 += s1;
              q0 += q1;
            }
            double* d = lst + ((size_t)(row0 / PPI) * C::BN + col) * 2;
            atomicAdd(d, (double)s0);
            atomicAdd(d + 1, (double)q0);
            if (PPI < 32) {
              double* d2 = lst + ((size_t)((row0 + 16) / PPI) * C::BN + col) * 2;
              atomicAdd(d2, (double)s1);
              atomicAdd(d2 + 1, (double)q1);
            }
          }
          continue;
        }
        S2_STAMP(20 + 4 * (i * NT + c));  // block (i, c) of the epilogue starts
        // split-K: every split writes its raw partial sums to its own slab (a.out + split * slab); bias / emb / residual and the
        // statistics are applied by the fixed-order reduction that follows (splitk_reduce_kernel)
        const bool first = ks == 1;
        const float bias = (a.bias && first) ? a.bias[co] : 0.f;
        float rv[16], ev[4], sv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          ev[g] = (a.emb && first) ? a.emb[(size_t)nimg[g] * a.emb_stride + co] : 0.f;
          // weight and (per image) input staging factors, all 2^-k (attention GEMMs: the "weights" are per image, and in_inv carries alpha)
          sv[g] = (a.w_inv_img ? a.w_inv_img[nimg[g]] : inv_scale) * (a.in_inv ? a.in_inv[nimg[g]] : 1.0f);
        }
#ifdef DRM_S2_STAMP
        const bool abl_noload = a.stamp_block & 0x200000;  // timing experiments (tools/epi_ablate.sh): no residual loads
#else
        constexpr bool abl_noload = false;
#endif
        if (a.res && first && !abl_noload) {
#pragma unroll
          for (int e = 0; e < 16; ++e) rv[e] = a.res[(pixb[e >> 2] + (e & 3)) * a.Cout + co];
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) rv[e] = 0.f;
        }
        float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;  // rows 0-15 / 16-31 of this tile (two images when PPI == 16)
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int g = e >> 2;
          v[e] = acc[i][c][e] * sv[g] + bias + ev[g] + rv[e];
          acc[i][c][e] = 0.f;  // ready for the next tile
#ifdef DRM_S2_STAMP
          if (a.stamp_block & 0x100000) continue;  // timing experiment: no statistics arithmetic
#endif
          if (okg[g]) {
            if (e < 8) {
              s0 += v[e];
              q0 += v[e] * v[e];
            } else {
              s1 += v[e];
              q1 += v[e] * v[e];
            }
          }
        }
        S2_STAMP(21 + 4 * (i * NT + c));  // values ready (bias / emb / residual landed and applied, statistics partials)
        if constexpr (POOL) {
          // v[4 g + k] = pixel (row g, column 4 h + k) of this block's 4 x 8 patch: pooled pixel (gy, 2 h + kx) = mean of rows 2 gy, 2 gy + 1 and
          // columns 2 kx, 2 kx + 1, summed in avgpool2_kernel's order ((a + b) + c) + d
          float pv[4];
          float ps = 0.f, pq = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int gy = q >> 1, kx = q & 1;
            pv[q] = (v[8 * gy + 2 * kx] + v[8 * gy + 2 * kx + 1] + v[8 * gy + 4 + 2 * kx] + v[8 * gy + 4 + 2 * kx + 1]) * 0.25f;
            ps += pv[q];
            pq += pv[q] * pv[q];
          }
          quad_transpose(pv[0], pv[1], pv[2], pv[3], r);  // -> pooled pixel (r & 3) of this lane's four channels
          {
            const int row0 = (wm * MT + i) * 32;
            int img0, py0, px0;
            C::rowmap(row0, img0, py0, px0);
            const int q = r & 3;
            const int ppy = ((cur.ty0 + py0) >> 1) + (q >> 1), ppx = ((cur.tx0 + px0) >> 1) + 2 * h + (q & 1);
            const int cq4 = cur.co0 + (wn * NT + c) * 32 + (r & ~3);
            const int n = cur.n0;
            *reinterpret_cast<float4*>(&a.pool_out[(((size_t)n * (a.H >> 1) + ppy) * (a.W >> 1) + ppx) * a.Cout + cq4]) = make_float4(pv[0], pv[1], pv[2], pv[3]);
            double* d2 = lst2 + (size_t)col * 2;
            atomicAdd(d2, (double)ps);
            atomicAdd(d2 + 1, (double)pq);
          }
        }
        if (a.out_nchw) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int g = e >> 2;
            if (!okg[g] || co >= a.cout_valid) continue;
            const size_t pix = pixb[g] + (e & 3);
            const size_t hw = (size_t)a.H * a.W;
            a.out[((size_t)nimg[g] * a.cout_valid + co) * hw + (pix - (size_t)nimg[g] * hw)] = v[e];
          }
        } else {
          // NHWC store, 16 bytes per lane: the accumulator layout gives a lane ONE channel of four consecutive pixels per row
          // group g; a 4 x 4 transpose inside every quad of lanes (two DPP butterfly stages) turns that into FOUR channels of one
          // pixel, so a wave writes a 32-channel row group with 4 global_store_dwordx4 instead of 16 global_store_dword -- the store
          // tail was issue-bound (64 store instructions per wave and tile; stamp timeline: 19 % of a K = 1152 tile).
          const int cq = cur.co0 + (wn * NT + c) * 32 + (r & ~3);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
#ifdef DRM_S2_STAMP
            if (a.stamp_block & 0x400000) {  // timing experiment: no transposes, no stores
              if (v[4 * g] == 12345.678f) out_s[0] = v[4 * g + 1] + v[4 * g + 2] + v[4 * g + 3];
              continue;
            }
#endif
            quad_transpose(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3], r);
#ifdef DRM_S2_STAMP
            if (a.stamp_block & 0x10000) {
              if (v[4 * g] == 12345.678f) out_s[0] = v[4 * g + 1] + v[4 * g + 2] + v[4 * g + 3];
              continue;
            }
#endif
            if constexpr (FUSE) {
              if (fused) {  // the slab of a fused split-K launch: agent-scope stores (see gstore16_agent)
                if (okg[g]) gstore16_agent(&out_s[(pixb[g] + (r & 3)) * a.Cout + cq], f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]});
                continue;
              }
            }
            if (okg[g]) *reinterpret_cast<float4*>(&out_s[(pixb[g] + (r & 3)) * a.Cout + cq]) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
          }
        }
        S2_STAMP(22 + 4 * (i * NT + c));  // stores issued
#ifdef DRM_S2_STAMP
        const bool abl_noatom = a.stamp_block & 0x80000;  // timing experiment: no LDS statistics atomics
#else
        constexpr bool abl_noatom = false;
#endif
        if (st && first && !abl_noatom) {
          const int row0 = (wm * MT + i) * 32;
          if (PPI >= 32) {
            s0 += s1;
            q0 += q1;
          }
          double* d = lst + ((size_t)(row0 / PPI) * C::BN + col) * 2;
          atomicAdd(d, (double)s0);
          atomicAdd(d + 1, (double)q0);
          if (PPI < 32) {
            double* d2 = lst + ((size_t)((row0 + 16) / PPI) * C::BN + col) * 2;
            atomicAdd(d2, (double)s1);
            atomicAdd(d2 + 1, (double)q1);
          }
        }
        S2_STAMP(23 + 4 * (i * NT + c));  // statistics atomics issued
      }
      if constexpr (FUSE) {
        // Fused split-K finish: hand this split's slab over (every storing wave drains its agent-scope slab stores, the workgroup meets, ONE lane
        // draws the tile's ticket with an agent-scope atomic); the workgroup that arrives LAST reads the ks slabs of its tile with agent-scope
        // loads, sums them IN SLAB ORDER (its own included: the result does not depend on who was last) and runs the full epilogue -- bias, emb,
        // residual, output statistics -- into the real output.  No second launch, no per-block statistics table.
        // [r4] The slabs used to be ordinary stores / loads around an agent-scope release fence (buffer_wbl2: the XCD's whole L2 written back) and
        // acquire fences (buffer_inv: its L2 invalidated, every wave): with 512 workgroups doing so at once the hand-off measured 40k cycles
        // (19 us) of a 48 us batch-1 launch at the 64x64 level -- drain 7k, release + ticket 10k, acquire 12k.  Data that only ever moves through
        // sc1 accesses needs no cache maintenance, only the ordering: stores complete (vmcnt(0)) -> barrier -> ticket -> barrier -> loads.
        if (fused) {
          __shared__ int s_last_tile;
          S2_STAMP(70);  // slab stores issued
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          S2_STAMP(71);  // ... and complete at agent scope
          __builtin_amdgcn_s_barrier();
          if (tid == 0) {
#ifdef DRM_SK_FENCED
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // (the workgroup's slab stores are complete -- vmcnt(0) + barrier above -- and now written back)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (cdna_hip_programming.md G16 pitfall 12: the fence's own wait may be dropped)
#endif
            s_last_tile = __hip_atomic_fetch_add(a.tile_ticket + (x_start + k_tile), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)ks - 1;
          }
          __syncthreads();
          S2_STAMP(72);  // ticket drawn
          if (s_last_tile) {
#ifdef DRM_SK_FENCED
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (every wave that loads slab bytes: its CU's L1 lines of them are invalidated)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            S2_STAMP(73);
            const int col = wn * 32 + r, co = cur.co0 + col;
            const int cq = cur.co0 + wn * 32 + (r & ~3);
            float v[16];
            // (bias / emb / residual requested first: their latency rides under the slab loads)
            const float bias = a.bias ? a.bias[co] : 0.f;
            float rv[16], ev[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) ev[g] = a.emb ? a.emb[(size_t)nimg[g] * a.emb_stride + co] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) rv[e] = a.res ? a.res[(pixb[e >> 2] + (e & 3)) * a.Cout + co] : 0.f;
            // 16 agent-scope 16-byte loads in flight per round (they come from memory, ~5k cycles a round -- the stamp timeline): all four row
            // groups at once when ks <= 4, two at a time beyond
            auto sum_groups = [&](auto G0c, auto NGc, auto NKc) {
              constexpr int G0 = decltype(G0c)::value, NGR = decltype(NGc)::value, NK = decltype(NKc)::value;
              f32x4 t[NGR][NK];
#pragma unroll
              for (int gg = 0; gg < NGR; ++gg)
#pragma unroll
                for (int k = 0; k < NK; ++k)
                  // (always issued, on slab 0 beyond ks: an asm load under a branch leaves its destination to a phi copy that may run before it lands)
                  gload16_agent(t[gg][k], &a.split_ws[(size_t)(k < ks ? k : 0) * a.split_stride + (pixb[G0 + gg] + (r & 3)) * a.Cout + cq]);
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
              for (int gg = 0; gg < NGR; ++gg)
#pragma unroll
                for (int k = 0; k < NK; ++k) tie_reg(t[gg][k]);
#pragma unroll
              for (int gg = 0; gg < NGR; ++gg) {
                f32x4 sacc = t[gg][0];
#pragma unroll
                for (int k = 1; k < NK; ++k)
                  if (k < ks) { sacc.x += t[gg][k].x; sacc.y += t[gg][k].y; sacc.z += t[gg][k].z; sacc.w += t[gg][k].w; }
                const int g = G0 + gg;
                v[4 * g] = sacc.x; v[4 * g + 1] = sacc.y; v[4 * g + 2] = sacc.z; v[4 * g + 3] = sacc.w;
                quad_transpose(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3], r);  // back to the accumulator layout: one channel, four pixels
              }
            };
            using I0 = std::integral_constant<int, 0>;
            using I2 = std::integral_constant<int, 2>;
            using I4 = std::integral_constant<int, 4>;
            using I8 = std::integral_constant<int, 8>;
            if (ks <= 4) {
              sum_groups(I0{}, I4{}, I4{});
            } else {
              sum_groups(I0{}, I2{}, I8{});
              sum_groups(I2{}, I2{}, I8{});
            }
            S2_STAMP(74);  // slabs summed
            float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int g = e >> 2;
              const float x = v[e] + bias + ev[g] + rv[e];  // (the order of the one-launch epilogue)
              v[e] = x;
              if (okg[g]) {
                if (e < 8) { s0 += x; q0 += x * x; } else { s1 += x; q1 += x * x; }
              }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              quad_transpose(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3], r);
              if (okg[g]) *reinterpret_cast<float4*>(&a.out[(pixb[g] + (r & 3)) * a.Cout + cq]) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
            }
            if (st) {  // LDS fold like every full epilogue; the global fold below runs for this workgroup
              const int row0 = wm * 32;
              if (PPI >= 32) {
                s0 += s1;
                q0 += q1;
              }
              double* d = lst + ((size_t)(row0 / PPI) * C::BN + col) * 2;
              atomicAdd(d, (double)s0);
              atomicAdd(d + 1, (double)q0);
              if (PPI < 32) {
                double* d2 = lst + ((size_t)((row0 + 16) / PPI) * C::BN + col) * 2;
                atomicAdd(d2, (double)s1);
                atomicAdd(d2 + 1, (double)q1);
              }
            }
            fused_last = true;
            S2_STAMP(75);  // finishing epilogue issued
          }
        }
      }
    }
    // The LDS statistics keep accumulating while the workgroup's next tile covers the same images and output channels (on the big maps a
    // workgroup visits several tiles of one image in a row): the fold -- two barriers and TN * BN * 2 global fp64 atomics -- runs only
    // when that changes, not once per tile.
    if (st && (ks == 1 || fused_last) && (!has_next || nxt.n0 != cur.n0 || nxt.co0 != cur.co0)) {
      // fold of the accumulated statistics: LDS -> one global fp64 atomic per (image, channel, moment); re-zero for the next tile
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      for (int k = tid; k < C::TN * C::BN * 2; k += C::NTHR) {
        const int img = k / (C::BN * 2), rem = k % (C::BN * 2);
        const int n = cur.n0 + img;
#ifdef DRM_S2_STAMP
        if (a.stamp_block & 0x40000) {  // timing experiment: plain stores in place of the contended global atomics (wrong statistics)
          if (n < a.N) *(reinterpret_cast<double*>(a.stat_out + (size_t)n * a.Cout + cur.co0) + rem) = lst[k];
        } else
#endif
        if (n < a.N) atomicAdd(reinterpret_cast<double*>(a.stat_out + (size_t)n * a.Cout + cur.co0) + rem, lst[k]);
        lst[k] = 0.0;
        if constexpr (POOL) {
          atomicAdd(reinterpret_cast<double*>(a.pool_stat + (size_t)n * a.Cout + cur.co0) + rem, lst2[k]);
          lst2[k] = 0.0;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    S2_STAMP(10);  // tile end (epilogue + statistics fold done)
    if (!has_next) break;
    k_tile += J;
    cur = nxt;
  }
  wait_vmcnt<0>();  // drain the tail DMAs before the workgroup's LDS can be re-assigned
#ifdef DRM_S2_STAMP
  if (a.stamp_out && (int)blockIdx.x == (a.stamp_block & 0xFFFF)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int k = tid; k < C::NW * 240; k += C::NTHR) a.stamp_out[k] = stamp_lds[k];
  }
#endif
}

template <int TAPS, int TH, int TW, int WM, int WN, int MT, int NT, int R, int TPS, int TERMS = 3, bool RAG = false, bool SK = false, bool POOL = false>
static int launch_s2(const ConvArgs& a, hipStream_t s) {
  using C = S2Cfg<TAPS, TH, TW, WM, WN, MT, NT, R, TPS, TERMS>;
  auto kern = conv_split2_kernel<TAPS, TH, TW, WM, WN, MT, NT, R, TPS, TERMS, RAG, SK, POOL>;
  DRM_REQUIRE(RAG || (a.H % TH == 0 && a.W % TW == 0), "conv tile does not divide the map");
  DRM_REQUIRE(POOL == (a.pool_out != nullptr), "pooled output: only launches conv_split_pool_applicable accepts");
  DRM_REQUIRE(!POOL || (a.pool_stat && a.stat_out && a.ksplit <= 1 && !a.out_nchw), "pooled output: needs both statistics tables, no split-K");
  const size_t lds_bytes = (size_t)(C::LDS_F4 + (POOL ? C::ST_F4 : 0)) * sizeof(float4);
  static_assert((C::LDS_F4 + (POOL ? C::ST_F4 : 0)) * 16 <= 160 * 1024, "LDS budget");
  const DeviceInfo* di = device_info();  // fails loudly on anything that is not an MI355X-shaped gfx950 (256 CUs, 160 KiB LDS)
  if (!di) return DRM_ERR_STATE;
  // the opt-in LDS size is a per-device function attribute: one bit per device ordinal, set idempotently (a racing second
  // thread at worst repeats the call), so the entry points stay re-entrant across streams, threads and devices
  static std::atomic<uint64_t> attr_mask{0};
  if (lds_bytes > 48 * 1024 && !(attr_mask.load(std::memory_order_acquire) >> di->ordinal & 1)) {
    DRM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_mask.fetch_or(uint64_t(1) << di->ordinal, std::memory_order_release);
  }
  const int groups = (a.N + C::TN - 1) / C::TN;
  const long long tiles = (long long)groups * ((a.H + TH - 1) / TH) * ((a.W + TW - 1) / TW) * (a.Cout / C::BN);
  DRM_REQUIRE(tiles > 0 && tiles < (1ll << 31), "conv grid size");
  // persistent grid: as many workgroups as stay resident (256 CUs x workgroups per CU by LDS), a multiple of 8 (XCDs)
  const int per_cu = std::max(1, std::min((int)(di->lds_per_cu / lds_bytes), 8 / C::NW));
  long long grid = (long long)di->cus * per_cu;
  const int ngt = ((a.C0 + a.C1) / C::KC) * C::NG;  // weight groups per tile
  const int ks = a.ksplit > 1 ? a.ksplit : 1;
  constexpr bool FUSE = SK;
  DRM_REQUIRE(ks == 1 || (!a.out_nchw && a.split_stride > 0 && a.w_img_stride_f4 == 0 && ((a.C0 + a.C1) / C::KC) / ks >= 1 && ks <= 8 &&
                          (a.split_ws ? (FUSE && a.tile_ticket != nullptr) : !a.stat_out)),
              "split-K launch contract");
  if (tiles < grid || ngt < R - 1 || (TAPS == 1 && ngt < 2) || ks > 1) grid = tiles;  // (a prefetch may only reach into the NEXT tile)
  {
    const double cin = a.cin_real > 0 ? a.cin_real : (a.C0 + a.C1), cout = a.out_nchw ? a.cout_valid : a.Cout;
    const double px = (double)a.N * a.H * a.W;
    const double px_in = (double)a.N * ((a.H >> a.up0) * (a.W >> a.up0)) * a.C0 + px * a.C1;
    prof_tag(a.N, a.H, a.W, a.C0 + a.C1, a.Cout);
    if (prof_enabled()) {  // the instantiation's name as rocprofv3 prints it: per-variant totals next to the per-family ones
      static const std::string vname = [] {
        char b[192];
        snprintf(b, sizeof b, "void drm::conv_split2_kernel<%d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %s, %s, %s>", TAPS, TH, TW, WM, WN, MT, NT, R, TPS, TERMS,
                 RAG ? "true" : "false", SK ? "true" : "false", POOL ? "true" : "false");
        return std::string(b);
      }();
      prof_variant(vname.c_str());
    }
    DRM_REQUIRE(a.w_img_stride_f4 == 0 || (C::TN == 1 && ks == 1), "per-image weights need one image per tile (H*W a multiple of the 256-pixel tile)");
    ProfScope ps(a.prof_kind == PROF_KINDS ? -1 : (a.prof_kind >= 0 ? a.prof_kind : (TAPS == 9 ? PROF_CONV3 : PROF_CONV1)), 2.0 * px * TAPS * cin * cout,
                 4.0 * (px_in + px * cout * (a.res ? 2 : 1) + (double)TAPS * cin * cout), s);
#ifdef DRM_S2_STAMP
    // DRM_S2_STAMP_FILE=<path> [DRM_S2_STAMP_BLOCK=<workgroup>] [DRM_S2_STAMP_TILE0=<first recorded tile>]: appends one record
    // per 8-wave launch that has the 7.5 KiB of LDS to spare
    static const char* stamp_file = getenv("DRM_S2_STAMP_FILE");
    if (stamp_file && lds_bytes + 7680 <= di->lds_per_cu) {
      static unsigned* dbuf = nullptr;
      if (!dbuf) DRM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dbuf), 8192));
      DRM_HIP_CHECK(hipMemsetAsync(dbuf, 0, 8192, s));
      ConvArgs at = a;
      at.stamp_out = dbuf;
      at.stamp_block = getenv("DRM_S2_STAMP_BLOCK") ? atoi(getenv("DRM_S2_STAMP_BLOCK")) : 8;
      at.stamp_tile0 = getenv("DRM_S2_STAMP_TILE0") ? atoi(getenv("DRM_S2_STAMP_TILE0")) : 2;
      DRM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes + 7680));
      hipLaunchKernelGGL(kern, dim3((unsigned)grid, (unsigned)ks), dim3(C::NTHR), lds_bytes + 7680, s, at);
      DRM_HIP_CHECK(hipStreamSynchronize(s));
      std::vector<unsigned> h(2048);
      DRM_HIP_CHECK(hipMemcpy(h.data(), dbuf, 8192, hipMemcpyDeviceToHost));
      if (FILE* f = fopen(stamp_file, "a")) {
        fprintf(f, "launch taps %d tile %dx%d C %d->%d map %dx%d N %d grid %lld\n", TAPS, TH, TW, a.C0 + a.C1, a.Cout, a.H, a.W, a.N, grid);
        for (int w = 0; w < 8; ++w) {
          fprintf(f, "wave %d:", w);
          for (int k = 0; k < 120 && h[(w * 120 + k) * 2]; ++k) fprintf(f, " %u:%u", h[(w * 120 + k) * 2], h[(w * 120 + k) * 2 + 1]);
          fprintf(f, "\n");
        }
        fclose(f);
      }
    } else if (static const char* fl = getenv("DRM_S2_FLAGS"); fl) {  // timing experiments without a timeline: 65536 no output stores, 131072 no epilogue, 262144 statistics fold by plain stores
      ConvArgs at = a;
      at.stamp_block = atoi(fl) & ~0xFFFF;
      hipLaunchKernelGGL(kern, dim3((unsigned)grid, (unsigned)ks), dim3(C::NTHR), lds_bytes, s, at);
    } else
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, (unsigned)ks), dim3(C::NTHR), lds_bytes, s, a);
  }
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

constexpr long long S2_MIN_WIDE_TILES = 176;  // 256 x 128 tiles are used from this many workgroups on (of 256 CUs)

// Deep levels (maps of at most 128 pixels: the 8x16 and 4x8 levels of a batch-32 step): too few GEMM rows for the 128-channel-wide tiles to fill
// the chip, and the narrow tiles that do fill it re-stage (GroupNorm + SiLU + split) every activation tile once per 32 or 64 output channels.
// Split-K over the WIDE tiles instead: every split writes its slab, splitk_reduce_small_kernel finishes (two-launch form).  Returns the
// split factor (1 = not this form).
inline int conv_split_wide_ksplit(const ConvArgs& a) {
#ifdef DRM_NO_WIDE_SPLIT
  return 1;
#else
  // (1x1 convs of these levels on the same form measured neutral: 5.86 -> 5.54 ms of 1x1 time per step, given back by the reduction launches)
  if (a.taps != 9) return 1;
  if (a.out_nchw || a.w_img_stride_f4 != 0 || a.Cout % 128 != 0) return 1;
  const int hw = a.H * a.W;
  if (hw > 128 || !((a.H % 8 == 0 && a.W % 16 == 0) || (a.H % 4 == 0 && a.W % 8 == 0))) return 1;
  const long long rows = (long long)a.N * hw;
  if (rows < 1024) return 1;  // (sparse launches keep the narrow tiles: a batch-1 step would fill half a wide tile)
  const int bm = (a.taps == 1 || (a.H % 8 == 0 && a.W % 16 == 0)) ? 256 : 128;  // (3x3 on 4x8 maps: 128-pixel x 128-channel tiles on 4 waves, dispatch_s2_bn)
  const long long tiles = ((rows + bm - 1) / bm) * (a.Cout / 128);
  if (tiles >= S2_MIN_WIDE_TILES) return 1;
  const int nch = (a.C0 + a.C1) / 32;
  const long long ks = std::min<long long>(std::min<long long>(8, nch / 2), std::max<long long>(1, 256 / tiles));
  return (int)std::max<long long>(ks, 1);
#endif
}

// 256-pixel x {128, 64}-channel tiles on 8 waves, 128-pixel x {64, 32}-channel tiles on 4 waves
template <int TAPS, int TH, int TW, int TH4, int TW4, int TERMS>
static int dispatch_s2_bn(const ConvArgs& a, hipStream_t s) {
  // Deep U-Net levels (4x8 .. 8x16 maps) have few GEMM rows: with 256x128 tiles they launch far fewer workgroups than
  // the chip has CUs.  Shrink the tile (128 rows, then 64 / 32 channels) until the grid covers the 256 CUs.
  const long long rows = (long long)a.N * a.H * a.W;
  auto wgs = [&](int bm, int bn) { return ((rows + bm - 1) / bm) * (a.Cout / bn); };
  // 3x3: one barrier per kernel ROW (3 taps, 48 KB of weights per step, double-buffered); 1x1: one tap per step, ring of 4.
  // 256-row tiles on 4x8 / 4x4 maps keep the one-tap ring (their multi-image halo tiles leave no room for 96 KB of weights).
  constexpr int TPS = (TAPS == 9) ? 3 : 1;
  constexpr int RG = (TAPS == 9) ? 2 : 4;
  constexpr bool big_ok = (TAPS == 1) || (TH >= 8);
  if constexpr (TAPS == 1 && TH == 16 && TW == 16 && TERMS == 3) {
    // 1x1 on big maps with Cout a multiple of 256: 256 output channels per tile (64 x 128 per wave, weight ring of 2).  The 1x1 form pays
    // its activation staging and its barrier once per 24 MFMAs of a wave; here per 48 (the 64x128-map skip convs: 11-15 % faster).  The
    // 1x1 kernel has the registers for it (176 -> 256 VGPRs, 2 spilled); the 3x3 kernel does not.
    if (a.Cout % 256 == 0 && wgs(256, 256) >= 512) return launch_s2<TAPS, TH, TW, 4, 2, 2, 4, 2, 1, TERMS>(a, s);
    // ... and 192 channels per tile (64 x 96 per wave, ring of 3) for Cout = 384 / 1152 / 1536 ...: qkv 512->1536 @16x32 and the 32x64-map skip
    // convs 12-16 % faster
    if (a.Cout % 192 == 0 && wgs(256, 192) >= 512) return launch_s2<TAPS, TH, TW, 4, 2, 2, 3, 3, 1, TERMS>(a, s);
  }
  // (one round of 128-wide tiles on >= 176 of the 256 CUs beats two rounds of the less efficient 64-wide ones: qkv 640->1920 @8x16 27 %,
  //  512->1536 @8x16 24 %, 3x3 384->384 @16x32 12 % faster than with the old "fill every CU" rule)
  if constexpr (TAPS == 9 && TH == 16 && TW == 16 && (TERMS == 3 || TERMS == 2)) {
    // 3x3 with 192 output channels per tile (64 x 96 per wave; one-tap weight ring of 3: the three-tap groups would not fit next to the
    // halo tile) for Cout = 384 on big maps: 12 fragment reads per 18 MFMA products instead of 8 per 12 -- 5 % faster there.  Fits since
    // the scalar-base DMA addressing took the kernel from 256 to 207 VGPRs.
    if (a.Cout % 192 == 0 && wgs(256, 192) >= 512) return launch_s2<TAPS, TH, TW, 4, 2, 2, 3, 3, 1, TERMS>(a, s);
  }
  if (a.Cout % 128 == 0 && (wgs(256, 128) >= S2_MIN_WIDE_TILES || (a.ksplit > 1 && !a.split_ws && conv_split_wide_ksplit(a) > 1))) {
    if constexpr (TAPS == 9 && TH == 16 && TW == 16) {  // (conv_split_pool_applicable: exactly the launches that reach this line on 16 x 16 tiles)
      if (a.pool_out) return launch_s2<TAPS, TH, TW, 4, 2, 2, 2, RG, TPS, TERMS, false, false, true>(a, s);
    }
    if constexpr (big_ok) return launch_s2<TAPS, TH, TW, 4, 2, 2, 2, RG, TPS, TERMS>(a, s);
    else return launch_s2<TAPS, TH4, TW4, 2, 2, 2, 2, 3, 1, TERMS>(a, s);
  }
  if (a.Cout % 64 == 0 && wgs(256, 64) >= 256) {
    if constexpr (big_ok) return launch_s2<TAPS, TH, TW, 4, 2, 2, 1, RG, TPS, TERMS>(a, s);
    else return launch_s2<TAPS, TH4, TW4, 2, 2, 2, 1, RG, TPS, TERMS>(a, s);
  }
  constexpr int RS = RG;  // ([r4] a ring of 4 three-tap groups on these sparse-launch tiles: 5.31 vs 5.38 ms on the batch-1 step, later 5.06 vs 5.08 -- not adopted)
  // ([r4] 128 x 32 tiles from 512 workgroups on -- two workgroups per CU on the batch-1 step's top level instead of one: 4.74 vs 4.755 ms there,
  //  780 vs 800 steps/s at batch 32 -- not adopted: co-resident workgroups do not hide what a sparse launch waits for)
  if (a.Cout % 64 == 0 && wgs(128, 64) >= 256) return launch_s2<TAPS, TH4, TW4, 2, 2, 2, 1, RS, TPS, TERMS>(a, s);
  if (a.ksplit > 1 && a.split_ws) return launch_s2<TAPS, TH4, TW4, 4, 1, 1, 1, RS, TPS, TERMS, false, true>(a, s);  // (conv_split_ksplit: only ever here)
  return launch_s2<TAPS, TH4, TW4, 4, 1, 1, 1, RS, TPS, TERMS>(a, s);
}

// Maps that are not a whole number of tiles (any H, W the reference accepts other than the shipped sizes): 128-row tiles on 4 waves,
// 64 or 32 channels wide, in three pixel-tile families, edge tiles masked (RAG instantiations; no split-K, no per-image weights).
template <int TAPS, int TH, int TW, int TERMS>
static int dispatch_s2_ragged(const ConvArgs& a, hipStream_t s) {
  constexpr int TPS = (TAPS == 9) ? 3 : 1;
  constexpr int RG = (TAPS == 9) ? 2 : 4;
  if (a.Cout % 64 == 0) return launch_s2<TAPS, TH, TW, 2, 2, 2, 1, RG, TPS, TERMS, true>(a, s);
  return launch_s2<TAPS, TH, TW, 4, 1, 1, 1, RG, TPS, TERMS, true>(a, s);
}

static bool s2_exact(const ConvArgs& a) { return a.H % 4 == 0 && a.W % 4 == 0; }

template <int TAPS, int TERMS>
int dispatch_s2_tile(const ConvArgs& a, hipStream_t s) {
  if (a.H % 16 == 0 && a.W % 16 == 0) return dispatch_s2_bn<TAPS, 16, 16, 8, 16, TERMS>(a, s);
  if (a.H % 8 == 0 && a.W % 16 == 0) return dispatch_s2_bn<TAPS, 8, 16, 8, 16, TERMS>(a, s);
  if (a.H % 8 == 0 && a.W % 8 == 0) return dispatch_s2_bn<TAPS, 8, 8, 8, 8, TERMS>(a, s);
  if (a.H % 4 == 0 && a.W % 8 == 0) return dispatch_s2_bn<TAPS, 4, 8, 4, 8, TERMS>(a, s);
  if (a.H % 4 == 0 && a.W % 4 == 0) return dispatch_s2_bn<TAPS, 4, 4, 4, 4, TERMS>(a, s);
  DRM_REQUIRE(a.w_img_stride_f4 == 0 && a.ksplit <= 1, "per-image weights / split-K need a map that is a whole number of tiles");
  if constexpr (TERMS == 0) {
    set_error("fp32 mode: ragged maps run on conv_igemm_kernel");
    return DRM_ERR_INVALID;
  } else {
  static const int fam[3][2] = {{8, 16}, {8, 8}, {4, 4}};
  switch (conv_tile_family(a.H, a.W, fam, 3)) {
    case 0: return dispatch_s2_ragged<TAPS, 8, 16, TERMS>(a, s);
    case 1: return dispatch_s2_ragged<TAPS, 8, 8, TERMS>(a, s);
    default: return dispatch_s2_ragged<TAPS, 4, 4, TERMS>(a, s);
  }
  }
}

// Build units: this file is compiled once per (TAPS, TERMS) pair with -DDRM_S2_UNIT=<10 * TAPS + TERMS> (the kernel instantiations of that pair,
// in parallel: drmnet_amd/build.py) and once without (the host-side rest below, which only declares them).
#ifdef DRM_S2_UNIT
template int dispatch_s2_tile<DRM_S2_UNIT / 10, DRM_S2_UNIT % 10>(const ConvArgs&, hipStream_t);
#else
extern template int dispatch_s2_tile<9, 0>(const ConvArgs&, hipStream_t);
extern template int dispatch_s2_tile<1, 0>(const ConvArgs&, hipStream_t);
extern template int dispatch_s2_tile<9, 1>(const ConvArgs&, hipStream_t);
extern template int dispatch_s2_tile<1, 1>(const ConvArgs&, hipStream_t);
extern template int dispatch_s2_tile<9, 4>(const ConvArgs&, hipStream_t);
extern template int dispatch_s2_tile<1, 4>(const ConvArgs&, hipStream_t);
extern template int dispatch_s2_tile<9, 2>(const ConvArgs&, hipStream_t);
extern template int dispatch_s2_tile<9, 3>(const ConvArgs&, hipStream_t);
extern template int dispatch_s2_tile<1, 3>(const ConvArgs&, hipStream_t);

// Who finishes a split-K conv.  Maps of at least 128 pixels: the launch itself -- the workgroup that arrives last at an output tile sums the slabs in
// slab order and runs the full epilogue (SK instantiation; the hand-off costs ~10 us whatever the shape).  Smaller maps: splitk_reduce_small_kernel
// in a second launch (5.8 us on the 4x8 maps of the batch-32 step, where the fused finish measured 1 % slower on the whole step; at batch 1 the maps
// of 128 .. 1024 pixels are where the second launch cost 12 .. 22 us: 6.64 -> 6.0 ms per step with the fused finish).
// ([r4] with the agent-scope hand-off the fused finish was tried on the smaller maps too: batch 1 4.79 vs 4.755 ms, batch 32 759 vs 763 steps/s -- the second launch stays)
// mirrors dispatch_s2_tile / dispatch_s2_bn: a 3x3 launch on 16 x 16 pixel tiles that takes the 128-channel-wide 8-wave variant without split-K
bool conv_split_pool_applicable(const ConvArgs& a) {
  if (a.taps != 9 || a.out_nchw || a.w_img_stride_f4 != 0 || a.H % 16 != 0 || a.W % 16 != 0 || a.Cout % 128 != 0 || (a.C0 + a.C1) % 32 != 0 || a.C0 % 32 != 0) return false;
  const long long rows = (long long)a.N * a.H * a.W;
  auto wgs = [&](int bm, int bn) { return ((rows + bm - 1) / bm) * (a.Cout / bn); };
  if ((a.terms == 3 || a.terms == 2) && a.Cout % 192 == 0 && wgs(256, 192) >= 512) return false;  // (the 192-wide variant goes first there)
  return wgs(256, 128) >= S2_MIN_WIDE_TILES;
}

bool conv_split_fused_finish(const ConvArgs& a) { return a.H * a.W >= 128 && conv_split_wide_ksplit(a) <= 1; }

// Split-K factor for a launch (1 = none).  Mirrors dispatch_s2_bn: only the 128-row x 32-channel fallback tiles qualify, when
// their grid leaves most of the 256 CUs idle and the reduction is long.
int conv_split_ksplit(const ConvArgs& a) {
  if (a.out_nchw || !s2_exact(a) || a.w_img_stride_f4 != 0) return 1;
  if (const int wide = conv_split_wide_ksplit(a); wide > 1) return wide;
  // (1x1 convs too [r3]: on the deep, small maps a K = 768 .. 1536 reduction is 24 .. 48 serial one-chunk steps of a handful of workgroups --
  //  30 us at batch 1 whatever the map; split, they are ~5 chunks each plus the small-map reduction)
  const long long rows = (long long)a.N * a.H * a.W;
  auto wgs = [&](int bm, int bn) { return ((rows + bm - 1) / bm) * (a.Cout / bn); };
  if (a.Cout % 128 == 0 && wgs(256, 128) >= S2_MIN_WIDE_TILES) return 1;
  if (a.Cout % 64 == 0 && wgs(256, 64) >= 256) return 1;
  if (a.Cout % 64 == 0 && wgs(128, 64) >= 256) return 1;
  const long long tiles = wgs(128, 32);
  const int nch = (a.C0 + a.C1) / 32;
  // ([r4] with the cheaper hand-off: splits of >= 2 chunks up to 1024 workgroups measured 5.13 vs 5.08 ms on the batch-1 step -- not adopted)
  const long long ks = std::min<long long>(std::min<long long>(8, nch / 4), 640 / std::max<long long>(tiles, 1));
  return (int)std::max<long long>(ks, 1);
}

// Second launch of a split-K conv on the smallest maps (H * W < 128: conv_split_fused_finish says which): out = sum of the slabs in slab order (already
// un-scaled by their launch) + bias (+ emb[n] + residual), and the per-(image, channel) sums / sums of squares of the result for the next GroupNorm.
// The whole image sits in ONE block per 16 channel quads -- 16 quads x 16 pixel lanes, the lanes of a channel fold in LDS in lane order and the
// block's sums ARE the statistics: no partial table, no ticket, no fence, no atomics; 5.8 us at batch 32 on the 4x8 maps.  Larger maps finish inside
// the conv launch (conv_split2_kernel<..., SK>): there this kernel took 12 us (22 us beyond 256 pixels, in a second kernel with a ticket hand-off).
__global__ __launch_bounds__(256) void splitk_reduce_small_kernel(const float4* __restrict__ partial, size_t slab_f4, int ks, const float* __restrict__ bias,
                                                                  const float* __restrict__ emb, int emb_stride, const float* res /* may alias out */,
                                                                  float4* out, double2* __restrict__ stat, int HW, int Cout) {
  __shared__ float red[256][8];
  const int n = blockIdx.y, q4 = Cout >> 2;
  const int ql = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int q = blockIdx.x * 16 + ql;
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q < q4) {
    if (bias) b = reinterpret_cast<const float4*>(bias)[q];
    if (emb) {
      const float4 e = *reinterpret_cast<const float4*>(emb + (size_t)n * emb_stride + 4 * q);
      b.x += e.x; b.y += e.y; b.z += e.z; b.w += e.w;
    }
  }
  float s[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int p = pl; p < HW && q < q4; p += 16) {
    const size_t idx = ((size_t)n * HW + p) * q4 + q;
    float4 acc = partial[idx];
    for (int k = 1; k < ks; ++k) {
      const float4 t = partial[idx + (size_t)k * slab_f4];
      acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    float4 v = make_float4(acc.x + b.x, acc.y + b.y, acc.z + b.z, acc.w + b.w);
    if (res) {
      const float4 r = reinterpret_cast<const float4*>(res)[idx];
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    out[idx] = v;
    s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
    ss[0] += v.x * v.x; ss[1] += v.y * v.y; ss[2] += v.z * v.z; ss[3] += v.w * v.w;
  }
  if (stat) {  // (uniform per launch)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      red[threadIdx.x][k] = s[k];
      red[threadIdx.x][4 + k] = ss[k];
    }
    __syncthreads();
    if (pl == 0 && q < q4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double a = 0.0, c2 = 0.0;
        for (int j = 0; j < 16; ++j) {  // fixed order
          a += (double)red[j * 16 + ql][k];
          c2 += (double)red[j * 16 + ql][4 + k];
        }
        double2* d = stat + (size_t)n * Cout + 4 * q + k;
        *d = make_double2(d->x + a, d->y + c2);
      }
    }
  }
}

int launch_splitk_reduce(const ConvArgs& a, const float* partial, hipStream_t s) {
  const int HW = a.H * a.W;
  DRM_REQUIRE(a.ksplit > 1 && a.split_stride % 4 == 0 && a.Cout % 4 == 0 && HW <= 256, "split-K reduction arguments (maps of more than 256 pixels finish inside the conv launch)");
  hipLaunchKernelGGL(splitk_reduce_small_kernel, dim3((a.Cout / 4 + 15) / 16, a.N), dim3(256), 0, s, reinterpret_cast<const float4*>(partial),
                     a.split_stride / 4, a.ksplit, a.bias, a.emb, a.emb_stride, a.res, reinterpret_cast<float4*>(a.out), a.stat_out, HW, a.Cout);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

int launch_conv_split2(const ConvArgs& a, hipStream_t s) {
  if (a.terms == 0) {  // exact fp32 operands on the same pipeline (DRM_PREC_FP32; maps that are a whole number of tiles)
    DRM_REQUIRE(a.H % 4 == 0 && a.W % 4 == 0, "fp32 mode: maps that are not a whole number of 4x4 tiles run on conv_igemm_kernel");
    if (a.taps == 9) return dispatch_s2_tile<9, 0>(a, s);
    return dispatch_s2_tile<1, 0>(a, s);
  }
  if (a.terms == 1) {  // plain fp16 operands, one MFMA per product (DRM_PREC_F16)
    if (a.taps == 9) return dispatch_s2_tile<9, 1>(a, s);
    return dispatch_s2_tile<1, 1>(a, s);
  }
  if (a.terms == 4) {  // plain bf16 operands, one MFMA per product (DRM_PREC_BF16)
    if (a.taps == 9) return dispatch_s2_tile<9, 4>(a, s);
    return dispatch_s2_tile<1, 4>(a, s);
  }
  if (a.terms == 2) {  // fp16 hi*hi + block-scaled fp8 cross terms (DRM_PREC_F16MX): the GroupNorm-fed 3x3 convs only
    DRM_REQUIRE(a.taps == 9 && a.gn_scale && !a.in_inv && a.w_img_stride_f4 == 0, "f16mx: 3x3 convs on a GroupNorm-ed input only");
    return dispatch_s2_tile<9, 2>(a, s);
  }
  if (a.taps == 9) return dispatch_s2_tile<9, 3>(a, s);
  return dispatch_s2_tile<1, 3>(a, s);
}
#endif  // DRM_S2_UNIT

}  // namespace drm
