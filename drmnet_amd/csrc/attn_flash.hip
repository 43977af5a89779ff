// Single-kernel spatial self-attention for the long-sequence level (reference: QKVAttentionLegacy.forward, openaimodel.py:365-381 with
// n_heads = 1; ObsNet's ds = 4 level: T = 32 x 64 = 2048 keys at 3x128x256, C = 384): S = alpha q k^T, softmax over keys, O = P v, without
// the [N, T, T] score matrix ever leaving the chip.  The three-launch form (attn.hip: S GEMM, row softmax, P v GEMM on the conv pipeline)
// writes the scores once, reads and rewrites them once and reads them again: 2.1 GB of HBM traffic per attention block at batch 32 against
// 0.4 GB of q / k / v / o.
//
// Structure (CDNA4): ONE wave per SIMD on the whole 512-entry register file -- 256-thread workgroup, 4 waves, each wave owns 32 queries:
//   * S^T = k q^T (keys are the M rows, queries the N columns): in the 32x32 accumulator layout a lane holds ONE query (its column) and 16
//     keys per block in registers, so the running row maximum / row sum of the online softmax are plain per-lane register arithmetic plus one
//     lane ^ 32 exchange -- no LDS, no cross-wave traffic;
//   * the probabilities go from the S^T accumulators straight into O^T = v^T P^T as its B operand (cdna_hip_programming.md 3 "An accumulator
//     tile as the next MFMA's operand"): registers 8s .. 8s+7 of a block are the fragment of k-step s, in the key order 16s + 8(j>>2) + 4h +
//     (j&3); v^T is packed in that order (pack_attn_v_perm_kernel), so P never touches LDS either;
//   * accumulators: O^T = C/32 blocks (192 registers at C = 384) + S^T = 4 blocks (64) = the 256 AGPRs; MFMA A / B operands (k, q, v^T
//     fragments from LDS, P fragments from conversions) stay in the 256 architectural VGPRs, which is all hipcc allows them;
//   * k / q (per 32-channel chunk: 128 keys + the workgroup's 128 queries, hi + lo fp16 planes: 32 KiB) and v^T (per 16-key slab: C channels,
//     hi + lo: 24 KiB at C = 384) stream global -> LDS by LDS-DMA from the pre-split images (per image one power of two, like every
//     un-normalised operand of the split modes) into a ring of four 32-KiB slots, two steps ahead, retired by counted s_waitcnt vmcnt(N) + raw
//     s_barrier (one per step; every wave issues the same number of DMA instructions per step, so the counts are compile-time constants);
//   * arithmetic: the three-product fp16 split of every other matrix kernel here (hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16, fp32
//     accumulate) -- q, k, v AND the probabilities (staged as p * FA_PSCALE = p * 2^8 <= 2^14 with the lazy-rescale head-room 2^FA_TAU = 2^6; their fp16 lo halves are normal numbers for p >~ 2^-11 and go subnormal, then to zero, below -- a relative 2^-11 of a probability that is itself below 2^-11 of the row sum) are split; TERMS = 1 (DRM_PREC_F16)
//     keeps the hi product only.  exp through v_exp_f32 on (s - m) log2(e).
// Work: per 128-key tile and workgroup 2 x 1152 MFMAs (q k^T and P v at C = 384); k, v and the workgroup's q are re-read from L2 / the
// Infinity Cache per key tile (the q / k / v images of a batch-32 pass are 300 MB).
#include <algorithm>
#include <atomic>
#include <utility>

#include "common.h"
#include "profiler.h"

namespace drm {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union F4H8 {
  float4 f4;
  f16x8 h8;
  bf16x8 b8;
};
// one 32x32x16 product on the operand type of the mode (TERMS = 4: bf16, else fp16)
template <int TERMS>
__device__ __forceinline__ f32x16 fa_mma(const F4H8& a, const F4H8& b, f32x16 c) {
  if constexpr (TERMS == 4) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b8, b.b8, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h8, b.h8, c, 0, 0, 0);
}

constexpr int FA_QT = 128;          // queries per workgroup: 4 waves x 32
constexpr int FA_KT = 128;          // keys per tile
constexpr int FA_SLOT_F4 = 2048;    // 32 KiB ring slot
constexpr int FA_SLOTS = 4;
constexpr int FA_AHEAD = 3;         // DMA runs this many steps ahead of the MFMAs (a step is confirmed one step before its first fragment read)
constexpr float FA_TAU = 6.0f;       // the softmax reference maximum is raised only when a tile exceeds it by more than 2^6 (lazy rescaling of O^T)
constexpr float FA_PSCALE = 256.0f;  // probabilities (<= 2^FA_TAU) are staged as p * 2^8 <= 2^14 in fp16; their lo halves stay normal down to p ~ 2^-22

template <int N>
__device__ __forceinline__ void fa_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One LDS-DMA instruction (64 lanes x 16 B, wave-uniform 64-bit base in SGPRs + 32-bit per-lane byte offset -> LDS bytes [lds_dst, +1 KiB)),
// issued from inline asm so that hipcc neither drains vmcnt(0) in front of the LDS reads nor re-orders it; completion is tracked by the
// counted waits (conv_split2.hip glds16s; cdna_hip_programming.md 5.7).
__device__ __forceinline__ void fa_glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}

// ... with an immediate byte offset (0 .. 4095).  Measured (tools/probes/glds_offset_probe.hip): the instruction offset moves BOTH sides -- the
// global source and the LDS destination (M0 + offset + 16 * lane) -- so pieces whose source and destination advance together share one base pair.
template <int IMM>
__device__ __forceinline__ void fa_glds16i(const void* sbase, unsigned voff, unsigned lds_dst) {
  static_assert(IMM >= 0 && IMM < 4096, "13-bit signed immediate");
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst), "n"(IMM)
               : "memory");
}

template <int... I, typename F>
__device__ __forceinline__ void fa_static_for(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}

}  // namespace

// q / k images: [C/32 chunks][8 planes: hi seg 0..3, lo seg 0..3][T rows][8 halfs] per image (pack_attn_weight_kernel<true> of attn.hip; seg =
// slab * 2 + lane half); v image: [T/32 key chunks][8 planes][C channels][8 halfs] with the keys of a half-plane in accumulator order
// (pack_attn_v_perm_kernel below).  out [N][T][C] fp32.  s_scale[n] = alpha 2^-kq 2^-kk, o_scale[n] = 2^-kv.
// grid: N * T / 128 workgroups of 256 threads.
template <int C, int TERMS>
__global__ __launch_bounds__(256, 1) void attn_flash_kernel(const float4* __restrict__ qimg, const float4* __restrict__ kimg, const float4* __restrict__ vimg,
                                                            float* __restrict__ out, const float* __restrict__ qk_inv, const float* __restrict__ k_inv,
                                                            const float* __restrict__ v_inv, int N, int T) {
  constexpr int CB = C / 32;                // channel blocks of O^T
  constexpr int NCH = C / 32;               // k / q chunk steps per key tile
  constexpr int NSL = FA_KT / 16;           // v slab steps per key tile
  constexpr int NSTEP = NCH + NSL;
  constexpr int PC = 32 / 4;                // DMA pieces per wave, chunk step: (8 k planes + 8 q planes) x 2 KiB = 32 x 1 KiB over 4 waves
  constexpr int PV = (4 * C / 64) / 4;      // ... slab step: 4 planes x C x 16 B over 4 waves
  static_assert(C % 64 == 0 && 4 * C * 16 <= FA_SLOT_F4 * 16 && CB * 16 + 64 <= 256, "C: whole 1-KiB pieces per wave, a slab fits a slot, accumulators fit the AGPRs");
  extern __shared__ float4 lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // workgroups b and b + 8 share an XCD (and its L2): the query tiles of one image are dealt to one XCD, so k / v of the image are fetched
  // from HBM once per XCD that works on it
  const int qtiles = T / FA_QT;
  const int total = N * qtiles;
  int logical = blockIdx.x;
  if ((total & 7) == 0) logical = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
  const int n = logical / qtiles, q0 = (logical % qtiles) * FA_QT;
  const size_t img_f4 = (size_t)T * C / 4;  // float4 per image in each of the three images
  const float4* qi = qimg + (size_t)n * img_f4;
  const float4* ki = kimg + (size_t)n * img_f4;
  const float4* vi = vimg + (size_t)n * img_f4;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(lds));
  const unsigned voff = (unsigned)lane * 16u;
  const int ntiles = T / FA_KT;

  // ---- DMA of a step.  Every wave issues PC (chunk step) or PV (slab step) 1-KiB pieces: the counted waits below rely on that.  The addresses are
  // wave-uniform: a wave owns four consecutive planes of k (waves 0, 1) or of q (waves 2, 3) in a chunk step, and one plane of v^T in a slab
  // step, so a step needs one or two 64-bit bases and immediate offsets.
  const size_t rowbytes = (size_t)T * 16;  // one plane row range [T][8 halfs] of the q / k images
  const char* cbase = reinterpret_cast<const char*>(wave >= 2 ? qi + q0 : ki) + (size_t)(4 * (wave & 1)) * rowbytes;
  const unsigned c_tile_step = wave >= 2 ? 0u : (unsigned)FA_KT * 16u;  // k advances by 128 rows per key tile, q stays
  // i-th piece (0 .. PC-1) of a chunk step / (0 .. PV-1) of a slab step: inside the main loop the pieces go out ONE AT A TIME between the MFMAs
  // (a wave alone on its SIMD has no partner to cover a burst: eight back-to-back 1-KiB DMAs stall its instruction stream for hundreds of cycles
  // while the CU's address path works through them -- all four waves bursting at the same point of the step)
  auto issue_chunk_piece = [&](int kt, int u, int gslot, int i) {
    const unsigned dst = lds0 + (unsigned)(gslot & (FA_SLOTS - 1)) * (FA_SLOT_F4 * 16u) + (unsigned)wave * (PC * 1024u);
    const char* b = cbase + (size_t)u * 8 * rowbytes + (size_t)kt * c_tile_step + (size_t)(i >> 1) * rowbytes;  // plane 4 (wave & 1) + i / 2
    if (i & 1) fa_glds16i<1024>(b, voff, dst + (unsigned)(i & ~1) * 1024u);  // (the offset moves the LDS side too)
    else fa_glds16i<0>(b, voff, dst + (unsigned)i * 1024u);
  };
  auto issue_chunk = [&](int kt, int u, int gslot) {
#pragma unroll
    for (int i = 0; i < PC; ++i) issue_chunk_piece(kt, u, gslot, i);
  };
  const char* vbase = reinterpret_cast<const char*>(vi) + (size_t)((wave >> 1) * 4 + (wave & 1)) * C * 16;  // plane (hi | lo) x lane half of this wave
  auto issue_slab_piece = [&](int kt, int sl, int gslot, int i) {
    static_assert(PV == 6, "C = 384: six 1-KiB pieces per plane");
    const unsigned dst = lds0 + (unsigned)(gslot & (FA_SLOTS - 1)) * (FA_SLOT_F4 * 16u) + (unsigned)wave * (C * 16u);
    const char* b = vbase + ((size_t)(kt * (FA_KT / 32) + (sl >> 1)) * 8 + 2 * (sl & 1)) * C * 16;
    switch (i) {
      case 0: fa_glds16i<0>(b, voff, dst); break;
      case 1: fa_glds16i<1024>(b, voff, dst); break;
      case 2: fa_glds16i<2048>(b, voff, dst); break;
      case 3: fa_glds16i<3072>(b, voff, dst); break;
      case 4: fa_glds16i<0>(b + 4096, voff, dst + 4096u); break;
      default: fa_glds16i<1024>(b + 4096, voff, dst + 4096u); break;
    }
  };

  // per-lane byte offsets inside a slot: k rows (plane of lane half h, row r), q rows (+ this wave's 32 queries), v^T rows (plane h, channel r)
  const unsigned lp_k = (unsigned)(h * 128 + r) * 16u, lp_q = (unsigned)(h * 128 + wave * 32 + r) * 16u, lp_v = (unsigned)(h * C + r) * 16u;
  f32x16 O[CB], S[4];
#pragma unroll
  for (int c = 0; c < CB; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) O[c][e] = 0.f;
  bool rescale = false;
  float f_resc = 1.0f;
  float m_run = -INFINITY, l_run = 0.f;  // this lane's query (column r of the wave's 32), this lane half's 64 keys per tile for l
  const float s_scale = qk_inv[n] * k_inv[n];
  const float s2 = s_scale * 1.44269504088896340736f;  // scores in units of log2: p = 2^(s2 * acc - m)

  // ---- fragment registers, loaded ONE UNIT AHEAD of their MFMAs (a unit = one 16-deep k-step: 12 MFMAs of q k^T, or one 16-key slab of P v).
  // One wave per SIMD has no partner to cover an LDS round trip, so every fragment is requested while the previous unit's MFMAs run -- across
  // the step boundary too: the DMA runs three steps ahead and a step's data is confirmed one whole step before its first read.
  constexpr int NQK = 2 * NCH;            // q k^T units per key tile
  constexpr int NUNIT = NQK + NSL;
  constexpr int VB = 4;                   // v^T fragment ring (channel blocks in flight)
  static_assert(CB % VB == 0, "the ring runs on from one slab into the next");
  F4H8 kh[2][4], kl[2][4], qh[2], ql[2], vh[VB], vl[VB];
  auto slot_ptr = [&](int gstep) {
    unsigned slot_b = (unsigned)(gstep & (FA_SLOTS - 1)) * (FA_SLOT_F4 * 16u);
    asm volatile("" : "+s"(slot_b));  // (opaque: one address register per operand kind + immediate offsets, nothing hoisted per slot)
    return reinterpret_cast<const char*>(lds) + slot_b;
  };
  auto ld = [&](const char* slb, unsigned lane_part, int imm) { return *reinterpret_cast<const float4*>(slb + lane_part + imm); };
  // q k^T unit (chunk u, k-step s) of the step at `slb` -> buffer `buf`
  auto load_qk = [&](const char* slb, int s, int buf) {
    qh[buf].f4 = ld(slb, lp_q, (1024 + 2 * s * 128) * 16);
    if (TERMS == 3) ql[buf].f4 = ld(slb, lp_q, (1024 + (4 + 2 * s) * 128) * 16);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      kh[buf][b].f4 = ld(slb, lp_k, (2 * s * 128 + b * 32) * 16);
      if (TERMS == 3) kl[buf][b].f4 = ld(slb, lp_k, ((4 + 2 * s) * 128 + b * 32) * 16);
    }
  };
  auto load_v = [&](const char* slb, int c) {
    vh[c % VB].f4 = ld(slb, lp_v, c * 32 * 16);
    if (TERMS == 3) vl[c % VB].f4 = ld(slb, lp_v, (2 * C + c * 32) * 16);
  };

  static_assert(NCH >= FA_AHEAD && FA_AHEAD == 3 && (FA_SLOTS & (FA_SLOTS - 1)) == 0 && FA_SLOTS >= FA_AHEAD + 1, "prologue: the first three steps are chunk steps; ring arithmetic");
  issue_chunk(0, 0, 0);
  issue_chunk(0, 1, 1);
  issue_chunk(0, 2, 2);
  fa_wait_vmcnt<PC>();  // steps 0 and 1 landed (step 2's pieces may still be in flight)
  __builtin_amdgcn_s_barrier();
  load_qk(slot_ptr(0), 0, 0);

  int g = 0;  // DMA step index (ring slot = g & 3)
  for (int kt = 0; kt < ntiles; ++kt) {
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) S[b][e] = 0.f;
    fa_static_for(std::make_integer_sequence<int, NUNIT>{}, [&](auto qc) {
      constexpr int q = decltype(qc)::value;
      constexpr bool is_qk = q < NQK;
      constexpr int u = is_qk ? q / 2 : NCH + (q - NQK);           // DMA step of this unit inside the key tile
      [[maybe_unused]] constexpr bool first_of_step = is_qk ? (q % 2 == 0) : true;
      constexpr bool last_of_step = is_qk ? (q % 2 == 1) : true;
      constexpr int u3 = (u + FA_AHEAD) % NSTEP;                    // the step whose DMA goes out at the top of this one
      int kt3 = kt + (u + FA_AHEAD >= NSTEP ? 1 : 0);  // (beyond the last tile: the first tile again -- a harmless re-read into a slot nobody reads any more)
      if (kt3 == ntiles) kt3 = 0;
      // i-th DMA piece of step g + 3 (this step's share: a chunk step spreads its PC / PV pieces over its two units)
      auto dma = [&](int i) {
        if constexpr (u3 < NCH) issue_chunk_piece(kt3, u3, g + FA_AHEAD, i); else issue_slab_piece(kt3, u3 - NCH, g + FA_AHEAD, i);
      };
      constexpr int NP3 = u3 < NCH ? PC : PV;  // pieces of step g + 3
      // ---- request the NEXT unit's fragments (its step is already confirmed), then run this unit's MFMAs
      constexpr int qn = (q + 1) % NUNIT;
      constexpr bool n_qk = qn < NQK;
      const char* slb_n = slot_ptr(g + (last_of_step ? 1 : 0));
      auto preload_next = [&]() {  // (a slab unit followed by another slab unit keeps its ring going instead: see below)
        if constexpr (n_qk) {
          load_qk(slb_n, qn % 2, qn % 2);
        } else if constexpr (is_qk) {
#pragma unroll
          for (int c = 0; c < VB; ++c) load_v(slb_n, c);
        }
      };
      // (the last q k^T unit of a tile requests the first slab's fragments AFTER its softmax: the rare rescale of O^T wants the VGPRs -- with the
      // fragments live across it the register allocator spilled three blocks of O^T to scratch at every key tile)
      if constexpr (q != NQK - 1) preload_next();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (is_qk) {
        // ---- S^T += k q^T over 16 channels: A = k rows (keys), B = q columns (queries)
        constexpr int buf = q % 2;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (TERMS == 3) {
            S[b] = fa_mma<TERMS>(kl[buf][b], qh[buf], S[b]);
            S[b] = fa_mma<TERMS>(kh[buf][b], ql[buf], S[b]);
          }
          S[b] = fa_mma<TERMS>(kh[buf][b], qh[buf], S[b]);
          // this unit's half of the step's DMA pieces, one behind each key block's MFMAs
          constexpr int half = q % 2;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = half * (NP3 / 2) + b * (NP3 / 2) / 4; i < half * (NP3 / 2) + (b + 1) * (NP3 / 2) / 4; ++i) dma(i);
          if (half == 1 && b == 3)
#pragma unroll
            for (int i = 2 * (NP3 / 2); i < NP3; ++i) dma(i);  // (odd piece counts)
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (q == NQK - 1) {
          // ---- online softmax of the tile, in registers: this lane's query, 64 of the tile's 128 keys per lane half.  The reference maximum
          // m_run is raised (and O^T rescaled, in the first slab unit below: 3 x 192 register moves through the VGPRs) only when the tile's maximum exceeds it by more than
          // FA_TAU (in log2 units): below that the probabilities are simply 2^(s - m_run) <= 2^FA_TAU -- the final O / l does not depend on
          // which reference was used, and the fp16 staging of the probabilities (x FA_PSCALE) has the head-room for it
          float mt = S[0][0];
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) mt = fmaxf(mt, S[b][e]);
          mt *= s2;  // (s2 > 0)
          mt = fmaxf(mt, __shfl_xor(mt, 32));
          const bool raise = mt > m_run + FA_TAU;
          rescale = __any(raise);  // (wave-uniform; lanes that do not raise rescale by 1)
          f_resc = 1.0f;
          if (rescale) {
            const float m_new = raise ? mt : m_run;
            f_resc = __builtin_amdgcn_exp2f(m_run - m_new);  // first tile: 2^(-inf) = 0
            m_run = m_new;
            l_run *= f_resc;
          }
          // The raw scores stay in the S^T accumulators: every slab unit below turns the eight it needs into probabilities (exp, row-sum
          // share, fp16 hi / lo split) between its MFMAs.  The empty asm pins them to the accumulator file here -- without it the compiler keeps
          // the 64 values it has just read for the maximum (and then the 64 probabilities) in VGPRs through the whole P v phase: 128 of the
          // 256 architectural registers, and spills around them.
#pragma unroll
          for (int b = 0; b < 4; ++b) asm volatile("" : "+a"(S[b]));
          preload_next();
        }
      } else {
        // ---- O^T += v^T_slab P^T_slab : A = v^T rows (channels), B = the probabilities of key block kb, k-step s
        constexpr int slab = q - NQK, kb = slab >> 1, s = slab & 1;
        F4H8 ph, pl;  // registers 8s .. 8s+7 of key block kb: the B fragment of this k-step (keys in accumulator order, as v^T is packed)
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float pr = __builtin_amdgcn_exp2f(S[kb][8 * s + j] * s2 - m_run);
          psum += pr;
          const float ps = pr * FA_PSCALE;
          if constexpr (TERMS == 4) {
            ph.b8[j] = (__bf16)ps;
          } else {
            const _Float16 hi = (_Float16)ps;
            ph.h8[j] = hi;
            if (TERMS == 3) pl.h8[j] = (_Float16)(ps - (float)hi);
          }
        }
        l_run += psum;
        const char* slb = slot_ptr(g);
#pragma unroll
        for (int c = 0; c < CB; ++c) {
          if constexpr (slab == 0) {
            // the (rare) rescale of O^T by 2^(m_old - m_new) rides in the tile's first slab unit, one block at a time just ahead of the block's
            // MFMAs: as one 192-register pass behind the softmax it made the register allocator spill three blocks of O^T at every key tile
            if (rescale) {
              // (the pins make the block a fresh accumulator-file value on both sides: without the first the allocator treats the tile-start value
              // of the last three blocks as a VGPR-class live range across the whole q k^T phase and parks it in scratch at every key tile)
              asm volatile("" : "+a"(O[c]));
#pragma unroll
              for (int e = 0; e < 16; ++e) O[c][e] *= f_resc;
              asm volatile("" : "+a"(O[c]));
            }
          }
          if (TERMS == 3) {
            O[c] = fa_mma<TERMS>(vl[c % VB], ph, O[c]);
            O[c] = fa_mma<TERMS>(vh[c % VB], pl, O[c]);
          }
          O[c] = fa_mma<TERMS>(vh[c % VB], ph, O[c]);
          {  // the step's DMA pieces, spread over the channel blocks
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = c * NP3 / CB; i < (c + 1) * NP3 / CB; ++i) dma(i);
            __builtin_amdgcn_sched_barrier(0);
          }
          // into the ring entry block c has just released: block c + VB of this slab, or -- the ring runs on -- block c + VB - CB of the next one
          if (c + VB < CB) {
            __builtin_amdgcn_sched_barrier(0);
            load_v(slb, c + VB);
            __builtin_amdgcn_sched_barrier(0);
          } else if constexpr (!n_qk) {
            __builtin_amdgcn_sched_barrier(0);
            load_v(slb_n, c + VB - CB);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if constexpr (last_of_step) {
        // everything up to the pieces of step g + 2 has landed for this wave; allowed in flight: the pieces of step g + 3 issued at the top of this step
        if constexpr (u3 < NCH) fa_wait_vmcnt<PC>(); else fa_wait_vmcnt<PV>();
        __builtin_amdgcn_s_barrier();
        ++g;
      }
    });
  }
  fa_wait_vmcnt<0>();  // drain the wrapped prefetches before the workgroup's LDS can be re-assigned

  // ---- epilogue: O^T / (l FA_PSCALE 2^kv) (FA_PSCALE = 2^8, the staging factor of the probabilities): this lane's query, channels c * 32 + 8 g4 + 4 h + {0..3} per register group
  const float l = l_run + __shfl_xor(l_run, 32);
  const float fo = v_inv[n] / (FA_PSCALE * l);
  float* op = out + ((size_t)n * T + q0 + wave * 32 + r) * C;
#pragma unroll
  for (int c = 0; c < CB; ++c)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      *reinterpret_cast<float4*>(op + c * 32 + 8 * g4 + 4 * h) =
          make_float4(O[c][4 * g4] * fo, O[c][4 * g4 + 1] * fo, O[c][4 * g4 + 2] * fo, O[c][4 * g4 + 3] * fo);
}

// v[n] ([T keys][ld] fp32, channel c at column c) * scale[n] -> the v^T image of attn_flash_kernel: [T/32 key chunks][8 planes: hi seg 0..3, lo seg
// 0..3][C][8 halfs], seg = k-step s * 2 + lane half h, and half j of plane (s, h) holds key 32 kc + 16 s + 8 (j >> 2) + 4 h + (j & 3): the order in
// which the S^T accumulator registers 8s .. 8s+7 of lane half h enumerate their keys.  A thread produces one hi + one lo entry (8 keys of one
// channel); adjacent lanes take adjacent channels: every load is a coalesced row segment.   grid (blocks, N), block 256.
__global__ __launch_bounds__(256) void pack_attn_v_perm_kernel(const float* __restrict__ src, long long img_stride, int ld, const float* __restrict__ scale,
                                                               float4* __restrict__ dst, int C, int T, int bf16) {
  const int n = blockIdx.y;
  const float sc = scale[n];
  const float* sp = src + (size_t)n * img_stride;
  float4* dp = dst + (size_t)n * ((size_t)C * T / 4);
  const size_t units = (size_t)C * (T / 8);  // (channel, (key chunk, seg))
  for (size_t u = blockIdx.x * (size_t)blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(u % C);
    const int oct = (int)(u / C);  // key chunk * 4 + seg
    const int kc = oct >> 2, seg = oct & 3, s = seg >> 1, hh = seg & 1;
    F4H8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int key = 32 * kc + 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3);
      if (bf16) {
        hi.b8[j] = (__bf16)(sp[(size_t)key * ld + c] * sc);
        lo.h8[j] = (_Float16)0.f;
        continue;
      }
      const float v = __builtin_amdgcn_fmed3f(sp[(size_t)key * ld + c] * sc, -65504.0f, 65504.0f);
      const _Float16 x = (_Float16)v;
      hi.h8[j] = x;
      lo.h8[j] = (_Float16)(v - (float)x);
    }
    dp[((size_t)kc * 8 + seg) * C + c] = hi.f4;
    dp[((size_t)kc * 8 + 4 + seg) * C + c] = lo.f4;
  }
}

// (T = 512 -- RefNet's 16x32 level at the metric shape, 128 workgroups at batch 32 -- measured 0.106 ms per core against 0.114 ms for the
//  conv-pipeline form: the three packing launches and the half-empty chip eat the kernel's advantage; from T = 1024 it is 0.224 vs ~0.45 ms)
#ifndef FA_MIN_T
#define FA_MIN_T 1024
#endif
bool attention_flash_applicable(int T, int C, int terms) { return (terms == 3 || terms == 1 || terms == 4) && C == 384 && T % FA_KT == 0 && T >= FA_MIN_T; }

size_t attention_flash_workspace_floats(int N, int T, int C) {
  const size_t Z = (size_t)(C > T ? C : T);
  return 3 * (size_t)N * T * C + (size_t)N * (2 * C + T + Z) + 7 * (size_t)N + 64;
}

// (attn.hip)
void launch_attn_scales(const double2* mom, int N, int C, int T, float alpha, float* q_tab, float* p_tab, float* zero_tab, float* qk_inv, float* k_scale,
                        float* k_inv, float* pv_inv, float* v_scale, float* v_inv, float* o_tab, float* q_scale, hipStream_t s);
int launch_pack_attn_rows(const float* src, long long img_stride, int ld, const float* scale, float* dst, int rows, int cin, int N, hipStream_t s, bool bf16);

// qkv [N][T][3C] (+ its fused per-channel statistics), out [N][T][C], ws: attention_flash_workspace_floats.
// proj_guard (optional): receives the (scale, shift, inverse) tables that guard proj_out's read of `out`, as launch_attention_conv does
int launch_attention_flash(const float* qkv, const double2* qkv_mom, float* out, float* ws, int N, int T, int C, int terms, hipStream_t s, ConvArgs* proj_guard) {
  DRM_REQUIRE(attention_flash_applicable(T, C, terms) && qkv_mom, "single-kernel attention: shape");
  const size_t Z = (size_t)(C > T ? C : T);
  float* wq = ws;                           // three pre-split images, T * C * 4 bytes per image each
  float* wk = wq + (size_t)N * T * C;
  float* wv = wk + (size_t)N * T * C;
  float* q_tab = wv + (size_t)N * T * C;    // [N][C]   (tables of the conv-pipeline form: only o_tab / zero_tab / v_inv are read here, by proj_out)
  float* p_tab = q_tab + (size_t)N * C;     // [N][T]
  float* zero_tab = p_tab + (size_t)N * T;  // [N][max(C, T)]
  float* o_tab = zero_tab + (size_t)N * Z;  // [N][C]
  float* vec = o_tab + (size_t)N * C;       // 7 x [N]
  float *qk_inv = vec, *k_scale = vec + N, *k_inv = vec + 2 * N, *pv_inv = vec + 3 * N, *v_scale = vec + 4 * N, *v_inv = vec + 5 * N, *q_scale = vec + 6 * N;
  const float alpha = 1.0f / sqrtf((float)C);  // (C^-1/4)^2, applied once to the dot product
  prof_tag(N, T, 1, C, C);
  ProfScope ps(PROF_ATTN, 4.0 * N * (double)T * T * C, 4.0 * N * ((double)T * 4 * C), s);  // algorithmic bytes: q, k, v in, o out
  launch_attn_scales(qkv_mom, N, C, T, alpha, q_tab, p_tab, zero_tab, qk_inv, k_scale, k_inv, pv_inv, v_scale, v_inv, o_tab, q_scale, s);
  DRM_HIP_CHECK(hipGetLastError());
  if (proj_guard) {
    proj_guard->gn_scale = o_tab;
    proj_guard->gn_shift = zero_tab;
    proj_guard->in_inv = v_inv;
  }
  const long long sq = (long long)T * 3 * C;
  DRM_TRY(launch_pack_attn_rows(qkv, sq, 3 * C, q_scale, wq, T, C, N, s, terms == 4));
  DRM_TRY(launch_pack_attn_rows(qkv + C, sq, 3 * C, k_scale, wk, T, C, N, s, terms == 4));
  const unsigned pb = (unsigned)std::min<size_t>(((size_t)T * C / 8 + 255) / 256, 4096);
  hipLaunchKernelGGL(pack_attn_v_perm_kernel, dim3(pb, N), dim3(256), 0, s, qkv + 2 * C, sq, 3 * C, v_scale, reinterpret_cast<float4*>(wv), C, T, terms == 4 ? 1 : 0);
  DRM_HIP_CHECK(hipGetLastError());
  const size_t lds_bytes = (size_t)FA_SLOTS * FA_SLOT_F4 * sizeof(float4);
  const DeviceInfo* di = device_info();
  if (!di) return DRM_ERR_STATE;
  auto go = [&](auto kern) -> int {
    static std::atomic<uint64_t> attr_mask{0};
    if (!(attr_mask.load(std::memory_order_acquire) >> di->ordinal & 1)) {
      DRM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      attr_mask.fetch_or(uint64_t(1) << di->ordinal, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(N * (T / FA_QT))), dim3(256), lds_bytes, s, reinterpret_cast<const float4*>(wq), reinterpret_cast<const float4*>(wk),
                       reinterpret_cast<const float4*>(wv), out, qk_inv, k_inv, v_inv, N, T);
    DRM_HIP_CHECK(hipGetLastError());
    return DRM_OK;
  };
  if (terms == 4) return go(attn_flash_kernel<384, 4>);
  if (terms == 1) return go(attn_flash_kernel<384, 1>);
  return go(attn_flash_kernel<384, 3>);
}

}  // namespace drm
