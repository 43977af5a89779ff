// Split-precision fused GroupNorm+SiLU+conv3x3 implicit GEMM, third generation: PRODUCER / CONSUMER waves.
//
// What the phase timers of conv_split2.hip showed (tools/build_exp.sh): a wave64 that issues a
// vector-memory instruction -- an LDS-DMA of a weight tile or an activation load -- is held at that instruction until
// the CU's address path accepts it (~300 cycles per LDS-DMA with 8 waves asking, ~5000 cycles for a 41-KB halo tile),
// and while it is held it issues no MFMAs.  With every wave doing loads, staging and MFMAs in turn, the memory-side work
// (0.47 ms of a 1.0-ms launch when run alone) and the MFMA work (0.57 ms alone) simply added up.
//
// Here the eight waves of a 512-thread workgroup (one workgroup per CU) have fixed roles:
//   waves 0-3  CONSUMERS, one per SIMD: ds_read fragments + MFMAs + the epilogue.  Each owns 64 pixels x 128 (or 64)
//              output channels of the 256-pixel tile: 2 x NT accumulator tiles of 32x32 (128 / 64 VGPRs).  Fragments are
//              double-buffered in registers: the reads of the next 16-channel slab are issued before the MFMAs of the
//              current one, so a single wave keeps its SIMD's MFMA pipe busy.
//   waves 4-7  PRODUCERS, one per SIMD: the LDS-DMA weight ring (R = 4 taps in flight), the activation halo tile of the
//              NEXT 32-channel chunk (global loads -> GroupNorm affine + SiLU + fp16 hi/lo split -> the other LDS tile
//              buffer), dealt over the nine taps of the current chunk.  They never touch the MFMA pipe; their VALU work
//              co-issues in the shadow of the consumer's MFMAs.
// One s_barrier per tap keeps the roles in step: before barrier t the producers have seen tap t+1's weights land and the
// consumers have finished reading tap t's; the consumer places that barrier BETWEEN its two slabs, so its fragment
// prefetch for tap t+1 runs under the second slab's MFMAs.
// Arithmetic, packed weight layout, epilogue (bias / emb / residual / fused GroupNorm statistics) and the persistent
// XCD-contiguous tile walk are those of conv_split2.hip; results are bit-identical to it (same products, same order).
#include <utility>

#include "common.h"
#include "profiler.h"

namespace drm {
namespace s3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int TH, int TW, int NT>
struct Cfg {
  static constexpr int KC = 32, R = 4, NW = 8, NTHR = 512, NCW = 4, NPW = 4, PTHR = NPW * 64;
  static constexpr int MT = 2;
  static constexpr int BM = NCW * MT * 32;  // 256 GEMM rows (pixels)
  static constexpr int BN = NT * 32;
  static constexpr int TN = BM / (TH * TW);  // images per tile
  static constexpr int HT = TH + 2, WT = TW + 2, HPI = HT * WT;
  static constexpr bool SUB48 = (TW % 8 == 0) && (TH % 4 == 0);
  static constexpr int B_F4 = 8 * BN;  // one tap's weight tile: [hl 2][s 2][h 2][BN] 16-byte entries
  static constexpr int ST_F4 = TN * BN;
  static constexpr int WTP_TRY = (TW == 16) ? 24 : WT;  // conflict-free row stride (see conv_split2.hip)
  static constexpr bool PAD_FITS = (2 * 8 * TN * HT * WTP_TRY + R * B_F4 + ST_F4) * 16 <= 160 * 1024;
  static constexpr int WTP = PAD_FITS ? WTP_TRY : WT;
  static constexpr int HPIP = HT * WTP, HP = TN * HPIP;
  static constexpr int A1_F4 = 8 * HP;
  static constexpr int LDS_F4 = 2 * A1_F4 + R * B_F4 + ST_F4;
  static constexpr bool FITS = LDS_F4 * 16 <= 160 * 1024 && (TH * TW * TN == BM);
  static constexpr int TPI = PTHR / TN, OCT = 4;
  static constexpr int A_SLOTS = (HPI * OCT + TPI - 1) / TPI;
  static constexpr int DMA_PER = B_F4 / 64 / NPW;  // LDS-DMA instructions per producer wave per tap
  // producer schedule inside a chunk (taps 0..8): loads of the first L0 slots + GroupNorm scale/shift at tap 0, the other
  // slots at tap 1, staging of slot j at tap 3 + j % 6
  static constexpr int L0 = (A_SLOTS + 1) / 2;
  static constexpr int N0 = 2 * L0 + 4, N1 = 2 * (A_SLOTS - L0);
  static __device__ __forceinline__ void rowmap(int row, int& img, int& py, int& px) {
    if constexpr (SUB48) {
      constexpr int SPI = (TH / 4) * (TW / 8);
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
  static_assert(B_F4 % (64 * NPW) == 0 && DMA_PER >= 1, "weight tile splits evenly over the producer waves");
  static_assert(TPI % OCT == 0 && TPI >= OCT, "loader mapping");
  static_assert(2 * DMA_PER + N0 + N1 < 64, "vmcnt is a 6-bit counter");
};

__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ void split(float v, _Float16& hi, _Float16& lo) {
  const float c = __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
  hi = (_Float16)c;
  lo = (_Float16)(c - (float)hi);
}
union F4H8 {
  float4 f4;
  f16x8 h8;
};
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void gload16x2(f32x4& d0, f32x4& d1, const void* p) {
  asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(d0), "=&v"(d1) : "v"(p) : "memory");
}
__device__ __forceinline__ void tie_regs(f32x4& x0, f32x4& x1) { asm volatile("" : "+v"(x0), "+v"(x1)::"memory"); }
template <int... I, typename F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}

struct TilePos {
  int n0, ty0, tx0, co0;
};

template <int TH, int TW, int NT>
__global__ __launch_bounds__(512, 2) void conv_split3_kernel(ConvArgs a) {
  using C = Cfg<TH, TW, NT>;
  constexpr int R = C::R, MT = C::MT;
  extern __shared__ float4 lds[];
  float4* As = lds;                                                        // 2 x [hl 2][s 2][h 2][HP]
  float4* Bs = lds + 2 * C::A1_F4;                                         // R x [hl 2][s 2][h 2][BN]
  double* lst = reinterpret_cast<double*>(lds + 2 * C::A1_F4 + R * C::B_F4);  // [TN][BN][2]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= C::NCW;
  const int r = lane & 31, h = lane >> 5;

  // ---- persistent tile walk (as conv_split2.hip)
  const int tiles_x = a.W / TW, tiles_y = a.H / TH;
  const int n_tiles = a.Cout / C::BN;
  const int total = ((a.N + C::TN - 1) / C::TN) * tiles_y * tiles_x * n_tiles;
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int q8 = total >> 3, r8 = total & 7;
  const int x_start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int x_count = q8 + (xcd < r8 ? 1 : 0);
  const int J = ((int)gridDim.x - xcd + 7) >> 3;
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
    return t;
  };
  int k_tile = jx;
  if (k_tile >= x_count) return;
  TilePos cur = decode(x_start + k_tile);

  const int Ctot = a.C0 + a.C1;
  const int nchunks = Ctot / C::KC;
  const int NGT = nchunks * 9;  // taps (weight tiles) per output tile
  const bool has_gn = a.gn_scale != nullptr;
  const bool st = a.stat_out != nullptr;
  if (st) {
    for (int k = tid; k < C::TN * C::BN * 2; k += C::NTHR) lst[k] = 0.0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // zeroed before the first barrier
  }

  if (producer) {
    // =====================================================================================================
    // PRODUCER
    // =====================================================================================================
    const int pw = wave - C::NCW;
    const int pt = tid - C::NCW * 64;  // 0 .. 255
    const int l_img = pt / C::TPI, l_tid = pt % C::TPI, l_o = pt % C::OCT;
    f32x4 areg[C::A_SLOTS][2];
    f32x4 sc[2], sh[2];
    unsigned avalid = 0;
    const unsigned lds_bs = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(Bs));

    // pieces [p0, p1) of the activation request; piece A_SLOTS is the GroupNorm scale / shift
    auto load_A = [&](const TilePos& tp, int chunk, int p0, int p1) {
      const int c = chunk * C::KC;
      const float* src;
      int Cs, coff, up;
      if (c < a.C0) {
        src = a.src0; Cs = a.C0; coff = c; up = a.up0;
      } else {
        src = a.src1; Cs = a.C1; coff = c - a.C0; up = 0;
      }
      const int Hs = up ? (a.H >> 1) : a.H, Ws = up ? (a.W >> 1) : a.W;
      const int l_n = tp.n0 + l_img;
      const int l_nc = l_n < a.N ? l_n : a.N - 1;
      if (p0 == 0) avalid = 0;
#pragma unroll
      for (int j = 0; j < C::A_SLOTS; ++j) {
        if (j < p0 || j >= p1) continue;
        const int lidx = l_tid + C::TPI * j;
        const int hpl = lidx / C::OCT;
        const int hy = hpl / C::WT, hx = hpl % C::WT;
        const int y = tp.ty0 + hy - 1, x = tp.tx0 + hx - 1;
        const bool ok = (lidx < C::HPI * C::OCT) && (l_n < a.N) && (y >= 0) && (y < a.H) && (x >= 0) && (x < a.W);
        const int yc = min(max(y, 0), a.H - 1), xc = min(max(x, 0), a.W - 1);
        const int ys = up ? (yc >> 1) : yc, xs = up ? (xc >> 1) : xc;
        const size_t pix = ((size_t)l_nc * Hs + ys) * Ws + xs;
        gload16x2(areg[j][0], areg[j][1], src + pix * Cs + coff + 8 * l_o);
        if (ok) avalid |= 1u << j;
      }
      if (p1 > C::A_SLOTS) {  // always issued (a harmless re-read of the input without GroupNorm): static vmcnt bookkeeping
        const float* gs = has_gn ? a.gn_scale + (size_t)l_nc * Ctot + c + 8 * l_o : a.src0;
        const float* gb = has_gn ? a.gn_shift + (size_t)l_nc * Ctot + c + 8 * l_o : a.src0;
        gload16x2(sc[0], sc[1], gs);
        gload16x2(sh[0], sh[1], gb);
      }
    };
    auto tie_A = [&]() {
#pragma unroll
      for (int j = 0; j < C::A_SLOTS; ++j) tie_regs(areg[j][0], areg[j][1]);
      tie_regs(sc[0], sc[1]);
      tie_regs(sh[0], sh[1]);
    };
    auto stage_slot = [&](float4* Ad, int j) {
      const int lidx = l_tid + C::TPI * j;
      if (lidx < C::HPI * C::OCT) {
        F4H8 hi, lo;
        float v[8] = {areg[j][0].x, areg[j][0].y, areg[j][0].z, areg[j][0].w, areg[j][1].x, areg[j][1].y, areg[j][1].z, areg[j][1].w};
        if (has_gn) {
          const float s8[8] = {sc[0].x, sc[0].y, sc[0].z, sc[0].w, sc[1].x, sc[1].y, sc[1].z, sc[1].w};
          const float b8[8] = {sh[0].x, sh[0].y, sh[0].z, sh[0].w, sh[1].x, sh[1].y, sh[1].z, sh[1].w};
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = v[k] * s8[k] + b8[k];
        }
        if (a.silu) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = silu(v[k]);
        }
        const bool ok = (avalid >> j) & 1u;  // conv zero padding applies after norm + activation
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          _Float16 hh, ll;
          split(ok ? v[k] : 0.f, hh, ll);
          hi.h8[k] = hh;
          lo.h8[k] = ll;
        }
        const int hpl = lidx / C::OCT;
        const int pixel = l_img * C::HPIP + (hpl / C::WT) * C::WTP + (hpl % C::WT);
        Ad[l_o * C::HP + pixel] = hi.f4;
        Ad[(4 + l_o) * C::HP + pixel] = lo.f4;
      }
    };
    // this wave's DMA_PER instructions of weight tile `g_in_tile` (= chunk * 9 + tap) into ring slot gseq % R
    auto issue_G = [&](int gseq, int g_in_tile, int co0) {
      const int slot = gseq % R;
      const int chunk = g_in_tile / 9, tap = g_in_tile - chunk * 9;
      const float4* wp = reinterpret_cast<const float4*>(a.w) + (((size_t)tap * nchunks + chunk) * 8) * a.Cout + co0;
#pragma unroll
      for (int j = 0; j < C::DMA_PER; ++j) {
        const int base = (pw * C::DMA_PER + j) * 64;  // wave-uniform float4 index inside the tile
        const int idx = base + lane;
        const int seg = idx / C::BN, co = idx % C::BN;
        glds16(wp + (size_t)seg * a.Cout + co, lds_bs + (unsigned)(slot * C::B_F4 + base) * 16u);
      }
    };

    // ---- prologue: R-1 weight tiles in flight, first activation tile staged into buffer 0
    int gseq = 0;
    {
      int k2 = k_tile;
      TilePos tp = cur;
      int gi = 0;
#pragma unroll
      for (int G = 0; G < R - 1; ++G) {
        if (gi >= NGT) {
          gi = 0;
          if (k2 + J < x_count) {
            k2 += J;
            tp = decode(x_start + k2);
          }
        }
        issue_G(gseq++, gi++, tp.co0);
      }
    }
    load_A(cur, 0, 0, C::A_SLOTS + 1);
    wait_vmcnt<0>();
    tie_A();
#pragma unroll
    for (int j = 0; j < C::A_SLOTS; ++j) stage_slot(As, j);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // P: tile 0 / chunk 0 staged, weight tile 0 landed

    int abuf = 0;
    while (true) {
      const bool has_next = k_tile + J < x_count;
      const TilePos nxt = has_next ? decode(x_start + k_tile + J) : cur;
      for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        // the activation tile staged during this chunk: next chunk, else chunk 0 of the next tile, else (last chunk of the
        // last tile) a harmless repeat of this one -- the work is unconditional so the vmcnt counts are static
        const TilePos& atp = more ? cur : nxt;
        const int achunk = more ? chunk + 1 : (has_next ? 0 : chunk);
        float4* Anext = As + (abuf ^ 1) * C::A1_F4;
        static_for(std::make_integer_sequence<int, 9>{}, [&](auto gc) {
          constexpr int g = decltype(gc)::value;
          {  // weight tile R-1 taps ahead: inside this tile, else the matching one of the next tile (else a harmless re-read)
            int gi = chunk * 9 + g + (R - 1);
            int co0 = cur.co0;
            if (gi >= NGT) {
              gi -= NGT;
              if (gi >= NGT) gi %= NGT;
              co0 = nxt.co0;
            }
            issue_G(gseq++, gi, co0);
          }
          if constexpr (g == 0) load_A(atp, achunk, 0, C::L0);
          if constexpr (g == 0) load_A(atp, achunk, C::A_SLOTS, C::A_SLOTS + 1);
          if constexpr (g == 1) load_A(atp, achunk, C::L0, C::A_SLOTS);
          if constexpr (g == 3) {
            wait_vmcnt<2 * C::DMA_PER>();  // everything up to the last activation load; younger: the weight tiles of taps 2, 3
            tie_A();
          }
          if constexpr (g >= 3) {
#pragma unroll
            for (int j = 0; j < C::A_SLOTS; ++j)
              if (3 + (j % 6) == g) stage_slot(Anext, j);
          }
          // weight tile of tap g+1 landed; younger: the tiles issued at taps g-1 and g, and the activation loads after tap g-2's
          constexpr int extra = (g == 0) ? C::N0 : (g == 1 || g == 2) ? C::N0 + C::N1 : (g == 3) ? C::N1 : 0;
          wait_vmcnt<2 * C::DMA_PER + extra>();
          if constexpr (g >= 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staged slots visible
          __builtin_amdgcn_s_barrier();  // B_t
        });
        abuf ^= 1;
      }
      // epilogue barriers (statistics fold): the producers take part in the fold
      if (st) {
        __builtin_amdgcn_s_barrier();
        for (int k = tid; k < C::TN * C::BN * 2; k += C::NTHR) {
          const int img = k / (C::BN * 2), rem = k % (C::BN * 2);
          const int n = cur.n0 + img;
          if (n < a.N) atomicAdd(reinterpret_cast<double*>(a.stat_out + (size_t)n * a.Cout + cur.co0) + rem, lst[k]);
          lst[k] = 0.0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if (!has_next) break;
      k_tile += J;
      cur = nxt;
    }
    wait_vmcnt<0>();  // drain the tail DMAs before the workgroup's LDS can be re-assigned
    return;
  }

  // =======================================================================================================
  // CONSUMER
  // =======================================================================================================
  const int cw = wave;
  int a_base[MT], b_base[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = (cw * MT + i) * 32 + r;
    int img, py, px;
    C::rowmap(row, img, py, px);
    a_base[i] = img * C::HPIP + py * C::WTP + px;
  }
#pragma unroll
  for (int c = 0; c < NT; ++c) b_base[c] = c * 32 + r;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][c][e] = 0.f;
  const float inv_scale = a.w_inv_scale ? *a.w_inv_scale : 1.0f;
  constexpr int PPI = TH * TW;

  // MFMA units: one 16-channel slab x 2 of the NT column tiles = 12 MFMAs (~384 cycles of the pipe); the fragments of
  // the next unit are requested before the MFMAs of the current one (two register sets of 8 x 16 B).
  constexpr int U = NT;  // units per tap: 2 slabs x NT/2 column pairs
  struct Frags {
    F4H8 ah[MT], al[MT], bh[2], bl[2];
  };
  Frags f0, f1;
  auto read_unit = [&](Frags& f, const float4* Ab, const float4* Bc, int tapoff, int u) {
    const int s2 = u / (NT / 2), hf = u % (NT / 2);
    const int seg = s2 * 2 + h;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      f.ah[i].f4 = Ab[seg * C::HP + a_base[i] + tapoff];
      f.al[i].f4 = Ab[(4 + seg) * C::HP + a_base[i] + tapoff];
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      f.bh[c].f4 = Bc[seg * C::BN + b_base[2 * hf + c]];
      f.bl[c].f4 = Bc[(4 + seg) * C::BN + b_base[2 * hf + c]];
    }
  };
  auto mma_unit = [&](const Frags& f, auto uc) {
    constexpr int hf = decltype(uc)::value % (NT / 2);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        acc[i][2 * hf + c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i].h8, f.bh[c].h8, acc[i][2 * hf + c], 0, 0, 0);
        acc[i][2 * hf + c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i].h8, f.bl[c].h8, acc[i][2 * hf + c], 0, 0, 0);
        acc[i][2 * hf + c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i].h8, f.bh[c].h8, acc[i][2 * hf + c], 0, 0, 0);
      }
  };

  __builtin_amdgcn_s_barrier();  // P
  int step = 0, abuf = 0;
  while (true) {
    const bool has_next = k_tile + J < x_count;
    const TilePos nxt = has_next ? decode(x_start + k_tile + J) : cur;
    read_unit(f0, As + abuf * C::A1_F4, Bs + (step % R) * C::B_F4, 0, 0);  // first unit of the tile (exposed once per tile)
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      const bool last_chunk = chunk + 1 == nchunks;
      const float4* Ac = As + abuf * C::A1_F4;
      const float4* An = As + (abuf ^ 1) * C::A1_F4;
      static_for(std::make_integer_sequence<int, 9>{}, [&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int tapoff = (g / 3) * C::WTP + (g % 3);
        constexpr int tapoff_n = (g == 8) ? 0 : ((g + 1) / 3) * C::WTP + ((g + 1) % 3);
        const float4* Bc = Bs + (step % R) * C::B_F4;
        static_for(std::make_integer_sequence<int, U>{}, [&](auto uc) {
          constexpr int u = decltype(uc)::value;
          Frags& fc = (u & 1) ? f1 : f0;
          Frags& fn = (u & 1) ? f0 : f1;
          if constexpr (u < U - 1) {
            read_unit(fn, Ac, Bc, tapoff, u + 1);
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this tap's weight tile and activation tile are fully read
            __builtin_amdgcn_s_barrier();  // B_t: tap t+1's weights (and, at g == 8, the next activation tile) are visible
            ++step;
            if (!(g == 8 && last_chunk)) read_unit(fn, g == 8 ? An : Ac, Bs + (step % R) * C::B_F4, tapoff_n, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          mma_unit(fc, uc);
          __builtin_amdgcn_sched_barrier(0);
        });
      });
      abuf ^= 1;
    }

    // ---- epilogue (rows of a 32x32 accumulator tile held by this lane: 8g + 4h + k; see conv_split2.hip)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      size_t pixb[4];
      int nimg[4];
      bool okg[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = (cw * MT + i) * 32 + 8 * g + 4 * h;
        int img, py, px;
        C::rowmap(row, img, py, px);
        const int n = cur.n0 + img;
        okg[g] = n < a.N;
        nimg[g] = okg[g] ? n : a.N - 1;
        pixb[g] = ((size_t)nimg[g] * a.H + (cur.ty0 + py)) * a.W + (cur.tx0 + px);
      }
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        const int col = c * 32 + r;
        const int co = cur.co0 + col;
        const float bias = a.bias ? a.bias[co] : 0.f;
        float rv[16], ev[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) ev[g] = a.emb ? a.emb[(size_t)nimg[g] * a.emb_stride + co] : 0.f;
        if (a.res) {
#pragma unroll
          for (int e = 0; e < 16; ++e) rv[e] = a.res[(pixb[e >> 2] + (e & 3)) * a.Cout + co];
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) rv[e] = 0.f;
        }
        float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int g = e >> 2;
          const float v = acc[i][c][e] * inv_scale + bias + ev[g] + rv[e];
          acc[i][c][e] = 0.f;
          if (okg[g]) {
            if (a.out_nchw) {
              if (co < a.cout_valid) {
                const size_t pix = pixb[g] + (e & 3);
                const size_t hw = (size_t)a.H * a.W;
                a.out[((size_t)nimg[g] * a.cout_valid + co) * hw + (pix - (size_t)nimg[g] * hw)] = v;
              }
            } else {
              a.out[(pixb[g] + (e & 3)) * a.Cout + co] = v;
            }
            if (e < 8) {
              s0 += v;
              q0 += v * v;
            } else {
              s1 += v;
              q1 += v * v;
            }
          }
        }
        if (st) {
          const int row0 = (cw * MT + i) * 32;
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
      }
    }
    if (st) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      for (int k = tid; k < C::TN * C::BN * 2; k += C::NTHR) {
        const int img = k / (C::BN * 2), rem = k % (C::BN * 2);
        const int n = cur.n0 + img;
        if (n < a.N) atomicAdd(reinterpret_cast<double*>(a.stat_out + (size_t)n * a.Cout + cur.co0) + rem, lst[k]);
        lst[k] = 0.0;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (!has_next) break;
    k_tile += J;
    cur = nxt;
  }
}

template <int TH, int TW, int NT>
static int launch(const ConvArgs& a, hipStream_t s) {
  using C = Cfg<TH, TW, NT>;
  auto kern = conv_split3_kernel<TH, TW, NT>;
  const size_t lds_bytes = (size_t)C::LDS_F4 * sizeof(float4);
  static_assert(C::FITS, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    DRM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_set = true;
  }
  const int groups = (a.N + C::TN - 1) / C::TN;
  const long long tiles = (long long)groups * (a.H / TH) * (a.W / TW) * (a.Cout / C::BN);
  DRM_REQUIRE(tiles > 0 && tiles < (1ll << 31), "conv grid size");
  long long grid = 256;  // persistent: one workgroup per CU
  if (tiles < grid) grid = tiles;
  {
    const double cin = a.cin_real > 0 ? a.cin_real : (a.C0 + a.C1), cout = a.out_nchw ? a.cout_valid : a.Cout;
    const double px = (double)a.N * a.H * a.W;
    const double px_in = (double)a.N * ((a.H >> a.up0) * (a.W >> a.up0)) * a.C0 + px * a.C1;
    prof_tag(a.N, a.H, a.W, a.C0 + a.C1, a.Cout);
    ProfScope ps(PROF_CONV3, 2.0 * px * 9 * cin * cout, 4.0 * (px_in + px * cout * (a.res ? 2 : 1) + 9.0 * cin * cout), s);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::NTHR), lds_bytes, s, a);
  }
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

template <int TH, int TW>
static int dispatch_nt(const ConvArgs& a, hipStream_t s, bool& handled) {
  const long long rows = (long long)a.N * a.H * a.W;
  auto wgs = [&](int bn) { return ((rows + 255) / 256) * (a.Cout / bn); };
  handled = true;
  if constexpr (Cfg<TH, TW, 4>::FITS) {
    if (a.Cout % 128 == 0 && wgs(128) >= 256) return launch<TH, TW, 4>(a, s);
  }
  if constexpr (Cfg<TH, TW, 2>::FITS) {
    if (a.Cout % 64 == 0 && wgs(64) >= 256) return launch<TH, TW, 2>(a, s);
  }
  handled = false;
  return DRM_OK;
}

}  // namespace s3

// 3x3 convolutions whose grid fills the chip with 256-pixel tiles; `handled` = false leaves the launch to conv_split2.hip
int launch_conv_split3(const ConvArgs& a, hipStream_t s, bool& handled) {
  handled = false;
  if (a.taps != 9) return DRM_OK;
  if (a.H % 16 == 0 && a.W % 16 == 0) return s3::dispatch_nt<16, 16>(a, s, handled);
  if (a.H % 8 == 0 && a.W % 16 == 0) return s3::dispatch_nt<8, 16>(a, s, handled);
  if (a.H % 8 == 0 && a.W % 8 == 0) return s3::dispatch_nt<8, 8>(a, s, handled);
  return DRM_OK;
}

}  // namespace drm
