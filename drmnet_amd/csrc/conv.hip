// Fused GroupNorm-apply + SiLU + 3x3/1x1 convolution as an implicit GEMM on the gfx950 fp32 matrix cores.
//
// Replaces, per call, the reference's  GroupNorm32 -> SiLU -> conv_nd  chains of ResBlock.in_layers /
// out_layers (openaimodel.py:201-205,225-232,263,274), the stem/head convs (:534,:703-707), the 1x1
// skip_connection (:241) and the k=1 Conv1d qkv / proj_out of AttentionBlock (:306,:314), including the
// th.cat of U-Net skips (:762) and the nearest x2 Upsample (:116), neither of which is ever materialised:
// the A-tile loader reads two NHWC sources and an (y>>1, x>>1) address.
//
//   C[M = N*H*W pixels][Cout] = sum_{tap, ci} act(A[pixel + tap][ci]) * Wp[tap][ci][Cout]
//
// Tiling (CDNA4): one 256-thread workgroup (4 wave64) owns a 128-pixel x BN-channel output tile; the
// pixel tile is TH x TW pixels of TN = 128/(TH*TW) images so that 4x4 / 4x8 / 8x8 feature maps still fill
// 128 GEMM rows.  Per 32-channel K-chunk the (TH+2)x(TW+2) halo tile is loaded ONCE from HBM/L2, gets the
// GroupNorm affine + SiLU applied ONCE in registers and is parked in LDS as [kgroup][pixel][4]; the nine
// taps then re-read it at shifted pixel offsets with conflict-free ds_read_b128.  Weights stream through a
// double-buffered [kgroup][cout][4] LDS tile, one tap at a time, prefetched into registers under the MFMAs.
// Math: v_mfma_f32_32x32x2_f32 (exact fp32, 256 FLOP/clk/CU); each wave holds MT x NT 32x32 accumulators.
// K order inside a chunk is permuted (lane half h takes k-groups 2j+h) so one b128 read feeds 4 MFMAs.
#include <atomic>

#include "common.h"
#include "profiler.h"

namespace drm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int TAPS, int TH, int TW, int WM, int WN, int MT, int NT, int KC>
struct Cfg {
  static constexpr int BM = WM * MT * 32;
  static constexpr int BN = WN * NT * 32;
  static constexpr int TN = BM / (TH * TW);
  static constexpr int HALO = (TAPS == 9) ? 1 : 0;
  static constexpr int HT = TH + 2 * HALO, WT = TW + 2 * HALO;
  static constexpr int HPI = HT * WT;  // (halo) pixels per image in the A tile
  static constexpr int HP = TN * HPI;
  static constexpr int KG = KC / 4;  // float4 k-groups per chunk
  static constexpr int TPI = 256 / TN;  // loader threads per image
  static constexpr int A_SLOTS = (HPI * KG + TPI - 1) / TPI;
  static constexpr int B_F4 = KG * BN;
  static constexpr int B_SLOTS = (B_F4 + 255) / 256;
  static constexpr int LDS_F4 = KG * HP + 2 * B_F4;
  static_assert(BM == 128 && WM * WN == 4, "4 waves, 128 GEMM rows");
  static_assert(KG % 2 == 0 && TPI % KG == 0, "k-group split across lane halves");
};

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }

template <int TAPS, int TH, int TW, int WM, int WN, int MT, int NT, int KC>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs a) {
  using C = Cfg<TAPS, TH, TW, WM, WN, MT, NT, KC>;
  extern __shared__ float4 lds[];
  float4* As = lds;                  // [KG][HP]
  float4* Bs = lds + C::KG * C::HP;  // 2 x [KG][BN]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  // ---- workgroup -> tile (XCD-aware: consecutive logical ids share an XCD's L2, and consecutive logical
  //      ids are the Cout tiles of one pixel tile, then its spatial neighbours)
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;  // edge tiles of a map that is not a multiple of the tile are masked
  const int n_tiles = a.Cout / C::BN;
  int logical;
  {
    const int id = blockIdx.x, nwg = gridDim.x;
    const int q = nwg >> 3, rr = nwg & 7, xcd = id & 7;
    logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (id >> 3);
  }
  const int n_tile = logical % n_tiles;
  int m_tile = logical / n_tiles;
  const int tx = m_tile % tiles_x;
  m_tile /= tiles_x;
  const int ty = m_tile % tiles_y;
  const int n0 = (m_tile / tiles_y) * C::TN;
  const int ty0 = ty * TH, tx0 = tx * TW, co0 = n_tile * C::BN;

  const int Ctot = a.C0 + a.C1;
  const int nchunks = Ctot / KC;
  const int cin4 = Ctot / 4;

  // ---- A loader: this thread always serves image (tid / TPI) and k-group (tid % KG)
  const int l_img = tid / C::TPI;
  const int l_n = n0 + l_img;
  const int l_tid = tid % C::TPI;
  const int l_g = tid % C::KG;
  float4 areg[C::A_SLOTS];
  float4 breg[C::B_SLOTS];
  float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned avalid = 0;

  auto load_A = [&](int chunk) {
    const int c = chunk * KC;
    const float* src;
    int Cs, coff, up;
    if (c < a.C0) {
      src = a.src0; Cs = a.C0; coff = c; up = a.up0;
    } else {
      src = a.src1; Cs = a.C1; coff = c - a.C0; up = 0;
    }
    const int Hs = up ? (a.H >> 1) : a.H, Ws = up ? (a.W >> 1) : a.W;
    avalid = 0;
#pragma unroll
    for (int j = 0; j < C::A_SLOTS; ++j) {
      const int lidx = l_tid + C::TPI * j;
      const int hpl = lidx / C::KG;
      const int hy = hpl / C::WT, hx = hpl % C::WT;
      const int y = ty0 + hy - C::HALO, x = tx0 + hx - C::HALO;
      const bool ok = (lidx < C::HPI * C::KG) && (l_n < a.N) && (y >= 0) && (y < a.H) && (x >= 0) && (x < a.W);
      if (ok) {
        const int ys = up ? (y >> 1) : y, xs = up ? (x >> 1) : x;
        const size_t pix = ((size_t)l_n * Hs + ys) * Ws + xs;
        areg[j] = *reinterpret_cast<const float4*>(src + pix * Cs + coff + 4 * l_g);
        avalid |= 1u << j;
      } else {
        areg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (a.gn_scale != nullptr && l_n < a.N) {
      sc4 = *reinterpret_cast<const float4*>(a.gn_scale + (size_t)l_n * Ctot + c + 4 * l_g);
      sh4 = *reinterpret_cast<const float4*>(a.gn_shift + (size_t)l_n * Ctot + c + 4 * l_g);
    }
  };
  auto store_A = [&]() {
#pragma unroll
    for (int j = 0; j < C::A_SLOTS; ++j) {
      const int lidx = l_tid + C::TPI * j;
      if (lidx < C::HPI * C::KG) {
        float4 v = areg[j];
        if (avalid & (1u << j)) {  // zero padding is applied AFTER norm+activation (conv pads its input)
          if (a.gn_scale != nullptr) {
            v.x = v.x * sc4.x + sh4.x; v.y = v.y * sc4.y + sh4.y; v.z = v.z * sc4.z + sh4.z; v.w = v.w * sc4.w + sh4.w;
          }
          if (a.silu) {
            v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w);
          }
        }
        As[l_g * C::HP + l_img * C::HPI + lidx / C::KG] = v;
      }
    }
  };
  auto load_B = [&](int chunk, int tap) {
    const float4* wp = reinterpret_cast<const float4*>(a.w);
#pragma unroll
    for (int j = 0; j < C::B_SLOTS; ++j) {
      const int idx = tid + 256 * j;
      if (idx < C::B_F4) {
        const int g = idx / C::BN, co = idx % C::BN;
        breg[j] = wp[((size_t)tap * cin4 + chunk * C::KG + g) * a.Cout + co0 + co];
      }
    }
  };
  auto store_B = [&](int buf) {
#pragma unroll
    for (int j = 0; j < C::B_SLOTS; ++j) {
      const int idx = tid + 256 * j;
      if (idx < C::B_F4) Bs[buf * C::B_F4 + idx] = breg[j];
    }
  };

  // ---- MFMA fragment addressing
  int a_base[MT], b_base[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = (wm * MT + i) * 32 + r;
    const int img = row / (TH * TW), py = (row / TW) % TH, px = row % TW;
    a_base[i] = img * C::HPI + py * C::WT + px;
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

  // ---- prologue: chunk 0, tap 0
  load_A(0);
  load_B(0, 0);
  store_A();
  store_B(0);
  __syncthreads();

  int buf = 0;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
      const bool last_tap = (tap == TAPS - 1);
      const bool last = last_tap && (chunk == nchunks - 1);
      if (!last) load_B(last_tap ? chunk + 1 : chunk, last_tap ? 0 : tap + 1);
      if (last_tap && !last) load_A(chunk + 1);

      const int tapoff = (TAPS == 9) ? ((tap / 3) * C::WT + (tap % 3)) : 0;
      const float4* Bc = Bs + buf * C::B_F4;
#pragma unroll
      for (int j = 0; j < C::KG / 2; ++j) {
        const int gi = 2 * j + h;
        float4 af[MT], bf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = As[gi * C::HP + a_base[i] + tapoff];
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
      }
      if (!last) store_B(buf ^ 1);  // safe: every wave passed the barrier that ended the step which read buf^1
      if (last_tap && !last) {
        __syncthreads();  // all waves done with this chunk's A tile
        store_A();
      }
      __syncthreads();
      buf ^= 1;
    }
  }

  // ---- epilogue: bias (+ per-sample embedding) (+ residual), NHWC (or NCHW head) store.
  // C/D layout of v_mfma_f32_32x32x2_f32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma unroll
  for (int c = 0; c < NT; ++c) {
    const int co = co0 + (wn * NT + c) * 32 + r;
    const float bias = a.bias ? a.bias[co] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (wm * MT + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int img = row / (TH * TW), py = (row / TW) % TH, px = row % TW;
        const int n = n0 + img;
        const int y = ty0 + py, x = tx0 + px;
        if (n < a.N && y < a.H && x < a.W) {
          float v = acc[i][c][e] + bias;
          if (a.emb) v += a.emb[(size_t)n * a.emb_stride + co];
          const size_t pix = ((size_t)n * a.H + y) * a.W + x;
          if (a.res) v += a.res[pix * a.Cout + co];
          if (a.out_nchw) {
            if (co < a.cout_valid) a.out[(((size_t)n * a.cout_valid + co) * a.H + y) * a.W + x] = v;
          } else {
            a.out[pix * a.Cout + co] = v;
          }
        }
      }
    }
  }
}

template <int TAPS, int TH, int TW, int WM, int WN, int MT, int NT, int KC>
static int launch_variant(const ConvArgs& a, hipStream_t s) {
  using C = Cfg<TAPS, TH, TW, WM, WN, MT, NT, KC>;
  auto kern = conv_igemm_kernel<TAPS, TH, TW, WM, WN, MT, NT, KC>;
  const size_t lds_bytes = (size_t)C::LDS_F4 * sizeof(float4);
  // the opt-in LDS size is a per-device function attribute: one bit per device ordinal, set idempotently (as in conv_split2.hip)
  const DeviceInfo* di = device_info();
  if (!di) return DRM_ERR_STATE;
  static std::atomic<uint64_t> attr_mask{0};
  if (lds_bytes > 48 * 1024 && !(attr_mask.load(std::memory_order_acquire) >> di->ordinal & 1)) {
    DRM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_mask.fetch_or(uint64_t(1) << di->ordinal, std::memory_order_release);
  }
  const int groups = (a.N + C::TN - 1) / C::TN;
  const long long blocks = (long long)groups * ((a.H + TH - 1) / TH) * ((a.W + TW - 1) / TW) * (a.Cout / C::BN);
  DRM_REQUIRE(blocks > 0 && blocks < (1ll << 31), "conv grid size");
  {
    // algorithmic work: 2 FLOP per MAC on un-padded channels; bytes = input read once + output written once + weights once
    const double cin = a.cin_real > 0 ? a.cin_real : (a.C0 + a.C1), cout = a.out_nchw ? a.cout_valid : a.Cout;
    const double px = (double)a.N * a.H * a.W;
    const double px_in = (double)a.N * ((a.H >> a.up0) * (a.W >> a.up0)) * a.C0 + px * a.C1;
    prof_tag(a.N, a.H, a.W, a.C0 + a.C1, a.Cout);
    ProfScope ps(TAPS == 9 ? PROF_CONV3 : PROF_CONV1, 2.0 * px * TAPS * cin * cout,
                 4.0 * (px_in + px * cout * (a.res ? 2 : 1) + (double)TAPS * cin * cout), s);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds_bytes, s, a);
  }
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

template <int TAPS, int TH, int TW, int KC>
static int dispatch_bn(const ConvArgs& a, hipStream_t s) {
  if (a.Cout % 128 == 0) return launch_variant<TAPS, TH, TW, 2, 2, 2, 2, KC>(a, s);
  if (a.Cout % 64 == 0) return launch_variant<TAPS, TH, TW, 2, 2, 2, 1, KC>(a, s);
  return launch_variant<TAPS, TH, TW, 4, 1, 1, 1, KC>(a, s);
}

// Pixel-tile family for an H x W map: the one that wastes the fewest GEMM rows on masked edge pixels (ties: the larger tile).  Maps
// that are whole multiples of a tile (every shipped configuration) pick exactly what they always did; any other size the reference
// accepts (openaimodel.py:731-768 is fully convolutional) runs with masked edge tiles.
int conv_tile_family(int H, int W, const int (*fam)[2], int n_fam) {
  int best = 0;
  long long best_area = -1;
  for (int k = 0; k < n_fam; ++k) {
    const long long th = fam[k][0], tw = fam[k][1];
    const long long area = ((H + th - 1) / th) * th * ((W + tw - 1) / tw) * tw;
    if (best_area < 0 || area < best_area) {
      best = k;
      best_area = area;
    }
  }
  return best;
}

template <int TAPS, int KC>
static int dispatch_tile(const ConvArgs& a, hipStream_t s) {
  static const int fam[4][2] = {{8, 16}, {8, 8}, {4, 8}, {4, 4}};
  switch (conv_tile_family(a.H, a.W, fam, 4)) {
    case 0: return dispatch_bn<TAPS, 8, 16, KC>(a, s);
    case 1: return dispatch_bn<TAPS, 8, 8, KC>(a, s);
    case 2: return dispatch_bn<TAPS, 4, 8, KC>(a, s);
    default: return dispatch_bn<TAPS, 4, 4, KC>(a, s);
  }
}

int launch_conv(const ConvArgs& a, hipStream_t s) {
  const int Ctot = a.C0 + a.C1;
  DRM_REQUIRE(a.taps == 9 || a.taps == 1, "conv taps must be 9 or 1");
  DRM_REQUIRE(a.Cout % 32 == 0, "conv Cout must be padded to a multiple of 32");
  DRM_REQUIRE(Ctot % 8 == 0, "conv Cin must be padded to a multiple of 8");
  DRM_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0, "conv shape");
  DRM_REQUIRE(!a.up0 || (a.H % 2 == 0 && a.W % 2 == 0), "upsampled source needs even output size");
  DRM_REQUIRE((a.gn_scale == nullptr) == (a.gn_shift == nullptr), "gn scale/shift must come together");
  if (Ctot % 32 == 0 && a.C0 % 32 == 0) {
    if (a.taps == 9) return dispatch_tile<9, 32>(a, s);
    return dispatch_tile<1, 32>(a, s);
  }
  DRM_REQUIRE(a.C0 % 8 == 0, "conv C0 must be a multiple of 8");
  DRM_REQUIRE(a.taps == 9, "1x1 conv needs Cin % 32 == 0");
  return dispatch_tile<9, 8>(a, s);
}

// ------------------------------------------------------------------------------------------------
// weight repack: PyTorch [Cout][Cin][taps] -> [tap][CinP/4][CoutP][4], zero padded
// ------------------------------------------------------------------------------------------------
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ p, int Cout, int Cin, int taps, int CoutP, int CinP) {
  const size_t total = (size_t)taps * CinP * CoutP;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k4 = i % 4;
    size_t t = i / 4;
    const int co = t % CoutP;
    t /= CoutP;
    const int cg = t % (CinP / 4);
    const int tap = t / (CinP / 4);
    const int ci = cg * 4 + k4;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * taps + tap];
    p[i] = v;
  }
}

size_t packed_conv_weight_floats(int taps, int CoutP, int CinP) { return (size_t)taps * CoutP * CinP; }

int launch_pack_conv_weight(const float* w, float* packed, int Cout, int Cin, int taps, int CoutP, int CinP, hipStream_t s) {
  DRM_REQUIRE(CinP % 4 == 0 && CoutP >= Cout && CinP >= Cin, "pack_conv_weight padding");
  const size_t total = packed_conv_weight_floats(taps, CoutP, CinP);
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(blocks), dim3(256), 0, s, w, packed, Cout, Cin, taps, CoutP, CinP);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

}  // namespace drm
