// Split-precision variant of the fused GroupNorm+SiLU+conv implicit GEMM (see conv.hip for the structure):
// fp32-accurate products on the gfx950 f16 matrix cores.
//
// Every fp32 operand x is split as x = hi + lo with hi = fp16(x), lo = fp16(x - hi) (22 significant bits together),
// and a product is evaluated as hi_a*hi_b + hi_a*lo_b + lo_a*hi_b with three v_mfma_f32_32x32x16_f16 accumulating
// into the same fp32 accumulator (the dropped lo*lo term is 2^-22 relative, the level of fp32 rounding itself).
// Cost per 16-deep K slab: 3 x 32 cycles instead of 8 x 64 cycles of v_mfma_f32_32x32x2_f32  =>  16/3 x the fp32
// matrix rate, at unchanged HBM traffic (activations stay fp32 in HBM; the split happens once per staged element, in
// registers, right after the GroupNorm affine + SiLU).
//   * activations: split unscaled (|x| is O(1) after GroupNorm; tiny elements lose relative precision in the fp16
//     subnormal range but their absolute error (<= 2^-25) is below the 2^-22 relative error of the O(1) elements);
//     values are clamped to +-65504 before the split so an outlier saturates instead of becoming inf.
//   * weights: pre-split at load time after an exact power-of-two scaling that moves max|w| to [8192, 16384), so the lo
//     halves stay in the fp16 normal range; the epilogue multiplies by the inverse scale (exact).
// LDS images: A = [hi|lo][slab s][lane-half h][pixel][8 halfs], B = [hi|lo][s][h][cout][8 halfs]; one ds_read_b128 per
// operand fragment (lane (r,h) of slab s needs channels 16s + 8h .. +7 of row/col r).
#include "common.h"
#include "profiler.h"

namespace drm {

int launch_conv_split2(const ConvArgs& a, hipStream_t s);
int launch_conv_split3(const ConvArgs& a, hipStream_t s, bool& handled);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int TAPS, int TH, int TW, int WM, int WN, int MT, int NT>
struct SCfg {
  static constexpr int KC = 32;
  static constexpr int BM = WM * MT * 32;
  static constexpr int BN = WN * NT * 32;
  static constexpr int TN = BM / (TH * TW);
  static constexpr int HALO = (TAPS == 9) ? 1 : 0;
  static constexpr int HT = TH + 2 * HALO, WT = TW + 2 * HALO;
  static constexpr int HPI = HT * WT;
  static constexpr int HP = TN * HPI;
  static constexpr int TPI = 256 / TN;                 // loader threads per image
  static constexpr int OCT = 4;                        // 8-channel groups per 32-channel chunk
  static constexpr int A_SLOTS = (HPI * OCT + TPI - 1) / TPI;
  static constexpr int A_F4 = 8 * HP;                  // [hl 2][s 2][h 2][HP] 16-byte entries
  static constexpr int B_F4 = 8 * BN;                  // [hl 2][s 2][h 2][BN]
  static constexpr int B_SLOTS = (B_F4 + 255) / 256;
  static constexpr int LDS_F4 = A_F4 + 2 * B_F4;
  static_assert(BM == 128 && WM * WN == 4, "4 waves, 128 GEMM rows");
  static_assert(TPI % OCT == 0, "loader mapping");
};

__device__ __forceinline__ float silu_s(float v) { return v / (1.0f + __expf(-v)); }

__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo) {
  const float c = __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
  hi = (_Float16)c;
  lo = (_Float16)(c - (float)hi);
}

union F4H8 {
  float4 f4;
  f16x8 h8;
};

template <int TAPS, int TH, int TW, int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(256, 2) void conv_igemm_split_kernel(ConvArgs a) {
  using C = SCfg<TAPS, TH, TW, WM, WN, MT, NT>;
  extern __shared__ float4 lds[];
  float4* As = lds;              // [hl][s][h][HP]
  float4* Bs = lds + C::A_F4;    // 2 x [hl][s][h][BN]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  const int tiles_x = a.W / TW, tiles_y = a.H / TH;
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
  const int nchunks = Ctot / C::KC;

  // ---- A loader: this thread always serves image (tid / TPI) and channel octet (tid % 4) of the chunk
  const int l_img = tid / C::TPI;
  const int l_n = n0 + l_img;
  const int l_tid = tid % C::TPI;
  const int l_o = tid % C::OCT;  // channels 8*l_o .. +7  ->  slab s = l_o>>1, lane-half h = l_o&1
  float4 areg[C::A_SLOTS][2];
  float4 breg[C::B_SLOTS];
  float4 sc[2], sh[2];
  unsigned avalid = 0;

  auto load_A = [&](int chunk) {
    const int c = chunk * C::KC;
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
      const int hpl = lidx / C::OCT;
      const int hy = hpl / C::WT, hx = hpl % C::WT;
      const int y = ty0 + hy - C::HALO, x = tx0 + hx - C::HALO;
      const bool ok = (lidx < C::HPI * C::OCT) && (l_n < a.N) && (y >= 0) && (y < a.H) && (x >= 0) && (x < a.W);
      if (ok) {
        const int ys = up ? (y >> 1) : y, xs = up ? (x >> 1) : x;
        const size_t pix = ((size_t)l_n * Hs + ys) * Ws + xs;
        const float4* p = reinterpret_cast<const float4*>(src + pix * Cs + coff + 8 * l_o);
        areg[j][0] = p[0];
        areg[j][1] = p[1];
        avalid |= 1u << j;
      }
    }
    if (a.gn_scale != nullptr && l_n < a.N) {
      const float4* ps = reinterpret_cast<const float4*>(a.gn_scale + (size_t)l_n * Ctot + c + 8 * l_o);
      const float4* pb = reinterpret_cast<const float4*>(a.gn_shift + (size_t)l_n * Ctot + c + 8 * l_o);
      sc[0] = ps[0]; sc[1] = ps[1]; sh[0] = pb[0]; sh[1] = pb[1];
    }
  };
  auto store_A = [&]() {
#pragma unroll
    for (int j = 0; j < C::A_SLOTS; ++j) {
      const int lidx = l_tid + C::TPI * j;
      if (lidx < C::HPI * C::OCT) {
        F4H8 hi, lo;
        if (avalid & (1u << j)) {
          float v[8] = {areg[j][0].x, areg[j][0].y, areg[j][0].z, areg[j][0].w, areg[j][1].x, areg[j][1].y, areg[j][1].z, areg[j][1].w};
          if (a.gn_scale != nullptr) {
            const float s8[8] = {sc[0].x, sc[0].y, sc[0].z, sc[0].w, sc[1].x, sc[1].y, sc[1].z, sc[1].w};
            const float b8[8] = {sh[0].x, sh[0].y, sh[0].z, sh[0].w, sh[1].x, sh[1].y, sh[1].z, sh[1].w};
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = v[k] * s8[k] + b8[k];
          }
          if (a.silu) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = silu_s(v[k]);
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            _Float16 hh, ll;
            split_f16(v[k], hh, ll);
            hi.h8[k] = hh;
            lo.h8[k] = ll;
          }
        } else {  // conv zero padding (applied after norm + activation)
          hi.f4 = make_float4(0.f, 0.f, 0.f, 0.f);
          lo.f4 = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int pixel = l_img * C::HPI + lidx / C::OCT;
        As[l_o * C::HP + pixel] = hi.f4;             // [hl=0][s][h] == l_o
        As[(4 + l_o) * C::HP + pixel] = lo.f4;       // [hl=1]
      }
    }
  };
  auto load_B = [&](int chunk, int tap) {
    const float4* wp = reinterpret_cast<const float4*>(a.w);
#pragma unroll
    for (int j = 0; j < C::B_SLOTS; ++j) {
      const int idx = tid + 256 * j;
      if (idx < C::B_F4) {
        const int seg = idx / C::BN, co = idx % C::BN;
        breg[j] = wp[(((size_t)tap * nchunks + chunk) * 8 + seg) * a.Cout + co0 + co];
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

  // De-phase the workgroups: all of them stream the SAME weight tiles, and in lockstep they would hit the same L2
  // channels at the same time.  Each workgroup therefore starts at its own (chunk, tap) and wraps around; the sum is
  // order independent up to fp32 re-association.
  const int rot_t = (a.dbg & 16) ? 0 : logical % TAPS;
  const int rot_c = (a.dbg & 16) ? 0 : (logical / TAPS) % nchunks;
  auto tp = [&](int tap) { const int t = tap + rot_t; return t >= TAPS ? t - TAPS : t; };
  auto cp = [&](int chunk) { const int c = chunk + rot_c; return c >= nchunks ? c - nchunks : c; };

  load_A(cp(0));
  load_B(cp(0), tp(0));
  store_A();
  store_B(0);
  __syncthreads();

  int buf = 0;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
      const bool last_tap = (tap == TAPS - 1);
      const bool last = last_tap && (chunk == nchunks - 1);
      if (!last && !(a.dbg & 1)) load_B(cp(last_tap ? chunk + 1 : chunk), tp(last_tap ? 0 : tap + 1));
      if (last_tap && !last && !(a.dbg & 2)) load_A(cp(chunk + 1));

      const int ptap = tp(tap);
      const int tapoff = (TAPS == 9) ? ((ptap / 3) * C::WT + (ptap % 3)) : 0;
      const float4* Bc = Bs + buf * C::B_F4;
      if (!(a.dbg & 4))
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int seg = s * 2 + h;
        F4H8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          ah[i].f4 = As[seg * C::HP + a_base[i] + tapoff];
          al[i].f4 = As[(4 + seg) * C::HP + a_base[i] + tapoff];
        }
#pragma unroll
        for (int c = 0; c < NT; ++c) {
          bh[c].f4 = Bc[seg * C::BN + b_base[c]];
          bl[c].f4 = Bc[(4 + seg) * C::BN + b_base[c]];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int c = 0; c < NT; ++c) {
            acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i].h8, bh[c].h8, acc[i][c], 0, 0, 0);
            acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i].h8, bl[c].h8, acc[i][c], 0, 0, 0);
            acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i].h8, bh[c].h8, acc[i][c], 0, 0, 0);
          }
      }
      if (!last && !(a.dbg & 1)) store_B(buf ^ 1);
      if (last_tap && !last && !(a.dbg & 2)) {
        if (!(a.dbg & 8)) __syncthreads();
        store_A();
      }
      if (!(a.dbg & 8)) __syncthreads();
      buf ^= 1;
    }
  }

  const float inv_scale = a.w_inv_scale ? *a.w_inv_scale : 1.0f;
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
        if (n < a.N) {
          const int y = ty0 + py, x = tx0 + px;
          float v = acc[i][c][e] * inv_scale + bias;
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

template <int TAPS, int TH, int TW, int WM, int WN, int MT, int NT>
static int launch_split_variant(const ConvArgs& a, hipStream_t s) {
  using C = SCfg<TAPS, TH, TW, WM, WN, MT, NT>;
  auto kern = conv_igemm_split_kernel<TAPS, TH, TW, WM, WN, MT, NT>;
  const size_t lds_bytes = (size_t)C::LDS_F4 * sizeof(float4);
  static bool attr_set = false;
  if (!attr_set && lds_bytes > 48 * 1024) {
    DRM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_set = true;
  }
  const int groups = (a.N + C::TN - 1) / C::TN;
  const long long blocks = (long long)groups * (a.H / TH) * (a.W / TW) * (a.Cout / C::BN);
  DRM_REQUIRE(blocks > 0 && blocks < (1ll << 31), "conv grid size");
  {
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

template <int TAPS, int TH, int TW>
static int dispatch_split_bn(const ConvArgs& a, hipStream_t s) {
  if (a.Cout % 128 == 0) return launch_split_variant<TAPS, TH, TW, 2, 2, 2, 2>(a, s);
  if (a.Cout % 64 == 0) return launch_split_variant<TAPS, TH, TW, 2, 2, 2, 1>(a, s);
  return launch_split_variant<TAPS, TH, TW, 4, 1, 1, 1>(a, s);
}

template <int TAPS>
static int dispatch_split_tile(const ConvArgs& a, hipStream_t s) {
  if (a.H % 8 == 0 && a.W % 16 == 0) return dispatch_split_bn<TAPS, 8, 16>(a, s);
  if (a.H % 8 == 0 && a.W % 8 == 0) return dispatch_split_bn<TAPS, 8, 8>(a, s);
  if (a.H % 4 == 0 && a.W % 8 == 0) return dispatch_split_bn<TAPS, 4, 8>(a, s);
  if (a.H % 4 == 0 && a.W % 4 == 0) return dispatch_split_bn<TAPS, 4, 4>(a, s);
  set_error("conv: feature map " + std::to_string(a.H) + "x" + std::to_string(a.W) + " is not a multiple of 4x4");
  return DRM_ERR_INVALID;
}

static int split_v1() {
  static int v1 = -1;
  if (v1 < 0) v1 = getenv("DRM_SPLIT_V1") ? 1 : 0;
  return v1;
}
bool conv_split_fuses_stats() { return !split_v1() && getenv("DRM_NO_FUSED_STATS") == nullptr; }

int launch_conv_split(const ConvArgs& a_in, hipStream_t s) {
  ConvArgs a = a_in;
  {
    static int dbg = -1;
    if (dbg < 0) {
      const char* e = getenv("DRM_DBG");
      dbg = e ? atoi(e) : 0;
    }
    a.dbg = dbg;
  }
  const int Ctot = a.C0 + a.C1;
  DRM_REQUIRE(a.taps == 9 || a.taps == 1, "conv taps must be 9 or 1");
  DRM_REQUIRE(a.Cout % 32 == 0 && Ctot % 32 == 0 && a.C0 % 32 == 0, "split conv needs channels % 32 == 0");
  DRM_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0, "conv shape");
  DRM_REQUIRE(!a.up0 || (a.H % 2 == 0 && a.W % 2 == 0), "upsampled source needs even output size");
  if (!split_v1()) {
    static const int use_s3 = getenv("DRM_S3") ? atoi(getenv("DRM_S3")) : 0;
    if (use_s3 && a.taps == 9) {  // producer / consumer waves (conv_split3.hip) for the 3x3 layers that fill the chip
      bool handled = false;
      const int rc = launch_conv_split3(a, s, handled);
      if (handled) return rc;
    }
    return launch_conv_split2(a, s);  // LDS-DMA weight ring, 256-pixel tiles (conv_split2.hip)
  }
  if (a.taps == 9) return dispatch_split_tile<9>(a, s);
  return dispatch_split_tile<1>(a, s);
}

// ------------------------------------------------------------------------------------------------
// weight pre-split: PyTorch [Cout][Cin][taps] fp32 -> [tap][chunk][hl][s][h][CoutP][8] fp16, scaled by 2^k
// ------------------------------------------------------------------------------------------------
__global__ void absmax_kernel(const float* __restrict__ w, size_t n, unsigned* __restrict__ out_bits) {
  float m = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(m));  // non-negative floats order like their bit patterns
}

// scales[0] = 2^k (applied to the weights), scales[1] = 2^-k (applied in the conv epilogue); max|w|*2^k in [8192, 16384)
__global__ void split_scale_kernel(const unsigned* __restrict__ absmax_bits, float* __restrict__ scales) {
  const float m = __uint_as_float(*absmax_bits);
  int k = 0;
  if (m > 0.f && m < INFINITY) {
    int e;
    frexpf(m, &e);  // m = f * 2^e, f in [0.5, 1)
    k = 14 - e;
    if (k > 30) k = 30;
    if (k < -30) k = -30;
  }
  scales[0] = ldexpf(1.0f, k);
  scales[1] = ldexpf(1.0f, -k);
}

__global__ void pack_conv_weight_split_kernel(const float* __restrict__ w, _Float16* __restrict__ p, const float* __restrict__ scales, int Cout,
                                              int Cin, int taps, int CoutP, int CinP) {
  const int nchunks = CinP / 32;
  const size_t total = (size_t)taps * nchunks * 2 * 2 * 2 * CoutP * 8;
  const float scale = scales[0];
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i % 8;
    size_t t = i / 8;
    const int co = t % CoutP; t /= CoutP;
    const int hh = t % 2; t /= 2;
    const int s = t % 2; t /= 2;
    const int hl = t % 2; t /= 2;
    const int q = t % nchunks;
    const int tap = t / nchunks;
    const int ci = 32 * q + 16 * s + 8 * hh + j;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * taps + tap] * scale;
    const _Float16 hi = (_Float16)v;
    p[i] = hl ? (_Float16)(v - (float)hi) : hi;
  }
}

size_t packed_conv_weight_split_floats(int taps, int CoutP, int CinP) {
  return (size_t)taps * CoutP * CinP;  // hi + lo fp16 = 4 bytes per weight, same footprint as fp32
}

int launch_pack_conv_weight_split(const float* w, float* packed, float* scales /*[2] device*/, unsigned* scratch /*1 uint device*/, int Cout,
                                  int Cin, int taps, int CoutP, int CinP, hipStream_t s) {
  DRM_REQUIRE(CinP % 32 == 0 && CoutP >= Cout && CinP >= Cin, "pack_conv_weight_split padding");
  const size_t n = (size_t)Cout * Cin * taps;
  DRM_HIP_CHECK(hipMemsetAsync(scratch, 0, sizeof(unsigned), s));
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0, s, w, n, scratch);
  DRM_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(split_scale_kernel, dim3(1), dim3(1), 0, s, scratch, scales);
  DRM_HIP_CHECK(hipGetLastError());
  const size_t total = (size_t)taps * CoutP * CinP * 2;
  hipLaunchKernelGGL(pack_conv_weight_split_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, w,
                     reinterpret_cast<_Float16*>(packed), scales, Cout, Cin, taps, CoutP, CinP);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

}  // namespace drm
