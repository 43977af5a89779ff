// The stem of a U-Net as a kernel of its own ("Stem (Cin = 6) and head (Cout = 3) get thin special cases").
//
//   stem_conv_kernel  input_blocks.0.0 = conv3x3(in_channels -> model_channels) on the raw network input (openaimodel.py:534, :757).  Through the
//                     generic pipeline kernel the 6 input channels were padded to one 32-channel chunk (K = 288 for 54 real taps x channels) behind
//                     a pack launch and a range-guard launch: 0.27 + 0.05 ms at batch 32 for a layer whose floor is its 537 MB output write.  Here:
//                     one launch reads x and cond in their NCHW boundary layout (cat([x, cond], 1) of DiffusionWrapper, ddpm.py:1527-1529, and the
//                     DRMNet active-row gather, drmnet.py:810-813, fused as before), builds the im2col rows (K = 9 * Cin, tap-major) in LDS and runs
//                     its products on the matrix pipe -- EXACT fp32 (v_mfma_f32_32x32x2_f32, 28 MFMAs of 64 cycles per block) in the fp32 mode; in
//                     every other mode the fp16 hi / lo split (12 v_mfma_f32_32x32x16_f16 per block) behind a per-TILE power of two taken from the
//                     tile's max |v| (exact, undone in the epilogue; a tile holding NaN / Inf goes through unclamped so the poison propagates) --
//                     then bias, NHWC store and the fused GroupNorm statistics of the output.
// (The head keeps the generic pipeline kernel: a kernel of its own -- one pixel per lane, exact fp32 FMAs -- measured 0.36 ms against 0.31 ms and
//  lives in tools/experiments/head_conv_kernel_r5.hip.txt, not in the library.)
#include <algorithm>
#include <atomic>

#include "common.h"
#include "profiler.h"

namespace drm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

// ---------------------------------------------------------------------------------------------- stem
constexpr int STEM_BM = 128;      // pixels per tile: 8 rows x 16 columns, four 4 x 8 patches = the four 32-row MFMA blocks
constexpr int STEM_TH = 8, STEM_TW = 16;
constexpr int STEM_COUT = 128;
constexpr int STEM_LD = 60;       // exact form, floats per LDS row: [half 0: KHP][half 1: KHP] padded so that 16 lanes of a ds_read_b128 hit 16 distinct 16-byte slots
constexpr int STEM_LDS = 68;      // split form, floats per LDS row: 64 fp16 hi + 64 fp16 lo (K padded to 64) + 16 bytes (same reason)
typedef _Float16 sh_f16x8 __attribute__((ext_vector_type(8)));
union ShF4H8 {
  float4 f4;
  sh_f16x8 h8;
};

template <int DPP_CTRL>
__device__ __forceinline__ float sh_quad_perm(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), DPP_CTRL, 0xF, 0xF, false));
}
// 4 x 4 transpose across the lanes of a quad (conv_split2.hip quad_transpose)
__device__ __forceinline__ void sh_quad_transpose(float& x0, float& x1, float& x2, float& x3, int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
  float t;
  t = sh_quad_perm<0xB1>(b0 ? x0 : x1);
  if (b0) x0 = t; else x1 = t;
  t = sh_quad_perm<0xB1>(b0 ? x2 : x3);
  if (b0) x2 = t; else x3 = t;
  t = sh_quad_perm<0x4E>(b1 ? x0 : x2);
  if (b1) x0 = t; else x2 = t;
  t = sh_quad_perm<0x4E>(b1 ? x1 : x3);
  if (b1) x1 = t; else x3 = t;
}

// GEMM row of a tile (0 .. 127) -> pixel inside the 8 x 16 tile: 32-row blocks are 4 x 8 patches, so the rows 8g + 4h + k a lane holds of an
// accumulator block are four consecutive pixels of one image row
__device__ __forceinline__ void stem_rowmap(int row, int& py, int& px) {
  const int q = row >> 5, rr = row & 31;
  py = (q >> 1) * 4 + (rr >> 3);
  px = (q & 1) * 8 + (rr & 7);
}

// K = 9 * CIN products per output.  The fp32 MFMA takes k = h from lane half h, so the K axis is cut in two halves of KH = 9 * CIN / 2 and
// MFMA j of a block multiplies element j of half h: lane half h owns the channels [h * CIN / 2, (h + 1) * CIN / 2) of every tap
// (j = tap * CIN / 2 + c'), and a lane's operands are KH consecutive floats of its LDS row -- read four at a time.
template <int CIN>
struct StemCfg {
  static_assert(CIN % 2 == 0, "the two lane halves split the input channels");
  static constexpr int CH = CIN / 2;
  static constexpr int KH = 9 * CH;
  static constexpr int KHP = (KH + 3) / 4 * 4;  // per-half length in LDS (zero padded: the padding MFMAs add 0 * 0)
  static_assert(2 * KHP <= STEM_LD, "LDS row holds both halves");
};

// EXACT: fp32 operands on v_mfma_f32_32x32x2_f32 (DRM_PREC_FP32: 28 MFMAs of 64 cycles per 32 x 32 block -- the matrix pipe then bounds the
// launch at ~0.25 ms at batch 32).  Otherwise the fp16 hi / lo split of the pipeline kernel (three v_mfma_f32_32x32x16_f16 per 16-deep K step,
// 22 operand bits, K padded 54 -> 64: 12 MFMAs of 32 cycles per block) behind a per-TILE power-of-two range guard: the tile's values are staged
// times 2^k with max |v| 2^k in [2^14, 2^15) (exact; undone in the epilogue), so the split keeps its 22 bits whatever the input's scale -- what
// engine.hip raw_input_guard does per image for the generic path.
template <int CIN, bool EXACT>
__global__ __launch_bounds__(256, 2) void stem_conv_kernel(const float* __restrict__ x, int Cx, const float* __restrict__ cond, int Cc,
                                                            const int* __restrict__ rows, const float* __restrict__ wimg /* [128][LD] (+ 2^-kw) */,
                                                            const float* __restrict__ bias, float* __restrict__ out, double2* __restrict__ stat,
                                                            int N, int H, int W) {
  using C = StemCfg<CIN>;
  constexpr int LD = EXACT ? STEM_LD : STEM_LDS;
  extern __shared__ float4 stem_lds[];
  float* As = reinterpret_cast<float*>(stem_lds);              // [128 rows][LD]
  float* Bs = As + STEM_BM * LD;                               // [128 couts][LD]
  double* lst = reinterpret_cast<double*>(Bs + STEM_COUT * LD);  // [128][2] per-channel (sum, sum of squares) of the current image
  float* wmax = reinterpret_cast<float*>(lst + 256);           // [4] per-wave max |value| of the tile being staged (split form)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_x = (W + STEM_TW - 1) / STEM_TW, tiles_y = (H + STEM_TH - 1) / STEM_TH;
  const int tiles_img = tiles_x * tiles_y;
  const long long total = (long long)N * tiles_img;
  // contiguous tile ranges per workgroup: an image's tiles are consecutive, so the statistics are flushed when the image changes
  const long long t0 = total * blockIdx.x / gridDim.x, t1 = total * (blockIdx.x + 1) / gridDim.x;
  if (t0 >= t1) return;

  // weights: the packed image is the LDS image
  for (int k = tid; k < STEM_COUT * LD / 4; k += 256) reinterpret_cast<float4*>(Bs)[k] = reinterpret_cast<const float4*>(wimg)[k];
  lst[tid] = 0.0;
  const float w_inv = EXACT ? 1.0f : wimg[STEM_COUT * LD];  // 2^-kw of the weight pre-scaling

  // loader: thread = (tile row p, half lh): the CIN / 2 channels of half lh at the nine taps of pixel p, straight from the NCHW planes of x / cond
  const int l_p = tid >> 1, l_h = tid & 1;
  int l_py, l_px;
  stem_rowmap(l_p, l_py, l_px);
  float av[C::KHP];
  const size_t HW = (size_t)H * W;
  auto load_tile = [&](long long t) {
    const int n = (int)(t / tiles_img), ti = (int)(t % tiles_img);
    const int y0 = (ti / tiles_x) * STEM_TH + l_py, x0 = (ti % tiles_x) * STEM_TW + l_px;
    const int src = rows ? rows[n] : n;
    const float* pl[C::CH];
#pragma unroll
    for (int cc = 0; cc < C::CH; ++cc) {
      const int c = l_h * C::CH + cc;
      pl[cc] = c < Cx ? x + ((size_t)src * Cx + c) * HW : cond + ((size_t)src * Cc + (c - Cx)) * HW;
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int y = y0 + tap / 3 - 1, xx = x0 + tap % 3 - 1;
      const bool ok = y >= 0 && y < H && xx >= 0 && xx < W;
      const size_t off = (size_t)min(max(y, 0), H - 1) * W + min(max(xx, 0), W - 1);  // always loaded (clamped), zeroed outside the map
#pragma unroll
      for (int cc = 0; cc < C::CH; ++cc) {
        const float v = pl[cc][off];
        av[tap * C::CH + cc] = ok ? v : 0.f;
      }
    }
#pragma unroll
    for (int j = C::KH; j < C::KHP; ++j) av[j] = 0.f;
  };
  float t_inv = 1.0f;  // split form: 2^-k of the staged tile
  auto store_tile = [&]() {
    if constexpr (EXACT) {
      float4* d = reinterpret_cast<float4*>(As + l_p * LD + C::KHP * l_h);
#pragma unroll
      for (int j = 0; j < C::KHP / 4; ++j) d[j] = make_float4(av[4 * j], av[4 * j + 1], av[4 * j + 2], av[4 * j + 3]);
    } else {
      static_assert(C::KH <= 32, "split form: a lane half's K elements fit 32 fp16");
      // the tile's power of two: wave maxima meet in LDS (one barrier), every thread derives the same k
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < C::KH; ++j) m = fmaxf(m, fabsf(av[j]) == fabsf(av[j]) ? fabsf(av[j]) : INFINITY);  // (NaN input: no finite bound)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      if (lane == 0) wmax[wave] = m;
      __syncthreads();
      const float bound = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
      int k = 0;
      if (bound > 0.f && bound < INFINITY) {
        int e;
        frexpf(bound, &e);  // bound = f * 2^e, f in [0.5, 1)
        k = 15 - e;
        k = k > 90 ? 90 : (k < -90 ? -90 : k);
      }
      const float sc = ldexpf(1.0f, k);
      t_inv = ldexpf(1.0f, -k);
      const bool finite_tile = bound < INFINITY;
      ShF4H8 hi[4], lo[4];
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        _Float16 hh = (_Float16)0.f, ll = (_Float16)0.f;
        if (j < C::KH) {
          // (a tile that holds a NaN / Inf pixel has no finite bound: its values go through unclamped, so the poison propagates into the outputs of
          //  that tile as it does through F.conv2d -- a downstream isfinite() guard sees it; finite tiles are untouched: ADVICE r5)
          const float v = finite_tile ? __builtin_amdgcn_fmed3f(av[j] * sc, -65504.0f, 65504.0f) : av[j];
          hh = (_Float16)v;
          ll = (_Float16)(v - (float)hh);
        }
        hi[j >> 3].h8[j & 7] = hh;
        lo[j >> 3].h8[j & 7] = ll;
      }
      float4* d = reinterpret_cast<float4*>(As + l_p * LD) + 4 * l_h;  // hi plane: 32 fp16 of half l_h; lo plane 8 float4 further
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        d[j] = hi[j].f4;
        d[8 + j] = lo[j].f4;
      }
    }
  };

  f32x16 acc[2][2];
  int n_cur = (int)(t0 / tiles_img);
  auto flush_stats = [&](int n) {  // (between barriers: every lane's LDS atomics of the finished tiles are in)
    if (stat) {
      const int c = tid >> 1, m = tid & 1;
      atomicAdd(reinterpret_cast<double*>(stat + (size_t)n * STEM_COUT + c) + m, lst[tid]);
    }
    lst[tid] = 0.0;
  };

  load_tile(t0);
  for (long long t = t0; t < t1; ++t) {
    const int n = (int)(t / tiles_img), ti = (int)(t % tiles_img);
    __syncthreads();  // the previous tile's fragment reads (and its statistics atomics) are done
    if (n != n_cur) {
      flush_stats(n_cur);
      n_cur = n;
      __syncthreads();
    }
    store_tile();
    __syncthreads();
    if (t + 1 < t1) load_tile(t + 1);  // in flight under the MFMAs
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][c][e] = 0.f;
    if constexpr (EXACT) {
      const float* ap = As + ((wm * 2) * 32 + r) * LD + C::KHP * h;
      const float* bp = Bs + ((wn * 2) * 32 + r) * LD + C::KHP * h;
#pragma unroll
      for (int g4 = 0; g4 < C::KHP / 4; ++g4) {
        float4 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const float4*>(ap + i * 32 * LD + 4 * g4);
#pragma unroll
        for (int c = 0; c < 2; ++c) b[c] = *reinterpret_cast<const float4*>(bp + c * 32 * LD + 4 * g4);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[c].x, acc[i][c], 0, 0, 0);
            acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[c].y, acc[i][c], 0, 0, 0);
            acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[c].z, acc[i][c], 0, 0, 0);
            acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[c].w, acc[i][c], 0, 0, 0);
          }
      }
    } else {
      // K step s2: lane half h takes the eight fp16 at K = 16 s2 + 8 h of its row (hi plane; lo plane 128 bytes further)
      const float4* ap = reinterpret_cast<const float4*>(As + ((wm * 2) * 32 + r) * LD) + h;
      const float4* bp = reinterpret_cast<const float4*>(Bs + ((wn * 2) * 32 + r) * LD) + h;
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        ShF4H8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          ah[i].f4 = ap[i * 32 * (LD / 4) + 2 * s2];
          al[i].f4 = ap[i * 32 * (LD / 4) + 8 + 2 * s2];
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          bh[c].f4 = bp[c * 32 * (LD / 4) + 2 * s2];
          bl[c].f4 = bp[c * 32 * (LD / 4) + 8 + 2 * s2];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i].h8, bh[c].h8, acc[i][c], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i].h8, bl[c].h8, acc[i][c], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i].h8, bh[c].h8, acc[i][c], 0, 0, 0);
      }
    }
    const float un = EXACT ? 1.0f : t_inv * w_inv;
    // epilogue: + bias, statistics, NHWC store (16 bytes per lane after the quad transpose: four channels of one pixel)
    const int ty0 = (ti / tiles_x) * STEM_TH, tx0 = (ti % tiles_x) * STEM_TW;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int col = (wn * 2 + c) * 32 + r;
      const float bv = bias[col];
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int blk = wm * 2 + i;
        const int py0 = (blk >> 1) * 4, px0 = (blk & 1) * 8 + 4 * h;
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          v[e] = EXACT ? acc[i][c][e] + bv : acc[i][c][e] * un + bv;
          const int y = ty0 + py0 + (e >> 2), xx = tx0 + px0 + (e & 3);
          if (y < H && xx < W) {
            s += v[e];
            q += v[e] * v[e];
          }
        }
        const int cq = (wn * 2 + c) * 32 + (r & ~3);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          sh_quad_transpose(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3], r);
          const int y = ty0 + py0 + g, xx = tx0 + px0 + (r & 3);
          if (y < H && xx < W)
            *reinterpret_cast<float4*>(out + (((size_t)n * H + y) * W + xx) * STEM_COUT + cq) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
        }
      }
      atomicAdd(lst + 2 * col, (double)s);
      atomicAdd(lst + 2 * col + 1, (double)q);
    }
  }
  __syncthreads();
  flush_stats(n_cur);
}

// PyTorch [Cout][Cin][3][3] -> the stem kernel's LDS image [Cout][STEM_LD]: half hh = channels [hh * Cin / 2, ...), element j = tap * Cin / 2 + c'
// at float hh * KHP + j, zeros elsewhere
__global__ void pack_stem_weight_kernel(const float* __restrict__ w, float* __restrict__ img, int Cout, int Cin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cout * STEM_LD) return;
  const int co = i / STEM_LD, f = i % STEM_LD;
  const int CH = Cin / 2, KH = 9 * CH, KHP = (KH + 3) / 4 * 4;
  float v = 0.f;
  if (f < 2 * KHP) {
    const int hh = f / KHP, j = f % KHP;
    if (j < KH) {
      const int tap = j / CH, c = hh * CH + j % CH;
      v = w[((size_t)co * Cin + c) * 9 + tap];
    }
  }
  img[i] = v;
}

// split form: one block; max |w| -> 2^kw with max |w| 2^kw in [2^13, 2^14) (conv_split.hip split_scale_kernel), rows of 64 fp16 hi + 64 fp16 lo
// (K element 32 hh + tap * Cin / 2 + c'), then 2^-kw at float Cout * STEM_LDS
__global__ __launch_bounds__(256) void pack_stem_weight_split_kernel(const float* __restrict__ w, float* __restrict__ img, int Cout, int Cin) {
  __shared__ float red[4];
  const int t = threadIdx.x, n = Cout * Cin * 9;
  float m = 0.f;
  for (int i = t; i < n; i += 256) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((t & 63) == 0) red[t >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  int k = 0;
  if (m > 0.f && m < INFINITY) {
    int e;
    frexpf(m, &e);
    k = 14 - e;
    k = k > 30 ? 30 : (k < -30 ? -30 : k);
  }
  const float sc = ldexpf(1.0f, k);
  const int CH = Cin / 2, KH = 9 * CH;
  _Float16* o = reinterpret_cast<_Float16*>(img);
  for (int i = t; i < Cout * 64; i += 256) {
    const int co = i >> 6, kk = i & 63, hh = kk >> 5, j = kk & 31;
    float v = 0.f;
    if (j < KH) v = w[((size_t)co * Cin + hh * CH + j % CH) * 9 + j / CH] * sc;
    const _Float16 hi = (_Float16)v;
    o[(size_t)co * (STEM_LDS * 2) + kk] = hi;
    o[(size_t)co * (STEM_LDS * 2) + 64 + kk] = (_Float16)(v - (float)hi);
  }
  for (int i = t; i < Cout; i += 256)
    for (int p = 0; p < 8; ++p) o[(size_t)i * (STEM_LDS * 2) + 128 + p] = (_Float16)0.f;  // the row padding
  if (t == 0) img[(size_t)Cout * STEM_LDS] = ldexpf(1.0f, -k);
}

}  // namespace

bool stem_direct_applicable(int Cin, int Cout) { return Cout == STEM_COUT && (Cin == 6 || Cin == 4); }
size_t stem_weight_floats() { return (size_t)STEM_COUT * STEM_LDS + 64; }  // (the larger of the two images + the 2^-kw word)

int launch_pack_stem_weight(const float* w, float* img, int Cout, int Cin, bool exact, hipStream_t s) {
  DRM_REQUIRE(stem_direct_applicable(Cin, Cout), "stem weight image: unsupported channel counts");
  if (exact) {
    const int n = Cout * STEM_LD;
    hipLaunchKernelGGL(pack_stem_weight_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, img, Cout, Cin);
  } else {
    hipLaunchKernelGGL(pack_stem_weight_split_kernel, dim3(1), dim3(256), 0, s, w, img, Cout, Cin);
  }
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

int launch_stem_conv(const float* x, int Cx, const float* cond, int Cc, const int* rows, const float* wimg, const float* bias, float* out, double2* stat,
                     int N, int H, int W, int Cout, bool exact, hipStream_t s) {
  const int Cin = Cx + Cc;
  DRM_REQUIRE(stem_direct_applicable(Cin, Cout), "stem kernel: unsupported channel counts");
  const DeviceInfo* di = device_info();
  if (!di) return DRM_ERR_STATE;
  const size_t lds = (size_t)(STEM_BM + STEM_COUT) * (exact ? STEM_LD : STEM_LDS) * sizeof(float) + 256 * sizeof(double) + 16;
  const long long tiles = (long long)N * ((H + STEM_TH - 1) / STEM_TH) * ((W + STEM_TW - 1) / STEM_TW);
  const int grid = (int)std::min<long long>(tiles, 2ll * di->cus);
  prof_tag(N, H, W, Cin, Cout);
  ProfScope ps(PROF_CONV3, 2.0 * N * H * W * 9.0 * Cin * Cout, 4.0 * ((double)N * H * W * (Cin + Cout) + 9.0 * Cin * Cout), s);
  // one opt-in mask PER kernel variant (all four share a function-pointer type, so a static inside the generic lambda would be shared: ADVICE r5)
  static std::atomic<uint64_t> attr_masks[4];
  auto go = [&](auto kern, int variant) -> int {
    std::atomic<uint64_t>& attr_mask = attr_masks[variant];
    if (!(attr_mask.load(std::memory_order_acquire) >> di->ordinal & 1)) {
      DRM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_mask.fetch_or(uint64_t(1) << di->ordinal, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, x, Cx, cond, Cc, rows, wimg, bias, out, stat, N, H, W);
    DRM_HIP_CHECK(hipGetLastError());
    return DRM_OK;
  };
  if (Cin == 4) return exact ? go(stem_conv_kernel<4, true>, 0) : go(stem_conv_kernel<4, false>, 1);
  return exact ? go(stem_conv_kernel<6, true>, 2) : go(stem_conv_kernel<6, false>, 3);
}

}  // namespace drm
