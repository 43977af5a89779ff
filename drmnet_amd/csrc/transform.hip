// The elementwise maps either side of the samplers and the envmap warp after them, as HBM-bound
// kernels on NCHW boundary tensors.
//
// Reference semantics restated (paths relative to the reference root):
//   BaseDataset.transform / rescale        dataset/basedataset.py:29-112   (a chain of named maps, applied right-to-left / inverted)
//   DRMNet.get_input_for_predict scaling   models/drmnet.py:1017-1034      (per-image luminance geometric mean -> refmap_input_scaler)
//   DRMNet.r0toenvmap / mirmap2envmap      models/drmnet.py:931-941, utils/transform.py:106-144
//   hdr2ldr                                utils/tonemap.py:4-9
// One launch applies a whole chain of maps per element (the reference makes one pass over the tensor per map); the two
// data-dependent pieces -- the masked min/max of `normalizedLogarithmic` and the luminance mean -- are one workgroup per image.
#include "common.h"

namespace drm {

// ------------------------------------------------------------------------------------------------ map chain
// op codes of include/drmnet_hip.h (DRM_MAP_*)
enum MapOp {
  MAP_LOG_P1 = 0,         // log10(x + 0.1) + 1                       "log"
  MAP_LOG10 = 1,          // log10(x)                                 "log10"
  MAP_LOWERBOUND = 2,     // clip(x, min = arg)                       "lowerbound<arg>"
  MAP_UNIT_TO_SIGNED = 3, // 2 x - 1                                  "0p1tom1p1"
  MAP_NORM_LOG = 4,       // (log10(x) - lo[b]) / (hi[b] - lo[b])     "normalizedLogarithmic"
  MAP_EXP_M1 = 5,         // 10^min(x - 1, arg) - 0.1                 inverse of "log"   (arg = clamp_before_exp or +inf)
  MAP_EXP10 = 6,          // 10^min(x, arg)                           inverse of "log10"
  MAP_SIGNED_TO_UNIT = 7, // (x + 1) / 2                              inverse of "0p1tom1p1"
  MAP_DENORM_LOG = 8,     // x (hi[b] - lo[b]) + lo[b]                first half of the inverse of "normalizedLogarithmic"
  MAP_IMG_MUL = 9,        // x * scale[b]                             normalizing_scale (drmnet.py:1027)
  MAP_IMG_DIV = 10,       // x / scale[b]                             scripts/estimate.py:99-100
  MAP_CLIP0 = 11,         // clip(x, min = 0)                         scripts/estimate.py:97
  MAP_COUNT = 12
};
constexpr int MAX_MAP_OPS = 8;
struct MapChain {
  int n;
  int op[MAX_MAP_OPS];
  float arg[MAX_MAP_OPS];
};

__device__ __forceinline__ float apply_map(int op, float arg, float v, float lo, float hi, float sc) {
  switch (op) {
    case MAP_LOG_P1: return log10f(v + 1e-1f) + 1.0f;
    case MAP_LOG10: return log10f(v);
    case MAP_LOWERBOUND: return v < arg ? arg : v;  // NaN stays NaN like torch.clip
    case MAP_UNIT_TO_SIGNED: return v * 2.0f - 1.0f;
    case MAP_NORM_LOG: return (log10f(v) - lo) / (hi - lo);
    case MAP_EXP_M1: { const float e = v - 1.0f; return powf(10.0f, e > arg ? arg : e) - 1e-1f; }
    case MAP_EXP10: return powf(10.0f, v > arg ? arg : v);
    case MAP_SIGNED_TO_UNIT: return (v + 1.0f) / 2.0f;
    case MAP_DENORM_LOG: return v * (hi - lo) + lo;
    case MAP_IMG_MUL: return v * sc;
    case MAP_IMG_DIV: return v / sc;
    case MAP_CLIP0: return v < 0.0f ? 0.0f : v;
  }
  return v;
}

// grid (blocks over one image's elements, B): every thread maps four consecutive elements through the whole chain
__global__ __launch_bounds__(256) void map_chain_kernel(const float* __restrict__ x, float* __restrict__ out, long long per_image, MapChain ch,
                                                        const float* __restrict__ lo, const float* __restrict__ hi,
                                                        const float* __restrict__ scale) {
  const int b = blockIdx.y;
  const float l = lo ? lo[b] : 0.f, h = hi ? hi[b] : 1.f, sc = scale ? scale[b] : 1.f;
  const long long base = (long long)b * per_image;
  const long long i0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i0 >= per_image) return;
  float v[4];
  const bool full = (i0 + 3 < per_image) && (((base + i0) & 3) == 0);
  if (full) {
    const float4 t = *reinterpret_cast<const float4*>(x + base + i0);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (i0 + k < per_image) ? x[base + i0 + k] : 0.f;
  }
  for (int j = 0; j < ch.n; ++j) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = apply_map(ch.op[j], ch.arg[j], v[k], l, h, sc);
  }
  if (full) {
    *reinterpret_cast<float4*>(out + base + i0) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i0 + k < per_image) out[base + i0 + k] = v[k];
  }
}

int launch_map_chain(const float* x, float* out, long long per_image, int B, const int32_t* ops, const float* args, int n_ops, const float* lo,
                     const float* hi, const float* scale, hipStream_t s) {
  DRM_REQUIRE(x && out && per_image > 0 && B > 0, "map chain: shape");
  DRM_REQUIRE(n_ops >= 0 && n_ops <= MAX_MAP_OPS && (n_ops == 0 || ops), "map chain: at most 8 maps");
  MapChain ch{};
  ch.n = n_ops;
  for (int j = 0; j < n_ops; ++j) {
    DRM_REQUIRE(ops[j] >= 0 && ops[j] < MAP_COUNT, "map chain: unknown op code " + std::to_string(ops[j]));
    DRM_REQUIRE((ops[j] != MAP_NORM_LOG && ops[j] != MAP_DENORM_LOG) || (lo && hi), "normalizedLogarithmic needs the per-image (log10 min, log10 max)");
    DRM_REQUIRE((ops[j] != MAP_IMG_MUL && ops[j] != MAP_IMG_DIV) || scale, "per-image scale map needs the scale vector");
    ch.op[j] = ops[j];
    ch.arg[j] = args ? args[j] : 0.f;
  }
  const long long quads = (per_image + 3) / 4;
  hipLaunchKernelGGL(map_chain_kernel, dim3((unsigned)((quads + 255) / 256), (unsigned)B), dim3(256), 0, s, x, out, per_image, ch, lo, hi, scale);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

// ------------------------------------------------------------------------------------------------ per-image reductions
template <typename T, typename F>
__device__ __forceinline__ T block_reduce(T v, F f, T* red /*[8]*/, T ident) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = f(v, __shfl_xor(v, o));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  T r = ident;
  const int nw = blockDim.x >> 6;
  for (int k = 0; k < nw; ++k) r = f(r, red[k]);
  return r;
}

// dynamic_normalize branch of "normalizedLogarithmic" (basedataset.py:63-69), one workgroup per image:
//   linearmax = max(x * mask);  hi = log10(linearmax);  lo = log10(min(x * mask + (1 - mask) * linearmax))
// x [B][C][HW] (already lower-bounded by the maps in front of it), mask [B][HW] fp32 0/1 (broadcast over channels).
__global__ __launch_bounds__(512) void masked_log_range_kernel(const float* __restrict__ x, const float* __restrict__ mask, int C, int HW,
                                                               float* __restrict__ lo, float* __restrict__ hi) {
  __shared__ float red[8];
  const int b = blockIdx.x;
  const float* xb = x + (size_t)b * C * HW;
  const float* mb = mask + (size_t)b * HW;
  const int n = C * HW;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < n; i += blockDim.x) mx = fmaxf(mx, xb[i] * mb[i % HW]);
  mx = block_reduce(mx, [](float a, float c) { return fmaxf(a, c); }, red, -INFINITY);
  float mn = INFINITY;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float m = mb[i % HW];
    mn = fminf(mn, xb[i] * m + (1.0f - m) * mx);
  }
  mn = block_reduce(mn, [](float a, float c) { return fminf(a, c); }, red, INFINITY);
  if (threadIdx.x == 0) {
    hi[b] = log10f(mx);
    lo[b] = log10f(mn);
  }
}

int launch_masked_log_range(const float* x, const float* mask, int B, int C, int HW, float* lo, float* hi, hipStream_t s) {
  DRM_REQUIRE(x && mask && lo && hi && B > 0 && C > 0 && HW > 0, "masked_log_range: shape");
  hipLaunchKernelGGL(masked_log_range_kernel, dim3(B), dim3(512), 0, s, x, mask, C, HW, lo, hi);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

// scale[b] = scaler / exp( sum(log(clip(L, 1e-5)) * (L > 0)) / count(L > 0) ),  L = Rec.709 luminance of x[b] (drmnet.py:1020-1026)
__global__ __launch_bounds__(512) void luminance_scale_kernel(const float* __restrict__ x, int HW, float scaler, float* __restrict__ scale) {
  __shared__ double red[8];
  const int b = blockIdx.x;
  const float* r = x + (size_t)b * 3 * HW;
  double sum = 0.0, cnt = 0.0;
  for (int i = threadIdx.x; i < HW; i += blockDim.x) {
    const float L = 0.212671f * r[i] + 0.715160f * r[HW + i] + 0.072169f * r[2 * HW + i];
    if (L > 0.f) {
      sum += (double)logf(fmaxf(L, 1e-5f));
      cnt += 1.0;
    }
  }
  auto add = [](double a, double c) { return a + c; };
  sum = block_reduce(sum, add, red, 0.0);
  cnt = block_reduce(cnt, add, red, 0.0);
  if (threadIdx.x == 0) scale[b] = scaler / expf((float)(sum / cnt));
}

int launch_luminance_scale(const float* x, int B, int HW, float scaler, float* scale, hipStream_t s) {
  DRM_REQUIRE(x && scale && B > 0 && HW > 0, "luminance_scale: shape");
  hipLaunchKernelGGL(luminance_scale_kernel, dim3(B), dim3(512), 0, s, x, HW, scaler, scale);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

// ------------------------------------------------------------------------------------------------ mirror map -> envmap
// mirmap2envmap (utils/transform.py:106-144) with view = +z, top = +y, zenith = +y, left edge = -z, reverse_azimuth: the
// envmap texel (i, j) looks along d(theta_i, phi_j); the mirror ball shows it at the pixel whose normal is the half vector
// n = normalize(d + view); (theta, phi) of n about (top, view) give the sample position, fetched with torch's
// grid_sample(bilinear, padding_mode="border", align_corners=False) arithmetic.  Optionally divides by basis_r0 first
// (DRMNet.r0toenvmap, models/drmnet.py:931-941) and stores channels-last ([B,OH,OW,C], what r0toenvmap returns).
__global__ __launch_bounds__(256) void mirmap2envmap_kernel(const float* __restrict__ mir, const float* __restrict__ basis, float* __restrict__ out,
                                                            int B, int C, int H, int W, int OH, int OW, int log_interp, int nhwc) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= OH * OW) return;
  const int i = idx / OW, j = idx % OW;
  const float PI = 3.14159265358979323846f;
  const float theta = ((float)i + 0.5f) * (PI / (float)OH);
  const float phi = -(((float)j + 0.5f) * (PI * 2.0f / (float)OW));
  const float st = sinf(theta);
  float nx = -st * sinf(phi), ny = cosf(theta), nz = -st * cosf(phi) + 1.0f;
  const float len = fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-12f);
  nx /= len; ny /= len; nz /= len;
  const float u = atan2f(nx, nz) * (2.0f / PI);
  const float v = acosf(ny) * (2.0f / PI) - 1.0f;
  // grid_sample: unnormalise (align_corners = False), clip to the border, bilinear
  float ix = ((u + 1.0f) * (float)W - 1.0f) / 2.0f, iy = ((v + 1.0f) * (float)H - 1.0f) / 2.0f;
  ix = fminf((float)(W - 1), fmaxf(ix, 0.0f));
  iy = fminf((float)(H - 1), fmaxf(iy, 0.0f));
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
  const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
  const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;  // (nw, ne, sw, se)
  for (int b = blockIdx.y; b < B; b += gridDim.y) {
    for (int c = 0; c < C; ++c) {
      const float* p = mir + ((size_t)b * C + c) * H * W;
      const float* q = basis ? basis + (size_t)c * H * W : nullptr;
      auto at = [&](int y, int x) {
        float t = p[y * W + x];
        if (q) t = t / q[y * W + x];
        return log_interp ? logf(fmaxf(t, 1e-7f)) : t;
      };
      float acc = 0.f;
      acc += at(y0, x0) * w00;
      if (x1 < W) acc += at(y0, x1) * w01;
      if (y1 < H) acc += at(y1, x0) * w10;
      if (x1 < W && y1 < H) acc += at(y1, x1) * w11;
      if (log_interp) acc = expf(acc);
      if (nhwc) out[(((size_t)b * OH + i) * OW + j) * C + c] = acc;
      else out[(((size_t)b * C + c) * OH + i) * OW + j] = acc;
    }
  }
}

int launch_mirmap2envmap(const float* mir, const float* basis, float* out, int B, int C, int H, int W, int OH, int OW, int log_interp, int nhwc,
                         hipStream_t s) {
  DRM_REQUIRE(mir && out && B > 0 && C > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "mirmap2envmap: shape");
  hipLaunchKernelGGL(mirmap2envmap_kernel, dim3((OH * OW + 255) / 256, std::min(B, 1024)), dim3(256), 0, s, mir, basis, out, B, C, H, W, OH, OW,
                     log_interp, nhwc);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

// ------------------------------------------------------------------------------------------------ tone map
// hdr2ldr (utils/tonemap.py:4-9): coeff = alpha / exp(mean over lit pixels of log(L + 1e-7)); out = clip(x coeff, 0, 1)^(1/gamma).
// x [HW][3] (channels last, one image), mask optional uint8 [HW].  Single workgroup: reduction, then the map.
__global__ __launch_bounds__(1024) void hdr2ldr_kernel(const float* __restrict__ x, const unsigned char* __restrict__ mask, int HW, float alpha,
                                                       float inv_gamma, float* __restrict__ out) {
  __shared__ double red[16];
  double sum = 0.0, cnt = 0.0;
  for (int i = threadIdx.x; i < HW; i += blockDim.x) {
    const float L = 0.212671f * x[3 * i] + 0.715160f * x[3 * i + 1] + 0.072169f * x[3 * i + 2];
    const bool lit = (L > 5e-5f) && (!mask || mask[i]);
    if (lit) {
      sum += (double)logf(fmaxf(L, 0.f) + 1e-7f);
      cnt += 1.0;
    }
  }
  auto add = [](double a, double c) { return a + c; };
  sum = block_reduce(sum, add, red, 0.0);
  cnt = block_reduce(cnt, add, red, 0.0);
  const float coeff = alpha / expf((float)(sum / cnt));
  for (int i = threadIdx.x; i < 3 * HW; i += blockDim.x) out[i] = powf(fminf(fmaxf(x[i] * coeff, 0.f), 1.f), inv_gamma);
}

int launch_hdr2ldr(const float* x, const unsigned char* mask, int HW, float alpha, float gamma, float* out, hipStream_t s) {
  DRM_REQUIRE(x && out && HW > 0 && gamma > 0.f, "hdr2ldr: shape");
  hipLaunchKernelGGL(hdr2ldr_kernel, dim3(1), dim3(1024), 0, s, x, mask, HW, alpha, 1.0f / gamma, out);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

// ------------------------------------------------------------------------------------------------ resize
// BaseDataset's "resize" map (dataset/basedataset.py:44-50): torchvision.transforms.functional.resize(x, (size, size), interpolation,
// antialias=True) on a float tensor = torch.nn.functional.interpolate(mode, align_corners=False, antialias=True) -- ATen's separable
// anti-aliased filter (aten/src/ATen/native/cpu/UpSampleKernel.cpp, _compute_indices_min_size_weights_aa; torchvision 0.13.1 /
// torch 1.12.1 are the reference's pins): per output index i along an axis of scale s = in / out,
//   support = (s >= 1 ? s : 1) * interp_size / 2,  centre = s (i + 0.5),  first = max(int(centre - support + 0.5), 0),
//   count = min(int(centre + support + 0.5), in) - first,  w_j = filter((j + first - centre + 0.5) * (s >= 1 ? 1 / s : 1)) / sum,
// filter = triangle (bilinear, interp_size 2) or Keys cubic a = -0.5 (bicubic, interp_size 4); the W axis is filtered first, then H,
// each in fp32 in tap order, as ATen's two passes do.  MODE_NEAREST is `interpolate(x, size)` with its default mode, the mask
// resize of ObsNetDiffusion.get_cond_for_predict (models/obsnet.py:691): src = min(int(floorf(dst * (float)in / out)), in - 1).
enum ResizeMode { RESIZE_NEAREST = 0, RESIZE_BILINEAR_AA = 1, RESIZE_BICUBIC_AA = 2 };
constexpr int RESIZE_MAX_TAPS = 96;  // per axis: covers scale <= 23 (bicubic) / 47 (bilinear)

__device__ __forceinline__ float aa_filter(int mode, float x) {
  x = fabsf(x);
  if (mode == RESIZE_BILINEAR_AA) return x < 1.0f ? 1.0f - x : 0.0f;
  const float a = -0.5f;
  if (x < 1.0f) return ((a + 2.0f) * x - (a + 3.0f)) * x * x + 1.0f;
  if (x < 2.0f) return (((x - 5.0f) * x + 8.0f) * x - 4.0f) * a;
  return 0.0f;
}

// taps of output index i along an axis: returns (first, count), weights (normalised) into w[]
__device__ __forceinline__ void aa_taps(int mode, int i, int in_len, float scale, int& first, int& count, float* w) {
  const float interp_half = mode == RESIZE_BILINEAR_AA ? 1.0f : 2.0f;
  const float support = scale >= 1.0f ? interp_half * scale : interp_half;
  const float invscale = scale >= 1.0f ? 1.0f / scale : 1.0f;
  const float centre = scale * ((float)i + 0.5f);
  first = max((int)(long long)(centre - support + 0.5f), 0);
  count = min((int)(long long)(centre + support + 0.5f), in_len) - first;
  count = min(count, RESIZE_MAX_TAPS);
  float total = 0.0f;
  for (int j = 0; j < count; ++j) {
    w[j] = aa_filter(mode, ((float)(j + first) - centre + 0.5f) * invscale);
    total += w[j];
  }
  if (total != 0.0f)
    for (int j = 0; j < count; ++j) w[j] /= total;
}

// grid (blocks over OH * OW, planes): one thread per output element; its row of horizontal weights lives in LDS per block column set
__global__ __launch_bounds__(256) void resize_kernel(const float* __restrict__ x, float* __restrict__ out, int planes, int IH, int IW, int OH, int OW,
                                                     int mode) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= OH * OW) return;
  const int oy = o / OW, ox = o - oy * OW;
  const float sh = (float)IH / (float)OH, sw = (float)IW / (float)OW;
  for (int p = blockIdx.y; p < planes; p += gridDim.y) {
    const float* src = x + (long long)p * IH * IW;
    float* dst = out + (long long)p * OH * OW;
    if (mode == RESIZE_NEAREST) {
      const int sy = min((int)floorf((float)oy * sh), IH - 1), sx = min((int)floorf((float)ox * sw), IW - 1);
      dst[o] = src[(long long)sy * IW + sx];
      continue;
    }
    float wx[RESIZE_MAX_TAPS], wy[RESIZE_MAX_TAPS];
    int x0, nx, y0, ny;
    aa_taps(mode, ox, IW, sw, x0, nx, wx);
    aa_taps(mode, oy, IH, sh, y0, ny, wy);
    float acc = 0.0f;
    for (int j = 0; j < ny; ++j) {
      const float* row = src + (long long)(y0 + j) * IW + x0;
      float h = row[0] * wx[0];  // ATen's horizontal pass: t = src[0] w[0]; t += src[k] w[k]
      for (int k = 1; k < nx; ++k) h += row[k] * wx[k];
      acc = j == 0 ? h * wy[0] : acc + h * wy[j];
    }
    dst[o] = acc;
  }
}

int launch_resize(const float* x, float* out, int planes, int IH, int IW, int OH, int OW, int mode, hipStream_t s) {
  DRM_REQUIRE(x && out && planes > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, "resize: shape");
  DRM_REQUIRE(mode >= RESIZE_NEAREST && mode <= RESIZE_BICUBIC_AA, "resize: mode must be DRM_RESIZE_NEAREST / _BILINEAR_AA / _BICUBIC_AA");
  const float half = mode == RESIZE_BICUBIC_AA ? 2.0f : 1.0f;
  const float smax = fmaxf(fmaxf((float)IH / OH, (float)IW / OW), 1.0f);
  DRM_REQUIRE(mode == RESIZE_NEAREST || 2.0f * half * smax + 2.0f <= (float)RESIZE_MAX_TAPS, "resize: down-scaling factor beyond the kernel's tap budget");
  hipLaunchKernelGGL(resize_kernel, dim3((OH * OW + 255) / 256, std::min(planes, 4096)), dim3(256), 0, s, x, out, planes, IH, IW, OH, OW, mode);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

}  // namespace drm
