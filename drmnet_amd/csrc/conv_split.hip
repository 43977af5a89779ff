// Split-precision arithmetic of the fused GroupNorm+SiLU+conv implicit GEMM -- entry point + weight pre-split (the kernel: conv_split2.hip):
// fp32-accurate products on the gfx950 f16 matrix cores.
//
// Every fp32 operand x is split as x = hi + lo with hi = fp16(x), lo = fp16(x - hi) (22 significant bits together),
// and a product is evaluated as hi_a*hi_b + hi_a*lo_b + lo_a*hi_b with three v_mfma_f32_32x32x16_f16 accumulating
// into the same fp32 accumulator (the dropped lo*lo term is 2^-22 relative, the level of fp32 rounding itself).
// Cost per 16-deep K slab: 3 x 32 cycles instead of 8 x 64 cycles of v_mfma_f32_32x32x2_f32  =>  16/3 x the fp32
// matrix rate, at unchanged HBM traffic (activations stay fp32 in HBM; the split happens once per staged element, in
// registers, right after the GroupNorm affine + SiLU).
//   * activations: values a GroupNorm has just normalised are O(1) and are split as they are; an UN-normalised input (skip_connection,
//     proj_out, stem, attention q / k / v) is first multiplied by a per-image power of two chosen from a rigorous bound of max |x|
//     (engine.hip raw_input_guard, gn.hip act_pow2_scale_kernel), so that nothing reaches fp16's limit and hi AND lo stay in the normal
//     range; the epilogue multiplies by the inverse power of two (exact).  A clamp to +-65504 in front of the split is the last resort.
//   * weights: pre-split at load time after an exact power-of-two scaling that moves max|w| to [8192, 16384), so the lo
//     halves stay in the fp16 normal range; the epilogue multiplies by the inverse scale (exact).
// LDS images: A = [hi|lo][slab s][lane-half h][pixel][8 halfs], B = [hi|lo][s][h][cout][8 halfs]; one ds_read_b128 per
// operand fragment (lane (r,h) of slab s needs channels 16s + 8h .. +7 of row/col r).
// This file holds the dispatcher and the weight pre-split / packing kernels; the conv kernel is conv_split2.hip.
#include "common.h"
#include "profiler.h"

namespace drm {

int launch_conv_split2(const ConvArgs& a, hipStream_t s);

// the split kernels accumulate ConvArgs::stat_out (GroupNorm statistics of their output) in the epilogue
bool conv_split_fuses_stats() { return true; }

int launch_conv_split(const ConvArgs& a, hipStream_t s) {
  const int Ctot = a.C0 + a.C1;
  DRM_REQUIRE(a.taps == 9 || a.taps == 1, "conv taps must be 9 or 1");
  DRM_REQUIRE(a.Cout % 32 == 0 && Ctot % 32 == 0 && a.C0 % 32 == 0, "split conv needs channels % 32 == 0");
  DRM_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0, "conv shape");
  DRM_REQUIRE(!a.up0 || (a.H % 2 == 0 && a.W % 2 == 0), "upsampled source needs even output size");
  return launch_conv_split2(a, s);  // LDS-DMA weight ring, 256-pixel tiles (conv_split2.hip)
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
                                              int Cin, int taps, int CoutP, int CinP, int bf16 /* hi plane as bf16 bits (DRM_PREC_BF16), lo plane zero */) {
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
    if (bf16) {
      const __bf16 b = (__bf16)v;
      p[i] = hl ? (_Float16)0.f : __builtin_bit_cast(_Float16, b);
      continue;
    }
    const _Float16 hi = (_Float16)v;
    p[i] = hl ? (_Float16)(v - (float)hi) : hi;
  }
}

// f16mx image (conv_split2.hip TERMS == 2): the hi planes as above; the four lo planes of every (tap, chunk) are replaced by the e4m3 images the
// block-scaled MFMA reads -- plane 4 + g: bh8 = e4m3(hi / 2^6), plane 6 + g: bl8 = e4m3(lo * 2^6) of the 16 channels of group g, one byte each
// (16 bytes per cout and plane, the same footprint).  |hi| < 2^14 and |lo| <= 4 by the pre-scaling, so both stay inside e4m3's +-448.
typedef short s16x2_t __attribute__((ext_vector_type(2)));
__global__ void pack_conv_weight_mx_kernel(const float* __restrict__ w, unsigned char* __restrict__ p, const float* __restrict__ scales, int Cout, int Cin,
                                           int taps, int CoutP, int CinP) {
  const int nchunks = CinP / 32;
  const size_t total = (size_t)taps * nchunks * 4 * CoutP * 16;  // bytes of the lo halves
  const float scale = scales[0];
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i % 16;
    size_t t = i / 16;
    const int co = t % CoutP; t /= CoutP;
    const int g = t % 2; t /= 2;
    const int blk = t % 2; t /= 2;
    const int q = t % nchunks;
    const int tap = t / nchunks;
    const int ci = 32 * q + 16 * g + j;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * taps + tap] * scale;
    const _Float16 hi = (_Float16)v;
    const float x = blk ? (v - (float)hi) : (float)hi;
    s16x2_t z = {0, 0};
    z = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, x, 0.f, blk ? (1.0f / 64.0f) : 64.0f, false);
    // byte address: ((tap * nchunks + q) * 8 + 4 + 2 * blk + g) planes of CoutP 16-byte entries
    p[((((size_t)tap * nchunks + q) * 8 + 4 + 2 * blk + g) * CoutP + co) * 16 + j] = (unsigned char)(z[0] & 0xff);
  }
}

size_t packed_conv_weight_split_floats(int taps, int CoutP, int CinP) {
  return (size_t)taps * CoutP * CinP;  // hi + lo fp16 = 4 bytes per weight, same footprint as fp32
}

int launch_pack_conv_weight_split(const float* w, float* packed, float* scales /*[2] device*/, unsigned* scratch /*1 uint device*/, int Cout,
                                  int Cin, int taps, int CoutP, int CinP, hipStream_t s, bool mx, bool bf16) {
  DRM_REQUIRE(CinP % 32 == 0 && CoutP >= Cout && CinP >= Cin, "pack_conv_weight_split padding");
  const size_t n = (size_t)Cout * Cin * taps;
  DRM_HIP_CHECK(hipMemsetAsync(scratch, 0, sizeof(unsigned), s));
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0, s, w, n, scratch);
  DRM_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(split_scale_kernel, dim3(1), dim3(1), 0, s, scratch, scales);
  DRM_HIP_CHECK(hipGetLastError());
  const size_t total = (size_t)taps * CoutP * CinP * 2;
  hipLaunchKernelGGL(pack_conv_weight_split_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, w,
                     reinterpret_cast<_Float16*>(packed), scales, Cout, Cin, taps, CoutP, CinP, bf16 ? 1 : 0);
  DRM_HIP_CHECK(hipGetLastError());
  if (mx) {  // (overwrites the lo planes the launch above filled)
    const size_t bytes = (size_t)taps * CoutP * CinP * 2;
    hipLaunchKernelGGL(pack_conv_weight_mx_kernel, dim3((unsigned)std::min<size_t>((bytes + 255) / 256, 4096)), dim3(256), 0, s, w,
                       reinterpret_cast<unsigned char*>(packed), scales, Cout, Cin, taps, CoutP, CinP);
    DRM_HIP_CHECK(hipGetLastError());
  }
  return DRM_OK;
}

}  // namespace drm
