// GroupNorm32 statistics for NHWC activations (reference: ldm/modules/diffusionmodules/util.py:199-216,
// torch.nn.GroupNorm with 32 groups, eps 1e-5, biased variance, fp32).
//
// The norm itself is never materialised: the consumer (conv.hip A-tile loader, misc.hip encoder head)
// applies  y = x * scale[n][c] + shift[n][c]  with scale = rstd*gamma, shift = beta - mean*rstd*gamma.
// Statistics are kept per CHANNEL ((mean, mean of squares) per (sample, channel)) so that
//   * a tensor's moments are computed once and reused by every GroupNorm that sees it (encoder output ->
//     next block AND the decoder concat);
//   * the concat of two tensors (openaimodel.py:762) and the nearest-x2 upsample (:116; replication leaves
//     per-channel moments unchanged) need no data pass at all: groups that straddle the concat boundary
//     are assembled from the two moment tables in gn_finalize.
// HBM-bound pass: reads the tensor once with 16-B/lane coalesced loads; wave/LDS tree reductions, partial
// sums per slab, final accumulation in double.
#include "common.h"
#include "gn_fold.h"
#include "profiler.h"

namespace drm {

int chan_moments_splits(int HW, int C) {
  int s = HW / 64;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return s;
}

// grid (splits, N), block (64, 4): x = channel quad within a 64-quad block, y = pixel lane
__global__ __launch_bounds__(256) void chan_moments_partial_kernel(const float* __restrict__ x, int HW, int C, int splits,
                                                                    double* __restrict__ partial) {
  __shared__ double red[2][4][64][4];
  const int s = blockIdx.x, n = blockIdx.y;
  const int qx = threadIdx.x, py = threadIdx.y;
  const int q4 = C >> 2;
  const int p0 = (int)(((long long)HW * s) / splits), p1 = (int)(((long long)HW * (s + 1)) / splits);
  const float4* xp = reinterpret_cast<const float4*>(x) + (size_t)n * HW * q4;
  for (int qb = 0; qb < q4; qb += 64) {
    const int q = qb + qx;
    // double accumulation: var = E[x^2] - E[x]^2 cancels badly in fp32 when |mean| >> std
    double sum[4] = {0.0, 0.0, 0.0, 0.0}, sq[4] = {0.0, 0.0, 0.0, 0.0};
    if (q < q4) {
      for (int p = p0 + py; p < p1; p += 4) {
        const float4 v = xp[(size_t)p * q4 + q];
        sum[0] += v.x; sum[1] += v.y; sum[2] += v.z; sum[3] += v.w;
        sq[0] += (double)v.x * v.x; sq[1] += (double)v.y * v.y; sq[2] += (double)v.z * v.z; sq[3] += (double)v.w * v.w;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      red[0][py][qx][k] = sum[k];
      red[1][py][qx][k] = sq[k];
    }
    __syncthreads();
    if (py == 0 && q < q4) {
      double* o = partial + (((size_t)n * splits + s) * C + 4 * q) * 2;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o[2 * k] = red[0][0][qx][k] + red[0][1][qx][k] + red[0][2][qx][k] + red[0][3][qx][k];
        o[2 * k + 1] = red[1][0][qx][k] + red[1][1][qx][k] + red[1][2][qx][k] + red[1][3][qx][k];
      }
    }
    __syncthreads();
  }
}

__global__ void chan_moments_reduce_kernel(const double* __restrict__ partial, int N, int C, int splits, double inv_hw, double2* __restrict__ mom) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i % C;
  double s = 0.0, q = 0.0;
  for (int k = 0; k < splits; ++k) {
    const double* p = partial + (((size_t)n * splits + k) * C + c) * 2;
    s += p[0];
    q += p[1];
  }
  mom[i] = make_double2(s * inv_hw, q * inv_hw);
}

int launch_chan_moments(const float* x, int N, int HW, int C, double* partial, double2* mom, hipStream_t s) {
  DRM_REQUIRE(C % 4 == 0, "chan_moments: C % 4");
  const int splits = chan_moments_splits(HW, C);
  prof_tag(N, HW, 1, C, C);
  ProfScope ps(PROF_GNSTATS, 3.0 * N * (double)HW * C, 4.0 * N * (double)HW * C, s);
  hipLaunchKernelGGL(chan_moments_partial_kernel, dim3(splits, N), dim3(64, 4), 0, s, x, HW, C, splits, partial);
  DRM_HIP_CHECK(hipGetLastError());
  const int total = N * C;
  hipLaunchKernelGGL(chan_moments_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, partial, N, C, splits, 1.0 / (double)HW, mom);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

// grid N, block 256.  32 groups over the concatenated channel axis [C0 | C1].
// Optional second product (guard_scale != null): the per-image power-of-two staging tables of the SAME concatenated tensor for a
// split-precision conv that reads it un-normalised (the ResBlock's skip_connection) -- see act_pow2_scale_kernel; cnt0 / cnt1 =
// pixels per table entry when a table holds means (0 = raw sums).
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double2* __restrict__ mom0, int C0, double inv0,
                                                           const double2* __restrict__ mom1, int C1, double inv1,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ scale, float* __restrict__ shift, double cnt0, double cnt1,
                                                           float* __restrict__ guard_scale, float* __restrict__ guard_shift,
                                                           float* __restrict__ guard_inv) {
  __shared__ __attribute__((aligned(16))) float scratch[72];
  GnFold f;
  f.mom0 = mom0; f.mom1 = mom1; f.C0 = C0; f.C1 = C1; f.inv0 = inv0; f.inv1 = inv1; f.cnt0 = cnt0; f.cnt1 = cnt1;
  f.gamma = gamma; f.beta = beta; f.scale = scale; f.shift = shift;
  f.guard_scale = guard_scale; f.guard_shift = guard_shift; f.guard_inv = guard_inv;
  gn_finalize_image(f, blockIdx.x, threadIdx.x, 256, scratch);  // (gn_fold.h: the same code runs in the prologue of sparse conv launches)
}

// Range guard of the split-precision path for UN-normalised conv inputs (ResBlock skip_connection, AttentionBlock proj_out, the
// stem: openaimodel.py:241, :314, :534).  The fp16 hi/lo split keeps 22 bits only while the staged values sit inside fp16's
// normal range, so such a tensor is staged through an exact power-of-two factor per image, undone in the epilogue:
//   bound[n] = a rigorous upper bound of max |x| over image n  --  sqrt(max_c sum_pixels x^2), from the per-channel (sum, sum of
//              squares) tables the GroupNorm fusion already keeps, or an explicit absmax word;
//   2^k[n]   = the largest power of two with bound * 2^k <= 2^15  (no element can reach the fp16 limit 65504; a typical element,
//              rms ~ bound / sqrt(H W), lands around 2^15 / sqrt(H W) >= 2^7.5: hi AND lo stay normal fp16 numbers).
// Written as the (scale, shift) = (2^k, 0) table the staging path already applies (an exact multiply) plus inv[n] = 2^-k.
// grid N, block 256.  cnt: pixels per table entry when the table holds means instead of raw sums (0 = raw sums).
__global__ __launch_bounds__(256) void act_pow2_scale_kernel(const double2* __restrict__ mom0, int C0, int lo0, int hi0, double cnt0,
                                                             const double2* __restrict__ mom1, int C1, double cnt1,
                                                             const unsigned* __restrict__ absmax_bits, int absmax_parts, int Ctab,
                                                             float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ inv) {
  __shared__ double red[4];
  __shared__ float s_scale;
  const int n = blockIdx.x, t = threadIdx.x;
  double m = 0.0;
  if (mom0)
    for (int c = lo0 + t; c < hi0; c += 256) m = fmax(m, mom0[(size_t)n * C0 + c].y * (cnt0 > 0 ? cnt0 : 1.0));
  if (mom1)
    for (int c = t; c < C1; c += 256) m = fmax(m, mom1[(size_t)n * C1 + c].y * (cnt1 > 0 ? cnt1 : 1.0));
  if (absmax_bits)  // explicit bounds (one word per producer block): squared, so they fold with the sum-of-squares bounds
    for (int i = t; i < absmax_parts; i += 256) {
      const double v = (double)__uint_as_float(absmax_bits[(size_t)n * absmax_parts + i]);
      m = fmax(m, v * v);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((t & 63) == 0) red[t >> 6] = m;
  __syncthreads();
  if (t == 0) {
    double bound = sqrt(fmax(fmax(red[0], red[1]), fmax(red[2], red[3])));
    int k = 0;
    if (bound > 0.0 && bound < INFINITY) {
      int e;
      frexp(bound, &e);  // bound = f * 2^e, f in [0.5, 1)
      k = 15 - e;        // bound * 2^k in [2^14, 2^15)
      k = k > 90 ? 90 : (k < -90 ? -90 : k);  // the epilogue multiplies 2^-k by the weights' 2^-k' (|k'| <= 30) in fp32
    }
    s_scale = ldexpf(1.0f, k);
    inv[n] = ldexpf(1.0f, -k);
  }
  __syncthreads();
  const float sc = s_scale;
  for (int c = t; c < Ctab; c += 256) {
    scale[(size_t)n * Ctab + c] = sc;
    shift[(size_t)n * Ctab + c] = 0.f;
  }
}

int launch_act_pow2_scale(const double2* mom0, int C0, int lo0, int hi0, double cnt0, const double2* mom1, int C1, double cnt1,
                          const unsigned* absmax_bits, int Ctab, int N, float* scale, float* shift, float* inv, hipStream_t s, int absmax_parts) {
  DRM_REQUIRE((mom0 || absmax_bits) && scale && shift && inv && N > 0 && Ctab > 0, "act_pow2_scale: arguments");
  hipLaunchKernelGGL(act_pow2_scale_kernel, dim3(N), dim3(256), 0, s, mom0, C0, lo0, hi0, cnt0, mom1, C1, cnt1, absmax_bits, absmax_parts, Ctab, scale,
                     shift, inv);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

int launch_gn_finalize(const double2* mom0, int C0, double inv0, const double2* mom1, int C1, double inv1, const float* gamma,
                       const float* beta, int N, float* scale, float* shift, hipStream_t s, double cnt0, double cnt1, float* guard_scale,
                       float* guard_shift, float* guard_inv) {
  DRM_REQUIRE((C0 + C1) % 32 == 0, "GroupNorm32 needs channels % 32 == 0");
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(N), dim3(256), 0, s, mom0, C0, inv0, mom1, C1, inv1, gamma, beta, scale, shift, cnt0, cnt1, guard_scale,
                     guard_shift, guard_inv);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

}  // namespace drm
