// GroupNorm32 finalisation of ONE image by the first 256 threads of a workgroup (reference: ldm/modules/diffusionmodules/util.py:199-216 --
// torch.nn.GroupNorm, 32 groups, eps 1e-5, biased variance): per-channel (sum, sum of squares) tables of up to two concatenated sources ->
// scale = rstd * gamma, shift = beta - mean * rstd * gamma per (image, channel), and optionally the power-of-two range-guard tables of the same
// tensor for a split-precision conv that reads it un-normalised (gn.hip act_pow2_scale_kernel).
// Shared by gn_finalize_kernel (gn.hip: one launch per GroupNorm) and by the prologue of conv_split2_kernel (GnFold: sparse launches, where a
// 5 us launch per GroupNorm is a tenth of the batch-1 step, finalise their own input tables -- every workgroup writes the same values).
#pragma once
#include <hip/hip_runtime.h>

namespace drm {

struct GnFold {
  const double2* mom0 = nullptr;  // [N][C0] (sum, sum of squares) or means; null = tables are finalised by their own launch
  const double2* mom1 = nullptr;  // [N][C1] second source of the concat, or null
  int C0 = 0, C1 = 0;
  double inv0 = 1.0, inv1 = 1.0;  // factor turning a table into per-pixel means
  double cnt0 = 0.0, cnt1 = 0.0;  // pixels per entry when a table holds means (0 = raw sums): the guard bound needs sums of squares
  const float* gamma = nullptr;
  const float* beta = nullptr;
  float* scale = nullptr;  // [N][C0 + C1] products
  float* shift = nullptr;
  float* guard_scale = nullptr;  // optional [N][C0 + C1] (2^k, 0) tables + guard_inv[N] = 2^-k
  float* guard_shift = nullptr;
  float* guard_inv = nullptr;
};

// workgroup barrier that orders LDS traffic only (a conv prologue has activation loads and table stores in flight that it does not wait for here)
__device__ __forceinline__ void gn_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Every thread of the workgroup calls this (barriers inside); threads t >= 256 only take part in the barriers and the table writes.
// lds: 72 floats of scratch, 8-byte aligned (32 means, 32 rstd, 4 doubles for the guard fold).
// lds_tab: optional [2][C0 + C1] copy of (scale, shift) in LDS -- the caller's first chunk reads it there instead of waiting for the global
// tables to be written and read back (conv prologue, one image).
constexpr int GN_FOLD_PRELOAD = 6;  // gamma / beta values a thread requests ahead of the statistics (C <= 6 * 256; beyond: after them)
__device__ __forceinline__ void gn_finalize_image(const GnFold& f, int n, int t, int nthreads, float* lds, float* lds_tab = nullptr) {
  float* g_mean = lds;
  float* g_rstd = lds + 32;
  const int C0 = f.C0, C1 = f.C1, C = C0 + C1, cpg = C / 32;
  // gamma / beta do not depend on the statistics: request them first, their round trip overlaps the moment loads and the fp64 math
  float gm[GN_FOLD_PRELOAD], bt[GN_FOLD_PRELOAD];
  const bool pre = C <= GN_FOLD_PRELOAD * nthreads;
  if (pre) {
#pragma unroll
    for (int k = 0; k < GN_FOLD_PRELOAD; ++k) {
      const int c = min(t + k * nthreads, C - 1);
      gm[k] = f.gamma[c];
      bt[k] = f.beta[c];
    }
  }
  if (f.guard_scale) {
    double* red = reinterpret_cast<double*>(lds + 64);  // 4 doubles
    double m = 0.0;
    if (t < 256) {
      for (int c = t; c < C; c += 256)
        m = fmax(m, c < C0 ? f.mom0[(size_t)n * C0 + c].y * (f.cnt0 > 0 ? f.cnt0 : 1.0) : f.mom1[(size_t)n * C1 + (c - C0)].y * (f.cnt1 > 0 ? f.cnt1 : 1.0));
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
      if ((t & 63) == 0) red[t >> 6] = m;
    }
    gn_lds_barrier();
    const double bound = sqrt(fmax(fmax(red[0], red[1]), fmax(red[2], red[3])));
    int k = 0;
    if (bound > 0.0 && bound < INFINITY) {
      int e;
      frexp(bound, &e);
      k = 15 - e;
      k = k > 90 ? 90 : (k < -90 ? -90 : k);
    }
    const float gs = ldexpf(1.0f, k);
    if (t == 0) f.guard_inv[n] = ldexpf(1.0f, -k);
    for (int c = t; c < C; c += nthreads) {
      f.guard_scale[(size_t)n * C + c] = gs;
      f.guard_shift[(size_t)n * C + c] = 0.f;
    }
  }
  if (t < 256) {
    // group sums in a fixed order, eight lanes per group (the 32 groups x 8 = 256 threads): lane j of a group takes its channels j, j + 8, ... and
    // the eight partial sums meet in three shuffle steps
    const int g = t >> 3, j = t & 7;
    double m = 0.0, q = 0.0;
    for (int k = j; k < cpg; k += 8) {
      const int c = g * cpg + k;
      if (c < C0) {
        const double2 v = f.mom0[(size_t)n * C0 + c];
        m += v.x * f.inv0;
        q += v.y * f.inv0;
      } else {
        const double2 v = f.mom1[(size_t)n * C1 + (c - C0)];
        m += v.x * f.inv1;
        q += v.y * f.inv1;
      }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
      m += __shfl_xor(m, o);
      q += __shfl_xor(q, o);
    }
    if (j == 0) {
      m /= cpg;
      q /= cpg;
      double var = q - m * m;
      if (var < 0.0) var = 0.0;
      g_mean[g] = (float)m;
      g_rstd[g] = (float)(1.0 / sqrt(var + 1e-5));
    }
  }
  gn_lds_barrier();
  if (pre) {
#pragma unroll
    for (int k = 0; k < GN_FOLD_PRELOAD; ++k) {
      const int c = t + k * nthreads;
      if (c < C) {
        const int g = c / cpg;
        const float sc = g_rstd[g] * gm[k], sh = bt[k] - g_mean[g] * sc;
        f.scale[(size_t)n * C + c] = sc;
        f.shift[(size_t)n * C + c] = sh;
        if (lds_tab) {
          lds_tab[c] = sc;
          lds_tab[C + c] = sh;
        }
      }
    }
  } else {
    for (int c = t; c < C; c += nthreads) {
      const int g = c / cpg;
      const float sc = g_rstd[g] * f.gamma[c], sh = f.beta[c] - g_mean[g] * sc;
      f.scale[(size_t)n * C + c] = sc;
      f.shift[(size_t)n * C + c] = sh;
      if (lds_tab) {
        lds_tab[c] = sc;
        lds_tab[C + c] = sh;
      }
    }
  }
  gn_lds_barrier();  // (the scratch is reused by the next image; lds_tab is complete)
}

}  // namespace drm
