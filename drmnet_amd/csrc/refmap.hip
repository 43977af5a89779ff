// Object image -> reflectance map gather, the step in front of the samplers.
//
// Reference: refmap_mask_make (utils/img2refmap.py:6-37) with xyz2thetaphi(normal = [0,1,0], tangent = [-1,0,0])
// (utils/transform.py:55-89), i.e. theta = acos(n.y), phi = atan2(n.z, -n.x); mask erosion scripts/estimate.py:43-50.
// The reference builds, for 32 batches of 512 texels, a [512, n_pixels] L-inf angle matrix and takes a nanmedian over
// each row: O(res^2 * n).  Here every pixel is binned once into the (few) texels whose predicate it passes -- the predicate
// itself, |centre - angle| > threshold in fp32 with the reference's centre = (i + 0.5f) * fl32(pi / res), is evaluated
// verbatim on the neighbouring rows / columns -- and one wave per texel selects the lower median of the colour sums by
// rank counting.  HBM-bound: 24 B read per pixel, a few atomics, 13 B written per texel.
//   count -> exclusive scan -> fill -> select      (all on the caller's stream; one host read-back validates capacity)
#include <algorithm>
#include <cmath>

#include "common.h"

namespace drm {

struct PixRec {
  float theta, phi, s;
};

__device__ __forceinline__ bool inside(float centre, float angle, float thr) { return !(fabsf(centre - angle) > thr); }  // NaN => inside

// rows / columns a pixel can belong to: the reference's predicate on a window around the nearest texel (all of them for NaN)
__device__ __forceinline__ void cand_range(float angle, float step, float thr, int res, int& lo, int& hi) {
  if (isnan(angle)) {
    lo = 0;
    hi = res - 1;
    return;
  }
  const int c = (int)floorf(angle / step);
  const int w = (int)ceilf(thr / step) + 1;
  lo = max(0, min(res - 1, c - w));
  hi = max(0, min(res - 1, c + w));
  if (c + w < 0 || c - w > res - 1) {  // far outside the grid: no candidates
    lo = 1;
    hi = 0;
  }
}

template <bool FILL>
__global__ void refmap_bin_kernel(const float* __restrict__ colors, const float* __restrict__ normals, long long n, int C, int res, float thr,
                                  PixRec* __restrict__ rec, int* __restrict__ counts, const int* __restrict__ offsets, int* __restrict__ cursor,
                                  int* __restrict__ list, long long capacity) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  PixRec r;
  if (!FILL) {
    const float nx = normals[p * 3 + 0], ny = normals[p * 3 + 1], nz = normals[p * 3 + 2];
    r.theta = (float)acos((double)ny);
    r.phi = (float)atan2((double)nz, -(double)nx);
    float s = colors[p * C];
    for (int c = 1; c < C; ++c) s += colors[p * C + c];
    r.s = s;
    rec[p] = r;
  } else {
    r = rec[p];
  }
  const float step = (float)(M_PI / (double)res);
  int i0, i1, j0, j1;
  cand_range(r.theta, step, thr, res, i0, i1);
  cand_range(r.phi, step, thr, res, j0, j1);
  for (int i = i0; i <= i1; ++i) {
    if (!inside(((float)i + 0.5f) * step, r.theta, thr)) continue;
    for (int j = j0; j <= j1; ++j) {
      if (!inside(((float)j + 0.5f) * step, r.phi, thr)) continue;
      const int t = i * res + j;
      if (!FILL) {
        atomicAdd(&counts[t], 1);
      } else {
        const long long at = (long long)offsets[t] + atomicAdd(&cursor[t], 1);
        if (at < capacity) list[at] = (int)p;
      }
    }
  }
}

// exclusive scan of `counts` (T entries) by one workgroup; total -> offsets[T]
__global__ __launch_bounds__(1024) void refmap_scan_kernel(const int* __restrict__ counts, int* __restrict__ offsets, int T) {
  __shared__ long long part[1024];
  const int tid = threadIdx.x;
  const int per = (T + 1023) / 1024;
  const int b = tid * per, e = min(T, b + per);
  long long s = 0;
  for (int k = b; k < e; ++k) s += counts[k];
  part[tid] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const long long v = (tid >= o) ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  long long run = part[tid] - s;
  for (int k = b; k < e; ++k) {
    offsets[k] = (int)min(run, (long long)INT_MAX);
    run += counts[k];
  }
  if (tid == 1023) offsets[T] = (int)min(part[1023], (long long)INT_MAX);
}

// one wave per texel: lower median (torch.nanmedian) of the colour sums of its pixels, ties ordered by pixel index
__global__ __launch_bounds__(256) void refmap_select_kernel(const float* __restrict__ colors, const PixRec* __restrict__ rec,
                                                            const int* __restrict__ offsets, const int* __restrict__ list, int T, int C,
                                                            int min_points, float* __restrict__ refmap, unsigned char* __restrict__ refmask) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (t >= T) return;
  const int off = offsets[t], k = offsets[t + 1] - off;
  int pick = -1;
  if (k > 0 && k >= min_points) {
    int nvalid = 0;
    for (int b = lane; b < k; b += 64) nvalid += isnan(rec[list[off + b]].s) ? 0 : 1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nvalid += __shfl_xor(nvalid, o);
    if (nvalid > 0) {
      const int target = (nvalid - 1) / 2;
      for (int b = lane; b < k; b += 64) {
        const int p = list[off + b];
        const float s = rec[p].s;
        if (isnan(s)) continue;
        int rank = 0;
        for (int m = 0; m < k; ++m) {
          const int pm = list[off + m];
          const float sm = rec[pm].s;
          rank += (sm < s || (sm == s && pm < p)) ? 1 : 0;  // NaN compares false: not counted
        }
        if (rank == target) pick = p;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pick = max(pick, __shfl_xor(pick, o));
  if (lane < C) refmap[(size_t)t * C + lane] = (pick >= 0) ? colors[(size_t)pick * C + lane] : 0.f;
  if (lane == 0) refmask[t] = pick >= 0 ? 1 : 0;
}

__global__ void erode_mask_kernel(const unsigned char* __restrict__ mask, int H, int W, int k, unsigned char* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i % W;
  unsigned char keep = mask[i] ? 1 : 0;
  if (keep) {
    const int left = (k - 1) / 2;  // torch padding = "same"
    const float half = (float)k / 2.0f;
    for (int a = 0; a < k && keep; ++a)
      for (int b = 0; b < k; ++b) {
        const float da = (float)a + 0.5f - half, db = (float)b + 0.5f - half;
        if (!(sqrtf(da * da + db * db) <= half)) continue;
        const int yy = y + a - left, xx = x + b - left;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W && !mask[yy * W + xx]) {
          keep = 0;
          break;
        }
      }
  }
  out[i] = keep;
}

static size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

// texel memberships budgeted per pixel: the candidate window of cand_range squared (1 texel for the estimate.py setting of
// half a texel, up to 4 with border ties), capped at the whole map; NaN normals (member of every row / column) are not budgeted
static long long refmap_capacity(long long n, int res, float thr) {
  const double step = M_PI / (double)res;
  const double w = std::ceil((double)thr / step) + 1.0;
  const double side = std::min((double)res, 2.0 * w + 1.0);
  const double per = std::min((double)res * res, std::max(4.0, side * side));
  return (long long)(per * (double)n) + (long long)res * res;
}

size_t refmap_workspace_bytes(long long n, int res, float thr) {
  const size_t T = (size_t)res * res;
  return align256((size_t)n * sizeof(PixRec)) + 3 * align256((T + 1) * sizeof(int)) + align256((size_t)refmap_capacity(n, res, thr) * sizeof(int)) + 256;
}

int launch_refmap_mask_make(const float* colors, const float* normals, long long n, int C, int res, float thr, int min_points, float* refmap,
                            unsigned char* refmask, void* ws, size_t ws_bytes, hipStream_t s) {
  DRM_REQUIRE(n >= 0 && n < (1ll << 31) && res > 0 && res <= 4096 && C >= 1 && C <= 64, "refmap_mask_make shape");
  DRM_REQUIRE(thr >= 0.f && ws_bytes >= refmap_workspace_bytes(n, res, thr), "refmap_mask_make workspace too small");
  const int T = res * res;
  char* w = reinterpret_cast<char*>(ws);
  PixRec* rec = reinterpret_cast<PixRec*>(w);
  w += align256((size_t)n * sizeof(PixRec));
  int* counts = reinterpret_cast<int*>(w);
  w += align256((size_t)(T + 1) * sizeof(int));
  int* offsets = reinterpret_cast<int*>(w);
  w += align256((size_t)(T + 1) * sizeof(int));
  int* cursor = reinterpret_cast<int*>(w);
  w += align256((size_t)(T + 1) * sizeof(int));
  int* list = reinterpret_cast<int*>(w);
  const long long capacity = refmap_capacity(n, res, thr);
  DRM_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)(T + 1) * sizeof(int), s));
  DRM_HIP_CHECK(hipMemsetAsync(cursor, 0, (size_t)(T + 1) * sizeof(int), s));
  const unsigned pb = (unsigned)((n + 255) / 256);
  if (n > 0) {
    hipLaunchKernelGGL(refmap_bin_kernel<false>, dim3(pb), dim3(256), 0, s, colors, normals, n, C, res, thr, rec, counts, nullptr, nullptr, nullptr,
                       0ll);
    DRM_HIP_CHECK(hipGetLastError());
  }
  hipLaunchKernelGGL(refmap_scan_kernel, dim3(1), dim3(1024), 0, s, counts, offsets, T);
  DRM_HIP_CHECK(hipGetLastError());
  int total = 0;
  DRM_HIP_CHECK(hipMemcpyAsync(&total, offsets + T, sizeof(int), hipMemcpyDeviceToHost, s));
  DRM_HIP_CHECK(hipStreamSynchronize(s));
  DRM_REQUIRE((long long)total <= capacity, "refmap_mask_make: " + std::to_string(total) + " texel memberships exceed the workspace budget of " +
                                                std::to_string(capacity) + " (angle_threshold far above the texel size, or NaN normals)");
  if (n > 0) {
    hipLaunchKernelGGL(refmap_bin_kernel<true>, dim3(pb), dim3(256), 0, s, colors, normals, n, C, res, thr, rec, counts, offsets, cursor, list,
                       capacity);
    DRM_HIP_CHECK(hipGetLastError());
  }
  hipLaunchKernelGGL(refmap_select_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, s, colors, rec, offsets, list, T, C, min_points, refmap,
                     refmask);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

int launch_erode_mask(const unsigned char* mask, int H, int W, int k, unsigned char* out, hipStream_t s) {
  DRM_REQUIRE(H > 0 && W > 0 && k >= 0 && k <= 255, "erode_mask shape");
  if (k == 0) {
    DRM_HIP_CHECK(hipMemcpyAsync(out, mask, (size_t)H * W, hipMemcpyDeviceToDevice, s));
    return DRM_OK;
  }
  hipLaunchKernelGGL(erode_mask_kernel, dim3((unsigned)((H * W + 255) / 256)), dim3(256), 0, s, mask, H, W, k, out);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

}  // namespace drm
