// Single-head spatial self-attention core (reference: QKVAttentionLegacy.forward, openaimodel.py:365-381,
// with n_heads = 1): given qkv [N][T][3C] (q | k | v channel thirds, channels contiguous -- the NHWC view of
// the reference's [N, 3C, T]),
//     S = (q * C^-1/4) . (k * C^-1/4)^T   -> fp32 softmax over keys -> out = P . v        out: [N][T][C]
// Round-1 structure: two batched MFMA GEMMs (fp32 v_mfma_f32_32x32x2_f32, 64x64 tiles) around a row softmax,
// with the [N][T][T] score matrix in a workspace.  Attention is <= 4 % (IllNet) / 9 % (ObsNet) of the FLOPs
// (SURVEY.md 8a), so the fused flash-style kernel is a later-round item; the GroupNorm, qkv and proj_out
// projections and the residual add run in conv.hip (taps = 1).
#include "common.h"
#include "profiler.h"

namespace drm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// C[b][m][n] = alpha * sum_k A[b][m][k] * B[b](k, n)
//   A: row-major [M][K], leading dim lda.
//   BT = true : B given as [Ncols][K] row-major (ldb)  -> C = A . B^T   (Q . K^T)
//   BT = false: B given as [K][Ncols] row-major (ldb)  -> C = A . B     (P . V)
// 64x64 tile, 4 waves (2x2), one 32x32 accumulator per wave, K chunk 32.
template <bool BT>
__global__ __launch_bounds__(256) void bgemm64_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ Cm, int M,
                                                      int Ncols, int K, int lda, int ldb, int ldc, long long sA, long long sB, long long sC,
                                                      float alpha) {
  constexpr int KG = 8, LD = 65;  // LD: padded row stride (float4 units) to spread LDS banks on the staging writes
  __shared__ float4 As[KG * LD];
  __shared__ float4 Bs[KG * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  A += (size_t)blockIdx.z * sA;
  B += (size_t)blockIdx.z * sB;
  Cm += (size_t)blockIdx.z * sC;

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;

  for (int k0 = 0; k0 < K; k0 += 32) {
    // stage A: 64 rows x 8 quads
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = tid + 256 * j;
      const int row = idx >> 3, g = idx & 7;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int k = k0 + 4 * g;
      if (m0 + row < M && k < K) {
        const float* p = A + (size_t)(m0 + row) * lda + k;
        if (k + 3 < K) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          v.x = p[0];
          if (k + 1 < K) v.y = p[1];
          if (k + 2 < K) v.z = p[2];
        }
      }
      As[g * LD + row] = v;
    }
    if (BT) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int idx = tid + 256 * j;
        const int col = idx >> 3, g = idx & 7;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int k = k0 + 4 * g;
        if (n0 + col < Ncols && k < K) {
          const float* p = B + (size_t)(n0 + col) * ldb + k;
          if (k + 3 < K) {
            v = *reinterpret_cast<const float4*>(p);
          } else {
            v.x = p[0];
            if (k + 1 < K) v.y = p[1];
            if (k + 2 < K) v.z = p[2];
          }
        }
        Bs[g * LD + col] = v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int idx = tid + 256 * j;
        const int g = idx >> 6, col = idx & 63;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int k = k0 + 4 * g;
        if (n0 + col < Ncols) {
          const float* p = B + (size_t)k * ldb + n0 + col;
          if (k < K) v.x = p[0];
          if (k + 1 < K) v.y = p[ldb];
          if (k + 2 < K) v.z = p[2 * (size_t)ldb];
          if (k + 3 < K) v.w = p[3 * (size_t)ldb];
        }
        Bs[g * LD + col] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KG / 2; ++j) {
      const int gi = 2 * j + h;
      const float4 af = As[gi * LD + wm * 32 + r];
      const float4 bf = Bs[gi * LD + wn * 32 + r];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int col = n0 + wn * 32 + r;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
    if (row < M && col < Ncols) Cm[(size_t)row * ldc + col] = alpha * acc[e];
  }
}

// Split-precision form of the same batched GEMM (see conv_split.hip): both operands are split into fp16 hi + lo while
// they are staged, each 32x32x16 product is hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation
// (~2^-22 relative per product).  A is multiplied by a_scale (a power of two) before the split -- softmax probabilities
// would otherwise sit in fp16's subnormal range -- and alpha carries the inverse.  Needs K % 32 == 0.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
union AF4H8 {
  float4 f4;
  f16x8 h8;
};
__device__ __forceinline__ void split8(const float (&v)[8], float scale, float4& hi, float4& lo) {
  AF4H8 h, l;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float c = __builtin_amdgcn_fmed3f(v[k] * scale, -65504.0f, 65504.0f);
    const _Float16 hh = (_Float16)c;
    h.h8[k] = hh;
    l.h8[k] = (_Float16)(c - (float)hh);
  }
  hi = h.f4;
  lo = l.f4;
}

template <bool BT, int TERMS>
__global__ __launch_bounds__(256) void bgemm64s_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ Cm, int M,
                                                       int Ncols, int K, int lda, int ldb, int ldc, long long sA, long long sB, long long sC,
                                                       float alpha, float a_scale) {
  __shared__ float4 As[2 * 4 * 64];  // [hl][octet of the 32-wide K chunk][row]
  __shared__ float4 Bs[2 * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  A += (size_t)blockIdx.z * sA;
  B += (size_t)blockIdx.z * sB;
  Cm += (size_t)blockIdx.z * sC;

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;

  for (int k0 = 0; k0 < K; k0 += 32) {
    {  // A: one (row, octet) per thread, 32 contiguous bytes
      const int row = tid >> 2, oct = tid & 3;
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (m0 + row < M) {
        const float4* p = reinterpret_cast<const float4*>(A + (size_t)(m0 + row) * lda + k0 + 8 * oct);
        const float4 x = p[0], y = p[1];
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
      }
      split8(v, a_scale, As[oct * 64 + row], As[(4 + oct) * 64 + row]);
    }
    if (BT) {
      const int col = tid >> 2, oct = tid & 3;
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (n0 + col < Ncols) {
        const float4* p = reinterpret_cast<const float4*>(B + (size_t)(n0 + col) * ldb + k0 + 8 * oct);
        const float4 x = p[0], y = p[1];
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
      }
      split8(v, 1.0f, Bs[oct * 64 + col], Bs[(4 + oct) * 64 + col]);
    } else {
      const int oct = tid >> 6, col = tid & 63;
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (n0 + col < Ncols) {
        const float* p = B + (size_t)(k0 + 8 * oct) * ldb + n0 + col;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[(size_t)j * ldb];
      }
      split8(v, 1.0f, Bs[oct * 64 + col], Bs[(4 + oct) * 64 + col]);
    }
    __syncthreads();
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int oct = 2 * s2 + h;
      AF4H8 ah, al, bh, bl;
      ah.f4 = As[oct * 64 + wm * 32 + r];
      al.f4 = As[(4 + oct) * 64 + wm * 32 + r];
      bh.f4 = Bs[oct * 64 + wn * 32 + r];
      bl.f4 = Bs[(4 + oct) * 64 + wn * 32 + r];
      if (TERMS == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.h8, bh.h8, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h8, bl.h8, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h8, bh.h8, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int col = n0 + wn * 32 + r;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
    if (row < M && col < Ncols) Cm[(size_t)row * ldc + col] = alpha * acc[e];
  }
}

// in-place softmax over the last axis; one wave per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ S, long long rows, int T) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float* p = S + row * T;
  float m = -INFINITY;
  for (int i = lane; i < T; i += 64) m = fmaxf(m, p[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float sum = 0.f;
  for (int i = lane; i < T; i += 64) {
    const float e = expf(p[i] - m);
    p[i] = e;
    sum += e;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = 1.0f / sum;
  for (int i = lane; i < T; i += 64) p[i] *= inv;
}

int launch_attention(const float* qkv, float* scores, float* out, int N, int T, int C, hipStream_t s, int terms) {
  bool split = terms != 0;
  DRM_REQUIRE(C % 4 == 0 && T > 0 && N > 0, "attention shape");
  const float alpha = 1.0f / sqrtf((float)C);  // (C^-1/4)^2, applied once to the dot product
  const int tb = (T + 63) / 64;
  prof_tag(N, T, 1, C, C);
  ProfScope ps(PROF_ATTN, 4.0 * N * (double)T * T * C, 4.0 * N * ((double)T * 4 * C + 4.0 * T * T), s);
  split = split && (C % 32 == 0) && (T % 32 == 0);
  if (split && terms == 1)
    hipLaunchKernelGGL((bgemm64s_kernel<true, 1>), dim3(tb, tb, N), dim3(256), 0, s, qkv, qkv + C, scores, T, T, C, 3 * C, 3 * C, T,
                       (long long)T * 3 * C, (long long)T * 3 * C, (long long)T * T, alpha, 1.0f);
  else if (split)
    hipLaunchKernelGGL((bgemm64s_kernel<true, 3>), dim3(tb, tb, N), dim3(256), 0, s, qkv, qkv + C, scores, T, T, C, 3 * C, 3 * C, T,
                       (long long)T * 3 * C, (long long)T * 3 * C, (long long)T * T, alpha, 1.0f);
  else
    hipLaunchKernelGGL(bgemm64_kernel<true>, dim3(tb, tb, N), dim3(256), 0, s, qkv, qkv + C, scores, T, T, C, 3 * C, 3 * C, T,
                       (long long)T * 3 * C, (long long)T * 3 * C, (long long)T * T, alpha);
  DRM_HIP_CHECK(hipGetLastError());
  const long long rows = (long long)N * T;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, scores, rows, T);
  DRM_HIP_CHECK(hipGetLastError());
  if (split && terms == 1)  // probabilities are scaled by 2^12 before the fp16 conversion (largest 4096, smallest normal 2^-26)
    hipLaunchKernelGGL((bgemm64s_kernel<false, 1>), dim3((C + 63) / 64, tb, N), dim3(256), 0, s, scores, qkv + 2 * C, out, T, C, T, T, 3 * C, C,
                       (long long)T * T, (long long)T * 3 * C, (long long)T * C, 1.0f / 4096.0f, 4096.0f);
  else if (split)
    hipLaunchKernelGGL((bgemm64s_kernel<false, 3>), dim3((C + 63) / 64, tb, N), dim3(256), 0, s, scores, qkv + 2 * C, out, T, C, T, T, 3 * C, C,
                       (long long)T * T, (long long)T * 3 * C, (long long)T * C, 1.0f / 4096.0f, 4096.0f);
  else
    hipLaunchKernelGGL(bgemm64_kernel<false>, dim3((C + 63) / 64, tb, N), dim3(256), 0, s, scores, qkv + 2 * C, out, T, C, T, T, 3 * C, C,
                       (long long)T * T, (long long)T * 3 * C, (long long)T * C, 1.0f);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

}  // namespace drm
