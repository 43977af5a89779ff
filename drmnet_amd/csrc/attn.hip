// Single-head spatial self-attention core (reference: QKVAttentionLegacy.forward, openaimodel.py:365-381,
// with n_heads = 1): given qkv [N][T][3C] (q | k | v channel thirds, channels contiguous -- the NHWC view of
// the reference's [N, 3C, T]),
//     S = (q * C^-1/4) . (k * C^-1/4)^T   -> fp32 softmax over keys -> out = P . v        out: [N][T][C]
// Round-1 structure: two batched MFMA GEMMs (fp32 v_mfma_f32_32x32x2_f32, 64x64 tiles) around a row softmax,
// with the [N][T][T] score matrix in a workspace.  Attention is <= 4 % (IllNet) / 9 % (ObsNet) of the FLOPs
// (SURVEY.md 8a), so the fused flash-style kernel is a later-round item; the GroupNorm, qkv and proj_out
// projections and the residual add run in conv.hip (taps = 1).
#include <algorithm>

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
typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
union AF4H8 {
  float4 f4;
  f16x8 h8;
  abf16x8 b8;
};
// bf16 operands (DRM_PREC_BF16): round to nearest even, no range clamp (fp32's exponent); the lo image is unused
__device__ __forceinline__ void round8_bf16(const float (&v)[8], float scale, float4& hi, float4& lo) {
  AF4H8 h;
#pragma unroll
  for (int k = 0; k < 8; ++k) h.b8[k] = (__bf16)(v[k] * scale);
  hi = h.f4;
  lo = make_float4(0.f, 0.f, 0.f, 0.f);
}
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

// SM (with BT = false: O = softmax(S) v): A holds raw scores and the row softmax is applied while A is staged -- the workgroup first takes the
// maximum and the sum of exp of its 64 rows (four lanes per row, whole rows: K = T), then stages expf(s - max) / sum: the arithmetic of
// softmax_rows_kernel, one launch and one pass over the scores less per attention block.  (The short-sequence levels have since moved to
// pv_small_kernel, which applies the softmax the same way with the keys split over the waves of a workgroup; this form stays for callers of the
// 64x64-tile GEMM.)
template <bool BT, int TERMS, bool SM = false>
__global__ __launch_bounds__(256) void bgemm64s_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ Cm, int M,
                                                       int Ncols, int K, int lda, int ldb, int ldc, long long sA, long long sB, long long sC,
                                                       float alpha, float a_scale, const float* __restrict__ b_scale_img = nullptr,
                                                       const float* __restrict__ alpha_img = nullptr) {
  // b_scale_img / alpha_img: optional per-image (blockIdx.z) power-of-two staging factor of B and its inverse (the range guard of v: attn_scales_kernel)
  const float b_scale = b_scale_img ? b_scale_img[blockIdx.z] : 1.0f;
  if (alpha_img) alpha *= alpha_img[blockIdx.z];
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

  [[maybe_unused]] float row_max = 0.f, row_inv = 0.f;  // SM: of row tid >> 2 (the row this thread stages)
  if constexpr (SM) {
    const int row = tid >> 2, part = tid & 3;
    const bool in = m0 + row < M;
    const float4* p = reinterpret_cast<const float4*>(A + (size_t)min(m0 + row, M - 1) * lda);
    float m = -INFINITY;
    for (int q = part; q < K / 4; q += 4) {
      const float4 x = p[q];
      m = fmaxf(fmaxf(m, fmaxf(x.x, x.y)), fmaxf(x.z, x.w));
    }
    m = fmaxf(m, __shfl_xor(m, 1));
    m = fmaxf(m, __shfl_xor(m, 2));
    float sum = 0.f;
    for (int q = part; q < K / 4; q += 4) {
      const float4 x = p[q];
      sum += (expf(x.x - m) + expf(x.y - m)) + (expf(x.z - m) + expf(x.w - m));
    }
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    row_max = in ? m : 0.f;
    row_inv = in ? 1.0f / sum : 0.f;
  }

  for (int k0 = 0; k0 < K; k0 += 32) {
    {  // A: one (row, octet) per thread, 32 contiguous bytes
      const int row = tid >> 2, oct = tid & 3;
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (m0 + row < M) {
        const float4* p = reinterpret_cast<const float4*>(A + (size_t)(m0 + row) * lda + k0 + 8 * oct);
        const float4 x = p[0], y = p[1];
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
        if constexpr (SM) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = expf(v[j] - row_max) * row_inv;
        }
      }
      if (TERMS == 4) round8_bf16(v, a_scale, As[oct * 64 + row], As[(4 + oct) * 64 + row]);
      else split8(v, a_scale, As[oct * 64 + row], As[(4 + oct) * 64 + row]);
    }
    if (BT) {
      const int col = tid >> 2, oct = tid & 3;
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (n0 + col < Ncols) {
        const float4* p = reinterpret_cast<const float4*>(B + (size_t)(n0 + col) * ldb + k0 + 8 * oct);
        const float4 x = p[0], y = p[1];
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
      }
      if (TERMS == 4) round8_bf16(v, b_scale, Bs[oct * 64 + col], Bs[(4 + oct) * 64 + col]);
      else split8(v, b_scale, Bs[oct * 64 + col], Bs[(4 + oct) * 64 + col]);
    } else {
      const int oct = tid >> 6, col = tid & 63;
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (n0 + col < Ncols) {
        const float* p = B + (size_t)(k0 + 8 * oct) * ldb + n0 + col;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[(size_t)j * ldb];
      }
      if (TERMS == 4) round8_bf16(v, b_scale, Bs[oct * 64 + col], Bs[(4 + oct) * 64 + col]);
      else split8(v, b_scale, Bs[oct * 64 + col], Bs[(4 + oct) * 64 + col]);
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
      if (TERMS == 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.b8, bh.b8, acc, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h8, bh.h8, acc, 0, 0, 0);
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

// S = alpha q k^T of the short-sequence levels (T <= 256: the 16x16, 8x8 and 4x4 maps), latency form.  The 64x64-tile kernel above is a chain
// of (load -> split -> LDS -> barrier -> MFMA -> barrier) rounds: 22 - 26 us per launch at any batch size, K = C = 384 ... 768 deep on ONE
// workgroup per image at T = 64.  Here a workgroup owns a 32x32 score tile and its four waves split K: a lane loads the eight consecutive
// channels of "its" q row and k row per 16-wide step straight into registers (the operand layout of v_mfma_f32_32x32x16: lane (r, h) holds
// k = 8h .. 8h + 7 of row r), QK_UNR steps of loads in flight at a time, no LDS and no barrier in the loop; the four partial tiles meet in LDS
// and are added in wave order (deterministic).  Same arithmetic as bgemm64s_kernel<true, TERMS> (fp16 hi / lo split of both operands), which it
// replaces for S (deepening that kernel's K chunk to 128 per barrier pair was measured first: 23 -> 22 us, the rounds are not what it waits for).
constexpr int QK_UNR = 4;
template <int TERMS>
__global__ __launch_bounds__(256) void qk_small_kernel(const float* __restrict__ Q, const float* __restrict__ Kp, float* __restrict__ S, int T, int C,
                                                       int ld, long long sQ, long long sS, float alpha, const float* __restrict__ q_scale,
                                                       const float* __restrict__ k_scale, const float* __restrict__ qk_inv,
                                                       const float* __restrict__ k_inv) {
  // q_scale .. k_inv: optional per-image (blockIdx.z) power-of-two staging factors of q and k and their inverses (qk_inv carries alpha): the
  // range guard of attn_scales_kernel, as on the conv-pipeline form
  const float sq_img = q_scale ? q_scale[blockIdx.z] : 1.0f, sk_img = q_scale ? k_scale[blockIdx.z] : 1.0f;
  if (q_scale) alpha = qk_inv[blockIdx.z] * k_inv[blockIdx.z];
  __shared__ float part[4][32 * 33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  Q += (size_t)blockIdx.z * sQ;
  Kp += (size_t)blockIdx.z * sQ;
  S += (size_t)blockIdx.z * sS;
  const bool qin = m0 + r < T, kin = n0 + r < T;
  const int steps = C / 16, per = (steps + 3) / 4;  // a contiguous K range per wave: consecutive steps share cache lines
  const int s0 = wave * per, s1 = min(steps, s0 + per);
  const float4* qp = reinterpret_cast<const float4*>(Q + (size_t)min(m0 + r, T - 1) * ld + 8 * h);
  const float4* kp = reinterpret_cast<const float4*>(Kp + (size_t)min(n0 + r, T - 1) * ld + 8 * h);
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int sb = s0; sb < s1; sb += QK_UNR) {
    float4 qa[QK_UNR][2], ka[QK_UNR][2];
#pragma unroll
    for (int u = 0; u < QK_UNR; ++u) {
      const int st = min(sb + u, s1 - 1);  // (a step past the range re-reads the last one and is not accumulated)
      qa[u][0] = qp[4 * st]; qa[u][1] = qp[4 * st + 1];
      ka[u][0] = kp[4 * st]; ka[u][1] = kp[4 * st + 1];
    }
#pragma unroll
    for (int u = 0; u < QK_UNR; ++u) {
      if (sb + u < s1) {
        float qv[8] = {qa[u][0].x, qa[u][0].y, qa[u][0].z, qa[u][0].w, qa[u][1].x, qa[u][1].y, qa[u][1].z, qa[u][1].w};
        float kv[8] = {ka[u][0].x, ka[u][0].y, ka[u][0].z, ka[u][0].w, ka[u][1].x, ka[u][1].y, ka[u][1].z, ka[u][1].w};
        if (!qin) {
#pragma unroll
          for (int j = 0; j < 8; ++j) qv[j] = 0.f;
        }
        if (!kin) {
#pragma unroll
          for (int j = 0; j < 8; ++j) kv[j] = 0.f;
        }
        AF4H8 ah, al, bh, bl;
        if (TERMS == 4) {
          round8_bf16(qv, sq_img, ah.f4, al.f4);
          round8_bf16(kv, sk_img, bh.f4, bl.f4);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.b8, bh.b8, acc, 0, 0, 0);
        } else {
          split8(qv, sq_img, ah.f4, al.f4);
          split8(kv, sk_img, bh.f4, bl.f4);
          if (TERMS == 3) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.h8, bh.h8, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h8, bl.h8, acc, 0, 0, 0);
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h8, bh.h8, acc, 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) part[wave][((e & 3) + 8 * (e >> 2) + 4 * h) * 33 + r] = acc[e];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = tid + 256 * j, row = idx >> 5, col = idx & 31;
    const float v = ((part[0][row * 33 + col] + part[1][row * 33 + col]) + part[2][row * 33 + col]) + part[3][row * 33 + col];
    if (m0 + row < T && n0 + col < T) S[(size_t)(m0 + row) * T + n0 + col] = alpha * v;
  }
}

// O = softmax(S) v of the short-sequence levels, latency form (the counterpart of qk_small_kernel; replaces bgemm64s_kernel<false, ., SM> there:
// T = 256 took 20 us on 32 workgroups, eight load -> split -> LDS -> barrier -> MFMA rounds each).  A workgroup owns 32 queries x 32 channels;
// it first takes the maximum and the sum of exp of its 32 score rows (eight lanes per row), then its four waves split the keys: a lane loads
// eight consecutive scores of "its" row and eight v values of "its" channel per 16-key step straight into the MFMA operand layout, applies
// expf(s - max) / sum * 2^12 to the scores and the per-image power of two to v (attn_scales_kernel), no LDS and no barrier in the loop; the four
// partial tiles meet in LDS and are added in wave order (deterministic).
constexpr int PV_UNR = 2;
template <int TERMS>
__global__ __launch_bounds__(256) void pv_small_kernel(const float* __restrict__ S, const float* __restrict__ V, float* __restrict__ O, int T, int C, int ldv,
                                                       long long sS, long long sV, long long sO, float alpha, float a_scale,
                                                       const float* __restrict__ v_scale_img, const float* __restrict__ v_inv_img) {
  __shared__ float part[4][32 * 33];
  __shared__ float row_max[32], row_inv[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  S += (size_t)blockIdx.z * sS;
  V += (size_t)blockIdx.z * sV;
  O += (size_t)blockIdx.z * sO;
  const float v_scale = v_scale_img ? v_scale_img[blockIdx.z] : 1.0f;
  if (v_inv_img) alpha *= v_inv_img[blockIdx.z];
  {  // row softmax statistics: row = tid >> 3, eight lanes per row over whole rows (T % 32 == 0)
    const int row = tid >> 3, part8 = tid & 7;
    const float4* p = reinterpret_cast<const float4*>(S + (size_t)min(m0 + row, T - 1) * T);
    float m = -INFINITY;
    for (int q = part8; q < T / 4; q += 8) {
      const float4 x = p[q];
      m = fmaxf(fmaxf(m, fmaxf(x.x, x.y)), fmaxf(x.z, x.w));
    }
    m = fmaxf(m, __shfl_xor(m, 1));
    m = fmaxf(m, __shfl_xor(m, 2));
    m = fmaxf(m, __shfl_xor(m, 4));
    float sum = 0.f;
    for (int q = part8; q < T / 4; q += 8) {
      const float4 x = p[q];
      sum += (expf(x.x - m) + expf(x.y - m)) + (expf(x.z - m) + expf(x.w - m));
    }
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    sum += __shfl_xor(sum, 4);
    if (part8 == 0) {
      row_max[row] = m;
      row_inv[row] = 1.0f / sum;
    }
  }
  __syncthreads();
  const bool qin = m0 + r < T, cin = c0 + r < C;
  const float rm = row_max[r], ri = qin ? row_inv[r] * a_scale : 0.f;
  const int steps = T / 16, per = (steps + 3) / 4;
  const int s0 = wave * per, s1 = min(steps, s0 + per);
  const float4* sp = reinterpret_cast<const float4*>(S + (size_t)min(m0 + r, T - 1) * T + 8 * h);
  const float* vp = V + (size_t)(8 * h) * ldv + min(c0 + r, C - 1);
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int sb = s0; sb < s1; sb += PV_UNR) {
    float4 sa[PV_UNR][2];
    float va[PV_UNR][8];
#pragma unroll
    for (int u = 0; u < PV_UNR; ++u) {
      const int st = min(sb + u, s1 - 1);  // (a step past the range re-reads the last one and is not accumulated)
      sa[u][0] = sp[4 * st];
      sa[u][1] = sp[4 * st + 1];
#pragma unroll
      for (int j = 0; j < 8; ++j) va[u][j] = vp[(size_t)(16 * st + j) * ldv];
    }
#pragma unroll
    for (int u = 0; u < PV_UNR; ++u) {
      if (sb + u < s1) {
        float pv[8] = {sa[u][0].x, sa[u][0].y, sa[u][0].z, sa[u][0].w, sa[u][1].x, sa[u][1].y, sa[u][1].z, sa[u][1].w};
        float vv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          pv[j] = expf(pv[j] - rm) * ri;
          vv[j] = cin ? va[u][j] : 0.f;
        }
        AF4H8 ah, al, bh, bl;
        if (TERMS == 4) {
          round8_bf16(pv, 1.0f, ah.f4, al.f4);
          round8_bf16(vv, v_scale, bh.f4, bl.f4);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.b8, bh.b8, acc, 0, 0, 0);
        } else {
          split8(pv, 1.0f, ah.f4, al.f4);
          split8(vv, v_scale, bh.f4, bl.f4);
          if (TERMS == 3) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.h8, bh.h8, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h8, bl.h8, acc, 0, 0, 0);
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h8, bh.h8, acc, 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) part[wave][((e & 3) + 8 * (e >> 2) + 4 * h) * 33 + r] = acc[e];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = tid + 256 * j, row = idx >> 5, col = idx & 31;
    const float v = ((part[0][row * 33 + col] + part[1][row * 33 + col]) + part[2][row * 33 + col]) + part[3][row * 33 + col];
    if (m0 + row < T && c0 + col < C) O[(size_t)(m0 + row) * C + c0 + col] = alpha * v;
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

// ------------------------------------------------------------------------------------------------------------------------------
// Split-precision attention core on the fused 1x1 conv pipeline (conv_split2.hip): both batched GEMMs are 1x1 "convolutions" whose
// weights differ per image --
//     S[n] = q[n] . k[n]^T : pixels = queries, Cin = C, Cout = T keys,   weights = k[n]      (rows of k are already [key][channel])
//     O[n] = P[n] . v[n]   : pixels = queries, Cin = T keys, Cout = C,   weights = v[n]^T
// -- so k and v^T are packed once per forward into the kernel's pre-split LDS-DMA weight image (hi/lo fp16 planes), each image
// scaled by its own power of two, and the persistent 256-pixel x 128-channel tiles, the LDS-DMA weight ring and the 16-byte
// transposed epilogue stores are reused as they are.  q and the probabilities are staged like any un-normalised activation
// (per-image 2^k for q from the qkv conv's fused statistics, a fixed 2^12 for the probabilities); alpha = C^-1/2 rides on the
// epilogue factor.  Needs one image per pixel tile: T = H*W a multiple of 256 (the T >= 512 levels, where the FLOPs are).
// ------------------------------------------------------------------------------------------------------------------------------

// per image: power-of-two factors of q, k, v from the per-channel sums of squares of the qkv tensor (bound * 2^k in [2^14, 2^15))
// and everything the two conv launches read: tables [N][C] (2^kq), [N][T] (2^12), zeros, and the epilogue / weight factors
__global__ __launch_bounds__(256) void attn_scales_kernel(const double2* __restrict__ mom, int C, int T, float alpha, float* __restrict__ q_tab,
                                                          float* __restrict__ p_tab, float* __restrict__ zero_tab, float* __restrict__ qk_inv,
                                                          float* __restrict__ k_scale, float* __restrict__ k_inv, float* __restrict__ pv_inv,
                                                          float* __restrict__ v_scale, float* __restrict__ v_inv, float* __restrict__ o_tab,
                                                          float* __restrict__ q_scale /* [N] 2^kq, or null */) {
  __shared__ double red[3][4];
  __shared__ float s_q, s_v;
  const int n = blockIdx.x, t = threadIdx.x;
  double m[3] = {0.0, 0.0, 0.0};
  for (int c = t; c < 3 * C; c += 256) m[c / C] = fmax(m[c / C], mom[(size_t)n * 3 * C + c].y);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m[j] = fmax(m[j], __shfl_xor(m[j], o));
    if ((t & 63) == 0) red[j][t >> 6] = m[j];
  }
  __syncthreads();
  if (t == 0) {
    float sc[3], iv[3];
    for (int j = 0; j < 3; ++j) {
      const double bound = sqrt(fmax(fmax(red[j][0], red[j][1]), fmax(red[j][2], red[j][3])));
      int k = 0;
      if (bound > 0.0 && bound < INFINITY) {
        int e;
        frexp(bound, &e);
        k = 15 - e;
        k = k > 40 ? 40 : (k < -40 ? -40 : k);
      }
      sc[j] = ldexpf(1.0f, k);
      iv[j] = ldexpf(1.0f, -k);
    }
    s_q = sc[0];
    s_v = sc[2];
    qk_inv[n] = alpha * iv[0];
    if (q_scale) q_scale[n] = sc[0];
    k_scale[n] = sc[1];
    k_inv[n] = iv[1];
    pv_inv[n] = 1.0f / 4096.0f;
    v_scale[n] = sc[2];
    v_inv[n] = iv[2];
  }
  __syncthreads();
  const float sq = s_q, sv = s_v;
  for (int c = t; c < C; c += 256) {
    q_tab[(size_t)n * C + c] = sq;
    o_tab[(size_t)n * C + c] = sv;  // the attention output is a convex combination of v rows: the same bound guards proj_out's input
  }
  for (int c = t; c < T; c += 256) p_tab[(size_t)n * T + c] = 4096.0f;  // probabilities <= 1: 2^12 keeps the small ones normal in fp16
  const int Z = C > T ? C : T;
  for (int c = t; c < Z; c += 256) zero_tab[(size_t)n * Z + c] = 0.f;
}

// One image's GEMM "weights" W[co][ci] * scale[n] -> the pre-split LDS image of conv_split.hip ([chunk][hi|lo][slab s][lane half]
// [Cout][8 halfs], taps = 1).  A thread produces one 16-byte (hi) + one 16-byte (lo) entry = 8 consecutive ci of one co.
//   ROWS = true : W[co][ci] = src[n][co * ld + ci]  (k: rows are keys, ci = channels contiguous): the four octets of a 32-channel
//                 chunk sit on four adjacent lanes, so a wave reads 16 whole 128-byte lines;
//   ROWS = false: W[co][ci] = src[n][ci * ld + co]  (v^T: co = channel contiguous, ci = key): adjacent lanes take adjacent channels,
//                 each of the 8 loads of a thread is a coalesced 256-byte row segment.
// grid (blocks, N), block 256.
template <bool ROWS>
__global__ __launch_bounds__(256) void pack_attn_weight_kernel(const float* __restrict__ src, long long img_stride, int ld,
                                                               const float* __restrict__ scale, float4* __restrict__ dst, int Cout, int Cin, int bf16) {
  const int n = blockIdx.y;
  const float sc = scale[n];
  const float* sp = src + (size_t)n * img_stride;
  float4* dp = dst + (size_t)n * ((size_t)Cout * Cin / 4);  // hi + lo fp16 = 4 bytes per weight
  const size_t units = (size_t)Cout * (Cin / 8);           // (co, octet of ci)
  for (size_t u = blockIdx.x * (size_t)blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
    int co, oct_all;
    if (ROWS) {
      oct_all = (int)(u % 4) + 4 * (int)(u / ((size_t)4 * Cout));  // lanes: 4 octets of a chunk, then co
      co = (int)((u / 4) % Cout);
    } else {
      co = (int)(u % Cout);
      oct_all = (int)(u / Cout);
    }
    const int q = oct_all >> 2, seg = oct_all & 3;  // seg = slab * 2 + lane half
    const int ci0 = 8 * oct_all;
    float v[8];
    if (ROWS) {
      const float4* p4 = reinterpret_cast<const float4*>(sp + (size_t)co * ld + ci0);
      const float4 x = p4[0], y = p4[1];
      v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = sp[(size_t)(ci0 + j) * ld + co];
    }
    AF4H8 hi, lo;
    if (bf16) {
      round8_bf16(v, sc, hi.f4, lo.f4);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float c = __builtin_amdgcn_fmed3f(v[j] * sc, -65504.0f, 65504.0f);
        const _Float16 hh = (_Float16)c;
        hi.h8[j] = hh;
        lo.h8[j] = (_Float16)(c - (float)hh);
      }
    }
    dp[((size_t)q * 8 + seg) * Cout + co] = hi.f4;
    dp[((size_t)q * 8 + 4 + seg) * Cout + co] = lo.f4;
  }
}

// One-pass row softmax for rows of T = 256 * Q floats (Q <= 8): a wave keeps its row in registers (16 bytes per lane and load),
// so the scores are read once and written once (the generic kernel above makes three passes over them).
template <int Q>
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(float* __restrict__ S, long long rows) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float4* p = reinterpret_cast<float4*>(S + row * (256ll * Q));
  float4 v[Q];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    v[j] = p[j * 64 + lane];
    m = fmaxf(m, fmaxf(fmaxf(v[j].x, v[j].y), fmaxf(v[j].z, v[j].w)));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    v[j].x = expf(v[j].x - m); v[j].y = expf(v[j].y - m); v[j].z = expf(v[j].z - m); v[j].w = expf(v[j].w - m);
    sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int j = 0; j < Q; ++j) p[j * 64 + lane] = make_float4(v[j].x * inv, v[j].y * inv, v[j].z * inv, v[j].w * inv);
}

static int launch_softmax_rows(float* S, long long rows, int T, hipStream_t s) {
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (T % 256 == 0 && T / 256 <= 8) {
    switch (T / 256) {
      case 1: hipLaunchKernelGGL(softmax_rows_reg_kernel<1>, grid, block, 0, s, S, rows); break;
      case 2: hipLaunchKernelGGL(softmax_rows_reg_kernel<2>, grid, block, 0, s, S, rows); break;
      case 3: hipLaunchKernelGGL(softmax_rows_reg_kernel<3>, grid, block, 0, s, S, rows); break;
      case 4: hipLaunchKernelGGL(softmax_rows_reg_kernel<4>, grid, block, 0, s, S, rows); break;
      case 5: hipLaunchKernelGGL(softmax_rows_reg_kernel<5>, grid, block, 0, s, S, rows); break;
      case 6: hipLaunchKernelGGL(softmax_rows_reg_kernel<6>, grid, block, 0, s, S, rows); break;
      case 7: hipLaunchKernelGGL(softmax_rows_reg_kernel<7>, grid, block, 0, s, S, rows); break;
      default: hipLaunchKernelGGL(softmax_rows_reg_kernel<8>, grid, block, 0, s, S, rows); break;
    }
  } else {
    hipLaunchKernelGGL(softmax_rows_kernel, grid, block, 0, s, S, rows, T);
  }
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

// Images per attention pass: the [T, T] score matrices of a group live in the workspace between the S GEMM, the softmax and the PV GEMM.
// The group bounds that workspace (1 GiB: 64 images at T = 2048 -- batch 256 used to need 4.3 GB per attention block) while keeping the
// launches large.  Small groups that would keep the scores in the 256 MiB Infinity Cache were measured and lose: at batch 32, T = 2048 the
// attention core takes 6.3 ms in one pass, 7.1 ms in 128 MiB groups, 8.8 ms in 64 MiB groups, 11.8 ms in 32 MiB groups (under-filled grids
// and launch tails cost more than the HBM round trips of the scores).
#ifndef DRM_ATTN_GROUP_MIB
#define DRM_ATTN_GROUP_MIB 1024
#endif
int attention_group(int N, int T) {
  const size_t per_image = (size_t)T * T * sizeof(float);
  const size_t g = ((size_t)DRM_ATTN_GROUP_MIB << 20) / (per_image ? per_image : 1);
  return (int)std::max<size_t>(1, std::min<size_t>(g, (size_t)N));
}
size_t attention_scores_floats(int N, int T) { return (size_t)attention_group(N, T) * T * T; }

size_t attention_conv_workspace_floats(int N, int T, int C) {
  const size_t Z = (size_t)(C > T ? C : T);
  return 2 * (size_t)N * T * C + (size_t)N * (2 * C + T + Z) + 6 * (size_t)N + 64;
}
bool attention_conv_applicable(int T, int C, int H, int W, int terms) {
  return terms != 0 && T % 256 == 0 && C % 128 == 0 && T % 128 == 0 && H % 16 == 0 && W % 16 == 0;
}

int launch_conv_split(const ConvArgs& a, hipStream_t s);

// (for attn_flash.hip: the per-image factor tables and the row-major pre-split image of q / k)
void launch_attn_scales(const double2* mom, int N, int C, int T, float alpha, float* q_tab, float* p_tab, float* zero_tab, float* qk_inv, float* k_scale,
                        float* k_inv, float* pv_inv, float* v_scale, float* v_inv, float* o_tab, float* q_scale, hipStream_t s) {
  hipLaunchKernelGGL(attn_scales_kernel, dim3(N), dim3(256), 0, s, mom, C, T, alpha, q_tab, p_tab, zero_tab, qk_inv, k_scale, k_inv, pv_inv, v_scale, v_inv,
                     o_tab, q_scale);
}
int launch_pack_attn_rows(const float* src, long long img_stride, int ld, const float* scale, float* dst, int rows, int cin, int N, hipStream_t s, bool bf16) {
  const unsigned pb = (unsigned)std::min<size_t>(((size_t)rows * cin / 8 + 255) / 256, 4096);
  hipLaunchKernelGGL(pack_attn_weight_kernel<true>, dim3(pb, N), dim3(256), 0, s, src, img_stride, ld, scale, reinterpret_cast<float4*>(dst), rows, cin, bf16 ? 1 : 0);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

// qkv [N][T][3C] (+ its fused per-channel statistics), scores workspace [N][T][T], out [N][T][C], ws: attention_conv_workspace_floats
// proj_guard (optional): receives the (scale, shift, inverse) tables that guard proj_out's read of `out`
int launch_attention_conv(const float* qkv, const double2* qkv_mom, float* scores, float* out, float* ws, int N, int H, int W, int C, int terms,
                          hipStream_t s, ConvArgs* proj_guard) {
  const int T = H * W;
  DRM_REQUIRE(attention_conv_applicable(T, C, H, W, terms) && qkv_mom, "attention on the conv pipeline: shape");
  const size_t Z = (size_t)(C > T ? C : T);
  float* wk = ws;                           // [N] packed k:   Cout = T, Cin = C
  float* wv = wk + (size_t)N * T * C;       // [N] packed v^T: Cout = C, Cin = T
  float* q_tab = wv + (size_t)N * T * C;    // [N][C]
  float* p_tab = q_tab + (size_t)N * C;     // [N][T]
  float* zero_tab = p_tab + (size_t)N * T;  // [N][max(C, T)]
  float* o_tab = zero_tab + (size_t)N * Z;  // [N][C]
  float* vec = o_tab + (size_t)N * C;       // 6 x [N]
  float *qk_inv = vec, *k_scale = vec + N, *k_inv = vec + 2 * N, *pv_inv = vec + 3 * N, *v_scale = vec + 4 * N, *v_inv = vec + 5 * N;
  const float alpha = 1.0f / sqrtf((float)C);  // (C^-1/4)^2, applied once to the dot product
  prof_tag(N, T, 1, C, C);
  ProfScope ps(PROF_ATTN, 4.0 * N * (double)T * T * C, 4.0 * N * ((double)T * 4 * C + 4.0 * T * T), s);  // one scope for the whole core
  hipLaunchKernelGGL(attn_scales_kernel, dim3(N), dim3(256), 0, s, qkv_mom, C, T, alpha, q_tab, p_tab, zero_tab, qk_inv, k_scale, k_inv, pv_inv,
                     v_scale, v_inv, o_tab, nullptr);
  DRM_HIP_CHECK(hipGetLastError());
  if (proj_guard) {
    proj_guard->gn_scale = o_tab;
    proj_guard->gn_shift = zero_tab;
    proj_guard->in_inv = v_inv;
  }
  const unsigned pb = (unsigned)std::min<size_t>(((size_t)T * C / 8 + 255) / 256, 4096);
  hipLaunchKernelGGL(pack_attn_weight_kernel<true>, dim3(pb, N), dim3(256), 0, s, qkv + C, (long long)T * 3 * C, 3 * C, k_scale,
                     reinterpret_cast<float4*>(wk), T, C, terms == 4 ? 1 : 0);
  DRM_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(pack_attn_weight_kernel<false>, dim3(pb, N), dim3(256), 0, s, qkv + 2 * C, (long long)T * 3 * C, 3 * C, v_scale,
                     reinterpret_cast<float4*>(wv), C, T, terms == 4 ? 1 : 0);
  DRM_HIP_CHECK(hipGetLastError());
  const int NB = attention_group(N, T);
  for (int n0 = 0; n0 < N; n0 += NB) {  // one pass per image group (scores = the group's [nb, T, T] buffer)
    const int nb = std::min(NB, N - n0);
    ConvArgs a;  // S = alpha q k^T
    a.src0 = qkv + (size_t)n0 * T * 3 * C; a.C0 = C; a.ld0 = 3 * C; a.N = nb; a.H = H; a.W = W;
    a.gn_scale = q_tab + (size_t)n0 * C; a.gn_shift = zero_tab + (size_t)n0 * Z; a.silu = 0;
    a.w = wk + (size_t)n0 * T * C; a.w_img_stride_f4 = (long long)T * C / 4; a.w_inv_img = k_inv + n0; a.in_inv = qk_inv + n0;
    a.taps = 1; a.Cout = T; a.out = scores; a.terms = terms; a.prof_kind = PROF_KINDS;  // (inside the scope above)
    DRM_TRY(launch_conv_split(a, s));
    DRM_TRY(launch_softmax_rows(scores, (long long)nb * T, T, s));
    ConvArgs b;  // O = P v
    b.src0 = scores; b.C0 = T; b.N = nb; b.H = H; b.W = W;
    b.gn_scale = p_tab + (size_t)n0 * T; b.gn_shift = zero_tab + (size_t)n0 * Z; b.silu = 0;
    b.w = wv + (size_t)n0 * T * C; b.w_img_stride_f4 = (long long)T * C / 4; b.w_inv_img = v_inv + n0; b.in_inv = pv_inv + n0;
    b.taps = 1; b.Cout = C; b.out = out + (size_t)n0 * T * C; b.terms = terms; b.prof_kind = PROF_KINDS;
    DRM_TRY(launch_conv_split(b, s));
  }
  return DRM_OK;
}

size_t attention_small_workspace_floats(int N, int T, int C) {
  const size_t Z = (size_t)(C > T ? C : T);
  return (size_t)N * (2 * C + T + Z) + 7 * (size_t)N + 64;
}

// Short-sequence form (T <= 256 off the conv pipeline, and every level of a sparse launch): S by qk_small_kernel, row softmax, P v by the 64x64-tile
// GEMM.  qkv_mom + ws (attention_small_workspace_floats): the split modes stage q, k and v through their per-image powers of two (attn_scales_kernel --
// the same range guard as the conv-pipeline form: |v| or |q| beyond fp16's range is exact), and proj_guard receives the guard tables of proj_out's
// input (|attention output| <= max |v|) from the same launch.
int launch_attention(const float* qkv, float* scores, float* out, int N, int T, int C, hipStream_t s, int terms, const double2* qkv_mom, float* ws,
                     ConvArgs* proj_guard) {
  bool split = terms != 0;
  DRM_REQUIRE(C % 4 == 0 && T > 0 && N > 0, "attention shape");
  const float alpha = 1.0f / sqrtf((float)C);  // (C^-1/4)^2, applied once to the dot product
  const int tb = (T + 63) / 64;
  prof_tag(N, T, 1, C, C);
  ProfScope ps(PROF_ATTN, 4.0 * N * (double)T * T * C, 4.0 * N * ((double)T * 4 * C + 4.0 * T * T), s);
  // S = q k^T needs whole 32-chunks of C only (rows beyond T are masked); P v needs them of T: the 4x4 level (T = 16) of a 128x128 input keeps
  // the exact-fp32 form for P v and takes the split form for S like every other level
  const bool split_qk = split && (C % 32 == 0);
  split = split_qk && (T % 32 == 0);
  const bool guard = split_qk && qkv_mom && ws;
  float *q_scale = nullptr, *k_scale = nullptr, *qk_inv = nullptr, *k_inv = nullptr, *v_scale = nullptr, *v_inv = nullptr;
  if (guard) {
    const size_t Z = (size_t)(C > T ? C : T);
    float* q_tab = ws;                        // [N][C]
    float* p_tab = q_tab + (size_t)N * C;     // [N][T]
    float* zero_tab = p_tab + (size_t)N * T;  // [N][max(C, T)]
    float* o_tab = zero_tab + (size_t)N * Z;  // [N][C]
    float* vec = o_tab + (size_t)N * C;       // 7 x [N]
    qk_inv = vec; k_scale = vec + N; k_inv = vec + 2 * N; float* pv_inv = vec + 3 * N; v_scale = vec + 4 * N; v_inv = vec + 5 * N; q_scale = vec + 6 * N;
    hipLaunchKernelGGL(attn_scales_kernel, dim3(N), dim3(256), 0, s, qkv_mom, C, T, alpha, q_tab, p_tab, zero_tab, qk_inv, k_scale, k_inv, pv_inv, v_scale,
                       v_inv, o_tab, q_scale);
    DRM_HIP_CHECK(hipGetLastError());
    if (proj_guard) {
      proj_guard->gn_scale = o_tab;
      proj_guard->gn_shift = zero_tab;
      proj_guard->in_inv = v_inv;
    }
  }
  const int NB = attention_group(N, T), t32 = (T + 31) / 32;
  const long long sq = (long long)T * 3 * C;  // image stride of qkv
  auto at = [&](float* p, int n0) { return p ? p + n0 : nullptr; };
  for (int n0 = 0; n0 < N; n0 += NB) {
    const int nb = std::min(NB, N - n0);
    const float* qg = qkv + (size_t)n0 * sq;
    float* og = out + (size_t)n0 * T * C;
    if (split_qk && terms == 4)
      hipLaunchKernelGGL(qk_small_kernel<4>, dim3(t32, t32, nb), dim3(256), 0, s, qg, qg + C, scores, T, C, 3 * C, sq, (long long)T * T, alpha, at(q_scale, n0), at(k_scale, n0), at(qk_inv, n0), at(k_inv, n0));
    else if (split_qk && terms == 1)
      hipLaunchKernelGGL(qk_small_kernel<1>, dim3(t32, t32, nb), dim3(256), 0, s, qg, qg + C, scores, T, C, 3 * C, sq, (long long)T * T, alpha, at(q_scale, n0), at(k_scale, n0), at(qk_inv, n0), at(k_inv, n0));
    else if (split_qk)
      hipLaunchKernelGGL(qk_small_kernel<3>, dim3(t32, t32, nb), dim3(256), 0, s, qg, qg + C, scores, T, C, 3 * C, sq, (long long)T * T, alpha, at(q_scale, n0), at(k_scale, n0), at(qk_inv, n0), at(k_inv, n0));
    else
      hipLaunchKernelGGL(bgemm64_kernel<true>, dim3(tb, tb, nb), dim3(256), 0, s, qg, qg + C, scores, T, T, C, 3 * C, 3 * C, T, sq, sq, (long long)T * T, alpha);
    DRM_HIP_CHECK(hipGetLastError());
    if (!split) DRM_TRY(launch_softmax_rows(scores, (long long)nb * T, T, s));  // (the split P v applies the row softmax while it stages the scores)
    if (split && terms == 4)
      hipLaunchKernelGGL(pv_small_kernel<4>, dim3((C + 31) / 32, t32, nb), dim3(256), 0, s, scores, qg + 2 * C, og, T, C, 3 * C, (long long)T * T, sq,
                         (long long)T * C, 1.0f, 1.0f, at(v_scale, n0), at(v_inv, n0));
    else if (split && terms == 1)  // probabilities are scaled by 2^12 before the fp16 conversion (largest 4096, smallest normal 2^-26)
      hipLaunchKernelGGL(pv_small_kernel<1>, dim3((C + 31) / 32, t32, nb), dim3(256), 0, s, scores, qg + 2 * C, og, T, C, 3 * C, (long long)T * T, sq,
                         (long long)T * C, 1.0f / 4096.0f, 4096.0f, at(v_scale, n0), at(v_inv, n0));
    else if (split)
      hipLaunchKernelGGL(pv_small_kernel<3>, dim3((C + 31) / 32, t32, nb), dim3(256), 0, s, scores, qg + 2 * C, og, T, C, 3 * C, (long long)T * T, sq,
                         (long long)T * C, 1.0f / 4096.0f, 4096.0f, at(v_scale, n0), at(v_inv, n0));
    else
      hipLaunchKernelGGL(bgemm64_kernel<false>, dim3((C + 63) / 64, tb, nb), dim3(256), 0, s, scores, qg + 2 * C, og, T, C, T, T, 3 * C, C,
                         (long long)T * T, sq, (long long)T * C, 1.0f);
    DRM_HIP_CHECK(hipGetLastError());
  }
  return DRM_OK;
}

}  // namespace drm
