// Small HBM-bound kernels around the convolutions: boundary layout conversion, 2x2 average pool,
// embedding MLPs, sinusoidal timestep embedding, RefNet pooled head.
#include "common.h"

namespace drm {

__device__ __forceinline__ float silu_m(float v) { return v / (1.0f + __expf(-v)); }

// ---------------------------------------------------------------------------------------------
// cat([x, cond], dim=1) (DiffusionWrapper concat branch, ldm/models/diffusion/ddpm.py:1527-1529;
// ZEmbDiffusionWrapper, models/drmnet.py:59-61) fused with NCHW -> NHWC and zero padding to CP channels.
// idx (optional) gathers rows: out row j reads sample idx[j] (DRMNet active-set compaction,
// models/drmnet.py:810-813).
// ---------------------------------------------------------------------------------------------
// One thread per (pixel, 4-channel quad): the NHWC writes are fully coalesced 16-byte stores; only the first
// (Cx + Cc + 3) / 4 quads read anything.
// grid (blocks over one image, N): a block never straddles two images, so the optional per-image absmax is a block reduction and
// at most ONE atomicMax per block (non-negative floats order like their bit patterns) -- skipped when the word already holds a
// value at least as large (a stale read only costs a redundant atomic; same-address atomics serialise at the memory side).
__global__ __launch_bounds__(256) void pack_input_kernel(const float* __restrict__ x, const float* __restrict__ cond, const int* __restrict__ idx,
                                                          float4* __restrict__ out, int N, int HW, int Cx, int Cc, int Q,
                                                          unsigned* __restrict__ absmax_bits) {
  const int n = blockIdx.y;
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // (pixel, quad) inside image n
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  const bool live = j < (long long)HW * Q;
  const int q = (int)(j % Q);
  const int p = (int)(j / Q);
  const long long i = (long long)n * HW * Q + j;
  if (live && 4 * q < Cx + Cc) {
    const int src = idx ? idx[n] : n;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = 4 * q + k;
      if (c < Cx) v[k] = x[((size_t)src * Cx + c) * HW + p];
      else if (c < Cx + Cc) v[k] = cond[((size_t)src * Cc + (c - Cx)) * HW + p];
    }
  }
  if (live) out[i] = make_float4(v[0], v[1], v[2], v[3]);
  if (absmax_bits) {
    __shared__ float wmax[4];
    float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    if (!(m == m)) m = INFINITY;  // NaN input: no finite bound
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
      if (m > 0.f && __float_as_uint(m) > __hip_atomic_load(absmax_bits + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(absmax_bits + n, __float_as_uint(m));
    }
  }
}

int launch_pack_input(const float* x, const float* cond, const int* idx, float* out, int N, int H, int W, int Cx, int Cc, int CP, hipStream_t s,
                      unsigned* absmax_bits) {
  DRM_REQUIRE(CP % 4 == 0, "packed input channels must be a multiple of 4");
  const long long per_image = (long long)H * W * (CP / 4);
  hipLaunchKernelGGL(pack_input_kernel, dim3((unsigned)((per_image + 255) / 256), (unsigned)N), dim3(256), 0, s, x, cond, idx,
                     reinterpret_cast<float4*>(out), N, H * W, Cx, Cc, CP / 4, absmax_bits);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

// Downsample without conv = AvgPool2d(2,2) (openaimodel.py:154-160), NHWC
__global__ void avgpool2_kernel(const float4* __restrict__ x, float4* __restrict__ out, int N, int Ho, int Wo, int q4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)N * Ho * Wo * q4;
  if (i >= total) return;
  const int q = (int)(i % q4);
  long long t = i / q4;
  const int xo = (int)(t % Wo);
  t /= Wo;
  const int yo = (int)(t % Ho);
  const int n = (int)(t / Ho);
  const int W = Wo * 2;
  const size_t base = (((size_t)n * Ho * 2 + yo * 2) * W + xo * 2) * q4 + q;
  const float4 a = x[base], b = x[base + q4], c = x[base + (size_t)W * q4], d = x[base + (size_t)W * q4 + q4];
  float4 r;
  r.x = (a.x + b.x + c.x + d.x) * 0.25f;
  r.y = (a.y + b.y + c.y + d.y) * 0.25f;
  r.z = (a.z + b.z + c.z + d.z) * 0.25f;
  r.w = (a.w + b.w + c.w + d.w) * 0.25f;
  out[i] = r;
}

int launch_avgpool2(const float* x, float* out, int N, int H, int W, int C, hipStream_t s) {
  DRM_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "avgpool2 shape");
  const long long total = (long long)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(avgpool2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(x),
                     reinterpret_cast<float4*>(out), N, H / 2, W / 2, C / 4);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

// out[n][o] = act_out( b[o] + sum_i act_in(in[n][i]) * W[o][i] ),  W in PyTorch nn.Linear layout [O][I].
// One wave per output feature, looping over the batch: each weight row is read once.
// Used for time_embed (openaimodel.py:521-526), all ResBlock emb_layers of a network in ONE launch
// (weights concatenated along O at load time; openaimodel.py:218-224) and z_emb_layer (models/drmnet.py:38-45).
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ b,
                                                     float* __restrict__ out, int N, int I, int O, int silu_in, int silu_out) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (o >= O) return;
  float wr[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = lane + 64 * k;
    wr[k] = (i < I) ? w[(size_t)o * I + i] : 0.f;
  }
  const float bias = b ? b[o] : 0.f;
  for (int n = 0; n < N; ++n) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = lane + 64 * k;
      if (i < I) {
        float v = in[(size_t)n * I + i];
        if (silu_in) v = silu_m(v);
        acc += v * wr[k];
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) {
      float v = acc + bias;
      if (silu_out) v = silu_m(v);
      out[(size_t)n * O + o] = v;
    }
  }
}

// Matrix-core form for in_features % 16 == 0: out[n][o] = act_out(bias[o] + sum_i act_in(in[n][i]) * w[o][i]) as a
// 32 (batch rows) x 32 (output features) x I GEMM per wave on v_mfma_f32_32x32x2_f32 (exact fp32 products).  Every lane
// reads 8 consecutive floats of "its" input row and weight row per 16-wide K step; MFMA j of the step pairs the j-th of
// them from the two half-waves, a permutation of k that both operands share.  Weights are read once per 32 batch rows.
typedef float f32x16m __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void linear_mfma_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ b,
                                                          float* __restrict__ out, int N, int I, int O, int silu_in, int silu_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int o0 = (blockIdx.x * 4 + wave) * 32, n0 = blockIdx.y * 32;
  if (o0 >= O) return;
  const int n = min(n0 + r, N - 1), o = min(o0 + r, O - 1);
  const float4* pi = reinterpret_cast<const float4*>(in + (size_t)n * I + 8 * h);
  const float4* pw = reinterpret_cast<const float4*>(w + (size_t)o * I + 8 * h);
  f32x16m acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll 4
  for (int kb = 0; kb < I / 16; ++kb) {
    const float4 a0 = pi[4 * kb], a1 = pi[4 * kb + 1];
    const float4 w0 = pw[4 * kb], w1 = pw[4 * kb + 1];
    float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    if (silu_in) {
#pragma unroll
      for (int j = 0; j < 8; ++j) av[j] = silu_m(av[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], wv[j], acc, 0, 0, 0);
  }
  const int oc = o0 + r;
  if (oc < O) {
    const float bias = b ? b[oc] : 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = n0 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (row < N) {
        float v = acc[e] + bias;
        if (silu_out) v = silu_m(v);
        out[(size_t)row * O + oc] = v;
      }
    }
  }
}

int launch_linear(const float* in, const float* w, const float* b, float* out, int N, int I, int O, int silu_in, int silu_out, hipStream_t s) {
  if (I % 16 == 0 && I >= 16) {
    hipLaunchKernelGGL(linear_mfma_kernel, dim3((O + 127) / 128, (N + 31) / 32), dim3(256), 0, s, in, w, b, out, N, I, O, silu_in, silu_out);
    DRM_HIP_CHECK(hipGetLastError());
    return DRM_OK;
  }
  DRM_REQUIRE(I > 0 && I <= 512, "linear: in_features must be <= 512");
  hipLaunchKernelGGL(linear_kernel, dim3((O + 3) / 4), dim3(256), 0, s, in, w, b, out, N, I, O, silu_in, silu_out);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

// timestep_embedding (util.py:151-171): cat(cos(t f_j), sin(t f_j)), f_j = exp(-ln(1e4) j / half), fp32.
// t comes as int64 (reference: torch.long timesteps) or as fp32.
__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t, const float* __restrict__ tf, float* __restrict__ out, int N, int dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim / 2;
  if (i >= N * half) return;
  const int n = i / half, j = i % half;
  const float tv = t ? (float)t[n] : tf[n];
  const float freq = expf((-9.210340371976184f * (float)j) / (float)half);
  const float a = tv * freq;
  out[(size_t)n * dim + j] = cosf(a);
  out[(size_t)n * dim + half + j] = sinf(a);
  if ((dim & 1) && j == 0) out[(size_t)n * dim + dim - 1] = 0.f;
}

int launch_timestep_embedding(const int64_t* t, const float* tf, float* out, int N, int dim, hipStream_t s) {
  DRM_REQUIRE((t != nullptr) != (tf != nullptr), "timestep_embedding: exactly one of int64 / fp32 timesteps");
  const int total = N * (dim / 2);
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, s, t, tf, out, N, dim);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

// EncoderUNetModel head, pool="adaptive" (openaimodel.py:922-929):
//   GroupNorm -> SiLU -> AdaptiveAvgPool2d(1,1) -> Conv1x1(C -> O) -> Flatten.   One workgroup per sample.
__global__ __launch_bounds__(256) void encoder_head_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ out, int HW, int C, int O) {
  extern __shared__ float pooled[];  // [C] + 4 reduction slots
  float* red = pooled + C;
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < C; c += 256) {
    const float sc = scale[(size_t)n * C + c], sh = shift[(size_t)n * C + c];
    float acc = 0.f;
    for (int p = 0; p < HW; ++p) acc += silu_m(x[((size_t)n * HW + p) * C + c] * sc + sh);
    pooled[c] = acc / (float)HW;
  }
  __syncthreads();
  for (int o = 0; o < O; ++o) {
    float acc = 0.f;
    for (int c = tid; c < C; c += 256) acc += w[(size_t)o * C + c] * pooled[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) out[(size_t)n * O + o] = red[0] + red[1] + red[2] + red[3] + b[o];
    __syncthreads();
  }
}

int launch_encoder_head(const float* x, const float* scale, const float* shift, const float* w, const float* b, float* out, int N, int HW,
                        int C, int O, hipStream_t s) {
  hipLaunchKernelGGL(encoder_head_kernel, dim3(N), dim3(256), (C + 4) * sizeof(float), s, x, scale, shift, w, b, out, HW, C, O);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

// layout converters (tests / per-block entry points)
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int HW, int C) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)N * HW * C) return;
  const int p = (int)(i % HW);
  long long t = i / HW;
  const int c = (int)(t % C);
  const int n = (int)(t / C);
  out[i] = x[((size_t)n * HW + p) * C + c];
}
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int HW, int C) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)N * HW * C) return;
  const int c = (int)(i % C);
  long long t = i / C;
  const int p = (int)(t % HW);
  const int n = (int)(t / HW);
  out[i] = x[((size_t)n * C + c) * HW + p];
}
int launch_nhwc_to_nchw(const float* x, float* out, int N, int H, int W, int C, hipStream_t s) {
  const long long total = (long long)N * H * W * C;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, out, N, H * W, C);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}
int launch_nchw_to_nhwc(const float* x, float* out, int N, int H, int W, int C, hipStream_t s) {
  const long long total = (long long)N * H * W * C;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, out, N, H * W, C);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

}  // namespace drm
