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
// One thread per pixel: the NCHW reads are coalesced along the pixel axis in every channel plane (the previous (pixel, quad)
// mapping read 32-byte fragments: 216 us per launch at B = 32, 3x128x256), the NHWC row of CP floats is written as whole float4s.
// grid (blocks over one image, N): a block never straddles two images, so the optional per-image absmax is a block reduction; every
// block writes its own word of absmax_part[n][gridDim.x] and the consumer (act_pow2_scale_kernel) folds them -- the 128 blocks of an
// image run concurrently, so one atomicMax word per image meant 128 same-address atomics serialised at the memory side.
template <int Q>
__global__ __launch_bounds__(256) void pack_input_kernel(const float* __restrict__ x, const float* __restrict__ cond, const int* __restrict__ idx,
                                                          float4* __restrict__ out, int HW, int Cx, int Cc, unsigned* __restrict__ absmax_part) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = p < HW;
  const int src = idx ? idx[n] : n;
  float v[4 * Q];
#pragma unroll
  for (int c = 0; c < 4 * Q; ++c) {
    float t = 0.f;
    if (live) {
      if (c < Cx) t = x[((size_t)src * Cx + c) * HW + p];
      else if (c < Cx + Cc) t = cond[((size_t)src * Cc + (c - Cx)) * HW + p];
    }
    v[c] = t;
  }
  if (live) {
    float4* o = out + ((size_t)n * HW + p) * Q;
#pragma unroll
    for (int q = 0; q < Q; ++q) o[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
  }
  if (absmax_part) {
    __shared__ float wmax[4];
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < 4 * Q; ++c) m = fmaxf(m, fabsf(v[c]) == fabsf(v[c]) ? fabsf(v[c]) : INFINITY);  // NaN input: no finite bound
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) absmax_part[(size_t)n * gridDim.x + blockIdx.x] = __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));
  }
}

// any padded width (per-module test entry points with wide inputs): one thread per (pixel, 4-channel quad)
__global__ __launch_bounds__(256) void pack_input_generic_kernel(const float* __restrict__ x, const float* __restrict__ cond,
                                                                  const int* __restrict__ idx, float4* __restrict__ out, int HW, int Cx, int Cc, int Q) {
  const int n = blockIdx.y;
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= (long long)HW * Q) return;
  const int q = (int)(j % Q), p = (int)(j / Q);
  const int src = idx ? idx[n] : n;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = 4 * q + k;
    if (c < Cx) v[k] = x[((size_t)src * Cx + c) * HW + p];
    else if (c < Cx + Cc) v[k] = cond[((size_t)src * Cc + (c - Cx)) * HW + p];
  }
  out[(size_t)n * HW * Q + j] = make_float4(v[0], v[1], v[2], v[3]);
}

int pack_input_absmax_parts(int H, int W) { return (H * W + 255) / 256; }

int launch_pack_input(const float* x, const float* cond, const int* idx, float* out, int N, int H, int W, int Cx, int Cc, int CP, hipStream_t s,
                      unsigned* absmax_bits) {
  DRM_REQUIRE(CP % 4 == 0 && CP >= Cx + Cc, "packed input channels must be a multiple of 4 that holds x and cond");
  const int HW = H * W;
  const dim3 grid((unsigned)((HW + 255) / 256), (unsigned)N), block(256);
  float4* o = reinterpret_cast<float4*>(out);
  switch (CP / 4) {
    case 1: hipLaunchKernelGGL(pack_input_kernel<1>, grid, block, 0, s, x, cond, idx, o, HW, Cx, Cc, absmax_bits); break;
    case 2: hipLaunchKernelGGL(pack_input_kernel<2>, grid, block, 0, s, x, cond, idx, o, HW, Cx, Cc, absmax_bits); break;
    case 8: hipLaunchKernelGGL(pack_input_kernel<8>, grid, block, 0, s, x, cond, idx, o, HW, Cx, Cc, absmax_bits); break;
    default:
      DRM_REQUIRE(!absmax_bits, "packed input: the absmax output needs a 4-, 8- or 32-channel padding");
      hipLaunchKernelGGL(pack_input_generic_kernel, dim3((unsigned)(((long long)HW * (CP / 4) + 255) / 256), (unsigned)N), block, 0, s, x, cond, idx, o,
                         HW, Cx, Cc, CP / 4);
  }
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

// Downsample without conv = AvgPool2d(2,2) (openaimodel.py:154-160), NHWC, fused with the per-(image, channel) sums / sums of
// squares of its OUTPUT (the next block's GroupNorm statistics: no stand-alone moments pass over the pooled tensor).
// grid (pixel blocks, N), block 256 = QL channel-quad lanes x PL pixel lanes; a thread walks its pixels with fp32 partial sums and
// ends with one fp64 atomic per (channel, moment).
__global__ __launch_bounds__(256) void avgpool2_kernel(const float4* __restrict__ x, float4* __restrict__ out, double2* __restrict__ stat, int Ho, int Wo,
                                                       int q4, int QL, int px_per_block) {
  __shared__ float red[256][8];  // per-thread partial (4 sums, 4 sums of squares): the pixel lanes of a channel quad are folded here
  const int n = blockIdx.y;
  const int ql = threadIdx.x % QL, pl = threadIdx.x / QL, PL = 256 / QL;
  const int HWo = Ho * Wo, W = Wo * 2;
  const int p0 = blockIdx.x * px_per_block, p1 = min(p0 + px_per_block, HWo);
  // (every thread runs every pass: the statistics fold has barriers; blockIdx.z strides the channel passes when the launch is sparse)
  for (int q0 = blockIdx.z * QL; q0 < q4; q0 += QL * gridDim.z) {
    const int q = q0 + ql;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = p0 + pl; p < p1 && q < q4; p += PL) {
      const int yo = p / Wo, xo = p % Wo;
      const size_t base = (((size_t)n * Ho * 2 + yo * 2) * W + xo * 2) * q4 + q;
      const float4 a = x[base], b = x[base + q4], c = x[base + (size_t)W * q4], d = x[base + (size_t)W * q4 + q4];
      float4 r;
      r.x = (a.x + b.x + c.x + d.x) * 0.25f;
      r.y = (a.y + b.y + c.y + d.y) * 0.25f;
      r.z = (a.z + b.z + c.z + d.z) * 0.25f;
      r.w = (a.w + b.w + c.w + d.w) * 0.25f;
      out[((size_t)n * HWo + p) * q4 + q] = r;
      s[0] += r.x; s[1] += r.y; s[2] += r.z; s[3] += r.w;
      ss[0] += r.x * r.x; ss[1] += r.y * r.y; ss[2] += r.z * r.z; ss[3] += r.w * r.w;
    }
    if (stat) {  // (uniform per block: every thread reaches the barriers)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        red[threadIdx.x][k] = s[k];
        red[threadIdx.x][4 + k] = ss[k];
      }
      __syncthreads();
      if (pl == 0 && q < q4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          double a = 0.0, b = 0.0;
          for (int j = 0; j < PL; ++j) {  // fixed order
            a += (double)red[j * QL + ql][k];
            b += (double)red[j * QL + ql][4 + k];
          }
          double* dd = reinterpret_cast<double*>(stat + (size_t)n * q4 * 4 + 4 * q + k);
          atomicAdd(dd, a);
          atomicAdd(dd + 1, b);
        }
      }
      __syncthreads();
    }
  }
}

int launch_avgpool2(const float* x, float* out, int N, int H, int W, int C, hipStream_t s, double2* stat) {
  DRM_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "avgpool2 shape");
  const int q4 = C / 4, HWo = (H / 2) * (W / 2);
  int QL = q4 >= 64 ? 64 : (q4 >= 32 ? 32 : (q4 >= 16 ? 16 : (q4 >= 8 ? 8 : 4)));
  // One fp64 atomic per (pixel block, channel, moment), 256-pixel blocks on full launches.  A sparse launch (small batch / deep level: the
  // batch-1 step ran its poolings on 1 - 16 workgroups, 18 - 56 us each) first narrows the channel lanes -- more channel passes, each its own
  // workgroup (blockIdx.z), no extra atomics -- and then shortens the pixel blocks, up to 16 per image: the atomics of one (image, channel)
  // serialise in L2 at ~0.18 us each (512 pixel blocks per image measured 91 us).
  int ppb = 256;
  auto blocks = [&]() { return (long long)N * ((HWo + ppb - 1) / ppb) * ((q4 + QL - 1) / QL); };
  while (blocks() < 256 && QL > 8) QL >>= 1;
  while (blocks() < 256 && (HWo + ppb - 1) / ppb < 16 && ppb > 256 / QL) ppb >>= 1;
  const int passes = (q4 + QL - 1) / QL;
  hipLaunchKernelGGL(avgpool2_kernel, dim3((unsigned)((HWo + ppb - 1) / ppb), (unsigned)N, (unsigned)passes), dim3(256), 0, s, reinterpret_cast<const float4*>(x),
                     reinterpret_cast<float4*>(out), stat, H / 2, W / 2, q4, QL, ppb);
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

// Few batch rows (the batch-1 step of scripts/estimate.py): the matrix-core form above walks a weight row with ONE dependent load pair per 16
// features (26 us for the 512 -> 11.7k emb_layers product at N = 1); here a wave owns one output feature, its 64 lanes read the weight row as
// coalesced float4 and meet in a shuffle tree, so a launch has O waves in flight and streams the weights once.  LN_ROWS batch rows per pass.
constexpr int LN_ROWS = 4;
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ b,
                                                          float* __restrict__ out, int N, int I, int O, int silu_in, int silu_out) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (o >= O) return;
  const float4* wr = reinterpret_cast<const float4*>(w + (size_t)o * I);
  for (int n0 = 0; n0 < N; n0 += LN_ROWS) {
    float acc[LN_ROWS];
#pragma unroll
    for (int j = 0; j < LN_ROWS; ++j) acc[j] = 0.f;
    for (int q = lane; q < I / 4; q += 64) {
      const float4 wv = wr[q];
#pragma unroll
      for (int j = 0; j < LN_ROWS; ++j) {
        if (n0 + j < N) {
          float4 v = reinterpret_cast<const float4*>(in + (size_t)(n0 + j) * I)[q];
          if (silu_in) {
            v.x = silu_m(v.x); v.y = silu_m(v.y); v.z = silu_m(v.z); v.w = silu_m(v.w);
          }
          acc[j] += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < LN_ROWS; ++j) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc[j] += __shfl_xor(acc[j], off);
    }
    if (lane == 0) {
      const float bias = b ? b[o] : 0.f;
#pragma unroll
      for (int j = 0; j < LN_ROWS; ++j) {
        if (n0 + j < N) {
          float v = acc[j] + bias;
          if (silu_out) v = silu_m(v);
          out[(size_t)(n0 + j) * O + o] = v;
        }
      }
    }
  }
}

int launch_linear(const float* in, const float* w, const float* b, float* out, int N, int I, int O, int silu_in, int silu_out, hipStream_t s) {
  if (N <= LN_ROWS && I % 16 == 0) {  // (same 16-byte row alignment as the matrix-core form)
    hipLaunchKernelGGL(linear_rows_kernel, dim3((O + 3) / 4), dim3(256), 0, s, in, w, b, out, N, I, O, silu_in, silu_out);
    DRM_HIP_CHECK(hipGetLastError());
    return DRM_OK;
  }
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
  extern __shared__ float pooled[];  // [C] pooled activations + [4][C] pixel-lane partial sums
  float* part = pooled + C;
  const int n = blockIdx.x, tid = threadIdx.x;
  // 64 channel lanes x 4 pixel lanes: a thread walks every 4th pixel of its channel (one serial chain of HW dependent loads per
  // thread made this kernel 80 us at HW = 512); the four partial sums of a channel are added in a fixed order.  One barrier for
  // all channel passes and none in the projection (a wave per output feature): 25 -> 16 us in the batch-1 step.
  const int cl = tid & 63, pl = tid >> 6;
  const float4* x4 = reinterpret_cast<const float4*>(x + (size_t)n * HW * C);
  const float4* sc4 = reinterpret_cast<const float4*>(scale + (size_t)n * C);
  const float4* sh4 = reinterpret_cast<const float4*>(shift + (size_t)n * C);
  const int q4 = C >> 2;
  for (int q = cl; q < q4; q += 64) {  // a channel quad per lane: 16-byte loads, C / 256 rounds of load latency instead of C / 64
    const float4 sc = sc4[q], sh = sh4[q];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int p = pl; p < HW; p += 4) {
      const float4 v = x4[(size_t)p * q4 + q];
      acc.x += silu_m(v.x * sc.x + sh.x);
      acc.y += silu_m(v.y * sc.y + sh.y);
      acc.z += silu_m(v.z * sc.z + sh.z);
      acc.w += silu_m(v.w * sc.w + sh.w);
    }
    *reinterpret_cast<float4*>(&part[pl * C + 4 * q]) = acc;
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) pooled[c] = (((part[c] + part[C + c]) + part[2 * C + c]) + part[3 * C + c]) / (float)HW;
  __syncthreads();
  for (int o = pl; o < O; o += 4) {
    float acc = 0.f;
    for (int c = cl; c < C; c += 64) acc += w[(size_t)o * C + c] * pooled[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (cl == 0) out[(size_t)n * O + o] = acc + b[o];
  }
}

int launch_encoder_head(const float* x, const float* scale, const float* shift, const float* w, const float* b, float* out, int N, int HW,
                        int C, int O, hipStream_t s) {
  DRM_REQUIRE(C % 4 == 0, "encoder head: C % 4");
  hipLaunchKernelGGL(encoder_head_kernel, dim3(N), dim3(256), (size_t)5 * C * sizeof(float), s, x, scale, shift, w, b, out, HW, C, O);
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
