// Sampler loops on the device: DRMNet's residual reverse process, DDIM and ancestral DDPM.
//
// Reference semantics restated (paths relative to the reference root):
//   DRMNet.p_sample_loop / forward / get_brdf_out / get_schedule / check_convergence
//       models/drmnet.py:782-847, :452-456, :390-396, :458-501, :747-750
//   DDIMSampler.ddim_sampling / p_sample_ddim     ldm/models/diffusion/ddim.py:128-259
//   LatentDiffusion.p_sample / p_mean_variance    ldm/models/diffusion/ddpm.py:1079-1167
//   ObsNetDiffusion.p_sample_loop                 models/obsnet.py:500-564
// Each sampler step is: U-Net forward(s) (engine) + ONE fused elementwise update kernel here.
// Noise is either an external buffer (parity mode; the reference's draws are data dependent) or the
// library's counter-based Philox4x32-10 stream (throughput mode), selected by a null pointer.
#include "samplers.h"
#include "profiler.h"

#include <atomic>
#include <cmath>
#include <cstring>
#include <vector>

namespace drm {

// ------------------------------------------------------------------------------------------------ Philox4x32-10
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
  const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
  const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ float4 philox_normal4(uint64_t seed, uint64_t ctr) {
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const float s = 2.3283064365386963e-10f;  // 2^-32
  const float u0 = ((float)c[0] + 0.5f) * s, u1 = ((float)c[1] + 0.5f) * s;
  const float u2 = ((float)c[2] + 0.5f) * s, u3 = ((float)c[3] + 0.5f) * s;
  const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
  float s0, c0, s1, c1;
  sincosf(6.283185307179586f * u1, &s0, &c0);
  sincosf(6.283185307179586f * u3, &s1, &c1);
  return make_float4(r0 * c0, r0 * s0, r1 * c1, r1 * s1);
}
__device__ __forceinline__ float philox_normal1(uint64_t seed, uint64_t elem) {
  const float4 v = philox_normal4(seed, elem >> 2);
  const int k = (int)(elem & 3);
  return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w));
}

// one uniform in (0, 1) per element (dropout masks): the same counter-based stream, without the Box-Muller step
__device__ __forceinline__ float philox_uniform1(uint64_t seed, uint64_t elem) {
  const uint64_t ctr = elem >> 2;
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return ((float)c[elem & 3] + 0.5f) * 2.3283064365386963e-10f;
}

__global__ void randn_kernel(float* __restrict__ out, size_t n, uint64_t seed, uint64_t offset) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = philox_normal1(seed, offset + i);
}
int launch_randn(float* out, size_t n, uint64_t seed, uint64_t offset, hipStream_t s) {
  if (n == 0) return DRM_OK;
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, n, seed, offset);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

__global__ void fill_f32_kernel(float* p, size_t n, float v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void fill_i32_kernel(int32_t* p, size_t n, int32_t v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
static int fill_f32(float* p, size_t n, float v, hipStream_t s) {
  hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}
static int fill_i32(int32_t* p, size_t n, int32_t v, hipStream_t s) {
  hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

// ------------------------------------------------------------------------------------------------ DRMNet kernels

// Lr_k = LrK + delta * eps0   (models/drmnet.py:796-797)
__global__ void drmnet_init_kernel(const float* __restrict__ LrK, const float* __restrict__ noise0, float* __restrict__ Lr_k, size_t n,
                                   float delta, uint64_t seed) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float e = noise0 ? noise0[i] : philox_normal1(seed, i);
  Lr_k[i] = LrK[i] + delta * e;
}

// get_brdf_out (eval) + check_convergence: zk = clamp(z0 + gpow (z_out - z0), 0, 1); zK = clamp(z_out, 0, 1);
// dz = zk - z0; conv = ||zk - z0||_2 < eps or == 0.  gpow = float(exp(i * ln(gamma))) evaluated in fp64 on the host
// (models/drmnet.py:494-495).
__global__ void drmnet_brdf_kernel(const float* __restrict__ z_out, const float* __restrict__ z0, int n, int zd, float gpow, float eps,
                                   float* __restrict__ zk, float* __restrict__ dz, float* __restrict__ zKc, int32_t* __restrict__ conv) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  float d2 = 0.f;
  for (int k = 0; k < zd; ++k) {
    const float zo = z_out[j * zd + k], z = z0[k];
    float v = gpow * (zo - z) + z;
    v = fminf(fmaxf(v, 0.f), 1.f);
    const float d = v - z;
    zk[j * zd + k] = v;
    dz[j * zd + k] = d;
    zKc[j * zd + k] = fminf(fmaxf(zo, 0.f), 1.f);
    d2 += fabsf(d) * fabsf(d);
  }
  const float dist = sqrtf(d2);
  conv[j] = (dist < eps || dist == 0.f) ? 1 : 0;
}

// Lr_k[rows[j]] += out[j] (+ delta * noise[rows[j]] unless converged)   (models/drmnet.py:763,822-825)
__global__ void drmnet_update_kernel(float* __restrict__ Lr_k, const float* __restrict__ out, const int32_t* __restrict__ rows,
                                     const int32_t* __restrict__ conv, const float* __restrict__ noise, int n, size_t chw, float delta,
                                     uint64_t seed, uint64_t noise_off, int row0) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n * chw) return;
  const int j = (int)(i / chw);
  const size_t e = i % chw;
  const int row = rows ? rows[j] : row0 + j;
  const size_t g = (size_t)row * chw + e;
  float v = Lr_k[g] + out[i];
  if (!conv[j]) {
    const float nz = noise ? noise[g] : philox_normal1(seed, noise_off + g);
    v += nz * delta;
  }
  Lr_k[g] = v;
}

// K[rows[j]] = step + 1; zK[rows[j]] = zKc[j] for rows that converged this step (models/drmnet.py:836-838)
__global__ void drmnet_record_kernel(const int32_t* __restrict__ rows, const int32_t* __restrict__ conv, const float* __restrict__ zKc, int n,
                                     int zd, int step, float* __restrict__ zK, int32_t* __restrict__ K) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n || !conv[j]) return;
  const int row = rows ? rows[j] : j;
  K[row] = step + 1;
  for (int k = 0; k < zd; ++k) zK[row * zd + k] = zKc[j * zd + k];
}

// ------------------------------------------------------------------------------------------------ DDIM / DDPM kernels

// DDIM / DDPM loops keep every per-step scalar in a DEVICE table (row j: timestep, five coefficients) indexed by a device
// counter, so the launches of a step do not depend on j and one captured hipGraph of a step can be replayed for the whole chain
// (BASELINE configs[2]: "hipGraph-captured step").  Table row layout: STEP_ROW floats.
constexpr int STEP_ROW = 8;  // [0] timestep, [1..5] coefficients, [6] flag (DDPM: t > 0), [7] DDIM: 1 + slot of the intermediates log this step writes (0 = none)
constexpr int MAX_TABLE_STEPS = 4096;  // sampler_workspace_bytes budgets the table for this many steps; the chain entry points check it

__global__ void step_begin_kernel(const float* __restrict__ tab, const int* __restrict__ counter, float* __restrict__ tf, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) tf[i] = tab[(size_t)(*counter) * STEP_ROW];
}
__global__ void step_advance_kernel(int* counter) { *counter += 1; }

// pred_x0 = (x - sqrt(1-a_t) e) / sqrt(a_t);  x = sqrt(a_prev) pred_x0 + sqrt(1-a_prev-s^2) e + s * noise   (ddim.py:249-258)
// log_x / log_pred (optional): the reference's `intermediates` (ddim.py:171-204): slots of n floats, written by the steps the table marks
// noise_dropout (ddim.py:256-257, ddpm.py:1158-1159: F.dropout on the step noise): an element is kept with probability 1 - p and scaled by 1 / (1 - p);
// keep: [steps][n] 0 / 1 masks (parity runs) or null -> Philox under a key of its own
__device__ __forceinline__ float dropout_factor(float p, const float* __restrict__ keep, int j, size_t n, size_t i, uint64_t seed) {
  if (p <= 0.f) return 1.0f;
  const bool k = keep ? keep[(size_t)j * n + i] != 0.f : philox_uniform1(seed ^ 0xD1B54A32D192ED03ull, (uint64_t)(j + 1) * n + i) >= p;
  return k ? 1.0f / (1.0f - p) : 0.f;
}

__global__ void ddim_update_kernel(float* __restrict__ x, const float* __restrict__ e, const float* __restrict__ noise, size_t n,
                                   const float* __restrict__ tab, const int* __restrict__ counter, uint64_t seed, float* __restrict__ log_x,
                                   float* __restrict__ log_pred, float drop_p, const float* __restrict__ drop_keep) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int j = *counter;
  const float* c = tab + (size_t)j * STEP_ROW + 1;
  const float sa = c[0], s1m = c[1], sap = c[2], sdir = c[3], sig = c[4];
  const float xv = x[i], ev = e[i];
  const float pred = (xv - s1m * ev) / sa;
  float nz = 0.f;
  if (sig != 0.f) nz = (noise ? noise[(size_t)j * n + i] : philox_normal1(seed, (uint64_t)(j + 1) * n + i)) * dropout_factor(drop_p, drop_keep, j, n, i, seed);
  const float xn = sap * pred + sdir * ev + sig * nz;
  x[i] = xn;
  const int slot = (int)c[6];  // (row element 7)
  if (log_x && slot > 0) {
    log_x[(size_t)(slot - 1) * n + i] = xn;
    log_pred[(size_t)(slot - 1) * n + i] = pred;
  }
}

// x_recon = c0 x - c1 e; mean = c2 x_recon + c3 x; x = mean + [t>0] c4 noise   (ddpm.py:233-246,1156-1167)
__global__ void ddpm_update_kernel(float* __restrict__ x, float* __restrict__ pred_x0, const float* __restrict__ e,
                                   const float* __restrict__ noise, size_t n, const float* __restrict__ tab, const int* __restrict__ counter,
                                   int clip, uint64_t seed, float drop_p, const float* __restrict__ drop_keep) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int j = *counter;
  const float* c = tab + (size_t)j * STEP_ROW + 1;
  const float xv = x[i];
  float xr = c[0] * xv - c[1] * e[i];
  if (clip) xr = fminf(fmaxf(xr, -1.f), 1.f);
  const float mean = c[2] * xr + c[3] * xv;
  float v = mean;
  if (c[5] != 0.f) {
    const float nz = (noise ? noise[(size_t)j * n + i] : philox_normal1(seed, (uint64_t)(j + 1) * n + i)) * dropout_factor(drop_p, drop_keep, j, n, i, seed);
    v += c[4] * nz;
  }
  x[i] = v;
  if (pred_x0) pred_x0[i] = xr;
}

// classifier-free guidance (ddim.py:225-232): e = e_uncond + scale * (e_cond - e_uncond); the reference evaluates both in one batch of 2 N rows
// (rows do not interact: two forwards give the same numbers)
__global__ void cfg_combine_kernel(float* __restrict__ e, const float* __restrict__ e_uncond, float scale, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float u = e_uncond[i];
    e[i] = u + scale * (e[i] - u);
  }
}

// Known-region blending of the samplers' `mask` / `x0` arguments (inpainting-style conditioning):
//   img = q_sample(x0, t) * mask + (1 - mask) * img,   q_sample(x0, t) = sqrt(a_bar_t) x0 + sqrt(1 - a_bar_t) noise   (ddpm.py:1300-1302, :1052-1058)
// applied BEFORE the step's network forward by DDIMSampler.ddim_sampling (at the step's own t, ddim.py:175-178) and by ObsNetDiffusion.p_sample_loop
// (x0 itself at t == 0, else q_sample at t - 1, models/obsnet.py:545-547), and AFTER the update by LatentDiffusion.p_sample_loop (ddpm.py:1300-1302):
// the host builds the (a, b) pair of every step, the kernel reads row `*counter` like the update kernels do, so graph replay applies unchanged.
// mask: [N, mask_c, H, W] with mask_c == 1 (broadcast over the channels) or == C.  q-noise: [steps][n] injected, or Philox under its own key.
__global__ void mask_blend_kernel(float* __restrict__ x, const float* __restrict__ x0, const float* __restrict__ mask, int mask_c, int C, size_t hw,
                                  const float* __restrict__ qnoise, size_t n, const float* __restrict__ qtab, const int* __restrict__ counter, uint64_t seed) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int j = *counter;
  const float a = qtab[2 * (size_t)j], b = qtab[2 * (size_t)j + 1];
  float nz = 0.f;
  if (b != 0.f) nz = qnoise ? qnoise[(size_t)j * n + i] : philox_normal1(seed ^ 0x9E3779B97F4A7C15ull, (uint64_t)(j + 1) * n + i);
  const size_t p = i % hw, ch = (i / hw) % C, img = i / (hw * C);
  const float m = mask[(img * mask_c + (mask_c == 1 ? 0 : ch)) * hw + p];
  x[i] = (a * x0[i] + b * nz) * m + (1.0f - m) * x[i];
}

// ------------------------------------------------------------------------------------------------ DRMNet sampler

DrmnetSampler::~DrmnetSampler() {
  if (zemb) (void)hipFree(zemb);
  if (z0_dev) (void)hipFree(z0_dev);
  if (h_rows) (void)hipHostFree(h_rows);
  if (h_conv) (void)hipHostFree(h_conv);
  for (int k = 0; k < PART_MAX; ++k) {
    if (part_stream[k]) (void)hipStreamDestroy(part_stream[k]);
    if (part_done[k]) (void)hipEventDestroy(part_done[k]);
  }
  if (part_fork) (void)hipEventDestroy(part_fork);
}

int DrmnetSampler::init(UNet* ill, UNet* ref, const float* const* zw, const drm_drmnet_cfg& c) {
  DRM_REQUIRE(ill && ref && ill->desc.kind == 0 && ref->desc.kind == 1, "drmnet needs a UNetModel (IllNet) and an EncoderUNetModel (RefNet)");
  DRM_REQUIRE(c.z_dim >= 1 && c.z_dim <= 8 && ref->desc.out_channels == c.z_dim, "z_dim must match RefNet out_channels (<= 8)");
  DRM_REQUIRE(c.max_timesteps >= 1, "max_timesteps");
  illnet = ill; refnet = ref; cfg = c;
  const int mc = ill->desc.model_channels, hd = mc / 2;
  const size_t sizes[6] = {(size_t)hd * c.z_dim, (size_t)hd, (size_t)hd * hd, (size_t)hd, (size_t)mc * hd, (size_t)mc};
  size_t total = 0;
  for (int i = 0; i < 6; ++i) { zoff[i] = total; total += (sizes[i] + 63) & ~size_t(63); }
  DRM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&zemb), total * sizeof(float)));
  for (int i = 0; i < 6; ++i) {
    DRM_REQUIRE(zw[i] != nullptr, "null z_emb_layer parameter");
    DRM_HIP_CHECK(hipMemcpy(zemb + zoff[i], zw[i], sizes[i] * sizeof(float), hipMemcpyDeviceToDevice));
  }
  DRM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&z0_dev), 8 * sizeof(float)));
  DRM_HIP_CHECK(hipMemcpy(z0_dev, c.z0, 8 * sizeof(float), hipMemcpyHostToDevice));
  return DRM_OK;
}

struct StepBuffers {
  float *z_out, *zk, *dz, *zKc, *h1, *h2, *temb, *tf, *eps;
  int32_t* conv;
};
static StepBuffers step_buffers(Arena& ar, int n, int zd, int mc, size_t chw) {
  StepBuffers b;
  b.z_out = ar.alloc<float>((size_t)n * zd);
  b.zk = ar.alloc<float>((size_t)n * zd);
  b.dz = ar.alloc<float>((size_t)n * zd);
  b.zKc = ar.alloc<float>((size_t)n * zd);
  b.h1 = ar.alloc<float>((size_t)n * (mc / 2));
  b.h2 = ar.alloc<float>((size_t)n * (mc / 2));
  b.temb = ar.alloc<float>((size_t)n * mc);
  b.tf = ar.alloc<float>((size_t)n);
  b.conv = ar.alloc<int32_t>((size_t)n);
  b.eps = ar.alloc<float>((size_t)n * chw);
  return b;
}

// workspace bytes one batch part of nmax rows takes (the larger of the two networks' forwards; memoised: a dry walk of both networks)
size_t DrmnetSampler::part_need(int nmax, int H, int W) const {
  for (const auto& e : part_need_memo)
    if (e.n == nmax && e.H == H && e.W == W) return e.bytes;
  Arena p1; p1.dry = true;
  Arena p2; p2.dry = true;
  if (illnet->forward(nullptr, 3, nullptr, 3, nullptr, nullptr, nullptr, nullptr, nullptr, nmax, H, W, p1, nullptr) != DRM_OK) return ~size_t(0);
  if (refnet->forward(nullptr, 3, nullptr, 3, nullptr, nullptr, nullptr, nullptr, nullptr, nmax, H, W, p2, nullptr) != DRM_OK) return ~size_t(0);
  const size_t need = (std::max(p1.peak, p2.peak) + 255) & ~size_t(255);
  if (part_need_memo.size() >= 64) part_need_memo.clear();
  part_need_memo.push_back({nmax, H, W, need});
  return need;
}

size_t DrmnetSampler::workspace_bytes(int N, int H, int W) const {
  Arena ar;
  ar.dry = true;
  step_buffers(ar, N, cfg.z_dim, illnet->desc.model_channels, (size_t)3 * H * W);
  ar.alloc<int32_t>((size_t)N);  // rows
  ar.alloc_bytes(0);
  const size_t base = (ar.peak + 255) & ~size_t(255);
  Arena a1; a1.dry = true;
  Arena a2; a2.dry = true;
  if (illnet->forward(nullptr, 3, nullptr, 3, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, a1, nullptr) != DRM_OK) return 0;
  if (refnet->forward(nullptr, 3, nullptr, 3, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, a2, nullptr) != DRM_OK) return 0;
  size_t need = std::max(a1.peak, a2.peak);
  for (int np = 2; np <= std::min(std::min(parts, (int)PART_MAX), N / std::max(part_min, 1)); ++np) {  // the batch parts' workspace slices (step)
    const size_t pn = part_need((N + np - 1) / np, H, W);
    if (pn == ~size_t(0)) return 0;
    need = std::max(need, (size_t)np * (pn + 256));
  }
  return base + need + 512;
}

// One reverse step on rows[0..n) (device indices, or null = identity)
int DrmnetSampler::step(float* Lr_k, const float* LrK, const int32_t* rows, int n, int i, const float* noise, uint64_t seed, float* zk_out,
                        float* zK_out, int32_t* conv_out, int B, int H, int W, Arena& ar, hipStream_t s) {
  DRM_REQUIRE(n >= 1 && n <= B, "n_active");
  const int zd = cfg.z_dim, mc = illnet->desc.model_channels;
  const size_t chw = (size_t)3 * H * W;
  StepBuffers b = step_buffers(ar, n, zd, mc, chw);  // per-row buffers of the whole step: a batch part works on its row range of them
  if (ar.failed) { set_error("drmnet step: workspace too small"); return DRM_ERR_WORKSPACE; }
  // (not under the launch profiler: it brackets every launch with events on the launch stream, and a bracket on one part's stream would include
  //  the other part's kernels -- the profiled pass runs the whole batch on the caller's stream)
  int np = prof_enabled() ? 1 : std::min(std::min(parts, (int)PART_MAX), n / std::max(part_min, 1));
  // the parts' workspace slices: 256-byte aligned starts behind the step's own buffers, each as large as the largest part needs.  A workspace sized
  // before drm_drmnet_set_batch_parts raised the part count (or by hand) may not hold them: then the step runs on the caller's stream alone.
  const size_t a0 = (ar.off + 255) & ~size_t(255);
  size_t room = 0;
  if (np >= 2 && !ar.dry) {
    room = a0 < ar.cap ? ((ar.cap - a0) / np) & ~size_t(255) : 0;
    if (room < part_need((n + np - 1) / np, H, W)) np = 1;
  }
  if (np < 2 || ar.dry) {
    DRM_TRY(step_rows(Lr_k, LrK, rows, 0, n, i, noise, seed, b, 0, B, H, W, ar, s));
  } else {
    // fork: every part waits for the caller's stream, runs its rows on its own stream and workspace slice, and the caller's stream waits for all
    if (!part_fork) DRM_HIP_CHECK(hipEventCreateWithFlags(&part_fork, hipEventDisableTiming));
    for (int k = 0; k < np; ++k) {
      if (!part_stream[k]) DRM_HIP_CHECK(hipStreamCreateWithFlags(&part_stream[k], hipStreamNonBlocking));
      if (!part_done[k]) DRM_HIP_CHECK(hipEventCreateWithFlags(&part_done[k], hipEventDisableTiming));
    }
    DRM_HIP_CHECK(hipEventRecord(part_fork, s));
    int rc = DRM_OK;
    for (int k = 0; k < np && rc == DRM_OK; ++k) {
      const int j0 = (int)((long long)n * k / np), j1 = (int)((long long)n * (k + 1) / np);
      Arena sub;
      sub.base = ar.base + a0 + (size_t)k * room;
      sub.cap = room;
      hipStream_t ps = part_stream[k];
      DRM_HIP_CHECK(hipStreamWaitEvent(ps, part_fork, 0));
      rc = step_rows(Lr_k, LrK, rows ? rows + j0 : nullptr, rows ? 0 : j0, j1 - j0, i, noise, seed, b, j0, B, H, W, sub, ps);
      if (sub.failed && rc == DRM_OK) { set_error("drmnet step: workspace too small for the batch parts"); rc = DRM_ERR_WORKSPACE; }
      (void)hipEventRecord(part_done[k], ps);  // (joined even after a failed part: the caller's stream must not run ahead of work already queued)
      (void)hipStreamWaitEvent(s, part_done[k], 0);
    }
    if (rc != DRM_OK) return rc;
  }
  if (zk_out) DRM_HIP_CHECK(hipMemcpyAsync(zk_out, b.zk, (size_t)n * zd * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (zK_out) DRM_HIP_CHECK(hipMemcpyAsync(zK_out, b.zKc, (size_t)n * zd * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (conv_out) DRM_HIP_CHECK(hipMemcpyAsync(conv_out, b.conv, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  last = b.conv;
  last_zKc = b.zKc;
  return DRM_OK;
}

// rows [j0, j0 + n) of a step's row list (rows: that range of the device indices, or null = the identity rows row0 .. row0 + n)
int DrmnetSampler::step_rows(float* Lr_k, const float* LrK, const int32_t* rows, int row0, int n, int i, const float* noise, uint64_t seed,
                             const StepBuffers& bb, int j0, int B, int H, int W, Arena& ar, hipStream_t s) {
  const int zd = cfg.z_dim, mc = illnet->desc.model_channels, hd = mc / 2;
  const size_t chw = (size_t)3 * H * W;
  StepBuffers b = bb;
  b.z_out += (size_t)j0 * zd; b.zk += (size_t)j0 * zd; b.dz += (size_t)j0 * zd; b.zKc += (size_t)j0 * zd;
  b.h1 += (size_t)j0 * hd; b.h2 += (size_t)j0 * hd; b.temb += (size_t)j0 * mc; b.tf += j0; b.conv += j0; b.eps += (size_t)j0 * chw;
  float* x = Lr_k + (size_t)row0 * chw;  // identity rows: the range starts at row0 (row lists index the whole tensors)
  const float* c = LrK + (size_t)row0 * chw;
  const size_t net_mark = ar.mark();
  // RefNet(cat[Lr_k, LrK], timesteps = i)     (models/drmnet.py:453, :376-388)
  DRM_TRY(fill_f32(b.tf, n, (float)i, s));
  DRM_TRY(refnet->forward(x, 3, c, 3, rows, nullptr, nullptr, b.tf, b.z_out, n, H, W, ar, s));
  ar.release(net_mark);
  const float gpow = (float)std::exp((double)i * std::log(cfg.gamma));
  hipLaunchKernelGGL(drmnet_brdf_kernel, dim3((n + 63) / 64), dim3(64), 0, s, b.z_out, z0_dev, n, zd, gpow, cfg.epsilon, b.zk, b.dz, b.zKc, b.conv);
  DRM_HIP_CHECK(hipGetLastError());
  // z_emb_layer(zk - z0): Linear-SiLU x3   (models/drmnet.py:38-45,55)
  DRM_TRY(launch_linear(b.dz, zemb + zoff[0], zemb + zoff[1], b.h1, n, zd, hd, 0, 1, s));
  DRM_TRY(launch_linear(b.h1, zemb + zoff[2], zemb + zoff[3], b.h2, n, hd, hd, 0, 1, s));
  DRM_TRY(launch_linear(b.h2, zemb + zoff[4], zemb + zoff[5], b.temb, n, hd, mc, 0, 1, s));
  // IllNet(cat[Lr_k, LrK], t_emb)             (models/drmnet.py:455-456, :54-61)
  DRM_TRY(illnet->forward(x, 3, c, 3, rows, b.temb, nullptr, nullptr, b.eps, n, H, W, ar, s));
  ar.release(net_mark);
  const size_t total = (size_t)n * chw;
  hipLaunchKernelGGL(drmnet_update_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, Lr_k, b.eps, rows, b.conv, noise, n, chw,
                     cfg.delta, seed, (uint64_t)(i + 1) * (uint64_t)B * chw, row0);
  DRM_HIP_CHECK(hipGetLastError());
  return DRM_OK;
}

int DrmnetSampler::sample(const float* LrK, const float* cond, const float* noise0, const float* step_noise, uint64_t seed, int early_exit, float* Lr0, float* zK,
                          int32_t* K, int32_t* steps_done, int B, int H, int W, Arena& ar, hipStream_t s) {
  const int zd = cfg.z_dim;
  const size_t chw = (size_t)3 * H * W;
  if (h_cap < B) {
    if (h_rows) (void)hipHostFree(h_rows);
    if (h_conv) (void)hipHostFree(h_conv);
    DRM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_rows), (size_t)B * sizeof(int32_t)));
    DRM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_conv), (size_t)B * sizeof(int32_t)));
    h_cap = B;
  }
  int32_t* d_rows = ar.alloc<int32_t>((size_t)B);
  if (ar.failed) { set_error("drmnet sample: workspace too small"); return DRM_ERR_WORKSPACE; }
  const size_t mark = ar.mark();
  const size_t total = (size_t)B * chw;
  hipLaunchKernelGGL(drmnet_init_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, LrK, noise0, Lr0, total, cfg.delta, seed);
  DRM_HIP_CHECK(hipGetLastError());
  DRM_TRY(fill_f32(zK, (size_t)B * zd, NAN, s));
  DRM_TRY(fill_i32(K, (size_t)B, cfg.max_timesteps, s));
  std::vector<int32_t> active(B);
  for (int b = 0; b < B; ++b) active[b] = b;
  int steps = 0;
  for (int i = 0; i < cfg.max_timesteps; ++i) {
    const int n = (int)active.size();
    for (int j = 0; j < n; ++j) h_rows[j] = active[j];
    DRM_HIP_CHECK(hipMemcpyAsync(d_rows, h_rows, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    const float* nz = step_noise ? step_noise + (size_t)i * B * chw : nullptr;
    ar.release(mark);
    DRM_TRY(step(Lr0, cond, d_rows, n, i, nz, seed, nullptr, nullptr, nullptr, B, H, W, ar, s));
    ++steps;
    if (early_exit) {
      hipLaunchKernelGGL(drmnet_record_kernel, dim3((n + 63) / 64), dim3(64), 0, s, d_rows, last, last_zKc, n, zd, i, zK, K);
      DRM_HIP_CHECK(hipGetLastError());
      DRM_HIP_CHECK(hipMemcpyAsync(h_conv, last, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
      DRM_HIP_CHECK(hipStreamSynchronize(s));  // the reference syncs here too (torch.any, models/drmnet.py:841)
      std::vector<int32_t> next;
      next.reserve(n);
      for (int j = 0; j < n; ++j)
        if (!h_conv[j]) next.push_back(active[j]);
      active.swap(next);
      if (active.empty()) break;
    }
  }
  if (steps_done) *steps_done = steps;
  return DRM_OK;
}

// ------------------------------------------------------------------------------------------------ DDIM / DDPM drivers

size_t sampler_workspace_bytes(UNet* net, int N, int H, int W) {
  Arena a;
  a.dry = true;
  if (net->forward(nullptr, 3, nullptr, 3, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, a, nullptr) != DRM_OK) return 0;
  const size_t chw = (size_t)net->desc.out_channels * H * W;
  // U-Net arena + eps [N,C,H,W] + timesteps [N] + the per-step scalar table (<= 4096 steps) and its counter
  // (+ the mask-blend (a, b) pair per step, same bound)
  // (+ a second eps buffer: the unconditional branch of classifier-free guidance)
  return a.peak + 2 * ((size_t)N * chw * sizeof(float) + 256) + ((size_t)N * sizeof(float) + 256) + ((size_t)MAX_TABLE_STEPS * (STEP_ROW + 2) * sizeof(float) + 512) + 1024;
}

// Runs `steps` identical-launch steps: the first eagerly (it also sizes caches and sets per-kernel attributes), the second under
// stream capture, the rest as replays of that graph.  Falls back to eager launches when replay is switched off, the launch
// profiler is recording (its events do not belong in a graph) or the chain is too short to pay for an instantiation.
static std::atomic<bool> g_graph_replay{false};  // measured: no faster at B = 32 / 256 (kernels already cover ~95 % of the wall time), slower at B = 1
static std::atomic<long long> g_graph_launches{0};
void set_graph_replay(bool on) { g_graph_replay.store(on); }
long long graph_launches() { return g_graph_launches.load(); }

static bool graph_wanted(int steps) { return g_graph_replay.load() && !prof_enabled() && steps >= 4; }

// Per-(thread, device) helper objects: a private capture stream + ordering event, and a pinned staging buffer for the step tables.
struct ThreadDev {
  hipStream_t priv = nullptr;
  hipEvent_t ev = nullptr;
  void* pinned = nullptr;       // step-table staging (host, page-locked)
  size_t pinned_cap = 0;
  hipEvent_t pinned_ev = nullptr;  // recorded behind the last copy out of `pinned`
  bool pinned_busy = false;
};
static ThreadDev* thread_dev() {
  constexpr int MAX_DEV = 64;
  static thread_local ThreadDev td[MAX_DEV];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) {
    set_error("sampler: no current HIP device");
    return nullptr;
  }
  return &td[dev];
}

// The legacy default stream (what PyTorch hands over unless the caller set a stream) cannot be captured: a chain that will be
// replayed runs on a private non-blocking stream ordered behind the caller's stream by an event; run_steps drains it before it
// returns, so work the caller enqueues afterwards on its own stream is ordered behind the chain.
static int chain_stream(hipStream_t caller, int steps, hipStream_t* out) {
  *out = caller;
  if (caller != nullptr || !graph_wanted(steps)) return DRM_OK;
  ThreadDev* t = thread_dev();
  if (!t) return DRM_ERR_STATE;
  if (!t->priv) {
    DRM_HIP_CHECK(hipStreamCreateWithFlags(&t->priv, hipStreamNonBlocking));
    DRM_HIP_CHECK(hipEventCreateWithFlags(&t->ev, hipEventDisableTiming));
  }
  DRM_HIP_CHECK(hipEventRecord(t->ev, caller));
  DRM_HIP_CHECK(hipStreamWaitEvent(t->priv, t->ev, 0));
  *out = t->priv;
  return DRM_OK;
}

template <typename Body>
static int run_steps(int steps, hipStream_t s, Body&& body) {
  const bool use_graph = graph_wanted(steps);
  if (!use_graph) {
    for (int j = 0; j < steps; ++j) DRM_TRY(body());
    return DRM_OK;
  }
  DRM_TRY(body());
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  DRM_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  const int rc = body();
  const hipError_t ec = hipStreamEndCapture(s, &graph);
  if (rc != DRM_OK) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
  }
  DRM_HIP_CHECK(ec);
  hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (ei != hipSuccess) {
    (void)hipGraphDestroy(graph);
    DRM_HIP_CHECK(ei);
  }
  int status = DRM_OK;
  for (int j = 1; j < steps && status == DRM_OK; ++j) {
    if (hipGraphLaunch(exec, s) != hipSuccess) {
      set_error("hipGraphLaunch failed while replaying a sampler step");
      status = DRM_ERR_HIP;
    }
    g_graph_launches.fetch_add(1);
  }
  // the executable graph must outlive its last launch: the stream is drained before it is destroyed
  if (hipStreamSynchronize(s) != hipSuccess && status == DRM_OK) {
    set_error("stream error while replaying sampler steps");
    status = DRM_ERR_HIP;
  }
  (void)hipGraphExecDestroy(exec);
  (void)hipGraphDestroy(graph);
  return status;
}

// uploads the per-step table (host rows -> device) and zeroes the step counter.  The rows go through a page-locked staging buffer
// owned by (thread, device); the only wait is for the PREVIOUS chain's copy out of that buffer, long retired in steady state.
static int upload_step_table(const std::vector<float>& rows, float* tab, int* counter, hipStream_t s) {
  ThreadDev* t = thread_dev();
  if (!t) return DRM_ERR_STATE;
  const size_t bytes = rows.size() * sizeof(float);
  if (t->pinned_busy) {
    DRM_HIP_CHECK(hipEventSynchronize(t->pinned_ev));
    t->pinned_busy = false;
  }
  if (bytes > t->pinned_cap) {
    if (t->pinned) (void)hipHostFree(t->pinned);
    t->pinned = nullptr;
    t->pinned_cap = 0;
    DRM_HIP_CHECK(hipHostMalloc(&t->pinned, bytes, hipHostMallocDefault));
    t->pinned_cap = bytes;
  }
  if (!t->pinned_ev) DRM_HIP_CHECK(hipEventCreateWithFlags(&t->pinned_ev, hipEventDisableTiming));
  memcpy(t->pinned, rows.data(), bytes);
  DRM_HIP_CHECK(hipMemcpyAsync(tab, t->pinned, bytes, hipMemcpyHostToDevice, s));
  DRM_HIP_CHECK(hipEventRecord(t->pinned_ev, s));
  t->pinned_busy = true;
  DRM_HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(int), s));
  return DRM_OK;
}

// uploads the blend's per-step (a, b) pairs behind the step table (same stream, pageable source: hipMemcpyAsync returns after staging it)
static int upload_blend_table(const MaskBlend* blend, int steps, float* qtab, hipStream_t s) {
  DRM_REQUIRE(blend->mask && blend->x0 && blend->qcoef, "mask blending needs mask, x0 and the per-step q_sample coefficients");
  DRM_REQUIRE(blend->mask_channels >= 1 && (blend->when == 0 || blend->when == 1), "mask blending: mask_channels >= 1, when = 0 (before the forward) or 1 (after the update)");
  DRM_HIP_CHECK(hipMemcpyAsync(qtab, blend->qcoef, (size_t)steps * 2 * sizeof(float), hipMemcpyHostToDevice, s));
  DRM_HIP_CHECK(hipStreamSynchronize(s));  // (the caller's table may be freed when the call returns under graph replay / eager alike)
  return DRM_OK;
}

int ddim_sample(UNet* net, float* x, const float* cond, const int64_t* timesteps, const float* coef, int S, int num_steps, const float* noise,
                uint64_t seed, int N, int H, int W, Arena& ar, hipStream_t caller, int log_every_t, float* log_x, float* log_pred, int log_slots,
                int* n_logged, const MaskBlend* blend, const float* uncond, float guidance_scale, float drop_p, const float* drop_keep) {
  DRM_REQUIRE(net && net->desc.kind == 0, "ddim needs a UNetModel");
  DRM_REQUIRE(S >= 1 && timesteps && coef, "ddim schedule");
  DRM_REQUIRE(S <= MAX_TABLE_STEPS, "ddim: at most " + std::to_string(MAX_TABLE_STEPS) + " steps (the workspace budgets the step table for that many)");
  const int Cx = net->desc.out_channels, Cc = net->desc.in_channels - Cx;
  const size_t n = (size_t)N * Cx * H * W;
  const int steps = (num_steps > 0 && num_steps < S) ? num_steps : S;
  hipStream_t s;
  DRM_TRY(chain_stream(caller, steps, &s));
  float* e = ar.alloc<float>(n);
  float* tf = ar.alloc<float>((size_t)N);
  float* tab = ar.alloc<float>((size_t)steps * STEP_ROW);
  int* counter = ar.alloc<int>(1);
  float* qtab = blend ? ar.alloc<float>((size_t)steps * 2) : nullptr;
  float* e_u = uncond ? ar.alloc<float>(n) : nullptr;
  if (ar.failed) { set_error("ddim: workspace too small"); return DRM_ERR_WORKSPACE; }
  DRM_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "noise_dropout: 0 <= p < 1");
  DRM_REQUIRE(!blend || blend->mask_channels == 1 || blend->mask_channels == Cx, "mask blending: the mask has 1 or out_channels channels");
  std::vector<float> rows((size_t)steps * STEP_ROW, 0.f);
  int logged = 0;
  DRM_REQUIRE(!log_x || log_pred, "ddim: the intermediates log needs both buffers");
  for (int j = 0; j < steps; ++j) {
    const int index = S - 1 - j;
    rows[(size_t)j * STEP_ROW] = (float)timesteps[index];
    for (int k = 0; k < 5; ++k) rows[(size_t)j * STEP_ROW + 1 + k] = coef[5 * index + k];
    // the reference appends (img, pred_x0) after the step of `index` when index % log_every_t == 0 or at the first step (ddim.py:198-200)
    if (log_x && log_every_t > 0 && (index % log_every_t == 0 || index == S - 1)) {
      DRM_REQUIRE(logged < log_slots, "ddim: the intermediates log has too few slots");
      rows[(size_t)j * STEP_ROW + 7] = (float)(++logged);
    }
  }
  if (n_logged) *n_logged = logged;
  DRM_TRY(upload_step_table(rows, tab, counter, s));
  if (blend) DRM_TRY(upload_blend_table(blend, steps, qtab, s));
  const size_t mark = ar.mark();
  const size_t hw = (size_t)H * W;
  auto blend_now = [&]() -> int {
    hipLaunchKernelGGL(mask_blend_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, blend->x0, blend->mask, blend->mask_channels, Cx, hw, blend->qnoise, n,
                       qtab, counter, seed);
    DRM_HIP_CHECK(hipGetLastError());
    return DRM_OK;
  };
  return run_steps(steps, s, [&]() -> int {
    hipLaunchKernelGGL(step_begin_kernel, dim3((N + 255) / 256), dim3(256), 0, s, tab, counter, tf, N);
    DRM_HIP_CHECK(hipGetLastError());
    if (blend && blend->when == 0) DRM_TRY(blend_now());
    ar.release(mark);
    DRM_TRY(net->forward(x, Cx, cond, Cc, nullptr, nullptr, nullptr, tf, e, N, H, W, ar, s));
    if (uncond) {
      ar.release(mark);
      DRM_TRY(net->forward(x, Cx, uncond, Cc, nullptr, nullptr, nullptr, tf, e_u, N, H, W, ar, s));
      hipLaunchKernelGGL(cfg_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, e, e_u, guidance_scale, n);
      DRM_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(ddim_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, e, noise, n, tab, counter, seed, log_x, log_pred, drop_p, drop_keep);
    DRM_HIP_CHECK(hipGetLastError());
    if (blend && blend->when == 1) DRM_TRY(blend_now());
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, s, counter);
    DRM_HIP_CHECK(hipGetLastError());
    return DRM_OK;
  });
}

int ddpm_sample(UNet* net, float* x, float* pred_x0, const float* cond, const float* coef, int T_start, int clip, const float* noise,
                uint64_t seed, int N, int H, int W, Arena& ar, hipStream_t caller, const MaskBlend* blend, float drop_p, const float* drop_keep) {
  DRM_REQUIRE(net && net->desc.kind == 0, "ddpm needs a UNetModel");
  DRM_REQUIRE(T_start >= 1 && coef, "ddpm schedule");
  DRM_REQUIRE(T_start <= MAX_TABLE_STEPS, "ddpm: at most " + std::to_string(MAX_TABLE_STEPS) + " steps (the workspace budgets the step table for that many)");
  hipStream_t s;
  DRM_TRY(chain_stream(caller, T_start, &s));
  const int Cx = net->desc.out_channels, Cc = net->desc.in_channels - Cx;
  const size_t n = (size_t)N * Cx * H * W;
  float* e = ar.alloc<float>(n);
  float* tf = ar.alloc<float>((size_t)N);
  float* tab = ar.alloc<float>((size_t)T_start * STEP_ROW);
  int* counter = ar.alloc<int>(1);
  float* qtab = blend ? ar.alloc<float>((size_t)T_start * 2) : nullptr;
  if (ar.failed) { set_error("ddpm: workspace too small"); return DRM_ERR_WORKSPACE; }
  DRM_REQUIRE(!blend || blend->mask_channels == 1 || blend->mask_channels == Cx, "mask blending: the mask has 1 or out_channels channels");
  std::vector<float> rows((size_t)T_start * STEP_ROW, 0.f);
  for (int j = 0; j < T_start; ++j) {
    const int t = T_start - 1 - j;
    rows[(size_t)j * STEP_ROW] = (float)t;
    for (int k = 0; k < 5; ++k) rows[(size_t)j * STEP_ROW + 1 + k] = coef[5 * t + k];
    rows[(size_t)j * STEP_ROW + 6] = t != 0 ? 1.f : 0.f;  // no noise at t == 0 (ddpm.py:1161)
  }
  DRM_TRY(upload_step_table(rows, tab, counter, s));
  if (blend) DRM_TRY(upload_blend_table(blend, T_start, qtab, s));
  const size_t mark = ar.mark();
  const size_t hw = (size_t)H * W;
  auto blend_now = [&]() -> int {
    hipLaunchKernelGGL(mask_blend_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, blend->x0, blend->mask, blend->mask_channels, Cx, hw, blend->qnoise, n,
                       qtab, counter, seed);
    DRM_HIP_CHECK(hipGetLastError());
    return DRM_OK;
  };
  return run_steps(T_start, s, [&]() -> int {
    hipLaunchKernelGGL(step_begin_kernel, dim3((N + 255) / 256), dim3(256), 0, s, tab, counter, tf, N);
    DRM_HIP_CHECK(hipGetLastError());
    if (blend && blend->when == 0) DRM_TRY(blend_now());
    ar.release(mark);
    DRM_TRY(net->forward(x, Cx, cond, Cc, nullptr, nullptr, nullptr, tf, e, N, H, W, ar, s));
    hipLaunchKernelGGL(ddpm_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, pred_x0, e, noise, n, tab, counter, clip, seed, drop_p, drop_keep);
    DRM_HIP_CHECK(hipGetLastError());
    if (blend && blend->when == 1) DRM_TRY(blend_now());
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, s, counter);
    DRM_HIP_CHECK(hipGetLastError());
    return DRM_OK;
  });
}

}  // namespace drm
