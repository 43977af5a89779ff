/* drmnet_hip.h -- C ABI of the MI355X-native DRMNet reverse-diffusion hot path.
 *
 * One shared library (drmnet_amd/csrc/libdrmnet_hip.so), plain pointers and sizes, no torch types.
 * All tensor memory is BORROWED from the caller (device pointers, fp32, contiguous); the library owns
 * only the packed-weight storage behind its handles.  Every entry point takes an explicit hipStream_t
 * (passed as void*), launches asynchronously and never synchronises, except the documented per-step
 * convergence read-back inside drm_drmnet_sample.  Return value 0 = OK, non-zero = error code with the
 * text available from drm_last_error(); nothing throws across the ABI.
 *
 * The reference (kyotovision-public/DRMNet) is 100 % Python with no FFI of its own, so each entry point
 * names the reference Python interface it replaces (paths relative to the reference root).
 */
#ifndef DRMNET_HIP_H
#define DRMNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRM_ABI_VERSION 3
#define DRM_MAX_LEVELS 8

int drm_abi_version(void);
const char* drm_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * U-Net family.  Replaces ldm/modules/diffusionmodules/openaimodel.py:
 *   UNetModel.__init__/forward        :452-713 / :731-768   (kind 0; IllNet and ObsNet)
 *   EncoderUNetModel.__init__/forward :777-953 / :969-991   (kind 1; RefNet, pool="adaptive")
 * Only the configuration space the shipped YAMLs use is accepted (num_heads=1, conv_resample=False,
 * resblock_updown=False, use_scale_shift_norm=False, dims=2, fp32); anything else is rejected.
 * ------------------------------------------------------------------------------------------- */
typedef struct drm_unet_desc {
  int32_t kind;           /* 0 = UNetModel, 1 = EncoderUNetModel */
  int32_t in_channels;    /* channels of cat([x, cond], 1); 6 in every shipped config */
  int32_t model_channels;
  int32_t out_channels;
  int32_t num_res_blocks;
  int32_t n_levels;
  int32_t channel_mult[DRM_MAX_LEVELS];
  int32_t n_attn;
  int32_t attention_resolutions[DRM_MAX_LEVELS];
} drm_unet_desc;

typedef struct drm_unet drm_unet;

/* Builds the block topology and the parameter table (host only, no GPU needed). */
int drm_unet_create(const drm_unet_desc* desc, drm_unet** out);
void drm_unet_destroy(drm_unet* net);

/* Parameter table == the reference module's state_dict() order, names and shapes
 * (e.g. "input_blocks.1.0.in_layers.2.weight", [128,128,3,3]). */
int drm_unet_param_count(const drm_unet* net);
int drm_unet_param_info(const drm_unet* net, int index, char* name, int name_cap, int64_t shape[4], int* ndim);

/* Uploads/repacks all parameters. ptrs[i] = device pointer to fp32 tensor i in PyTorch layout.
 * Replaces nn.Module.load_state_dict + LitEma.copy_to (ldm/modules/ema.py:46-53): the host passes the
 * EMA tensors when sampling under ema_scope. May be called again to swap weights. */
int drm_unet_load_params(drm_unet* net, const float* const* ptrs, int count, void* stream);

/* Weight sets.  A handle keeps DRM_WEIGHT_SETS packed images side by side: set 0 for the module's live parameters, set 1 for
 * the EMA shadow that ema_scope swaps in around sampling (models/drmnet.py:242-258, ldm/models/diffusion/ddpm.py:189-202,
 * ldm/modules/ema.py:46-76).  Each set is packed (and, in the f16 modes, pre-split) once by drm_unet_load_params_set;
 * drm_unet_use_set selects the one the following forwards / sampler calls read -- entering and leaving the scope moves no
 * weights.  drm_unet_load_params loads into the currently selected set.  The selection is part of the handle's state: calls that
 * use one handle from several threads must agree on it. */
#define DRM_WEIGHT_SETS 2
int drm_unet_load_params_set(drm_unet* net, int set, const float* const* ptrs, int count, void* stream);
int drm_unet_use_set(drm_unet* net, int set);

/* Arithmetic of the convolution / projection kernels:
 *   0 = DRM_PREC_FP32 : v_mfma_f32_32x32x2_f32, exact fp32 products (default)
 *   1 = DRM_PREC_F16X3: every fp32 operand split into fp16 hi + lo, products evaluated as hi*hi + hi*lo + lo*hi on the f16
 *       matrix cores with fp32 accumulation (22-bit operands: fp32-level accuracy at 16/3 x the fp32 matrix rate).
 * Must be set before drm_unet_load_params (weights are pre-split at load time); drm_set_op_precision does the same for the
 * drm_op_* entry points. */
#define DRM_PREC_FP32 0
#define DRM_PREC_F16X3 1
/*   2 = DRM_PREC_F16  : REDUCED PRECISION.  Operands rounded to fp16 (weights after the same power-of-two pre-scaling), one
 *       f16 MFMA per product, fp32 accumulation, fp32 activations in HBM.  ~1e-3 rel-L2 on the full networks -- outside the
 *       1e-4 contract of the two modes above; offered for BASELINE configs[2] (the reference's reduced-precision sampling). */
#define DRM_PREC_F16 2
/*   3 = DRM_PREC_F16MX: DRM_PREC_F16X3 with the GroupNorm-fed 3x3 convs of the res blocks (80 % of a step's matrix work) evaluated as
 *       hi*hi on the f16 matrix cores + BOTH cross terms (hi*lo + lo*hi) in one block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, OCP
 *       e4m3 operands with power-of-two block factors): 2/3 of the matrix-pipe cycles of F16X3.  The cross terms are 2^-11 of a product and
 *       carry an e4m3 rounding: 7e-6 rel-L2 per res block, 2.4e-5 .. 4e-5 per network against the reference -- inside the 1e-4 contract
 *       (tests/test_gpu_f16mx.py), an order of magnitude above F16X3's ~2e-6.  Every other launch runs exactly as in F16X3.
 *       RANGE LIMIT of this mode: the input of such a conv AFTER GroupNorm + SiLU is clipped to +-3584 (= 448 * 2^3, the e4m3 image's range)
 *       while it is staged -- F16X3 clips at fp16's 65504, FP32 not at all.  A GroupNorm output reaches at most sqrt(channels per group) *
 *       |gamma| + |beta| (<= 7 |gamma| + |beta| here), so |gamma| would have to exceed ~500 before a value gets there; such an input comes out
 *       finite and saturated, not wrapped or NaN.  Networks with GroupNorm gains of that size belong in F16X3. */
#define DRM_PREC_F16MX 3
/*   4 = DRM_PREC_BF16 : REDUCED PRECISION, BASELINE configs[2] as written ("DDIM 50-step, batch 256, bf16").  Operands rounded to bf16 (8
 *       significant bits, fp32's exponent range: no range guard needed), one v_mfma_f32_32x32x16_bf16 per product, fp32 accumulation, on the
 *       same kernels as DRM_PREC_F16 -- same speed, three mantissa bits fewer (~1e-2 rel-L2 per network against ~1e-3).  Outside the 1e-4
 *       contract; tests hold it to 3e-2. */
#define DRM_PREC_BF16 4
int drm_unet_set_precision(drm_unet* net, int precision);
/* drm_set_op_precision: the mode of the drm_op_* per-module entry points -- the library's only process-wide setting (an atomic word, read once at
 * the entry of every drm_op_* call; set it before the calls it is meant for, not concurrently with them).  Networks and samplers carry their own. */
int drm_set_op_precision(int precision);

/* Workspace (activations, statistics, attention scores) needed by one forward of batch N at HxW. */
size_t drm_unet_workspace_bytes(const drm_unet* net, int N, int H, int W);

/* kind 0: UNetModel.forward(cat([x, cond],1), timesteps=t | t_emb=t_emb) -> out [N,out_channels,H,W] NCHW.
 * kind 1: EncoderUNetModel.forward(cat([x, cond],1), timesteps)           -> out [N,out_channels].
 *   x    : [*, Cx, H, W] NCHW,  cond: [*, Cc, H, W] NCHW (Cx + Cc == in_channels; cond may be NULL if Cc == 0)
 *   rows : optional int32[N] gather indices into x/cond (DRMNet active-set compaction,
 *          models/drmnet.py:810-813); NULL = identity.
 *   t_emb: [N, model_channels] or NULL;  timesteps: int64[N] or NULL;  timesteps_f: fp32[N] or NULL
 *          (exactly one of the three; kind 1 needs timesteps / timesteps_f). */
int drm_unet_forward(drm_unet* net, const float* x, int Cx, const float* cond, int Cc, const int32_t* rows, const float* t_emb,
                     const int64_t* timesteps, const float* timesteps_f, float* out, int N, int H, int W, void* workspace,
                     size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Primitive ops on reference-layout tensors (NCHW activations, PyTorch weights).  They allocate their
 * own scratch and are meant for parity tests / per-module drop-ins, not for the timed path.
 * ------------------------------------------------------------------------------------------- */
/* nn.Linear with optional SiLU before/after: out[N,O] = act(b + act(in[N,I]) W[O,I]^T)
 * (time_embed, emb_layers: openaimodel.py:521-526,218-224; z_emb_layer: models/drmnet.py:38-45) */
int drm_linear_forward(const float* in, const float* w, const float* b, float* out, int N, int I, int O, int silu_in, int silu_out, void* stream);
/* timestep_embedding(timesteps, dim) (ldm/modules/diffusionmodules/util.py:151-171) */
int drm_timestep_embedding(const int64_t* timesteps, float* out, int N, int dim, void* stream);
/* [GroupNorm32 -> [SiLU] ->] conv2d k x k (k in {1,3}, stride 1, pad k/2) [+ emb[n,co]] [+ residual]
 * (ResBlock.in_layers / out_layers / skip_connection: openaimodel.py:201-241). gamma/beta NULL = no norm. */
int drm_op_norm_act_conv(const float* x, const float* gamma, const float* beta, int silu, const float* w, const float* b, int ksize,
                         const float* emb, const float* residual, float* out, int N, int Cin, int Cout, int H, int W, void* stream);
/* ResBlock._forward (openaimodel.py:255-275) on cat([x0 (optionally nearest-x2 upsampled), x1], 1).
 * params: 10 (or 12 with skip_connection) pointers in state_dict order. emb: [N, emb_dim]. */
int drm_op_resblock(const float* x0, int C0, int up0, const float* x1, int C1, const float* emb, int emb_dim, const float* const* params,
                    int n_params, float* out, int N, int Cout, int H, int W, void* stream);
/* AttentionBlock._forward (openaimodel.py:325-333, QKVAttentionLegacy :365-381), params: norm.w, norm.b, qkv.w, qkv.b, proj.w, proj.b */
int drm_op_attention_block(const float* x, const float* const* params, float* out, int N, int C, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Samplers.
 * ------------------------------------------------------------------------------------------- */
typedef struct drm_drmnet_cfg {
  int32_t z_dim;          /* len(z0), 6 in the shipped config */
  int32_t max_timesteps;
  double gamma;           /* double: the reference evaluates gamma^i = exp(i ln gamma) in fp64 (models/drmnet.py:494-495) */
  float epsilon, delta;   /* compared / multiplied in fp32 like the reference's tensor-scalar ops */
  float z0[8];            /* mirror-reflectance code (configs/drmnet/eval_drmnet.yaml: z0) */
} drm_drmnet_cfg;

typedef struct drm_drmnet drm_drmnet;

/* DRMNet reverse process over (RefNet, IllNet, z_emb_layer). Borrows the two nets (caller keeps them alive).
 * zemb: 6 device pointers = ZEmbDiffusionWrapper.z_emb_layer.{0,2,4}.{weight,bias} (models/drmnet.py:38-45). */
int drm_drmnet_create(drm_unet* illnet, drm_unet* refnet, const float* const* zemb, const drm_drmnet_cfg* cfg, drm_drmnet** out);
void drm_drmnet_destroy(drm_drmnet* s);
size_t drm_drmnet_workspace_bytes(const drm_drmnet* s, int N, int H, int W);
/* Batch parts of a reverse step (default 2; 1 = off; at most 4): a step over at least 64 rows per part runs its row ranges on internal streams
 * forked from and joined back into `stream`, so the sparse launches of one range (deep levels, small kernels) overlap with the other's.  The rows
 * of the reference's loop are independent (models/drmnet.py:809-839): results are those of the ranges run one after the other.  Call before
 * drm_drmnet_workspace_bytes (the workspace covers the parts' slices); a workspace that does not hold the slices makes the step run unforked, not fail.
 * NOT bitwise: a range of n / parts rows can take other tile shapes and split-K forms than the whole batch, so a row's result depends on the batch size
 * and on `parts` at the level of the arithmetic mode's rounding (f16x3: < 2e-5 rel-L2 over a recorded loop). */
int drm_drmnet_set_batch_parts(drm_drmnet* s, int parts);
/* Rows per part from which a step is forked (default 64: below it the ranges fall to the narrow conv tiles and the overlap loses, 817 vs 852 steps/s at
 * 32 rows); tests set 1 to drive tiny batches through the forked form.  Replaces the DRM_BATCH_PARTS / DRM_BATCH_PART_MIN environment reads of r5. */
int drm_drmnet_set_batch_part_min(drm_drmnet* s, int rows);

/* One reverse step on the active rows (DRMNet.p_mean_variance + the loop body, models/drmnet.py:752-770,809-839):
 *   z_out = RefNet(cat[Lr_k, LrK], i); zk = clamp(z0 + gamma^i (z_out - z0)); out = IllNet(cat[Lr_k, LrK], z_emb(zk - z0));
 *   Lr_k[rows] += out (+ delta * noise on rows that did not converge).
 * Lr_k, LrK: [B,3,H,W]; rows: int32[n_active] (device); noise: [B,3,H,W] or NULL (then Philox(seed, step));
 * zk_out / zK_out: [n_active, z_dim]; converged_out: int32[n_active] (device). */
int drm_drmnet_step(drm_drmnet* s, float* Lr_k, const float* LrK, const int32_t* rows, int n_active, int step, const float* noise,
                    uint64_t seed, float* zk_out, float* zK_out, int32_t* converged_out, int B, int H, int W, void* workspace,
                    size_t workspace_bytes, void* stream);

/* DRMNet.p_sample_loop (models/drmnet.py:782-847): LrK [B,3,H,W] (+ cond [B,3,H,W], the concat conditioning of both
 * nets; == LrK unless sigma_for_cond_xK > 0, models/drmnet.py:1037-1043) -> Lr0 [B,3,H,W], zK [B,z_dim] (NaN if never
 * converged), K int32[B].  noise0 [B,3,H,W] / step_noise [max_timesteps,B,3,H,W] or NULL (Philox).
 * early_exit = 0 keeps every row active for max_timesteps steps (countable-steps benchmark mode).
 * Synchronises the stream once per step to read the convergence flags (the reference does the same,
 * models/drmnet.py:841).  steps_done returns the number of executed steps. */
int drm_drmnet_sample(drm_drmnet* s, const float* LrK, const float* cond, const float* noise0, const float* step_noise, uint64_t seed, int early_exit,
                      float* Lr0, float* zK, int32_t* K, int32_t* steps_done, int B, int H, int W, void* workspace, size_t workspace_bytes,
                      void* stream);

/* DDIM sampling (DDIMSampler.ddim_sampling + p_sample_ddim, ldm/models/diffusion/ddim.py:128-259).
 *   timesteps: int64[S] (host) ddim_timesteps; coef: float[S][5] (host) = sqrt(a_t), sqrt(1-a_t), sqrt(a_prev),
 *   sqrt(1-a_prev-sigma^2), sigma per index; runs index S-1 .. 0 (or the first num_steps of them).  Synchronises the stream once
 *   at entry (upload of the per-step table).
 *   x: [N,3,H,W] in = x_T, out = final x;  cond: [N,3,H,W];  noise: [steps,N,3,H,W] or NULL (Philox). */
int drm_ddim_sample(drm_unet* net, float* x, const float* cond, const int64_t* timesteps, const float* coef, int S, int num_steps,
                    const float* noise, uint64_t seed, int N, int H, int W, void* workspace, size_t workspace_bytes, void* stream);

/* The same chain with the reference's `intermediates` (ddim.py:171-204: after the step of `index`, (img, pred_x0) are appended when
 * index % log_every_t == 0 or at the first step): slot k of log_x / log_pred_x0 ([log_slots][N,3,H,W] each, device) receives the k-th appended
 * pair, *n_logged (host) the number of pairs.  The steps that log are marked in the device step table, so graph replay applies unchanged. */
int drm_ddim_sample_logged(drm_unet* net, float* x, const float* cond, const int64_t* timesteps, const float* coef, int S, int num_steps,
                           const float* noise, uint64_t seed, int log_every_t, float* log_x, float* log_pred_x0, int log_slots, int32_t* n_logged, int N,
                           int H, int W, void* workspace, size_t workspace_bytes, void* stream);

/* Ancestral DDPM (LatentDiffusion.p_sample via ObsNetDiffusion.p_sample_loop, ldm/models/diffusion/ddpm.py:1079-1167,
 * models/obsnet.py:500-564).  coef: float[T][5] (host) = sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod,
 * posterior_mean_coef1, posterior_mean_coef2, exp(0.5*posterior_log_variance_clipped) for t = 0..T-1; runs t = T_start-1 .. 0.
 * x: in = x_T, out = last img; pred_x0: [N,3,H,W] out (what ObsNetDiffusion.p_sample_loop returns). */
int drm_ddpm_sample(drm_unet* net, float* x, float* pred_x0, const float* cond, const float* coef, int T_start, int clip_denoised,
                    const float* noise, uint64_t seed, int N, int H, int W, void* workspace, size_t workspace_bytes, void* stream);

/* The samplers' `mask` / `x0` arguments (known-region blending): img = q_sample(x0, t) * mask + (1 - mask) * img with
 * q_sample(x0, t) = sqrt(a_bar_t) x0 + sqrt(1 - a_bar_t) noise (ddpm.py:1052-1058).  Where and at which t it is applied differs between the three
 * reference loops, so the caller supplies the (a, b) pair of every chain step and the side of the step:
 *   DDIMSampler.ddim_sampling      ldm/models/diffusion/ddim.py:175-178   before the step's forward, at the step's t            -> when = 0
 *   ObsNetDiffusion.p_sample_loop  models/obsnet.py:545-547               before p_sample: x0 itself at t == 0, else t - 1       -> when = 0
 *   LatentDiffusion.p_sample_loop  ldm/models/diffusion/ddpm.py:1300-1302 after p_sample, at the step's t                       -> when = 1
 * (`temperature` of p_sample_ddim / p_sample, ddim.py:255 / ddpm.py:1157, needs no entry point: it scales the sigma column of `coef`.)  Passed through
 * drm_sampler_options below. */
typedef struct drm_mask_blend {
  const float* mask;   /* [N, mask_channels, H, W], device */
  int mask_channels;   /* 1 (broadcast over the channels) or 3 */
  const float* x0;     /* [N,3,H,W], device */
  const float* qcoef;  /* float[steps][2], host: (a, b) of chain step j (j = 0 is the first executed step) */
  const float* qnoise; /* [steps][N,3,H,W] device (the draws of q_sample), or NULL: Philox(seed) under a key of its own */
  int when;            /* 0 = before the step's network forward, 1 = after its update */
} drm_mask_blend;

/* The remaining per-step options of the reference's samplers, in one struct (zero-initialise; every member optional):
 *   blend          mask / x0 (above)
 *   uncond         DDIM only -- classifier-free guidance (p_sample_ddim, ldm/models/diffusion/ddim.py:225-232): the unconditional conditioning
 *                  [N,3,H,W]; every step evaluates the network on it as well and uses e = e_uncond + guidance_scale * (e_cond - e_uncond).  (The
 *                  reference batches both evaluations as 2 N rows; rows do not interact.)
 *   noise_dropout  F.dropout on the step noise (ddim.py:256-257, ddpm.py:1158-1159): an element is kept with probability 1 - p and scaled by
 *                  1 / (1 - p); dropout_keep [steps][N,3,H,W] holds 0 / 1 keep masks (parity runs), NULL draws them from Philox(seed) under a key of
 *                  its own.
 * Not offered: score correctors, quantize_denoised, callbacks (host hooks of the reference with no device meaning here). */
typedef struct drm_sampler_options {
  const drm_mask_blend* blend;
  const float* uncond;
  float guidance_scale;
  float noise_dropout;
  const float* dropout_keep;
} drm_sampler_options;

/* drm_ddim_sample_logged / drm_ddpm_sample with the options above (log_every_t <= 0: no intermediates).  Graph replay applies unchanged (every
 * per-step quantity lives in a device table read through the step counter). */
int drm_ddim_sample_ex(drm_unet* net, float* x, const float* cond, const int64_t* timesteps, const float* coef, int S, int num_steps,
                       const float* noise, uint64_t seed, const drm_sampler_options* opt, int log_every_t, float* log_x, float* log_pred_x0, int log_slots,
                       int32_t* n_logged, int N, int H, int W, void* workspace, size_t workspace_bytes, void* stream);
int drm_ddpm_sample_ex(drm_unet* net, float* x, float* pred_x0, const float* cond, const float* coef, int T_start, int clip_denoised,
                       const float* noise, uint64_t seed, const drm_sampler_options* opt, int N, int H, int W, void* workspace, size_t workspace_bytes,
                       void* stream);

size_t drm_sampler_workspace_bytes(const drm_unet* net, int N, int H, int W);

/* drm_ddim_sample / drm_ddpm_sample keep every per-step scalar (timestep, coefficients, noise offset) in a device table indexed
 * by a device counter, so all steps issue the same launches: the first step runs eagerly, the second is captured into a hipGraph
 * and the rest of the chain replays it (BASELINE configs[2] "hipGraph-captured step").  Off by default (measured on MI355X:
 * within +-0.5 % of eager launches at batch 32 / 256, 4-8 % slower at batch 1 where the instantiation is not amortised); when on
 * it applies to chains of >= 4 steps and steps aside while the launch profiler records (events do not belong in a graph).  With replay the call returns after the chain has
 * finished (the executable graph is destroyed behind its last launch).  drm_graph_launches counts hipGraphLaunch calls so far. */
int drm_set_graph_replay(int on);
int64_t drm_graph_launches(void);

/* Launch profiler (HIP events on the launch stream around each kernel family; used by bench.py for the roofline
 * object).  Kinds: 0 conv3x3 (fused GN+SiLU+conv implicit GEMM), 1 conv1x1 (skip / qkv / proj), 2 attention core,
 * 3 GroupNorm statistics, 4 other.  drm_profile_enable(1) instruments every family, (2) only kind 0 -- the dominant
 * kernel, ~90 instead of ~600 event pairs per step, which perturbs a timed region by < 0.5 % instead of ~3 %; (0) off.
 * drm_profile_collect synchronises the recorded events and returns totals since the last drm_profile_reset: each array
 * has DRM_PROFILE_KINDS entries. */
#define DRM_PROFILE_KINDS 5
void drm_profile_enable(int on);
void drm_profile_reset(void);
int drm_profile_collect(double* ms, double* flops, double* bytes, int64_t* launches);
/* Per kernel INSTANTIATION of the conv families (the name rocprofv3 prints): one text line "name\tkind\tlaunches\tms\tflops\tbytes\n" each,
 * totals since the last drm_profile_reset as of the last drm_profile_collect.  Writes at most cap - 1 characters + NUL into buf (may be
 * NULL) and returns the size needed.  Lets a per-launch PMC figure of ONE instantiation (HBM bytes) be set against the algorithmic bytes
 * of the same launches rather than a family mean. */
size_t drm_profile_variants(char* buf, size_t cap);

/* Standard-normal fill from the library's Philox4x32-10 stream (throughput mode noise source). */
int drm_randn(float* out, size_t n, uint64_t seed, uint64_t offset, void* stream);

/* Object image -> reflectance map, the step in front of the samplers (reference: refmap_mask_make,
 * utils/img2refmap.py:6-37, with xyz2thetaphi(normal = [0,1,0], tangent = [-1,0,0]), utils/transform.py:55-89).
 *   colors  [n][channels], normals [n][3]: the object pixels (fp32, device).  res: the map is res x res texels over
 *   (theta, phi) in (0, pi)^2.  A texel takes the colour of the pixel whose colour SUM is the lower median
 *   (torch.nanmedian) among the pixels with max(|theta - theta_i|, |phi - phi_j|) <= angle_threshold (fp32 compare);
 *   fewer than min_points such pixels, or none with a non-NaN sum, leaves it zero / unmasked.
 *   refmap [res][res][channels] fp32, refmask [res][res] uint8 (0/1).
 * Synchronises the stream once (capacity check of the binning lists). */
size_t drm_refmap_workspace_bytes(int64_t n, int res, float angle_threshold);
int drm_refmap_mask_make(const float* colors, const float* normals, int64_t n, int channels, int res, float angle_threshold, int min_points,
                         float* refmap, uint8_t* refmask, void* workspace, size_t workspace_bytes, void* stream);

/* Mask erosion of scripts/estimate.py:43-50: a mask pixel is dropped when a non-mask pixel lies inside the disk
 * footprint of diameter kernel_size around it (zero "same" padding: the image border does not erode).  uint8 0/1, [H][W]. */
int drm_erode_mask(const uint8_t* mask, int H, int W, int kernel_size, uint8_t* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The elementwise maps either side of the samplers, and the envmap warp / tone map after them.
 * ------------------------------------------------------------------------------------------- */
/* BaseDataset.transform / rescale (dataset/basedataset.py:29-112): a transform_func string such as
 * "resize_0p1tom1p1_normalizedLogarithmic_lowerbound1e-6" is a chain of named maps (applied right to left; rescale applies
 * the inverses left to right).  drm_map_chain applies up to 8 maps per element in one pass: x, out [B][per_image] fp32
 * (out may alias x), ops[n_ops] the DRM_MAP_* codes in application order, args[n_ops] their scalar argument.
 *   lo / hi : fp32[B] per-image (log10 min, log10 max) of "normalizedLogarithmic" (NULL unless DRM_MAP_NORM_LOG / _DENORM_LOG is used)
 *   scale   : fp32[B] per-image factor of DRM_MAP_IMG_MUL / _IMG_DIV (DRMNet.normalizing_scale, models/drmnet.py:1020-1027;
 *             scripts/estimate.py:99-100) */
#define DRM_MAP_LOG_P1 0          /* "log":    log10(x + 0.1) + 1                                   basedataset.py:52-53  */
#define DRM_MAP_LOG10 1           /* "log10":  log10(x)                                             :54-55                */
#define DRM_MAP_LOWERBOUND 2      /* "lowerbound<b>": clip(x, min = arg)                            :56-58                */
#define DRM_MAP_UNIT_TO_SIGNED 3  /* "0p1tom1p1": 2 x - 1                                           :59-60                */
#define DRM_MAP_NORM_LOG 4        /* "normalizedLogarithmic": (log10 x - lo[b]) / (hi[b] - lo[b])   :61-76                */
#define DRM_MAP_EXP_M1 5          /* inverse of "log":  10^min(x - 1, arg) - 0.1  (arg = clamp_before_exp, +inf = none)  :88-92 */
#define DRM_MAP_EXP10 6           /* inverse of "log10": 10^min(x, arg)                             :93-97                */
#define DRM_MAP_SIGNED_TO_UNIT 7  /* inverse of "0p1tom1p1": (x + 1) / 2                            :100-101              */
#define DRM_MAP_DENORM_LOG 8      /* x (hi[b] - lo[b]) + lo[b]  (followed by DRM_MAP_EXP10)         :102-112              */
#define DRM_MAP_IMG_MUL 9         /* x * scale[b] */
#define DRM_MAP_IMG_DIV 10        /* x / scale[b] */
#define DRM_MAP_CLIP0 11          /* clip(x, min = 0)                                               scripts/estimate.py:97 */
int drm_map_chain(const float* x, float* out, int64_t per_image, int B, const int32_t* ops, const float* args, int n_ops, const float* lo,
                  const float* hi, const float* scale, void* stream);
/* dynamic_normalize branch of "normalizedLogarithmic" (basedataset.py:63-69): per image b of x [B][C][HW] with mask [B][HW]
 * (fp32 0/1, broadcast over channels): linearmax = max(x mask); hi[b] = log10(linearmax); lo[b] = log10(min(x mask + (1 - mask) linearmax)). */
int drm_masked_log_range(const float* x, const float* mask, int B, int C, int HW, float* lo, float* hi, void* stream);
/* models/drmnet.py:1020-1026: scale[b] = scaler / exp(mean over {L > 0} of log(clip(L, 1e-5))), L = Rec.709 luminance of x[b] ([B][3][HW]). */
int drm_luminance_scale(const float* x, int B, int HW, float scaler, float* scale, void* stream);
/* mirmap2envmap (utils/transform.py:106-144; view +z, top +y, zenith +y, left edge -z, reverse_azimuth -- the only configuration
 * the reference supports) fused with the basis_r0 division of DRMNet.r0toenvmap (models/drmnet.py:931-941; basis [C][H][W] or
 * NULL): mirmap [B][C][H][W] -> out [B][C][OH][OW], or [B][OH][OW][C] when channels_last (what r0toenvmap returns).
 * Bilinear, border padding, align_corners = False (torch.nn.functional.grid_sample arithmetic). */
int drm_mirmap2envmap(const float* mirmap, const float* basis, float* out, int B, int C, int H, int W, int OH, int OW, int log_scale_interpolation,
                      int channels_last, void* stream);
/* hdr2ldr (utils/tonemap.py:4-9): x [HW][3] one channels-last image, mask uint8[HW] or NULL -> out [HW][3] in [0, 1]. */
int drm_hdr2ldr(const float* x, const uint8_t* mask, int HW, float alpha, float gamma, float* out, void* stream);
/* The "resize" map of BaseDataset.transform (dataset/basedataset.py:44-50: torchvision.transforms.functional.resize(x, (size, size),
 * interpolation, antialias=True) = torch's anti-aliased separable bilinear / bicubic filter, align_corners = False) and the nearest
 * mask resize of ObsNetDiffusion.get_cond_for_predict (models/obsnet.py:691: torch.nn.functional.interpolate(mask, size)).
 * x [planes][IH][IW] -> out [planes][OH][OW] fp32 (planes = every leading dimension flattened).  Down-scaling factors up to 23
 * (bicubic) / 47 (bilinear); beyond that DRM_ERR_ARG. */
#define DRM_RESIZE_NEAREST 0
#define DRM_RESIZE_BILINEAR_AA 1
#define DRM_RESIZE_BICUBIC_AA 2
int drm_resize(const float* x, float* out, int planes, int IH, int IW, int OH, int OW, int mode, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DRMNET_HIP_H */
