// Sampler drivers (see samplers.hip).
#pragma once
#include <vector>

#include "engine.h"

namespace drm {

int launch_randn(float* out, size_t n, uint64_t seed, uint64_t offset, hipStream_t s);

struct StepBuffers;
class DrmnetSampler {
 public:
  UNet* illnet = nullptr;
  UNet* refnet = nullptr;
  drm_drmnet_cfg cfg{};
  int init(UNet* ill, UNet* ref, const float* const* zemb_params, const drm_drmnet_cfg& c);
  size_t workspace_bytes(int N, int H, int W) const;
  int step(float* Lr_k, const float* LrK, const int32_t* rows, int n, int i, const float* noise, uint64_t seed, float* zk_out, float* zK_out,
           int32_t* conv_out, int B, int H, int W, Arena& ar, hipStream_t s);
  int sample(const float* LrK, const float* cond, const float* noise0, const float* step_noise, uint64_t seed, int early_exit, float* Lr0, float* zK, int32_t* K,
             int32_t* steps_done, int B, int H, int W, Arena& ar, hipStream_t s);
  ~DrmnetSampler();
  // Batch parts of a step (1 = off): a step over n >= parts * part_min rows runs as `parts` independent row ranges on internal streams forked
  // from and joined back into the caller's stream -- the sparse launches of one range (deep levels, small kernels: a third of the 256 CUs busy)
  // overlap with the other range's.  Rows are independent (the reference's loop has no cross-row term); results are those of the ranges run one
  // after the other.  Measured [r5]: two parts at 128 rows 927 vs 911 steps/s; at 32 rows 817 vs 852 (two ranges of 16 run one after the other
  // reach 711: the levels below 64x128 drop to the narrow tiles, and the overlap does not win that back) -- hence part_min = 64.
  // NOT bitwise: a range of n / parts rows can take other tile shapes and split-K forms than the whole batch, so a row's result depends on the batch
  // size and the part count at the level of the arithmetic mode's rounding (tests hold 2e-5 / 5e-6 over three steps).  When the caller's workspace
  // does not hold the parts' slices (sized before the part count was raised) the step runs on the caller's stream alone.
  static constexpr int PART_MAX = 4;
  int parts = 2, part_min = 64;

 private:
  int step_rows(float* Lr_k, const float* LrK, const int32_t* rows, int row0, int n, int i, const float* noise, uint64_t seed, const struct StepBuffers& b,
                int j0, int B, int H, int W, Arena& ar, hipStream_t s);
  size_t part_need(int nmax, int H, int W) const;
  struct PartNeed { int n, H, W; size_t bytes; };
  mutable std::vector<PartNeed> part_need_memo;
  hipStream_t part_stream[PART_MAX] = {};
  hipEvent_t part_done[PART_MAX] = {};
  hipEvent_t part_fork = nullptr;
  float* zemb = nullptr;  // z_emb_layer.{0,2,4}.{weight,bias}, device copy
  size_t zoff[6] = {0, 0, 0, 0, 0, 0};
  float* z0_dev = nullptr;
  int32_t* h_rows = nullptr;  // pinned staging for the active-row list / convergence flags
  int32_t* h_conv = nullptr;
  int h_cap = 0;
  int32_t* last = nullptr;  // device convergence flags / clamped zK of the most recent step (inside the workspace)
  float* last_zKc = nullptr;
};

size_t sampler_workspace_bytes(UNet* net, int N, int H, int W);
void set_graph_replay(bool on);   // DDIM / DDPM chains: replay one captured hipGraph of a step (default off)
long long graph_launches();       // hipGraphLaunch calls made so far (tests / bench read it)
// mask / x0 blending of a DDIM / DDPM chain (drm_mask_blend of the C ABI; see mask_blend_kernel)
struct MaskBlend {
  const float* mask = nullptr;    // [N, mask_channels, H, W] device
  int mask_channels = 1;          // 1 (broadcast) or out_channels
  const float* x0 = nullptr;      // [N, C, H, W] device
  const float* qcoef = nullptr;   // [steps][2] host: (sqrt(a_bar), sqrt(1 - a_bar)) of q_sample for the blend of chain step j
  const float* qnoise = nullptr;  // [steps][N, C, H, W] device or null (Philox)
  int when = 0;                   // 0 = before the step's network forward, 1 = after its update
};
// log_every_t > 0 with log_x / log_pred ([log_slots][N,3,H,W] each): the reference's intermediates (ddim.py:198-200), *n_logged = slots written
int ddim_sample(UNet* net, float* x, const float* cond, const int64_t* timesteps, const float* coef, int S, int num_steps, const float* noise,
                uint64_t seed, int N, int H, int W, Arena& ar, hipStream_t s, int log_every_t = 0, float* log_x = nullptr, float* log_pred = nullptr,
                int log_slots = 0, int* n_logged = nullptr, const MaskBlend* blend = nullptr, const float* uncond = nullptr, float guidance_scale = 1.0f,
                float drop_p = 0.f, const float* drop_keep = nullptr);
int ddpm_sample(UNet* net, float* x, float* pred_x0, const float* cond, const float* coef, int T_start, int clip, const float* noise,
                uint64_t seed, int N, int H, int W, Arena& ar, hipStream_t s, const MaskBlend* blend = nullptr, float drop_p = 0.f, const float* drop_keep = nullptr);

}  // namespace drm
