// Network engine: block topology, parameter table, packed-weight storage and the forward schedule of
// kernel launches for the reference U-Net family (ldm/modules/diffusionmodules/openaimodel.py).
#pragma once
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/drmnet_hip.h"
#include "common.h"

namespace drm {

// Bump allocator over the caller's workspace; "dry" mode only measures (drm_unet_workspace_bytes).
struct Arena {
  char* base = nullptr;
  size_t cap = 0, off = 0, peak = 0;
  bool dry = false;
  bool failed = false;
  void* alloc_bytes(size_t bytes) {
    const size_t a = (off + 255) & ~size_t(255);
    off = a + bytes;
    if (off > peak) peak = off;
    if (dry) return nullptr;
    if (off > cap) {
      failed = true;
      return nullptr;
    }
    return base + a;
  }
  template <typename T>
  T* alloc(size_t n) {
    return reinterpret_cast<T*>(alloc_bytes(n * sizeof(T)));
  }
  size_t mark() const { return off; }
  void release(size_t m) { off = m; }
  // GroupNorm statistics pool of one forward pass: the [N][C] (sum, sumsq) tables of all activations, carved from one
  // region that is zeroed by a single memset at the start of the pass (they are accumulated into by conv epilogues).
  char* st_base = nullptr;
  size_t st_cap = 0, st_off = 0;
  bool st_active = false;  // dry mode: count the pool; real mode: st_base is valid and zeroed
  void* alloc_stats(size_t bytes, bool* zeroed) {
    if (!st_active) {
      *zeroed = false;
      return alloc_bytes(bytes);
    }
    const size_t a = (st_off + 255) & ~size_t(255);
    st_off = a + bytes;
    *zeroed = true;
    if (dry) return nullptr;
    if (st_off > st_cap) {
      failed = true;
      return nullptr;
    }
    return st_base + a;
  }
};

struct Act {  // NHWC activation (+ cached per-channel moments for GroupNorm)
  float* p = nullptr;
  int C = 0, H = 0, W = 0;  // stored size
  int up = 0;               // logically nearest-x2 upsampled (consumer reads (y>>1, x>>1))
  double2* mom = nullptr;   // [N][C] (mean, mean of squares)
  bool mom_valid = false;
  bool mom_sums = false;    // table holds raw sums over the stored pixels (fused conv-epilogue statistics) instead of means
  bool mom_zeroed = false;  // table comes from the pass's pre-zeroed statistics pool
};

enum ParamKind { PK_COPY, PK_CONV };

struct ParamSlot {
  std::string name;
  std::vector<int64_t> shape;
  ParamKind kind = PK_COPY;
  size_t dst = 0;          // float offset into the packed weight buffer
  size_t count = 0;        // floats copied (PK_COPY)
  int cout = 0, cin = 0, taps = 0, coutp = 0, cinp = 0;  // PK_CONV
  size_t scale_dst = 0;    // PK_CONV: 2 floats (2^k, 2^-k) of the split-precision weight pre-scaling
  bool mx_site = false;    // PK_CONV: a GroupNorm-fed 3x3 conv of a res block (PREC_F16MX packs the f16mx image for it)
};

struct ResLayer {
  int cin = 0, cout = 0;
  size_t n1_w = 0, n1_b = 0, c1_w = 0, c1_b = 0, n2_w = 0, n2_b = 0, c2_w = 0, c2_b = 0, sk_w = 0, sk_b = 0;
  size_t c1_s = 0, c2_s = 0, sk_s = 0;  // weight pre-scaling slots (split-precision path)
  int emb_off = 0;  // column offset of this block's emb_layers output in the fused embedding buffer
  bool has_skip = false;
};
struct AttnLayer {
  int ch = 0;
  size_t n_w = 0, n_b = 0, qkv_w = 0, qkv_b = 0, proj_w = 0, proj_b = 0, qkv_s = 0, proj_s = 0;
};
struct Layer {
  enum Kind { RES, ATTN, DOWN, UP } kind;
  ResLayer res;
  AttnLayer attn;
};

class UNet {
 public:
  drm_unet_desc desc{};
  std::vector<ParamSlot> params;  // reference state_dict() order
  std::vector<std::vector<Layer>> input_blocks, output_blocks;  // input_blocks[0] is the stem (empty list)
  std::vector<Layer> middle;
  int final_ch = 0, emb_dim = 0, emb_total = 0, in_cp = 0, out_cp = 0;
  size_t stem_s = 0;
  // packed-buffer offsets
  size_t te0_w = 0, te0_b = 0, te2_w = 0, te2_b = 0, stem_w = 0, stem_b = 0, embcat_w = 0, embcat_b = 0, on_w = 0, on_b = 0, oc_w = 0, oc_b = 0, oc_s = 0, scratch_off = 0;
  // direct stem kernel (stemhead.hip): its own weight image, packed next to the generic one; -1 = the generic conv path
  long long stem_direct_w = -1;
  int stem_param = -1;  // index of the source tensor in `params`
  size_t wbuf_floats = 0;
  // Packed weight images.  A network keeps up to DRM_WEIGHT_SETS of them side by side (set 0 = the live parameters, set 1 = the
  // EMA shadow the reference swaps in for sampling, ema.py:46-76): each is packed / pre-split ONCE and selected per forward, so
  // entering and leaving ema_scope costs no re-upload.
  static constexpr int NSETS = DRM_WEIGHT_SETS;
  float* wsets[NSETS] = {nullptr, nullptr};  // device
  bool loaded[NSETS] = {false, false};
  int loaded_precision[NSETS] = {-1, -1};
  int active = 0;                    // the set forward() reads
  int precision = PREC_FP32;         // arithmetic of the conv kernels: PREC_FP32 (v_mfma_f32_32x32x2_f32), PREC_F16X3 (split fp16) or PREC_F16

  int build(const drm_unet_desc& d);
  int load(const float* const* ptrs, int count, hipStream_t s, int set);
  std::map<std::tuple<int, int, int, int>, size_t> stats_pool_cache;  // (N, H, W, precision) -> bytes of the statistics pool (split-K tickets live in it)
  int forward(const float* x, int Cx, const float* cond, int Cc, const int32_t* rows, const float* t_emb, const int64_t* t, const float* tf,
              float* out, int N, int H, int W, Arena& ar, hipStream_t s);
  ~UNet();

 private:
  size_t add_copy(const std::string& name, std::vector<int64_t> shape, size_t padded_count = 0);
  size_t add_conv(const std::string& name, int cout, int cin, int k, int coutp, int cinp, bool conv1d = false, size_t* scale_off = nullptr, bool mx_site = false);
  void add_res(Layer& l, const std::string& prefix, int cin, int cout);
  void add_attn(Layer& l, const std::string& prefix, int ch);
};

// building blocks shared with the op-level ABI entry points
struct Ctx {
  Arena* ar;
  hipStream_t s;
  int N;
  int precision = PREC_FP32;
  bool dry() const { return ar->dry; }
  bool split() const { return precision == PREC_F16X3 || precision == PREC_F16 || precision == PREC_F16MX || precision == PREC_BF16; }
  // ConvArgs::terms of the pipeline kernel (PREC_F16MX: 3, and 2 on its mx_site launches)
  int terms() const { return precision == PREC_F16 ? 1 : (precision == PREC_BF16 ? 4 : (precision == PREC_FP32 ? 0 : 3)); }
  bool mx() const { return precision == PREC_F16MX; }
};
Act new_act(Ctx& c, int C, int H, int W);
int ensure_moments(Ctx& c, Act& a);
// pool (optional): the Downsample that follows this block -- when the out_layers conv can write the 2x2 average pool of `out` and its statistics from
// its own epilogue (conv_split_pool_applicable) it does, and *pooled is set; otherwise the caller runs launch_avgpool2
int run_resblock(Ctx& c, const float* wbuf, const ResLayer& r, Act& x0, Act* x1, const float* emb_all, int emb_stride, Act& out, Act* pool = nullptr,
                 bool* pooled = nullptr);
int run_attention(Ctx& c, const float* wbuf, const AttnLayer& a, Act& x, Act& out);
// dispatches to the fp32 or the split-precision conv kernel; scale_off = the conv's pre-scaling slot in wbuf
// `stats_for` (optional): the activation this conv completes -- its GroupNorm statistics are then accumulated in the epilogue
// `splitk_ws`: the partial-slab workspace plan_splitk returned for this launch (null = no split-K)
int run_conv(Ctx& c, ConvArgs& a, const float* wbuf, size_t scale_off, Act* stats_for = nullptr, float* splitk_ws = nullptr);
float* plan_splitk(Ctx& c, ConvArgs& a);  // sets a.ksplit / a.split_stride from the shape fields of `a`; allocates the slabs
// split-precision range guard for an un-normalised conv input (see engine.hip)
int raw_input_guard(Ctx& c, ConvArgs& a, Act* x0, int lo0, int hi0, Act* x1, const unsigned* absmax_bits, int Ctab, int absmax_parts = 1);
// ensure_moments on both sources + gn_finalize into (scale, shift)
// guard_for (optional): a split conv reading (x0 | x1) un-normalised gets its range-guard tables from the same launch
int gn_params(Ctx& c, Act& x0, Act* x1, const float* gamma, const float* beta, float* scale, float* shift, ConvArgs* guard_for = nullptr,
              ConvArgs* fold_into = nullptr);  // fold_into: the consumer conv finalises the tables itself on sparse launches (ConvArgs::gnf)

}  // namespace drm
