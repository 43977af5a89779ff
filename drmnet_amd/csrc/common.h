// Shared declarations for the DRMNet MI355X (gfx950) hot-path library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include "gn_fold.h"

namespace drm {

// Thread-local last error text, surfaced through drm_last_error() (C ABI never throws).
void set_error(const std::string& msg);
const char* last_error();

#define DRM_OK 0
#define DRM_ERR_INVALID 1
#define DRM_ERR_HIP 2
#define DRM_ERR_WORKSPACE 3
#define DRM_ERR_STATE 4

#define DRM_HIP_CHECK(expr)                                                                              \
  do {                                                                                                   \
    hipError_t _e = (expr);                                                                              \
    if (_e != hipSuccess) {                                                                              \
      ::drm::set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " at " + __FILE__ + ":" + \
                       std::to_string(__LINE__));                                                        \
      return DRM_ERR_HIP;                                                                                \
    }                                                                                                    \
  } while (0)

#define DRM_REQUIRE(cond, msg)                                    \
  do {                                                            \
    if (!(cond)) {                                                \
      ::drm::set_error(std::string("invalid argument: ") + (msg)); \
      return DRM_ERR_INVALID;                                     \
    }                                                             \
  } while (0)

#define DRM_TRY(expr)          \
  do {                         \
    int _s = (expr);           \
    if (_s != DRM_OK) return _s; \
  } while (0)

// The kernels are written for one device shape: gfx950 / MI355X = 256 CUs in 8 XCDs (workgroups b and b + 8 share an XCD and
// its L2), 160 KiB of LDS per CU.  device_info() reads the CURRENT device's properties once per device ordinal and returns
// null -- with the error text set -- when they do not match, so a launch on anything else fails loudly instead of running a
// mis-sized persistent grid.  Thread-safe; one process may drive several GPUs (one stream / handle set per device).
struct DeviceInfo {
  int ordinal = 0;
  int cus = 0;          // 256
  int xcds = 0;         // 8
  size_t lds_per_cu = 0;  // 163840
};
const DeviceInfo* device_info();

// ---------------------------------------------------------------------------------------------
// Activation layout: NHWC fp32 ("pixels x channels", channels contiguous) everywhere inside the
// network; NCHW only at the boundary (reference tensors are NCHW).
// ---------------------------------------------------------------------------------------------

// Fused conv / GEMM descriptor (see conv.hip).
struct ConvArgs {
  const float* src0 = nullptr;  // NHWC, C0 channels; if up0, stored at (H/2, W/2) and read nearest-upsampled
  const float* src1 = nullptr;  // NHWC, C1 channels at (H, W) (skip tensor of the U-Net concat), may be null
  int C0 = 0, C1 = 0, up0 = 0;
  int N = 0, H = 0, W = 0;       // conv input == output spatial size (stride 1, pad = taps/2)
  const float* gn_scale = nullptr;  // [N][C0+C1] per-(sample,channel) GroupNorm scale (rstd*gamma) or null
  const float* gn_shift = nullptr;  // [N][C0+C1] (beta - mean*rstd*gamma)
  int silu = 0;                     // apply x*sigmoid(x) after the affine
  const float* w = nullptr;         // packed [taps][Cin/4][Cout][4]
  const float* bias = nullptr;      // [Cout]
  int taps = 9;                     // 9 (3x3, pad 1) or 1 (1x1)
  int Cout = 0;                     // padded Cout (multiple of 32)
  const float* emb = nullptr;       // optional per-(sample, cout) add: emb[n*emb_stride + co]
  int emb_stride = 0;
  const float* res = nullptr;       // optional residual NHWC [N,H,W,Cout]; may alias out
  float* out = nullptr;             // NHWC [N,H,W,Cout], or NCHW [N,cout_valid,H,W] if out_nchw
  int out_nchw = 0, cout_valid = 0;
  int cin_real = 0;                 // un-padded Cin for FLOP accounting (0 = C0 + C1)
  int ksplit = 1;                      // split-K factor (conv_split_ksplit); > 1: split k writes its raw partial sums to out + k * split_stride
  size_t split_stride = 0;             // floats between the split-K slabs (0 unless ksplit > 1)
  float* split_ws = nullptr;           // fused split-K: slab workspace ([ksplit] slabs of split_stride floats); the workgroup that arrives LAST at an output tile
  unsigned* tile_ticket = nullptr;     // (arrival counter per output tile, zero before the launch) sums the slabs in slab order and runs the full epilogue into `out`
  int terms = 3;                       // split kernels: 3 = fp16 hi/lo (fp32 accuracy), 2 = fp16 hi*hi + fp8 cross terms, 1 = plain fp16 operands, 4 = plain bf16 operands
  int mx_site = 0;                     // PREC_F16MX: this launch is one of the 3x3 convs whose weights carry the f16mx image
#ifdef DRM_S2_STAMP
  unsigned* stamp_out = nullptr;       // diagnostic build: [8 waves][120][2] (id, s_memtime low word) of one workgroup
  int stamp_block = 0, stamp_tile0 = 0;
#endif
  double2* stat_out = nullptr;         // optional [N][Cout] (sum, sum of squares) of the OUTPUT, accumulated atomically (must be zeroed)
  float* pool_out = nullptr;           // optional: the 2x2 average pool of the output, NHWC [N][H/2][W/2][Cout] (Downsample, openaimodel.py:154-160), written by the
  double2* pool_stat = nullptr;        // same epilogue, with its [N][Cout] (sum, sum of squares) (zeroed) -- conv_split_pool_applicable says which launches can
  const float* w_inv_scale = nullptr;  // split-precision path: device scalar 2^-k undoing the weight pre-scaling
  int ld0 = 0;                         // channel stride of src0's pixels when it is a channel slice of a wider tensor (0 = C0)
  long long w_img_stride_f4 = 0;       // split 1x1 path: every image has its own packed weight set this many float4 apart (attention GEMMs)
  const float* w_inv_img = nullptr;    // ... and its own 2^-k weight factor [N] (replaces w_inv_scale)
  int prof_kind = -1;                  // launch-profiler family override (-1 = by tap count, PROF_KINDS = no scope of its own)
  const float* in_inv = nullptr;       // split-precision path: [N] per-image 2^-k undoing the input staging factor (launch_act_pow2_scale)
  GnFold gnf;                          // sparse launches (engine.hip gn_params): gn_scale / gn_shift (and a guard table set) are finalised by this launch's own prologue
};
int launch_conv(const ConvArgs& a, hipStream_t s);
// index of the pixel-tile family {TH, TW} that wastes the fewest GEMM rows on an H x W map (ties: the first = larger tile)
int conv_tile_family(int H, int W, const int (*fam)[2], int n_fam);
// split-precision (fp16 hi/lo x 3 MFMA, fp32-accurate) variant; a.w = pre-split weights (conv_split.hip)
int launch_conv_split(const ConvArgs& a, hipStream_t s);
int conv_split_ksplit(const ConvArgs& a);  // split-K factor the split kernels want for this launch (1 = none)
bool conv_split_fused_finish(const ConvArgs& a);  // a split-K launch of this shape finishes its tiles itself (ConvArgs::split_ws / tile_ticket); else launch_splitk_reduce follows
// deterministic second half of a split-K conv on maps of at most 256 pixels: out = sum of the slabs at `partial` (+ bias, emb, residual), statistics into a.stat_out
int launch_splitk_reduce(const ConvArgs& a, const float* partial, hipStream_t s);
bool conv_split_pool_applicable(const ConvArgs& a);  // this launch (shape fields set) can write ConvArgs::pool_out / pool_stat from its epilogue
bool conv_split_fuses_stats();  // true when the active split kernel accumulates ConvArgs::stat_out in its epilogue
size_t packed_conv_weight_split_floats(int taps, int CoutP, int CinP);
// mx: the f16mx image (fp16 hi planes + e4m3 planes of hi and lo) for the 3x3 convs that run with ConvArgs::terms == 2
int launch_pack_conv_weight_split(const float* w, float* packed, float* scales, unsigned* scratch, int Cout, int Cin, int taps, int CoutP,
                                  int CinP, hipStream_t s, bool mx = false, bool bf16 = false);
// PREC_F16MX: PREC_F16X3 with the GroupNorm-fed 3x3 convs on fp16 hi*hi + one block-scaled fp8 MFMA for both cross terms (~4e-5 per network)
// PREC_BF16: bf16 operands, fp32 accumulate (v_mfma_f32_32x32x16_bf16): BASELINE configs[2] as written; reduced precision like PREC_F16
enum Precision { PREC_FP32 = 0, PREC_F16X3 = 1, PREC_F16 = 2, PREC_F16MX = 3, PREC_BF16 = 4 };
inline bool precision_valid(int p) { return p >= PREC_FP32 && p <= PREC_BF16; }
// repack PyTorch conv weight [Cout][Cin][kh][kw] -> [taps][CinP/4][CoutP][4] (zero padded)
int launch_pack_conv_weight(const float* w, float* packed, int Cout, int Cin, int taps, int CoutP, int CinP, hipStream_t s);
size_t packed_conv_weight_floats(int taps, int CoutP, int CinP);

// GroupNorm statistics (gn.hip)
// per-(n,c) first/second moments of an NHWC tensor: mom[n][c] = (mean, mean of squares)
int launch_chan_moments(const float* x, int N, int HW, int C, double* partial /*[N][splits][C][2]*/, double2* mom /*[N][C]*/, hipStream_t s);
int chan_moments_splits(int HW, int C);
// combine moments of up to two concatenated sources into per-(n,c) scale/shift (32 groups, eps 1e-5)
// inv0 / inv1: factor turning a table into per-pixel means (1 for tables of means, 1/(H*W) for tables of raw sums)
// per-image power-of-two staging factor for un-normalised inputs of the split-precision convs (gn.hip)
int launch_act_pow2_scale(const double2* mom0, int C0, int lo0, int hi0, double cnt0, const double2* mom1, int C1, double cnt1,
                          const unsigned* absmax_bits, int Ctab, int N, float* scale, float* shift, float* inv, hipStream_t s, int absmax_parts = 1);
// guard_*: optional fused range-guard tables of the same tensor (act_pow2_scale_kernel's product), cnt0 / cnt1 as there
int launch_gn_finalize(const double2* mom0, int C0, double inv0, const double2* mom1, int C1, double inv1, const float* gamma,
                       const float* beta, int N, float* scale, float* shift, hipStream_t s, double cnt0 = 0.0, double cnt1 = 0.0,
                       float* guard_scale = nullptr, float* guard_shift = nullptr, float* guard_inv = nullptr);

// attention (attn.hip): qkv [N][T][3C] -> out [N][T][C]; scores workspace [N][T][T]
size_t refmap_workspace_bytes(long long n, int res, float thr);
int launch_refmap_mask_make(const float* colors, const float* normals, long long n, int C, int res, float thr, int min_points, float* refmap,
                            unsigned char* refmask, void* ws, size_t ws_bytes, hipStream_t s);
int launch_erode_mask(const unsigned char* mask, int H, int W, int k, unsigned char* out, hipStream_t s);
// terms: 0 = fp32 MFMA, 3 = fp16 hi/lo split, 1 = plain fp16 operands
// qkv_mom + ws (attention_small_workspace_floats floats): per-image range guard of q, k, v in the split modes; proj_guard then receives proj_out's guard tables
int launch_attention(const float* qkv, float* scores, float* out, int N, int T, int C, hipStream_t s, int terms = 0, const double2* qkv_mom = nullptr,
                     float* ws = nullptr, ConvArgs* proj_guard = nullptr);
size_t attention_small_workspace_floats(int N, int T, int C);
// split-precision attention core on the fused 1x1 conv pipeline (per-image weights = k, v^T); T = H*W must be a multiple of 256
bool attention_conv_applicable(int T, int C, int H, int W, int terms);
// single-kernel form (attn_flash.hip): the long-sequence level (T >= 1024, C = 384), no score matrix in HBM
bool attention_flash_applicable(int T, int C, int terms);
size_t attention_flash_workspace_floats(int N, int T, int C);
int launch_attention_flash(const float* qkv, const double2* qkv_mom, float* out, float* ws, int N, int T, int C, int terms, hipStream_t s, ConvArgs* proj_guard);
// images per attention pass and the score workspace that takes ([group, T, T] floats: independent of the batch beyond one group)
int attention_group(int N, int T);
size_t attention_scores_floats(int N, int T);
size_t attention_conv_workspace_floats(int N, int T, int C);
struct ConvArgs;
int launch_attention_conv(const float* qkv, const double2* qkv_mom, float* scores, float* out, float* ws, int N, int H, int W, int C, int terms,
                          hipStream_t s, ConvArgs* proj_guard = nullptr);

// boundary maps and the envmap warp (transform.hip)
int launch_map_chain(const float* x, float* out, long long per_image, int B, const int32_t* ops, const float* args, int n_ops, const float* lo,
                     const float* hi, const float* scale, hipStream_t s);
int launch_masked_log_range(const float* x, const float* mask, int B, int C, int HW, float* lo, float* hi, hipStream_t s);
int launch_luminance_scale(const float* x, int B, int HW, float scaler, float* scale, hipStream_t s);
int launch_mirmap2envmap(const float* mir, const float* basis, float* out, int B, int C, int H, int W, int OH, int OW, int log_interp, int nhwc,
                         hipStream_t s);
int launch_hdr2ldr(const float* x, const unsigned char* mask, int HW, float alpha, float gamma, float* out, hipStream_t s);
int launch_resize(const float* x, float* out, int planes, int IH, int IW, int OH, int OW, int mode, hipStream_t s);

// misc kernels (misc.hip)
// absmax_bits (optional): [N][pack_input_absmax_parts(H, W)] words, every one written: max |element| of one block of a packed image as fp32 bits
int pack_input_absmax_parts(int H, int W);
int launch_pack_input(const float* x, const float* cond, const int* idx, float* out, int N, int H, int W, int Cx, int Cc, int CP, hipStream_t s,
                      unsigned* absmax_bits = nullptr);
// stat (optional, zeroed): [N][C] (sum, sum of squares) of the pooled tensor, accumulated with fp64 atomics
int launch_avgpool2(const float* x, float* out, int N, int H, int W, int C, hipStream_t s, double2* stat = nullptr);
int launch_linear(const float* in, const float* w, const float* b, float* out, int N, int I, int O, int silu_in, int silu_out, hipStream_t s);
int launch_timestep_embedding(const int64_t* t, const float* tf, float* out, int N, int dim, hipStream_t s);
int launch_encoder_head(const float* x, const float* scale, const float* shift, const float* w, const float* b, float* out, int N, int HW,
                        int C, int O, hipStream_t s);
// the stem of a U-Net as a kernel of its own (stemhead.hip): conv3x3(in_channels -> 128) on the NCHW boundary tensors (cat + row gather fused,
// fused output statistics)
bool stem_direct_applicable(int Cin, int Cout);
size_t stem_weight_floats();
int launch_pack_stem_weight(const float* w, float* img, int Cout, int Cin, bool exact, hipStream_t s);  // exact: fp32 operands (PREC_FP32), else fp16 hi / lo
int launch_stem_conv(const float* x, int Cx, const float* cond, int Cc, const int* rows, const float* wimg, const float* bias, float* out, double2* stat,
                     int N, int H, int W, int Cout, bool exact, hipStream_t s);
int launch_nhwc_to_nchw(const float* x, float* out, int N, int H, int W, int C, hipStream_t s);
int launch_nchw_to_nhwc(const float* x, float* out, int N, int H, int W, int C, hipStream_t s);

}  // namespace drm
