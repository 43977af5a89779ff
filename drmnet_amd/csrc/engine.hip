// Network engine implementation (see engine.h).
//
// Topology enumeration restates the reference constructors (openaimodel.py:528-707 UNetModel,
// :824-929 EncoderUNetModel) so that the parameter table equals the reference state_dict() order;
// the forward schedule restates UNetModel.forward (:731-768) / EncoderUNetModel.forward (:969-991) as a
// fixed list of kernel launches on one HIP stream, with
//   * skip-concat and nearest-x2 upsample folded into the consumer's A-tile loader (never materialised),
//   * GroupNorm+SiLU folded into the consumer conv, statistics cached per tensor,
//   * all ResBlock emb_layers of the network evaluated by ONE linear launch per forward.
#include "engine.h"

#include <algorithm>
#include <atomic>
#include <deque>
#include <mutex>

namespace drm {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* last_error() { return g_err.c_str(); }

const DeviceInfo* device_info() {
  constexpr int MAX_DEV = 64;
  static DeviceInfo info[MAX_DEV];
  static std::atomic<int> state[MAX_DEV];  // 0 unknown, 1 valid, 2 rejected (zero-initialised)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) {
    set_error("device_info: no current HIP device");
    return nullptr;
  }
  int st = state[dev].load(std::memory_order_acquire);
  if (st == 0) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    st = state[dev].load(std::memory_order_acquire);
    if (st == 0) {
      hipDeviceProp_t p;
      if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
        set_error("device_info: hipGetDeviceProperties failed");
        return nullptr;
      }
      DeviceInfo d;
      d.ordinal = dev;
      d.cus = p.multiProcessorCount;
      d.lds_per_cu = p.maxSharedMemoryPerMultiProcessor;
      const std::string arch = p.gcnArchName;
      // XCD count is not a device property: it is 8 on the one part this library targets (gfx950 with 256 CUs)
      d.xcds = (arch.rfind("gfx950", 0) == 0 && d.cus == 256) ? 8 : 0;
      info[dev] = d;
      st = (d.xcds == 8 && d.lds_per_cu >= 160 * 1024) ? 1 : 2;
      state[dev].store(st, std::memory_order_release);
    }
  }
  if (st != 1) {
    set_error("unsupported device " + std::to_string(dev) + ": libdrmnet_hip is built for MI355X (gfx950, 256 CUs in 8 XCDs, 160 KiB LDS per CU); found " +
              std::to_string(info[dev].cus) + " CUs, " + std::to_string(info[dev].lds_per_cu) + " B LDS per CU");
    return nullptr;
  }
  return &info[dev];
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

UNet::~UNet() {
  for (float* w : wsets)
    if (w) (void)hipFree(w);
}

size_t UNet::add_copy(const std::string& name, std::vector<int64_t> shape, size_t padded_count) {
  ParamSlot p;
  p.name = name;
  p.shape = shape;
  p.kind = PK_COPY;
  size_t cnt = 1;
  for (auto v : shape) cnt *= (size_t)v;
  p.count = cnt;
  p.dst = wbuf_floats;
  wbuf_floats += (std::max(cnt, padded_count) + 63) & ~size_t(63);
  params.push_back(p);
  return p.dst;
}

size_t UNet::add_conv(const std::string& name, int cout, int cin, int k, int coutp, int cinp, bool conv1d, size_t* scale_off, bool mx_site) {
  ParamSlot p;
  p.name = name;
  if (conv1d) p.shape = {cout, cin, k};
  else p.shape = {cout, cin, k, k};
  p.kind = PK_CONV;
  p.cout = cout; p.cin = cin; p.taps = conv1d ? k : k * k; p.coutp = coutp; p.cinp = cinp;
  p.mx_site = mx_site;
  p.dst = wbuf_floats;
  wbuf_floats += (packed_conv_weight_floats(p.taps, coutp, cinp) + 63) & ~size_t(63);
  p.scale_dst = wbuf_floats;
  wbuf_floats += 64;
  if (scale_off) *scale_off = p.scale_dst;
  params.push_back(p);
  return p.dst;
}

void UNet::add_res(Layer& l, const std::string& px, int cin, int cout) {
  l.kind = Layer::RES;
  ResLayer& r = l.res;
  r.cin = cin; r.cout = cout; r.has_skip = (cin != cout);
  r.n1_w = add_copy(px + ".in_layers.0.weight", {cin});
  r.n1_b = add_copy(px + ".in_layers.0.bias", {cin});
  r.c1_w = add_conv(px + ".in_layers.2.weight", cout, cin, 3, cout, cin, false, &r.c1_s, true);
  r.c1_b = add_copy(px + ".in_layers.2.bias", {cout});
  r.emb_off = emb_total;
  emb_total += cout;
  // emb_layers.1.{weight,bias}: destinations are fixed up once emb_total is known (fused [sum Cout][emb_dim] matrix)
  ParamSlot ew; ew.name = px + ".emb_layers.1.weight"; ew.shape = {cout, emb_dim}; ew.kind = PK_COPY; ew.count = (size_t)cout * emb_dim; ew.dst = (size_t)-1; ew.cout = r.emb_off;
  params.push_back(ew);
  ParamSlot eb; eb.name = px + ".emb_layers.1.bias"; eb.shape = {cout}; eb.kind = PK_COPY; eb.count = (size_t)cout; eb.dst = (size_t)-2; eb.cout = r.emb_off;
  params.push_back(eb);
  r.n2_w = add_copy(px + ".out_layers.0.weight", {cout});
  r.n2_b = add_copy(px + ".out_layers.0.bias", {cout});
  r.c2_w = add_conv(px + ".out_layers.3.weight", cout, cout, 3, cout, cout, false, &r.c2_s, true);
  r.c2_b = add_copy(px + ".out_layers.3.bias", {cout});
  if (r.has_skip) {
    r.sk_w = add_conv(px + ".skip_connection.weight", cout, cin, 1, cout, cin, false, &r.sk_s);
    r.sk_b = add_copy(px + ".skip_connection.bias", {cout});
  }
}

void UNet::add_attn(Layer& l, const std::string& px, int ch) {
  l.kind = Layer::ATTN;
  AttnLayer& a = l.attn;
  a.ch = ch;
  a.n_w = add_copy(px + ".norm.weight", {ch});
  a.n_b = add_copy(px + ".norm.bias", {ch});
  a.qkv_w = add_conv(px + ".qkv.weight", 3 * ch, ch, 1, 3 * ch, ch, true, &a.qkv_s);
  a.qkv_b = add_copy(px + ".qkv.bias", {3 * ch});
  a.proj_w = add_conv(px + ".proj_out.weight", ch, ch, 1, ch, ch, true, &a.proj_s);
  a.proj_b = add_copy(px + ".proj_out.bias", {ch});
}

int UNet::build(const drm_unet_desc& d) {
  desc = d;
  DRM_REQUIRE(d.kind == 0 || d.kind == 1, "kind must be 0 (UNetModel) or 1 (EncoderUNetModel)");
  DRM_REQUIRE(d.n_levels >= 1 && d.n_levels <= DRM_MAX_LEVELS, "n_levels");
  DRM_REQUIRE(d.n_attn >= 0 && d.n_attn <= DRM_MAX_LEVELS, "n_attn");
  DRM_REQUIRE(d.model_channels > 0 && d.model_channels % 32 == 0, "model_channels must be a multiple of 32 (GroupNorm32)");
  DRM_REQUIRE(d.in_channels > 0 && d.in_channels <= 32, "in_channels");
  DRM_REQUIRE(d.out_channels > 0 && d.out_channels <= 32, "out_channels");
  DRM_REQUIRE(d.num_res_blocks >= 1, "num_res_blocks");
  const int mc = d.model_channels;
  emb_dim = 4 * mc;
  DRM_REQUIRE(emb_dim <= 512, "time_embed_dim (4*model_channels) must be <= 512");
  in_cp = round_up(d.in_channels, 32);  // one 32-channel K chunk: the stem runs on the same kernels as every other conv
  out_cp = 32;
  auto has_attn = [&](int ds) {
    for (int i = 0; i < d.n_attn; ++i)
      if (d.attention_resolutions[i] == ds) return true;
    return false;
  };

  te0_w = add_copy("time_embed.0.weight", {emb_dim, mc});
  te0_b = add_copy("time_embed.0.bias", {emb_dim});
  te2_w = add_copy("time_embed.2.weight", {emb_dim, emb_dim});
  te2_b = add_copy("time_embed.2.bias", {emb_dim});
  stem_w = add_conv("input_blocks.0.0.weight", mc, d.in_channels, 3, mc, in_cp, false, &stem_s);
  stem_param = (int)params.size() - 1;
  stem_b = add_copy("input_blocks.0.0.bias", {mc});
  input_blocks.emplace_back();

  std::vector<int> chans{mc};
  int ch = mc, ds = 1, idx = 1;
  for (int level = 0; level < d.n_levels; ++level) {
    const int m = d.channel_mult[level];
    DRM_REQUIRE(m >= 1, "channel_mult");
    for (int k = 0; k < d.num_res_blocks; ++k) {
      std::vector<Layer> ls(1);
      add_res(ls[0], "input_blocks." + std::to_string(idx) + ".0", ch, m * mc);
      ch = m * mc;
      if (has_attn(ds)) {
        ls.emplace_back();
        add_attn(ls.back(), "input_blocks." + std::to_string(idx) + ".1", ch);
      }
      input_blocks.push_back(ls);
      chans.push_back(ch);
      ++idx;
    }
    if (level != d.n_levels - 1) {
      std::vector<Layer> ls(1);
      ls[0].kind = Layer::DOWN;
      input_blocks.push_back(ls);
      chans.push_back(ch);
      ++idx;
      ds *= 2;
    }
  }
  middle.resize(3);
  add_res(middle[0], "middle_block.0", ch, ch);
  add_attn(middle[1], "middle_block.1", ch);
  add_res(middle[2], "middle_block.2", ch, ch);
  if (d.kind == 0) {
    int oidx = 0;
    for (int level = d.n_levels - 1; level >= 0; --level) {
      const int m = d.channel_mult[level];
      for (int i = 0; i <= d.num_res_blocks; ++i) {
        const int ich = chans.back();
        chans.pop_back();
        std::vector<Layer> ls(1);
        add_res(ls[0], "output_blocks." + std::to_string(oidx) + ".0", ch + ich, mc * m);
        ch = mc * m;
        if (has_attn(ds)) {
          ls.emplace_back();
          add_attn(ls.back(), "output_blocks." + std::to_string(oidx) + ".1", ch);
        }
        if (level && i == d.num_res_blocks) {
          ls.emplace_back();
          ls.back().kind = Layer::UP;
          ds /= 2;
        }
        output_blocks.push_back(ls);
        ++oidx;
      }
    }
  }
  final_ch = ch;
  on_w = add_copy("out.0.weight", {ch});
  on_b = add_copy("out.0.bias", {ch});
  if (d.kind == 0) {
    DRM_REQUIRE(ch == mc, "UNetModel head expects model_channels inputs");
    oc_w = add_conv("out.2.weight", d.out_channels, mc, 3, out_cp, mc, false, &oc_s);
    oc_b = add_copy("out.2.bias", {d.out_channels}, out_cp);
  } else {
    oc_w = add_copy("out.3.weight", {d.out_channels, ch, 1, 1});
    oc_b = add_copy("out.3.bias", {d.out_channels});
  }
  scratch_off = wbuf_floats;
  wbuf_floats += 64;
#ifndef DRM_NO_DIRECT_ENDS  // (A/B builds: tools/build_variant.sh)
  if (stem_direct_applicable(d.in_channels, mc)) {
    stem_direct_w = (long long)wbuf_floats;
    wbuf_floats += (stem_weight_floats() + 63) & ~size_t(63);
  }
#endif
  // fused embedding projection
  embcat_w = wbuf_floats;
  wbuf_floats += ((size_t)emb_total * emb_dim + 63) & ~size_t(63);
  embcat_b = wbuf_floats;
  wbuf_floats += ((size_t)emb_total + 63) & ~size_t(63);
  for (auto& p : params) {
    if (p.dst == (size_t)-1) p.dst = embcat_w + (size_t)p.cout * emb_dim;
    else if (p.dst == (size_t)-2) p.dst = embcat_b + (size_t)p.cout;
  }
  return DRM_OK;
}

int UNet::load(const float* const* ptrs, int count, hipStream_t s, int set) {
  DRM_REQUIRE(set >= 0 && set < NSETS, "weight set index");
  DRM_REQUIRE(count == (int)params.size(), "parameter count mismatch: got " + std::to_string(count) + ", expected " + std::to_string(params.size()));
  if (!wsets[set]) DRM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&wsets[set]), wbuf_floats * sizeof(float)));
  float* wbuf = wsets[set];
  loaded[set] = false;
  DRM_HIP_CHECK(hipMemsetAsync(wbuf, 0, wbuf_floats * sizeof(float), s));
  for (size_t i = 0; i < params.size(); ++i) {
    const ParamSlot& p = params[i];
    DRM_REQUIRE(ptrs[i] != nullptr, "null parameter pointer for " + p.name);
    if (p.kind == PK_COPY) {
      DRM_HIP_CHECK(hipMemcpyAsync(wbuf + p.dst, ptrs[i], p.count * sizeof(float), hipMemcpyDeviceToDevice, s));
    } else if (precision != PREC_FP32 && p.cinp % 32 == 0) {  // both fp16 modes use the pre-split, pre-scaled image
      DRM_TRY(launch_pack_conv_weight_split(ptrs[i], wbuf + p.dst, wbuf + p.scale_dst, reinterpret_cast<unsigned*>(wbuf + scratch_off), p.cout,
                                            p.cin, p.taps, p.coutp, p.cinp, s, precision == PREC_F16MX && p.mx_site, precision == PREC_BF16));
    } else {
      DRM_TRY(launch_pack_conv_weight(ptrs[i], wbuf + p.dst, p.cout, p.cin, p.taps, p.coutp, p.cinp, s));
    }
  }
  if (stem_direct_w >= 0) DRM_TRY(launch_pack_stem_weight(ptrs[stem_param], wbuf + stem_direct_w, desc.model_channels, desc.in_channels, precision == PREC_FP32, s));
  loaded[set] = true;
  loaded_precision[set] = precision;
  return DRM_OK;
}

// ------------------------------------------------------------------------------------------------ blocks

Act new_act(Ctx& c, int C, int H, int W) {
  Act a;
  a.C = C; a.H = H; a.W = W;
  a.p = c.ar->alloc<float>((size_t)c.N * H * W * C);
  a.mom = reinterpret_cast<double2*>(c.ar->alloc_stats((size_t)c.N * C * sizeof(double2), &a.mom_zeroed));
  return a;
}

int ensure_moments(Ctx& c, Act& a) {
  if (a.mom_valid) return DRM_OK;
  const int hs = a.H >> a.up, ws = a.W >> a.up;
  const size_t m = c.ar->mark();
  const int splits = chan_moments_splits(hs * ws, a.C);
  double* partial = c.ar->alloc<double>((size_t)c.N * splits * a.C * 2);
  if (!c.dry()) DRM_TRY(launch_chan_moments(a.p, c.N, hs * ws, a.C, partial, a.mom, c.s));
  c.ar->release(m);
  a.mom_valid = true;
  return DRM_OK;
}

// split-K factor the launch will use and the partial-slab workspace it then needs (allocated in sizing and real passes alike)
// The pipeline kernel (conv_split2.hip: LDS-DMA weight ring, persistent tiles, fused output statistics) takes every conv whose channel counts
// are whole 32-chunks: the two fp16 modes on any map, exact fp32 (TERMS = 0) on maps that are a whole number of 4x4 tiles.  The rest -- odd
// channel counts of the per-module entry points, ragged maps in fp32 -- runs on conv_igemm_kernel (conv.hip).
static bool on_pipeline(const Ctx& c, const ConvArgs& a) {
  return (a.C0 + a.C1) % 32 == 0 && a.C0 % 32 == 0 && (c.split() || (a.H % 4 == 0 && a.W % 4 == 0));
}

float* plan_splitk(Ctx& c, ConvArgs& a) {
  a.ksplit = 1;
  if (!on_pipeline(c, a)) return nullptr;
  a.ksplit = conv_split_ksplit(a);
  if (a.ksplit <= 1) return nullptr;
  a.split_stride = (size_t)a.N * a.H * a.W * a.Cout;
  a.tile_ticket = nullptr;
  if (conv_split_fused_finish(a)) {
    // one arrival counter per output tile (128 GEMM rows x 32 channels; an upper bound over the pixel-tile families), zeroed: from the pass's
    // statistics pool, else here -- the workgroup that arrives last at a tile sums the slabs and runs the full epilogue (conv_split2.hip)
    const size_t hw = (size_t)a.H * a.W;
    const size_t tickets = ((size_t)a.N * hw / 128 + hw / 16 + 2) * (size_t)(a.Cout / 32);
    bool zeroed = false;
    a.tile_ticket = reinterpret_cast<unsigned*>(c.ar->alloc_stats(tickets * sizeof(unsigned), &zeroed));
    if (!zeroed && !c.dry()) (void)hipMemsetAsync(a.tile_ticket, 0, tickets * sizeof(unsigned), c.s);
  }
  return c.ar->alloc<float>(a.split_stride * a.ksplit);
}

int run_conv(Ctx& c, ConvArgs& a, const float* Wb, size_t scale_off, Act* stats_for, float* splitk_ws) {
  if (on_pipeline(c, a)) {
    a.w_inv_scale = c.split() ? Wb + scale_off + 1 : nullptr;  // (fp32 weights are packed unscaled)
    a.terms = (c.mx() && a.mx_site) ? 2 : c.terms();
    if (!splitk_ws) {
      a.ksplit = 1;
      a.split_stride = 0;
    }
    double2* stat = nullptr;
    if (stats_for && !a.out_nchw && conv_split_fuses_stats()) {
      if (!stats_for->mom_zeroed) DRM_HIP_CHECK(hipMemsetAsync(stats_for->mom, 0, (size_t)c.N * stats_for->C * sizeof(double2), c.s));
      stat = stats_for->mom;
      stats_for->mom_valid = true;
      stats_for->mom_sums = true;
    }
    if (a.ksplit > 1 && !a.tile_ticket) {
      // smallest maps: every split writes its own slab, a fixed-order second launch sums them and applies bias / emb / residual / statistics
      ConvArgs part = a;
      part.out = splitk_ws;
      part.bias = nullptr; part.emb = nullptr; part.res = nullptr; part.stat_out = nullptr;
      DRM_TRY(launch_conv_split(part, c.s));
      ConvArgs red = a;
      red.stat_out = stat;
      return launch_splitk_reduce(red, splitk_ws, c.s);
    }
    // (other split-K launches: the workgroup that arrives last at an output tile sums the slabs in slab order and applies bias / emb /
    //  residual / statistics itself -- one launch, no atomics on the data path)
    a.split_ws = a.ksplit > 1 ? splitk_ws : nullptr;
    a.stat_out = stat;
    return launch_conv_split(a, c.s);
  }
  return launch_conv(a, c.s);
}

// Split-precision convs on an UN-normalised input (skip_connection, proj_out, stem): stage it through a per-image power of two
// derived from a rigorous bound of max |x| (gn.hip act_pow2_scale_kernel) so nothing saturates or underflows fp16; the factor rides
// on the (scale, shift) tables the staging path applies anyway and is undone per image in the epilogue.  Bound source: the
// per-channel sum-of-squares tables of x0 (channels [lo0, hi0)) and x1, or an absmax word per image.  No-op in fp32 mode.
int raw_input_guard(Ctx& c, ConvArgs& a, Act* x0, int lo0, int hi0, Act* x1, const unsigned* absmax_bits, int Ctab, int absmax_parts) {
  if (!c.split() || Ctab % 32 != 0) return DRM_OK;  // (channel counts that are not whole 32-chunks run on the exact-fp32 kernel)
  float* sc = c.ar->alloc<float>((size_t)c.N * Ctab);
  float* sh = c.ar->alloc<float>((size_t)c.N * Ctab);
  float* inv = c.ar->alloc<float>((size_t)c.N);
  if (x0) DRM_TRY(ensure_moments(c, *x0));
  if (x1) DRM_TRY(ensure_moments(c, *x1));
  if (c.dry()) return DRM_OK;
  auto cnt = [](const Act& t) { return t.mom_sums ? 0.0 : (double)(t.H >> t.up) * (t.W >> t.up); };
  DRM_TRY(launch_act_pow2_scale(x0 ? x0->mom : nullptr, x0 ? x0->C : 0, lo0, hi0, x0 ? cnt(*x0) : 0.0, x1 ? x1->mom : nullptr, x1 ? x1->C : 0,
                                x1 ? cnt(*x1) : 0.0, absmax_bits, Ctab, c.N, sc, sh, inv, c.s, absmax_parts));
  a.gn_scale = sc;
  a.gn_shift = sh;
  a.in_inv = inv;
  return DRM_OK;
}

// fold_into: the split-pipeline conv that stages through (scale, shift) -- its shape fields already set.  On sparse launches (few images) that
// launch finalises the tables in its own prologue (ConvArgs::gnf, gn_fold.h) and no gn_finalize launch is made: at batch 1 a ~5 us launch per
// GroupNorm was 111 launches = a tenth of the DRMNet step.
constexpr int GN_FOLD_MAX_N = 4;
int gn_params(Ctx& c, Act& x0, Act* x1, const float* gamma, const float* beta, float* scale, float* shift, ConvArgs* guard_for, ConvArgs* fold_into) {
  DRM_TRY(ensure_moments(c, x0));
  if (x1) DRM_TRY(ensure_moments(c, *x1));
  // the same launch can also produce the range-guard tables of a split conv that reads (x0 | x1) un-normalised (skip_connection)
  const int Ctot = x0.C + (x1 ? x1->C : 0);
  float *gs = nullptr, *gh = nullptr, *gi = nullptr;
  if (guard_for && c.split() && Ctot % 32 == 0) {
    gs = c.ar->alloc<float>((size_t)c.N * Ctot);
    gh = c.ar->alloc<float>((size_t)c.N * Ctot);
    gi = c.ar->alloc<float>((size_t)c.N);
    guard_for->gn_scale = gs;
    guard_for->gn_shift = gh;
    guard_for->in_inv = gi;
  }
  if (c.dry()) return DRM_OK;
  auto inv = [](const Act& a) { return a.mom_sums ? 1.0 / ((double)(a.H >> a.up) * (a.W >> a.up)) : 1.0; };
  auto cnt = [](const Act& t) { return t.mom_sums ? 0.0 : (double)(t.H >> t.up) * (t.W >> t.up); };
  if (fold_into && c.N <= GN_FOLD_MAX_N && on_pipeline(c, *fold_into)) {
    GnFold& f = fold_into->gnf;
    f.mom0 = x0.mom; f.C0 = x0.C; f.inv0 = inv(x0); f.cnt0 = cnt(x0);
    f.mom1 = x1 ? x1->mom : nullptr; f.C1 = x1 ? x1->C : 0; f.inv1 = x1 ? inv(*x1) : 1.0; f.cnt1 = x1 ? cnt(*x1) : 0.0;
    f.gamma = gamma; f.beta = beta; f.scale = scale; f.shift = shift;
    f.guard_scale = gs; f.guard_shift = gh; f.guard_inv = gi;
    return DRM_OK;
  }
  return launch_gn_finalize(x0.mom, x0.C, inv(x0), x1 ? x1->mom : nullptr, x1 ? x1->C : 0, x1 ? inv(*x1) : 1.0, gamma, beta, c.N, scale, shift, c.s,
                            cnt(x0), x1 ? cnt(*x1) : 0.0, gs, gh, gi);
}

// a real pass whose arena ran out hands out null tables: stop before any kernel is launched on them
static int arena_ok(const Ctx& c) {
  if (!c.ar->failed) return DRM_OK;
  set_error("workspace too small: need more than " + std::to_string(c.ar->cap) + " bytes (query drm_unet_workspace_bytes for this shape and precision)");
  return DRM_ERR_WORKSPACE;
}

int run_resblock(Ctx& c, const float* Wb, const ResLayer& r, Act& x0, Act* x1, const float* emb_all, int emb_stride, Act& out, Act* pool, bool* pooled) {
  DRM_TRY(arena_ok(c));
  const int H = x0.H, W = x0.W;
  const int C0 = x0.C, C1 = x1 ? x1->C : 0;
  DRM_REQUIRE(C0 + C1 == r.cin, "resblock input channels");
  DRM_REQUIRE(!x1 || (x1->H == H && x1->W == W && !x1->up), "resblock skip tensor shape");
  DRM_REQUIRE(r.has_skip || (!x1 && !x0.up), "identity skip on a concatenated / upsampled input is not supported");
  const size_t mark = c.ar->mark();
  float* sc1 = c.ar->alloc<float>((size_t)c.N * r.cin);
  float* sh1 = c.ar->alloc<float>((size_t)c.N * r.cin);
  Act h1 = new_act(c, r.cout, H, W);
  float* sc2 = c.ar->alloc<float>((size_t)c.N * r.cout);
  float* sh2 = c.ar->alloc<float>((size_t)c.N * r.cout);
  ConvArgs k;  // skip_connection: 1x1 conv on the raw (un-normalised) block input; its range-guard tables come out of the same launch
  ConvArgs a;  // in_layers conv: GroupNorm(x0 | x1) -> SiLU -> 3x3 + emb
  a.src0 = x0.p; a.src1 = x1 ? x1->p : nullptr; a.C0 = C0; a.C1 = C1; a.up0 = x0.up;
  a.N = c.N; a.H = H; a.W = W; a.taps = 9; a.Cout = r.cout; a.mx_site = 1;
  DRM_TRY(gn_params(c, x0, x1, Wb + r.n1_w, Wb + r.n1_b, sc1, sh1, r.has_skip ? &k : nullptr, &a));
  float* ws1 = plan_splitk(c, a);
  if (!c.dry()) {
    a.gn_scale = sc1; a.gn_shift = sh1; a.silu = 1;
    a.w = Wb + r.c1_w; a.bias = Wb + r.c1_b;
    a.emb = emb_all ? emb_all + r.emb_off : nullptr; a.emb_stride = emb_stride;
    a.out = h1.p;
    DRM_TRY(run_conv(c, a, Wb, r.c1_s, &h1, ws1));
  }
  ConvArgs b;  // out_layers conv: GroupNorm(h1) -> SiLU -> 3x3 + residual
  b.src0 = h1.p; b.C0 = r.cout; b.N = c.N; b.H = H; b.W = W; b.taps = 9; b.Cout = r.cout; b.mx_site = 1;
  DRM_TRY(gn_params(c, h1, nullptr, Wb + r.n2_w, Wb + r.n2_b, sc2, sh2, nullptr, &b));
  float* ws2 = plan_splitk(c, b);
  // the Downsample behind this block, from this conv's epilogue (decided on shapes and mode only: the sizing pass decides the same)
  b.terms = (c.mx() && b.mx_site) ? 2 : c.terms();
#ifdef DRM_NO_POOL_FUSION  // (A/B builds)
  pool = nullptr;
#endif
  const bool fuse_pool = pool && b.ksplit <= 1 && c.split() && conv_split_fuses_stats() && conv_split_pool_applicable(b);
  if (pooled) *pooled = fuse_pool;
  if (fuse_pool) {
    pool->mom_valid = true;
    pool->mom_sums = true;
  }
  float* wsk = nullptr;
  if (r.has_skip) {
    k.C0 = C0; k.C1 = C1; k.N = c.N; k.H = H; k.W = W; k.taps = 1; k.Cout = r.cout;
    wsk = plan_splitk(c, k);
  }
  if (!c.dry()) {
    const float* res = x0.p;
    if (r.has_skip) {
      k.src0 = x0.p; k.src1 = x1 ? x1->p : nullptr; k.up0 = x0.up;
      k.w = Wb + r.sk_w; k.bias = Wb + r.sk_b;
      k.out = out.p;
      DRM_TRY(run_conv(c, k, Wb, r.sk_s, nullptr, wsk));
      res = out.p;
    }
    b.gn_scale = sc2; b.gn_shift = sh2; b.silu = 1;
    b.w = Wb + r.c2_w; b.bias = Wb + r.c2_b;
    b.res = res; b.out = out.p;
    if (fuse_pool) {
      if (!pool->mom_zeroed) DRM_HIP_CHECK(hipMemsetAsync(pool->mom, 0, (size_t)c.N * pool->C * sizeof(double2), c.s));
      b.pool_out = pool->p;
      b.pool_stat = pool->mom;
    }
    DRM_TRY(run_conv(c, b, Wb, r.c2_s, &out, ws2));
  }
  c.ar->release(mark);
  return DRM_OK;
}

int run_attention(Ctx& c, const float* Wb, const AttnLayer& l, Act& x, Act& out) {
  DRM_REQUIRE(!x.up && x.C == l.ch, "attention input");
  DRM_TRY(arena_ok(c));
  const int H = x.H, W = x.W, T = H * W, C = l.ch;
  const size_t mark = c.ar->mark();
  float* sc = c.ar->alloc<float>((size_t)c.N * C);
  float* sh = c.ar->alloc<float>((size_t)c.N * C);
  ConvArgs a;  // qkv: GroupNorm(x) -> 1x1
  a.C0 = C; a.N = c.N; a.H = H; a.W = W; a.taps = 1; a.Cout = 3 * C;
  DRM_TRY(gn_params(c, x, nullptr, Wb + l.n_w, Wb + l.n_b, sc, sh, nullptr, &a));
  Act qkv_act = new_act(c, 3 * C, H, W);  // its per-channel sums (fused into the qkv conv's epilogue) bound |v| >= |attention output|
  float* qkv = qkv_act.p;
  const bool flash = c.split() && attention_flash_applicable(T, C, c.terms());  // the long-sequence level: one kernel, no score matrix (attn_flash.hip)
  float* scores = flash ? nullptr : c.ar->alloc<float>(attention_scores_floats(c.N, T));  // one image group at a time (attn.hip attention_group)
  float* att = c.ar->alloc<float>((size_t)c.N * T * C);
  // the T >= 256 levels: both GEMMs on the conv pipeline -- except on sparse launches (the 16x16 level of a batch-1 step: six launches, 51 us, where
  // the short-sequence form -- qk_small, softmax, P v -- takes three and ~25 us)
  const bool on_conv = flash || (c.split() && attention_conv_applicable(T, C, H, W, c.terms()) && (long long)c.N * T > 1024);
  // (the short-sequence form in the split modes: per-image guard factors + proj_out's guard tables, from one attn_scales launch)
  const bool small_guard = !on_conv && c.split() && C % 32 == 0;
  float* aws = flash ? c.ar->alloc<float>(attention_flash_workspace_floats(c.N, T, C))
               : on_conv ? c.ar->alloc<float>(attention_conv_workspace_floats(c.N, T, C))
               : small_guard ? c.ar->alloc<float>(attention_small_workspace_floats(c.N, T, C)) : nullptr;
  ConvArgs p;  // proj_out: 1x1 conv on the raw attention output, a convex combination of v rows: max |att| <= max |v|
  float* wsq = plan_splitk(c, a);
  p.C0 = C; p.N = c.N; p.H = H; p.W = W; p.taps = 1; p.Cout = C;
  float* wsp = plan_splitk(c, p);
  if (!c.dry()) {
    a.src0 = x.p;
    a.gn_scale = sc; a.gn_shift = sh; a.silu = 0;
    a.w = Wb + l.qkv_w; a.bias = Wb + l.qkv_b; a.out = qkv;
    DRM_TRY(run_conv(c, a, Wb, l.qkv_s, &qkv_act, wsq));
    if (flash) DRM_TRY(launch_attention_flash(qkv, qkv_act.mom, att, aws, c.N, T, C, c.terms(), c.s, &p));
    else if (on_conv) DRM_TRY(launch_attention_conv(qkv, qkv_act.mom, scores, att, aws, c.N, H, W, C, c.terms(), c.s, &p));
    else DRM_TRY(launch_attention(qkv, scores, att, c.N, T, C, c.s, c.split() ? c.terms() : 0, small_guard ? qkv_act.mom : nullptr, aws, small_guard ? &p : nullptr));
  } else {
    qkv_act.mom_valid = true;  // sizing pass: the table is filled by the conv epilogue, no stand-alone moments launch
    qkv_act.mom_sums = true;
  }
  if (!on_conv && !small_guard) DRM_TRY(raw_input_guard(c, p, &qkv_act, 2 * C, 3 * C, nullptr, nullptr, C));  // (the attention cores of the split modes hand p its guard tables)
  if (!c.dry()) {
    p.src0 = att;
    p.w = Wb + l.proj_w; p.bias = Wb + l.proj_b; p.res = x.p; p.out = out.p;
    DRM_TRY(run_conv(c, p, Wb, l.proj_s, &out, wsp));
  }
  c.ar->release(mark);
  return DRM_OK;
}

// ------------------------------------------------------------------------------------------------ forward

int UNet::forward(const float* x, int Cx, const float* cond, int Cc, const int32_t* rows, const float* t_emb, const int64_t* t,
                  const float* tf, float* out, int N, int H, int W, Arena& ar, hipStream_t s) {
  DRM_REQUIRE(ar.dry || loaded[active], "drm_unet_forward before drm_unet_load_params (weight set " + std::to_string(active) + ")");
  DRM_REQUIRE(N > 0, "batch size");
  DRM_REQUIRE(Cx + Cc == desc.in_channels, "x/cond channels must sum to in_channels");
  const int down = 1 << (desc.n_levels - 1);
  // any size the reference's fully convolutional forward accepts (openaimodel.py:731-768): every Downsample must see even sizes
  DRM_REQUIRE(H > 0 && W > 0 && H % down == 0 && W % down == 0,
              "H and W must be multiples of " + std::to_string(down) + " (2^(levels-1): each of the " + std::to_string(desc.n_levels - 1) + " Downsample layers halves the map)");
  const int n_t = (t_emb != nullptr) + (t != nullptr) + (tf != nullptr);
  if (!ar.dry) {
    if (desc.kind == 0) DRM_REQUIRE(n_t == 1, "timesteps and t_emb cannot be specified at the same time");
    else DRM_REQUIRE(n_t == 1 && t_emb == nullptr, "EncoderUNetModel takes timesteps");
  }
  DRM_REQUIRE(ar.dry || loaded_precision[active] == precision, "precision changed after drm_unet_load_params: reload the parameters");
  Ctx c{&ar, s, N, precision};
  const float* Wb = wsets[active];
  const int mc = desc.model_channels;

  // statistics pool: sized by a dry pass (cached per shape), zeroed once
  ar.st_active = true;
  ar.st_off = 0;
  if (!ar.dry) {
    const auto key = std::make_tuple(N, H, W, precision);  // the split modes carve their split-K ticket counters out of the pool
    auto it = stats_pool_cache.find(key);
    if (it == stats_pool_cache.end()) {
      Arena probe;
      probe.dry = true;
      DRM_TRY(forward(nullptr, Cx, nullptr, Cc, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, probe, s));
      it = stats_pool_cache.emplace(key, probe.st_off).first;
    }
    ar.st_cap = it->second;
    ar.st_base = reinterpret_cast<char*>(ar.alloc_bytes(ar.st_cap));
    if (ar.st_base) DRM_HIP_CHECK(hipMemsetAsync(ar.st_base, 0, ar.st_cap, s));
  }
  struct PoolScope {  // the pool belongs to this pass only
    Arena& a;
    ~PoolScope() {
      if (a.dry) a.peak += (a.st_off + 255) & ~size_t(255);
      a.st_active = false;
      a.st_base = nullptr;
    }
  } pool_scope{ar};

  const bool stem_direct = stem_direct_w >= 0;
  Act xin;
  const int amax_parts = pack_input_absmax_parts(H, W);
  unsigned* amax = nullptr;
  if (!stem_direct) {
    xin = new_act(c, in_cp, H, W);
    amax = c.ar->alloc<unsigned>((size_t)N * amax_parts);  // max |input| per (image, pack block): every word is written by the pack kernel
  }
  float* temb = c.ar->alloc<float>((size_t)N * mc);
  float* e1 = c.ar->alloc<float>((size_t)N * emb_dim);
  float* emb = c.ar->alloc<float>((size_t)N * emb_dim);
  float* emb_all = c.ar->alloc<float>((size_t)N * emb_total);
  DRM_TRY(arena_ok(c));
  if (!c.dry()) {
    if (!stem_direct) DRM_TRY(launch_pack_input(x, cond, rows, xin.p, N, H, W, Cx, Cc, in_cp, s, amax));
    const float* te = t_emb;
    if (!te) {
      DRM_TRY(launch_timestep_embedding(t, tf, temb, N, mc, s));
      te = temb;
    }
    DRM_TRY(launch_linear(te, Wb + te0_w, Wb + te0_b, e1, N, mc, emb_dim, 0, 1, s));
    DRM_TRY(launch_linear(e1, Wb + te2_w, Wb + te2_b, emb, N, emb_dim, emb_dim, 0, 0, s));
    DRM_TRY(launch_linear(emb, Wb + embcat_w, Wb + embcat_b, emb_all, N, emb_dim, emb_total, 1, 0, s));
  }

  std::deque<Act> acts;  // stable addresses: the skip stack and `h` share cached GroupNorm moments
  auto make = [&](int C_, int H_, int W_) -> Act* {
    acts.push_back(new_act(c, C_, H_, W_));
    return &acts.back();
  };
  std::vector<Act*> hs;
  Act* h = make(mc, H, W);
  if (stem_direct) {
    // stem conv on the NCHW boundary tensors (cat, row gather, exact fp32 products and the output's GroupNorm sums in one launch: stemhead.hip)
    if (!c.dry()) {
      if (!h->mom_zeroed) DRM_HIP_CHECK(hipMemsetAsync(h->mom, 0, (size_t)N * mc * sizeof(double2), s));
      DRM_TRY(launch_stem_conv(x, Cx, cond, Cc, rows, Wb + stem_direct_w, Wb + stem_b, h->p, h->mom, N, H, W, mc, precision == PREC_FP32, s));
    }
    h->mom_valid = true;
    h->mom_sums = true;
  } else {
    ConvArgs a;  // stem conv: raw network input
    DRM_TRY(raw_input_guard(c, a, nullptr, 0, 0, nullptr, amax, in_cp, amax_parts));
    if (!c.dry()) {
      a.src0 = xin.p; a.C0 = in_cp; a.N = N; a.H = H; a.W = W;
      a.w = Wb + stem_w; a.bias = Wb + stem_b; a.taps = 9; a.Cout = mc; a.out = h->p; a.cin_real = desc.in_channels;
      DRM_TRY(run_conv(c, a, Wb, stem_s, h));
    }
  }
  hs.push_back(h);

  Act* pool_buf = nullptr;   // the output tensor of the Downsample that follows the block being run (allocated ahead of it) ...
  bool pool_done = false;    // ... already written, statistics included, by that block's out_layers conv
  auto run_layers = [&](std::vector<Layer>& ls, Act* skip, bool down_next = false) -> int {
    for (size_t li = 0; li < ls.size(); ++li) {
      Layer& l = ls[li];
      if (l.kind == Layer::RES) {
        Act* o = make(l.res.cout, h->H, h->W);
        Act* po = nullptr;
        bool pooled = false;
        if (down_next && li + 1 == ls.size() && h->H % 2 == 0 && h->W % 2 == 0) po = make(l.res.cout, h->H / 2, h->W / 2);
        DRM_TRY(run_resblock(c, Wb, l.res, *h, (li == 0) ? skip : nullptr, emb_all, emb_total, *o, po, &pooled));
        pool_buf = po;
        pool_done = pooled;
        h = o;
      } else if (l.kind == Layer::ATTN) {
        Act* o = make(l.attn.ch, h->H, h->W);
        DRM_TRY(run_attention(c, Wb, l.attn, *h, *o));
        h = o;
      } else if (l.kind == Layer::DOWN) {
        DRM_REQUIRE(!h->up, "downsample of an upsampled tensor");
        Act* o = pool_buf ? pool_buf : make(h->C, h->H / 2, h->W / 2);
        const bool done = pool_buf && pool_done;  // written by the producing conv's epilogue, statistics included
        pool_buf = nullptr;
        pool_done = false;
        DRM_TRY(arena_ok(c));
        if (done) {
          h = o;
          continue;
        }
        if (!c.dry()) {
          if (!o->mom_zeroed) DRM_HIP_CHECK(hipMemsetAsync(o->mom, 0, (size_t)N * o->C * sizeof(double2), s));
          DRM_TRY(launch_avgpool2(h->p, o->p, N, h->H, h->W, h->C, s, o->mom));  // pooled tensor + its GroupNorm sums in one pass
          o->mom_valid = true;
          o->mom_sums = true;
        }
        h = o;
      } else {  // UP: nearest x2, folded into the consumer (moments are unchanged by replication)
        DRM_REQUIRE(!h->up, "double upsample");
        h->up = 1;
        h->H *= 2;
        h->W *= 2;
      }
    }
    return DRM_OK;
  };

  for (size_t b = 1; b < input_blocks.size(); ++b) {
    const bool down_next = b + 1 < input_blocks.size() && input_blocks[b + 1].size() == 1 && input_blocks[b + 1][0].kind == Layer::DOWN;
    DRM_TRY(run_layers(input_blocks[b], nullptr, down_next));
    hs.push_back(h);
  }
  DRM_TRY(run_layers(middle, nullptr));
  for (auto& blk : output_blocks) {
    Act* skip = hs.back();
    hs.pop_back();
    DRM_TRY(run_layers(blk, skip));
  }

  // head
  DRM_REQUIRE(!h->up && h->C == final_ch, "head input");
  float* sc = c.ar->alloc<float>((size_t)N * final_ch);
  float* sh = c.ar->alloc<float>((size_t)N * final_ch);
  DRM_TRY(arena_ok(c));
  DRM_TRY(gn_params(c, *h, nullptr, Wb + on_w, Wb + on_b, sc, sh, nullptr, nullptr));
  if (!c.dry()) {
    if (desc.kind == 0) {
      ConvArgs a;
      a.src0 = h->p; a.C0 = final_ch; a.N = N; a.H = h->H; a.W = h->W;
      a.gn_scale = sc; a.gn_shift = sh; a.silu = 1;
      a.w = Wb + oc_w; a.bias = Wb + oc_b; a.taps = 9; a.Cout = out_cp;
      a.out = out; a.out_nchw = 1; a.cout_valid = desc.out_channels;
      DRM_TRY(run_conv(c, a, Wb, oc_s));
    } else {
      DRM_TRY(launch_encoder_head(h->p, sc, sh, Wb + oc_w, Wb + oc_b, out, N, h->H * h->W, final_ch, desc.out_channels, s));
    }
  }
  if (ar.failed) {
    set_error("workspace too small: need " + std::to_string(ar.peak) + " bytes, got " + std::to_string(ar.cap));
    return DRM_ERR_WORKSPACE;
  }
  return DRM_OK;
}

}  // namespace drm
