// extern "C" surface of libdrmnet_hip.so (declared in include/drmnet_hip.h).
#include <atomic>
#include <cstring>
#include <memory>

#include "profiler.h"
#include "samplers.h"

using namespace drm;

struct drm_unet {
  UNet net;
};
struct drm_drmnet {
  DrmnetSampler s;
};

namespace {

// RAII device scratch for the op-level entry points (tests / per-module drop-ins only).
struct Scratch {
  void* p = nullptr;
  hipStream_t s;
  explicit Scratch(hipStream_t st) : s(st) {}
  int reserve(size_t bytes) { DRM_HIP_CHECK(hipMalloc(&p, bytes)); return DRM_OK; }
  ~Scratch() {
    if (p) {
      (void)hipStreamSynchronize(s);
      (void)hipFree(p);
    }
  }
};

int make_arena(Arena& ar, void* ws, size_t bytes, size_t need) {
  if (need == 0) return DRM_ERR_INVALID;  // error text already set by the dry run
  if (bytes < need || ws == nullptr) {
    set_error("workspace too small: need " + std::to_string(need) + " bytes, got " + std::to_string(bytes));
    return DRM_ERR_WORKSPACE;
  }
  ar.base = static_cast<char*>(ws);
  ar.cap = bytes;
  return DRM_OK;
}

template <typename F>
int guarded(F&& f) {
  try {
    return f();
  } catch (const std::exception& e) {
    set_error(std::string("exception: ") + e.what());
    return DRM_ERR_STATE;
  } catch (...) {
    set_error("unknown exception");
    return DRM_ERR_STATE;
  }
}

// shared driver for the two block-level ops: runs `body` twice (measure, then execute) over a private scratch arena
template <typename Body>
int with_scratch(hipStream_t s, Body&& body) {
  Arena dry;
  dry.dry = true;
  DRM_TRY(body(dry));
  Scratch sc(s);
  DRM_TRY(sc.reserve(dry.peak + 256));
  Arena ar;
  ar.base = static_cast<char*>(sc.p);
  ar.cap = dry.peak + 256;
  DRM_HIP_CHECK(hipMemsetAsync(sc.p, 0, ar.cap, s));
  DRM_TRY(body(ar));
  DRM_HIP_CHECK(hipStreamSynchronize(s));
  return DRM_OK;
}

// Arithmetic of the drm_op_* entry points (drm_set_op_precision).  The one piece of process-wide state behind the C ABI: an atomic word that every
// drm_op_* call reads ONCE at entry (a call runs in one mode from its weight packing to its last launch); a concurrent drm_set_op_precision takes
// effect for calls that start after it.  The network / sampler handles carry their own mode (drm_unet_set_precision).
std::atomic<int> g_op_precision{PREC_FP32};

// packs one conv weight for the op precision `prec`; `slot` = 64-float scale slot (2^k, 2^-k), `scratch` = 1 uint
int pack_for_ops(int prec, const float* w, float* dst, float* slot, float* scratch, int cout, int cin, int taps, int coutp, int cinp, hipStream_t s,
                 bool mx_site = false) {
  if (prec != PREC_FP32 && cinp % 32 == 0)
    return launch_pack_conv_weight_split(w, dst, slot, reinterpret_cast<unsigned*>(scratch), cout, cin, taps, coutp, cinp, s,
                                         prec == PREC_F16MX && mx_site, prec == PREC_BF16);
  return launch_pack_conv_weight(w, dst, cout, cin, taps, coutp, cinp, s);
}

size_t unet_ws(UNet& net, int N, int H, int W) {
  Arena a;
  a.dry = true;
  const int Cx = net.desc.kind == 0 ? net.desc.out_channels : net.desc.in_channels / 2;
  if (net.forward(nullptr, Cx, nullptr, net.desc.in_channels - Cx, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, a, nullptr) != DRM_OK) return 0;
  return a.peak + 256;
}

}  // namespace

extern "C" {

int drm_abi_version(void) { return DRM_ABI_VERSION; }
const char* drm_last_error(void) { return last_error(); }

int drm_unet_create(const drm_unet_desc* desc, drm_unet** out) {
  return guarded([&]() -> int {
    DRM_REQUIRE(desc && out, "null argument");
    std::unique_ptr<drm_unet> h(new drm_unet());
    DRM_TRY(h->net.build(*desc));
    *out = h.release();
    return DRM_OK;
  });
}
void drm_unet_destroy(drm_unet* net) { delete net; }

int drm_unet_param_count(const drm_unet* net) { return net ? (int)net->net.params.size() : -1; }

int drm_unet_param_info(const drm_unet* net, int index, char* name, int name_cap, int64_t shape[4], int* ndim) {
  return guarded([&]() -> int {
    DRM_REQUIRE(net && index >= 0 && index < (int)net->net.params.size(), "param index");
    const ParamSlot& p = net->net.params[index];
    if (name && name_cap > 0) {
      std::strncpy(name, p.name.c_str(), name_cap - 1);
      name[name_cap - 1] = 0;
    }
    if (ndim) *ndim = (int)p.shape.size();
    if (shape)
      for (size_t i = 0; i < 4; ++i) shape[i] = i < p.shape.size() ? p.shape[i] : 1;
    return DRM_OK;
  });
}

int drm_unet_load_params(drm_unet* net, const float* const* ptrs, int count, void* stream) {
  return guarded([&]() -> int {
    DRM_REQUIRE(net && ptrs, "null argument");
    return net->net.load(ptrs, count, static_cast<hipStream_t>(stream), net->net.active);
  });
}

int drm_unet_load_params_set(drm_unet* net, int set, const float* const* ptrs, int count, void* stream) {
  return guarded([&]() -> int {
    DRM_REQUIRE(net && ptrs, "null argument");
    return net->net.load(ptrs, count, static_cast<hipStream_t>(stream), set);
  });
}

int drm_unet_use_set(drm_unet* net, int set) {
  return guarded([&]() -> int {
    DRM_REQUIRE(net && set >= 0 && set < UNet::NSETS, "weight set index");
    net->net.active = set;
    return DRM_OK;
  });
}

size_t drm_unet_workspace_bytes(const drm_unet* net, int N, int H, int W) {
  if (!net) return 0;
  size_t r = 0;
  guarded([&]() -> int {
    r = unet_ws(const_cast<drm_unet*>(net)->net, N, H, W);
    return DRM_OK;
  });
  return r;
}

int drm_unet_forward(drm_unet* net, const float* x, int Cx, const float* cond, int Cc, const int32_t* rows, const float* t_emb,
                     const int64_t* timesteps, const float* timesteps_f, float* out, int N, int H, int W, void* workspace,
                     size_t workspace_bytes, void* stream) {
  return guarded([&]() -> int {
    DRM_REQUIRE(net && x && out, "null argument");
    DRM_REQUIRE(Cc == 0 || cond, "cond is null but Cc > 0");
    Arena ar;
    DRM_TRY(make_arena(ar, workspace, workspace_bytes, unet_ws(net->net, N, H, W)));
    return net->net.forward(x, Cx, cond, Cc, rows, t_emb, timesteps, timesteps_f, out, N, H, W, ar, static_cast<hipStream_t>(stream));
  });
}

// ------------------------------------------------------------------------------------------------ primitive ops

int drm_linear_forward(const float* in, const float* w, const float* b, float* out, int N, int I, int O, int silu_in, int silu_out, void* stream) {
  return guarded([&]() -> int { return launch_linear(in, w, b, out, N, I, O, silu_in, silu_out, static_cast<hipStream_t>(stream)); });
}

int drm_timestep_embedding(const int64_t* timesteps, float* out, int N, int dim, void* stream) {
  return guarded([&]() -> int { return launch_timestep_embedding(timesteps, nullptr, out, N, dim, static_cast<hipStream_t>(stream)); });
}

int drm_op_norm_act_conv(const float* x, const float* gamma, const float* beta, int silu, const float* w, const float* b, int ksize,
                         const float* emb, const float* residual, float* out, int N, int Cin, int Cout, int H, int W, void* stream) {
  return guarded([&]() -> int {
    hipStream_t s = static_cast<hipStream_t>(stream);
    DRM_REQUIRE(ksize == 1 || ksize == 3, "ksize");
    DRM_REQUIRE((gamma == nullptr) == (beta == nullptr), "gamma/beta");
    DRM_REQUIRE(!gamma || Cin % 32 == 0, "GroupNorm32 needs Cin % 32 == 0");
    const int taps = ksize * ksize;
    const int cinp = (ksize == 1) ? (Cin + 31) / 32 * 32 : (Cin + 7) / 8 * 8, coutp = (Cout + 31) / 32 * 32;
    DRM_REQUIRE((!residual && !emb) || Cout == coutp, "residual/emb need Cout % 32 == 0");
    const size_t hw = (size_t)H * W;
    return with_scratch(s, [&](Arena& ar) -> int {
      const int op_prec = g_op_precision.load(std::memory_order_relaxed);  // (read once: the whole call runs in this mode)
      Ctx c{&ar, s, N, op_prec};
      Act xa = new_act(c, cinp, H, W);
      float* wb = ar.alloc<float>(64);  // base for offsets: [0..63] = scale slot, then scratch
      float* scratch = ar.alloc<float>(64);
      float* wp = ar.alloc<float>(packed_conv_weight_floats(taps, coutp, cinp));
      float* bp = ar.alloc<float>(coutp);
      float* sc = ar.alloc<float>((size_t)N * cinp);
      float* sh = ar.alloc<float>((size_t)N * cinp);
      float* resn = ar.alloc<float>(N * hw * coutp);
      float* gp = ar.alloc<float>(cinp);
      float* bpn = ar.alloc<float>(cinp);
      ConvArgs a;
      if (ar.dry) {
        xa.mom_valid = false;
        DRM_TRY(ensure_moments(c, xa));
        if (!gamma) DRM_TRY(raw_input_guard(c, a, &xa, 0, cinp, nullptr, nullptr, cinp));
        return DRM_OK;
      }
      DRM_TRY(launch_pack_input(x, nullptr, nullptr, xa.p, N, H, W, Cin, 0, cinp, s));
      DRM_TRY(pack_for_ops(op_prec, w, wp, wb, scratch, Cout, Cin, taps, coutp, cinp, s));
      if (b) DRM_HIP_CHECK(hipMemcpyAsync(bp, b, Cout * sizeof(float), hipMemcpyDeviceToDevice, s));
      if (!gamma) {
        xa.mom_valid = false;
        DRM_TRY(ensure_moments(c, xa));
        DRM_TRY(raw_input_guard(c, a, &xa, 0, cinp, nullptr, nullptr, cinp));
      }
      if (gamma) {
        DRM_HIP_CHECK(hipMemcpyAsync(gp, gamma, Cin * sizeof(float), hipMemcpyDeviceToDevice, s));
        DRM_HIP_CHECK(hipMemcpyAsync(bpn, beta, Cin * sizeof(float), hipMemcpyDeviceToDevice, s));
        DRM_TRY(gn_params(c, xa, nullptr, gp, bpn, sc, sh));
        a.gn_scale = sc; a.gn_shift = sh;
      }
      if (residual) {
        DRM_TRY(launch_nchw_to_nhwc(residual, resn, N, H, W, Cout, s));
        a.res = resn;
      }
      a.src0 = xa.p; a.C0 = cinp; a.N = N; a.H = H; a.W = W; a.silu = silu;
      a.w = wp; a.bias = bp; a.taps = taps; a.Cout = coutp;
      a.emb = emb; a.emb_stride = Cout;
      a.out = out; a.out_nchw = 1; a.cout_valid = Cout; a.cin_real = Cin;
      return run_conv(c, a, wb, 0);
    });
  });
}

int drm_op_resblock(const float* x0, int C0, int up0, const float* x1, int C1, const float* emb, int emb_dim, const float* const* params,
                    int n_params, float* out, int N, int Cout, int H, int W, void* stream) {
  return guarded([&]() -> int {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int cin = C0 + C1;
    const bool has_skip = cin != Cout;
    DRM_REQUIRE(n_params == (has_skip ? 12 : 10), "resblock expects 10 params (12 with skip_connection)");
    DRM_REQUIRE(cin % 32 == 0 && Cout % 32 == 0 && C0 % 32 == 0, "channels % 32");
    DRM_REQUIRE(emb_dim <= 512, "emb_dim <= 512");
    return with_scratch(s, [&](Arena& ar) -> int {
      const int op_prec = g_op_precision.load(std::memory_order_relaxed);  // (read once: the whole call runs in this mode)
      Ctx c{&ar, s, N, op_prec};
      // packed weights laid out like UNet::add_res
      ResLayer r;
      r.cin = cin; r.cout = Cout; r.has_skip = has_skip; r.emb_off = 0;
      float* wb = ar.alloc<float>(1);  // base pointer for offsets
      auto off = [&](float* p) { return (size_t)(p - wb); };
      float* n1w = ar.alloc<float>(cin); float* n1b = ar.alloc<float>(cin);
      float* c1w = ar.alloc<float>(packed_conv_weight_floats(9, Cout, cin)); float* c1b = ar.alloc<float>(Cout);
      float* n2w = ar.alloc<float>(Cout); float* n2b = ar.alloc<float>(Cout);
      float* c2w = ar.alloc<float>(packed_conv_weight_floats(9, Cout, Cout)); float* c2b = ar.alloc<float>(Cout);
      float* skw = ar.alloc<float>(packed_conv_weight_floats(1, Cout, cin)); float* skb = ar.alloc<float>(Cout);
      float* s1 = ar.alloc<float>(64); float* s2 = ar.alloc<float>(64); float* s3 = ar.alloc<float>(64); float* scratch = ar.alloc<float>(64);
      float* e_out = ar.alloc<float>((size_t)N * Cout);
      Act a0 = new_act(c, C0, H, W);
      a0.up = up0;
      Act a1 = new_act(c, C1 > 0 ? C1 : 4, H, W);
      Act o = new_act(c, Cout, H, W);
      if (!ar.dry) {
        r.n1_w = off(n1w); r.n1_b = off(n1b); r.c1_w = off(c1w); r.c1_b = off(c1b); r.n2_w = off(n2w); r.n2_b = off(n2b);
        r.c2_w = off(c2w); r.c2_b = off(c2b); r.sk_w = off(skw); r.sk_b = off(skb);
        r.c1_s = off(s1); r.c2_s = off(s2); r.sk_s = off(s3);
        auto cp = [&](float* d, const float* sp, size_t n) -> int {
          DRM_HIP_CHECK(hipMemcpyAsync(d, sp, n * sizeof(float), hipMemcpyDeviceToDevice, s));
          return DRM_OK;
        };
        DRM_TRY(cp(n1w, params[0], cin)); DRM_TRY(cp(n1b, params[1], cin));
        DRM_TRY(pack_for_ops(op_prec, params[2], c1w, s1, scratch, Cout, cin, 9, Cout, cin, s, true)); DRM_TRY(cp(c1b, params[3], Cout));
        DRM_TRY(launch_linear(emb, params[4], params[5], e_out, N, emb_dim, Cout, 1, 0, s));
        DRM_TRY(cp(n2w, params[6], Cout)); DRM_TRY(cp(n2b, params[7], Cout));
        DRM_TRY(pack_for_ops(op_prec, params[8], c2w, s2, scratch, Cout, Cout, 9, Cout, Cout, s, true)); DRM_TRY(cp(c2b, params[9], Cout));
        if (has_skip) {
          DRM_TRY(pack_for_ops(op_prec, params[10], skw, s3, scratch, Cout, cin, 1, Cout, cin, s)); DRM_TRY(cp(skb, params[11], Cout));
        }
        DRM_TRY(launch_nchw_to_nhwc(x0, a0.p, N, H >> up0, W >> up0, C0, s));
        if (C1 > 0) DRM_TRY(launch_nchw_to_nhwc(x1, a1.p, N, H, W, C1, s));
      }
      DRM_TRY(run_resblock(c, wb, r, a0, C1 > 0 ? &a1 : nullptr, e_out, Cout, o));
      if (!ar.dry) DRM_TRY(launch_nhwc_to_nchw(o.p, out, N, H, W, Cout, s));
      return DRM_OK;
    });
  });
}

int drm_op_attention_block(const float* x, const float* const* params, float* out, int N, int C, int H, int W, void* stream) {
  return guarded([&]() -> int {
    hipStream_t s = static_cast<hipStream_t>(stream);
    DRM_REQUIRE(C % 32 == 0, "channels % 32");
    return with_scratch(s, [&](Arena& ar) -> int {
      const int op_prec = g_op_precision.load(std::memory_order_relaxed);  // (read once: the whole call runs in this mode)
      Ctx c{&ar, s, N, op_prec};
      AttnLayer l;
      l.ch = C;
      float* wb = ar.alloc<float>(1);
      auto off = [&](float* p) { return (size_t)(p - wb); };
      float* nw = ar.alloc<float>(C); float* nb = ar.alloc<float>(C);
      float* qw = ar.alloc<float>(packed_conv_weight_floats(1, 3 * C, C)); float* qb = ar.alloc<float>(3 * C);
      float* pw = ar.alloc<float>(packed_conv_weight_floats(1, C, C)); float* pb = ar.alloc<float>(C);
      float* s1 = ar.alloc<float>(64); float* s2 = ar.alloc<float>(64); float* scratch = ar.alloc<float>(64);
      Act a = new_act(c, C, H, W);
      Act o = new_act(c, C, H, W);
      if (!ar.dry) {
        l.n_w = off(nw); l.n_b = off(nb); l.qkv_w = off(qw); l.qkv_b = off(qb); l.proj_w = off(pw); l.proj_b = off(pb); l.qkv_s = off(s1); l.proj_s = off(s2);
        DRM_HIP_CHECK(hipMemcpyAsync(nw, params[0], C * sizeof(float), hipMemcpyDeviceToDevice, s));
        DRM_HIP_CHECK(hipMemcpyAsync(nb, params[1], C * sizeof(float), hipMemcpyDeviceToDevice, s));
        DRM_TRY(pack_for_ops(op_prec, params[2], qw, s1, scratch, 3 * C, C, 1, 3 * C, C, s));
        DRM_HIP_CHECK(hipMemcpyAsync(qb, params[3], 3 * C * sizeof(float), hipMemcpyDeviceToDevice, s));
        DRM_TRY(pack_for_ops(op_prec, params[4], pw, s2, scratch, C, C, 1, C, C, s));
        DRM_HIP_CHECK(hipMemcpyAsync(pb, params[5], C * sizeof(float), hipMemcpyDeviceToDevice, s));
        DRM_TRY(launch_nchw_to_nhwc(x, a.p, N, H, W, C, s));
      }
      DRM_TRY(run_attention(c, wb, l, a, o));
      if (!ar.dry) DRM_TRY(launch_nhwc_to_nchw(o.p, out, N, H, W, C, s));
      return DRM_OK;
    });
  });
}

// ------------------------------------------------------------------------------------------------ samplers

int drm_drmnet_create(drm_unet* illnet, drm_unet* refnet, const float* const* zemb, const drm_drmnet_cfg* cfg, drm_drmnet** out) {
  return guarded([&]() -> int {
    DRM_REQUIRE(illnet && refnet && zemb && cfg && out, "null argument");
    std::unique_ptr<drm_drmnet> h(new drm_drmnet());
    DRM_TRY(h->s.init(&illnet->net, &refnet->net, zemb, *cfg));
    *out = h.release();
    return DRM_OK;
  });
}
void drm_drmnet_destroy(drm_drmnet* s) { delete s; }

size_t drm_drmnet_workspace_bytes(const drm_drmnet* s, int N, int H, int W) {
  if (!s) return 0;
  size_t r = 0;
  guarded([&]() -> int {
    r = s->s.workspace_bytes(N, H, W);
    return DRM_OK;
  });
  return r;
}

int drm_drmnet_set_batch_parts(drm_drmnet* s, int parts) {
  return guarded([&]() -> int {
    DRM_REQUIRE(s != nullptr, "drm_drmnet_set_batch_parts: null handle");
    DRM_REQUIRE(parts >= 1 && parts <= DrmnetSampler::PART_MAX, "drm_drmnet_set_batch_parts: 1 .. 4 parts");
    s->s.parts = parts;
    return DRM_OK;
  });
}

int drm_drmnet_set_batch_part_min(drm_drmnet* s, int rows) {
  return guarded([&]() -> int {
    DRM_REQUIRE(s != nullptr, "drm_drmnet_set_batch_part_min: null handle");
    DRM_REQUIRE(rows >= 1, "drm_drmnet_set_batch_part_min: at least one row per part");
    s->s.part_min = rows;
    return DRM_OK;
  });
}

int drm_drmnet_step(drm_drmnet* s, float* Lr_k, const float* LrK, const int32_t* rows, int n_active, int step, const float* noise,
                    uint64_t seed, float* zk_out, float* zK_out, int32_t* converged_out, int B, int H, int W, void* workspace,
                    size_t workspace_bytes, void* stream) {
  return guarded([&]() -> int {
    DRM_REQUIRE(s && Lr_k && LrK, "null argument");
    Arena ar;
    DRM_TRY(make_arena(ar, workspace, workspace_bytes, s->s.workspace_bytes(n_active, H, W)));
    return s->s.step(Lr_k, LrK, rows, n_active, step, noise, seed, zk_out, zK_out, converged_out, B, H, W, ar, static_cast<hipStream_t>(stream));
  });
}

int drm_drmnet_sample(drm_drmnet* s, const float* LrK, const float* cond, const float* noise0, const float* step_noise, uint64_t seed, int early_exit,
                      float* Lr0, float* zK, int32_t* K, int32_t* steps_done, int B, int H, int W, void* workspace, size_t workspace_bytes,
                      void* stream) {
  return guarded([&]() -> int {
    DRM_REQUIRE(s && LrK && cond && Lr0 && zK && K, "null argument");
    Arena ar;
    DRM_TRY(make_arena(ar, workspace, workspace_bytes, s->s.workspace_bytes(B, H, W)));
    return s->s.sample(LrK, cond, noise0, step_noise, seed, early_exit, Lr0, zK, K, steps_done, B, H, W, ar, static_cast<hipStream_t>(stream));
  });
}

size_t drm_sampler_workspace_bytes(const drm_unet* net, int N, int H, int W) {
  if (!net) return 0;
  size_t r = 0;
  guarded([&]() -> int {
    r = sampler_workspace_bytes(&const_cast<drm_unet*>(net)->net, N, H, W);
    return DRM_OK;
  });
  return r;
}

int drm_ddim_sample(drm_unet* net, float* x, const float* cond, const int64_t* timesteps, const float* coef, int S, int num_steps,
                    const float* noise, uint64_t seed, int N, int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
  return guarded([&]() -> int {
    DRM_REQUIRE(net && x && cond, "null argument");
    Arena ar;
    DRM_TRY(make_arena(ar, workspace, workspace_bytes, sampler_workspace_bytes(&net->net, N, H, W)));
    return ddim_sample(&net->net, x, cond, timesteps, coef, S, num_steps, noise, seed, N, H, W, ar, static_cast<hipStream_t>(stream));
  });
}

int drm_ddim_sample_logged(drm_unet* net, float* x, const float* cond, const int64_t* timesteps, const float* coef, int S, int num_steps,
                           const float* noise, uint64_t seed, int log_every_t, float* log_x, float* log_pred_x0, int log_slots, int32_t* n_logged, int N,
                           int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
  return guarded([&]() -> int {
    DRM_REQUIRE(net && x && cond, "null argument");
    DRM_REQUIRE(log_every_t > 0 && log_x && log_pred_x0 && log_slots > 0, "ddim intermediates: log_every_t, both log buffers and their slot count");
    Arena ar;
    DRM_TRY(make_arena(ar, workspace, workspace_bytes, sampler_workspace_bytes(&net->net, N, H, W)));
    int logged = 0;
    const int rc = ddim_sample(&net->net, x, cond, timesteps, coef, S, num_steps, noise, seed, N, H, W, ar, static_cast<hipStream_t>(stream), log_every_t, log_x,
                               log_pred_x0, log_slots, &logged);
    if (n_logged) *n_logged = logged;
    return rc;
  });
}

int drm_ddpm_sample(drm_unet* net, float* x, float* pred_x0, const float* cond, const float* coef, int T_start, int clip_denoised,
                    const float* noise, uint64_t seed, int N, int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
  return guarded([&]() -> int {
    DRM_REQUIRE(net && x && cond, "null argument");
    Arena ar;
    DRM_TRY(make_arena(ar, workspace, workspace_bytes, sampler_workspace_bytes(&net->net, N, H, W)));
    return ddpm_sample(&net->net, x, pred_x0, cond, coef, T_start, clip_denoised, noise, seed, N, H, W, ar, static_cast<hipStream_t>(stream));
  });
}

static MaskBlend to_blend(const drm_mask_blend* b) {
  MaskBlend m;
  m.mask = b->mask; m.mask_channels = b->mask_channels; m.x0 = b->x0; m.qcoef = b->qcoef; m.qnoise = b->qnoise; m.when = b->when;
  return m;
}

int drm_ddim_sample_ex(drm_unet* net, float* x, const float* cond, const int64_t* timesteps, const float* coef, int S, int num_steps,
                       const float* noise, uint64_t seed, const drm_sampler_options* opt, int log_every_t, float* log_x, float* log_pred_x0, int log_slots,
                       int32_t* n_logged, int N, int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
  return guarded([&]() -> int {
    DRM_REQUIRE(net && x && cond && opt, "null argument");
    DRM_REQUIRE(log_every_t <= 0 || (log_x && log_pred_x0 && log_slots > 0), "ddim intermediates: both log buffers and their slot count");
    Arena ar;
    DRM_TRY(make_arena(ar, workspace, workspace_bytes, sampler_workspace_bytes(&net->net, N, H, W)));
    MaskBlend mb;
    if (opt->blend) mb = to_blend(opt->blend);
    int logged = 0;
    const int rc = ddim_sample(&net->net, x, cond, timesteps, coef, S, num_steps, noise, seed, N, H, W, ar, static_cast<hipStream_t>(stream),
                               log_every_t > 0 ? log_every_t : 0, log_every_t > 0 ? log_x : nullptr, log_every_t > 0 ? log_pred_x0 : nullptr, log_slots, &logged,
                               opt->blend ? &mb : nullptr, opt->uncond, opt->guidance_scale, opt->noise_dropout, opt->dropout_keep);
    if (n_logged) *n_logged = logged;
    return rc;
  });
}

int drm_ddpm_sample_ex(drm_unet* net, float* x, float* pred_x0, const float* cond, const float* coef, int T_start, int clip_denoised,
                       const float* noise, uint64_t seed, const drm_sampler_options* opt, int N, int H, int W, void* workspace, size_t workspace_bytes,
                       void* stream) {
  return guarded([&]() -> int {
    DRM_REQUIRE(net && x && cond && opt, "null argument");
    DRM_REQUIRE(opt->uncond == nullptr, "drm_ddpm_sample_ex: classifier-free guidance exists on the DDIM chain only (ddim.py:225-232; ddpm.py p_sample has none)");
    Arena ar;
    DRM_TRY(make_arena(ar, workspace, workspace_bytes, sampler_workspace_bytes(&net->net, N, H, W)));
    MaskBlend mb;
    if (opt->blend) mb = to_blend(opt->blend);
    return ddpm_sample(&net->net, x, pred_x0, cond, coef, T_start, clip_denoised, noise, seed, N, H, W, ar, static_cast<hipStream_t>(stream), opt->blend ? &mb : nullptr,
                       opt->noise_dropout, opt->dropout_keep);
  });
}

int drm_unet_set_precision(drm_unet* net, int precision) {
  return guarded([&]() -> int {
    DRM_REQUIRE(net && precision_valid(precision),
                "precision must be 0 (fp32 MFMA), 1 (split fp16 x3), 2 (plain fp16 operands), 3 (split fp16 with fp8 cross terms) or 4 (plain bf16 operands)");
    net->net.precision = precision;
    return DRM_OK;
  });
}
int drm_set_op_precision(int precision) {
  return guarded([&]() -> int {
    DRM_REQUIRE(precision_valid(precision),
                "precision must be 0 (fp32 MFMA), 1 (split fp16 x3), 2 (plain fp16 operands), 3 (split fp16 with fp8 cross terms) or 4 (plain bf16 operands)");
    g_op_precision.store(precision, std::memory_order_relaxed);
    return DRM_OK;
  });
}

int drm_set_graph_replay(int on) {
  set_graph_replay(on != 0);
  return DRM_OK;
}
int64_t drm_graph_launches(void) { return (int64_t)graph_launches(); }

void drm_profile_enable(int on) { prof_enable(on); }
void drm_profile_reset(void) { prof_reset(); }
int drm_profile_collect(double* ms, double* flops, double* bytes, int64_t* launches) {
  return guarded([&]() -> int {
    DRM_REQUIRE(ms && flops && bytes && launches, "null argument");
    prof_collect(ms, flops, bytes, launches);
    return DRM_OK;
  });
}

int drm_randn(float* out, size_t n, uint64_t seed, uint64_t offset, void* stream) {
  return guarded([&]() -> int { return launch_randn(out, n, seed, offset, static_cast<hipStream_t>(stream)); });
}

size_t drm_refmap_workspace_bytes(int64_t n, int res, float angle_threshold) {
  if (n < 0 || res <= 0 || !(angle_threshold >= 0.f)) return 0;
  return refmap_workspace_bytes(n, res, angle_threshold);
}

int drm_refmap_mask_make(const float* colors, const float* normals, int64_t n, int channels, int res, float angle_threshold, int min_points,
                         float* refmap, uint8_t* refmask, void* workspace, size_t workspace_bytes, void* stream) {
  return guarded([&]() -> int {
    DRM_REQUIRE(refmap && refmask && workspace && (n == 0 || (colors && normals)), "drm_refmap_mask_make: null pointer");
    return launch_refmap_mask_make(colors, normals, n, channels, res, angle_threshold, min_points, refmap, refmask, workspace, workspace_bytes,
                                   static_cast<hipStream_t>(stream));
  });
}

int drm_erode_mask(const uint8_t* mask, int H, int W, int kernel_size, uint8_t* out, void* stream) {
  return guarded([&]() -> int {
    DRM_REQUIRE(mask && out, "drm_erode_mask: null pointer");
    return launch_erode_mask(mask, H, W, kernel_size, out, static_cast<hipStream_t>(stream));
  });
}

// ------------------------------------------------------------------------------------------------ boundary maps (transform.hip)

int drm_map_chain(const float* x, float* out, int64_t per_image, int B, const int32_t* ops, const float* args, int n_ops, const float* lo,
                  const float* hi, const float* scale, void* stream) {
  return guarded([&]() -> int { return launch_map_chain(x, out, per_image, B, ops, args, n_ops, lo, hi, scale, static_cast<hipStream_t>(stream)); });
}

int drm_masked_log_range(const float* x, const float* mask, int B, int C, int HW, float* lo, float* hi, void* stream) {
  return guarded([&]() -> int { return launch_masked_log_range(x, mask, B, C, HW, lo, hi, static_cast<hipStream_t>(stream)); });
}

int drm_luminance_scale(const float* x, int B, int HW, float scaler, float* scale, void* stream) {
  return guarded([&]() -> int { return launch_luminance_scale(x, B, HW, scaler, scale, static_cast<hipStream_t>(stream)); });
}

int drm_mirmap2envmap(const float* mirmap, const float* basis, float* out, int B, int C, int H, int W, int OH, int OW, int log_scale_interpolation,
                      int channels_last, void* stream) {
  return guarded([&]() -> int {
    return launch_mirmap2envmap(mirmap, basis, out, B, C, H, W, OH, OW, log_scale_interpolation, channels_last, static_cast<hipStream_t>(stream));
  });
}

int drm_hdr2ldr(const float* x, const uint8_t* mask, int HW, float alpha, float gamma, float* out, void* stream) {
  return guarded([&]() -> int { return launch_hdr2ldr(x, mask, HW, alpha, gamma, out, static_cast<hipStream_t>(stream)); });
}

size_t drm_profile_variants(char* buf, size_t cap) { return prof_variants_text(buf, cap); }

int drm_resize(const float* x, float* out, int planes, int IH, int IW, int OH, int OW, int mode, void* stream) {
  return guarded([&]() -> int { return launch_resize(x, out, planes, IH, IW, OH, OW, mode, static_cast<hipStream_t>(stream)); });
}

}  // extern "C"
