"""Primitive ops of the HIP library on reference-layout tensors (NCHW activations, PyTorch-layout weights).

Per-module drop-ins for the reference's L1/L2 pieces (GroupNorm32+SiLU+conv_nd chains, ResBlock,
AttentionBlock, linear/SiLU embeddings, timestep_embedding).  GPU only.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import _lib


def set_precision(precision: str) -> None:
    """Arithmetic of the drm_op_* entry points: "fp32" (default), "f16x3" (split fp16, fp32-accurate), "f16mx" (f16x3 with the res blocks'
    3x3 convs on fp16 hi*hi + a block-scaled fp8 MFMA for the cross terms, ~7e-6 per block), "f16" or "bf16" (reduced precision)."""
    modes = {"fp32": 0, "f16x3": 1, "f16": 2, "f16mx": 3, "bf16": 4}
    if precision not in modes:
        raise ValueError(f"precision must be one of {list(modes)}")
    _lib.check(_lib.lib().drm_set_op_precision(modes[precision]))


def set_graph_replay(on: bool) -> None:
    """DDIM / DDPM chains replay one captured hipGraph of a step (default off); see include/drmnet_hip.h."""
    _lib.check(_lib.lib().drm_set_graph_replay(int(bool(on))))


def graph_launches() -> int:
    return int(_lib.lib().drm_graph_launches())


def _dev(t: torch.Tensor):
    return torch.cuda.device(t.device)


@torch.no_grad()
def linear(x, weight, bias=None, silu_in=False, silu_out=False):
    """act_out(F.linear(act_in(x), weight, bias)) -- openaimodel.py:218-224,521-526; models/drmnet.py:38-45."""
    x = _lib.require_gpu_tensor(x, "x")
    weight = _lib.require_gpu_tensor(weight, "weight")
    bias = None if bias is None else _lib.require_gpu_tensor(bias, "bias")
    n, i = x.shape
    o = weight.shape[0]
    out = torch.empty((n, o), dtype=torch.float32, device=x.device)
    with _dev(x):
        _lib.check(_lib.lib().drm_linear_forward(x.data_ptr(), weight.data_ptr(), _lib.ptr(bias), out.data_ptr(), n, i, o, int(silu_in), int(silu_out), _lib.stream_ptr(x.device)))
    return out


@torch.no_grad()
def timestep_embedding(timesteps, dim):
    """ldm/modules/diffusionmodules/util.py:151-171."""
    t = _lib.require_gpu_tensor(timesteps.long(), "timesteps", torch.int64)
    out = torch.empty((t.shape[0], dim), dtype=torch.float32, device=t.device)
    with _dev(t):
        _lib.check(_lib.lib().drm_timestep_embedding(t.data_ptr(), out.data_ptr(), t.shape[0], dim, _lib.stream_ptr(t.device)))
    return out


@torch.no_grad()
def norm_act_conv(x, weight, bias=None, gamma=None, beta=None, silu=False, emb=None, residual=None):
    """[GroupNorm32 -> [SiLU] ->] conv2d(k in {1,3}, pad k//2) [+ emb[:, :, None, None]] [+ residual]."""
    x = _lib.require_gpu_tensor(x, "x")
    weight = _lib.require_gpu_tensor(weight, "weight")
    n, cin, h, w = x.shape
    cout, k = weight.shape[0], weight.shape[-1]
    ts = [None if t is None else _lib.require_gpu_tensor(t, "arg") for t in (bias, gamma, beta, emb, residual)]
    bias, gamma, beta, emb, residual = ts
    out = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
    with _dev(x):
        _lib.check(_lib.lib().drm_op_norm_act_conv(x.data_ptr(), _lib.ptr(gamma), _lib.ptr(beta), int(silu), weight.data_ptr(), _lib.ptr(bias), k,
                                                    _lib.ptr(emb), _lib.ptr(residual), out.data_ptr(), n, cin, cout, h, w, _lib.stream_ptr(x.device)))
    return out


@torch.no_grad()
def resblock(params: Sequence[torch.Tensor], x0, emb, x1=None, up0=False):
    """ResBlock._forward (openaimodel.py:255-275) on cat([up(x0), x1], 1); params in state_dict order."""
    x0 = _lib.require_gpu_tensor(x0, "x0")
    emb = _lib.require_gpu_tensor(emb, "emb")
    params = [_lib.require_gpu_tensor(p, "param") for p in params]
    n, c0 = x0.shape[0], x0.shape[1]
    h, w = x0.shape[2] * (2 if up0 else 1), x0.shape[3] * (2 if up0 else 1)
    c1 = 0
    if x1 is not None:
        x1 = _lib.require_gpu_tensor(x1, "x1")
        c1 = x1.shape[1]
    cout = params[2].shape[0]
    out = torch.empty((n, cout, h, w), dtype=torch.float32, device=x0.device)
    arr = _lib.ptr_array(params)
    with _dev(x0):
        _lib.check(_lib.lib().drm_op_resblock(x0.data_ptr(), c0, int(up0), _lib.ptr(x1), c1, emb.data_ptr(), emb.shape[1], arr, len(params),
                                               out.data_ptr(), n, cout, h, w, _lib.stream_ptr(x0.device)))
    return out


@torch.no_grad()
def attention_block(params: Sequence[torch.Tensor], x):
    """AttentionBlock._forward (openaimodel.py:325-333); params = norm.w, norm.b, qkv.w, qkv.b, proj_out.w, proj_out.b."""
    x = _lib.require_gpu_tensor(x, "x")
    params = [_lib.require_gpu_tensor(p, "param") for p in params]
    n, c, h, w = x.shape
    out = torch.empty_like(x)
    arr = _lib.ptr_array(params)
    with _dev(x):
        _lib.check(_lib.lib().drm_op_attention_block(x.data_ptr(), arr, out.data_ptr(), n, c, h, w, _lib.stream_ptr(x.device)))
    return out


@torch.no_grad()
def randn(shape, seed: int, offset: int = 0, device="cuda"):
    """Standard normal from the library's Philox4x32-10 stream."""
    out = torch.empty(shape, dtype=torch.float32, device=device)
    with torch.cuda.device(out.device):
        _lib.check(_lib.lib().drm_randn(out.data_ptr(), out.numel(), seed, offset, _lib.stream_ptr(out.device)))
    return out


# ------------------------------------------------------------------------------------------------ boundary maps (csrc/transform.hip)

MAP_CODES = {"log_p1": 0, "log10": 1, "lowerbound": 2, "unit_to_signed": 3, "norm_log": 4, "exp_m1": 5, "exp10": 6, "signed_to_unit": 7,
             "denorm_log": 8, "img_mul": 9, "img_div": 10, "clip0": 11}
MAX_CHAIN = 8


def _per_image(x: torch.Tensor):
    """[B, C, H, W] -> (B, C*H*W); a 3-D [C, H, W] tensor is one image (the reference reduces over the last three dims)."""
    if x.ndim < 3:
        raise RuntimeError("expected a [(B,) C, H, W] tensor")
    b = 1
    for d in x.shape[:-3]:
        b *= int(d)
    return b, x.numel() // max(b, 1)


@torch.no_grad()
def map_chain(x, steps, lo=None, hi=None, scale=None, out=None):
    """Applies ``steps`` = [(map name, scalar argument), ...] to every element in ONE pass (drm_map_chain).
    lo / hi / scale: per-image fp32 vectors for the maps that need them.  Longer chains are cut into passes of 8."""
    import ctypes as C

    x = _lib.require_gpu_tensor(x, "x")
    b, per = _per_image(x)
    vec = lambda t, n: None if t is None else _lib.require_gpu_tensor(t.reshape(-1), n)
    lo, hi, scale = vec(lo, "lo"), vec(hi, "hi"), vec(scale, "scale")
    for n, t in (("lo", lo), ("hi", hi), ("scale", scale)):
        if t is not None and t.numel() != b:
            raise RuntimeError(f"{n} must have one entry per image ({b}), got {t.numel()}")
    res = torch.empty_like(x) if out is None else out
    src = x
    steps = list(steps)
    if not steps:
        res.copy_(x)
        return res
    with _dev(x):
        for k in range(0, len(steps), MAX_CHAIN):
            part = steps[k:k + MAX_CHAIN]
            ops_arr = (C.c_int32 * len(part))(*[MAP_CODES[n] for n, _ in part])
            arg_arr = (C.c_float * len(part))(*[float(a) for _, a in part])
            _lib.check(_lib.lib().drm_map_chain(src.data_ptr(), res.data_ptr(), per, b, ops_arr, arg_arr, len(part), _lib.ptr(lo), _lib.ptr(hi),
                                                _lib.ptr(scale), _lib.stream_ptr(x.device)))
            src = res
    return res


@torch.no_grad()
def masked_log_range(x, mask):
    """(log10 min, log10 max) per image of x under mask, the dynamic_normalize statistics of basedataset.py:63-69 -> two [B] tensors."""
    x = _lib.require_gpu_tensor(x, "x")
    b, per = _per_image(x)
    hw = x.shape[-1] * x.shape[-2]
    mask = _lib.require_gpu_tensor(mask.to(torch.float32), "mask")
    if mask.numel() != b * hw:
        raise NotImplementedError("mask must be [B, 1, H, W] (one plane per image, broadcast over channels)")
    lo = torch.empty((b,), dtype=torch.float32, device=x.device)
    hi = torch.empty_like(lo)
    with _dev(x):
        _lib.check(_lib.lib().drm_masked_log_range(x.data_ptr(), mask.data_ptr(), b, per // hw, hw, lo.data_ptr(), hi.data_ptr(), _lib.stream_ptr(x.device)))
    return lo, hi


@torch.no_grad()
def luminance_scale(x, scaler: float):
    """models/drmnet.py:1020-1026: scaler / geometric-mean luminance over the lit pixels, per image of x [B, 3, H, W] -> [B]."""
    x = _lib.require_gpu_tensor(x, "x")
    if x.ndim != 4 or x.shape[1] != 3:
        raise RuntimeError("luminance_scale expects [B, 3, H, W]")
    out = torch.empty((x.shape[0],), dtype=torch.float32, device=x.device)
    with _dev(x):
        _lib.check(_lib.lib().drm_luminance_scale(x.data_ptr(), x.shape[0], x.shape[2] * x.shape[3], float(scaler), out.data_ptr(), _lib.stream_ptr(x.device)))
    return out


@torch.no_grad()
def mirmap2envmap(mirmap, output_shape, basis=None, log_scale_interpolation=False, channels_last=False):
    """utils/transform.py:106-144 (+ the basis_r0 division of DRMNet.r0toenvmap when ``basis`` [C, H, W] is given)."""
    mirmap = _lib.require_gpu_tensor(mirmap, "mirmap")
    b, c, h, w = mirmap.shape
    oh, ow = int(output_shape[0]), int(output_shape[1])
    if basis is not None:
        basis = _lib.require_gpu_tensor(basis.expand(c, h, w), "basis_r0")
    out = torch.empty((b, oh, ow, c) if channels_last else (b, c, oh, ow), dtype=torch.float32, device=mirmap.device)
    with _dev(mirmap):
        _lib.check(_lib.lib().drm_mirmap2envmap(mirmap.data_ptr(), _lib.ptr(basis), out.data_ptr(), b, c, h, w, oh, ow, int(bool(log_scale_interpolation)),
                                                int(bool(channels_last)), _lib.stream_ptr(mirmap.device)))
    return out


@torch.no_grad()
def hdr2ldr(x, mask=None, alpha: float = 0.18, gamma: float = 2.2):
    """utils/tonemap.py:4-9 on a [H, W, 3] device tensor (mask: optional [H, W] bool / uint8)."""
    x = _lib.require_gpu_tensor(x, "x")
    if x.ndim != 3 or x.shape[2] != 3:
        raise RuntimeError("hdr2ldr expects [H, W, 3]")
    m = None if mask is None else _lib.require_gpu_tensor(mask.to(torch.uint8), "mask", torch.uint8)
    out = torch.empty_like(x)
    with _dev(x):
        _lib.check(_lib.lib().drm_hdr2ldr(x.data_ptr(), _lib.ptr(m), x.shape[0] * x.shape[1], float(alpha), float(gamma), out.data_ptr(), _lib.stream_ptr(x.device)))
    return out


RESIZE_MODES = {"nearest": 0, "bilinear": 1, "bicubic": 2}


@torch.no_grad()
def resize(x, size, mode: str = "bilinear"):
    """dataset/basedataset.py:44-50 (anti-aliased bilinear / bicubic, align_corners=False) and models/obsnet.py:691 (nearest) on the last
    two dimensions of a device tensor; every leading dimension is a plane."""
    x = _lib.require_gpu_tensor(x, "x")
    if mode not in RESIZE_MODES:
        raise NotImplementedError(f"resize mode {mode!r} (drm_resize implements {sorted(RESIZE_MODES)})")
    oh, ow = int(size[0]), int(size[1])
    ih, iw = int(x.shape[-2]), int(x.shape[-1])
    planes = x.numel() // (ih * iw)
    out = torch.empty(tuple(x.shape[:-2]) + (oh, ow), dtype=torch.float32, device=x.device)
    with _dev(x):
        _lib.check(_lib.lib().drm_resize(x.data_ptr(), out.data_ptr(), planes, ih, iw, oh, ow, RESIZE_MODES[mode], _lib.stream_ptr(x.device)))
    return out
