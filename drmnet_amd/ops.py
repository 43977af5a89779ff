"""Primitive ops of the HIP library on reference-layout tensors (NCHW activations, PyTorch-layout weights).

Per-module drop-ins for the reference's L1/L2 pieces (GroupNorm32+SiLU+conv_nd chains, ResBlock,
AttentionBlock, linear/SiLU embeddings, timestep_embedding).  GPU only.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import _lib


def set_precision(precision: str) -> None:
    """Arithmetic of the drm_op_* entry points: "fp32" (default) or "f16x3" (split fp16, fp32-accurate)."""
    modes = {"fp32": 0, "f16x3": 1, "f16": 2}
    if precision not in modes:
        raise ValueError(f"precision must be one of {list(modes)}")
    _lib.check(_lib.lib().drm_set_op_precision(modes[precision]))


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
