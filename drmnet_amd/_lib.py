"""ctypes binding of include/drmnet_hip.h (libdrmnet_hip.so).

The product path has NO fallback: if the library is missing or a call fails, a
RuntimeError is raised (the reference's only error style is Python exceptions /
asserts, e.g. openaimodel.py:740-749).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DRM_LIB_PATH: an explicitly named alternative build (tools/stamp_probe.sh uses it for its diagnostic library); never set by the product
LIB_PATH = os.environ.get("DRM_LIB_PATH") or os.path.join(_HERE, "csrc", "libdrmnet_hip.so")
ABI_VERSION = 3
MAX_LEVELS = 8

# every symbol include/drmnet_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "drm_abi_version", "drm_last_error",
    "drm_unet_create", "drm_unet_destroy", "drm_unet_param_count", "drm_unet_param_info", "drm_unet_load_params",
    "drm_unet_workspace_bytes", "drm_unet_forward",
    "drm_linear_forward", "drm_timestep_embedding", "drm_op_norm_act_conv", "drm_op_resblock", "drm_op_attention_block",
    "drm_drmnet_create", "drm_drmnet_destroy", "drm_drmnet_workspace_bytes", "drm_drmnet_set_batch_parts", "drm_drmnet_set_batch_part_min", "drm_drmnet_step", "drm_drmnet_sample",
    "drm_sampler_workspace_bytes", "drm_ddim_sample", "drm_ddim_sample_logged", "drm_ddpm_sample", "drm_ddim_sample_ex", "drm_ddpm_sample_ex", "drm_randn",
    "drm_profile_enable", "drm_profile_reset", "drm_profile_collect", "drm_unet_set_precision", "drm_set_op_precision",
    "drm_refmap_workspace_bytes", "drm_refmap_mask_make", "drm_erode_mask",
    "drm_unet_load_params_set", "drm_unet_use_set", "drm_set_graph_replay", "drm_graph_launches",
    "drm_map_chain", "drm_masked_log_range", "drm_luminance_scale", "drm_mirmap2envmap", "drm_hdr2ldr", "drm_resize", "drm_profile_variants",
]


class UNetDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("in_channels", C.c_int32), ("model_channels", C.c_int32), ("out_channels", C.c_int32),
        ("num_res_blocks", C.c_int32), ("n_levels", C.c_int32), ("channel_mult", C.c_int32 * MAX_LEVELS),
        ("n_attn", C.c_int32), ("attention_resolutions", C.c_int32 * MAX_LEVELS),
    ]


class MaskBlend(C.Structure):  # drm_mask_blend
    _fields_ = [("mask", C.c_void_p), ("mask_channels", C.c_int32), ("x0", C.c_void_p), ("qcoef", C.POINTER(C.c_float)), ("qnoise", C.c_void_p), ("when", C.c_int32)]


class SamplerOptions(C.Structure):  # drm_sampler_options
    _fields_ = [("blend", C.POINTER(MaskBlend)), ("uncond", C.c_void_p), ("guidance_scale", C.c_float), ("noise_dropout", C.c_float), ("dropout_keep", C.c_void_p)]


class DrmnetCfg(C.Structure):
    _fields_ = [
        ("z_dim", C.c_int32), ("max_timesteps", C.c_int32), ("gamma", C.c_double), ("epsilon", C.c_float), ("delta", C.c_float),
        ("z0", C.c_float * 8),
    ]


def make_mask_blend(mask, x0, qcoef, qnoise, when: int, img_shape):
    """drm_mask_blend for a chain on `img_shape` [N,C,H,W]: mask [N,1|C,H,W] (or broadcastable [1|N,1|C,H,W]), x0 like img, qcoef float32 [steps,2] (numpy, host),
    qnoise [steps,N,C,H,W] or None.  Returns (struct, keep-alive tuple)."""
    import numpy as np
    import torch

    n, c, h, w = img_shape
    x0 = require_gpu_tensor(x0, "x0").float().contiguous()
    if tuple(x0.shape) != tuple(img_shape):
        raise RuntimeError(f"x0 must be {tuple(img_shape)}")
    mask = require_gpu_tensor(mask, "mask").float()
    if mask.dim() != 4 or mask.shape[2:] != x0.shape[2:] or mask.shape[1] not in (1, c) or mask.shape[0] not in (1, n):
        raise RuntimeError(f"mask must be [N or 1, 1 or {c}, {h}, {w}]")
    mask = mask.expand(n, mask.shape[1], h, w).contiguous()
    qc = np.ascontiguousarray(np.asarray(qcoef, dtype=np.float32))
    qn = None if qnoise is None else require_gpu_tensor(qnoise, "mask_noise").float().contiguous()
    if qn is not None and tuple(qn.shape) != (qc.shape[0],) + tuple(img_shape):
        raise RuntimeError(f"mask_noise must be [steps={qc.shape[0]}, N, C, H, W]")
    b = MaskBlend()
    b.mask, b.mask_channels, b.x0 = mask.data_ptr(), int(mask.shape[1]), x0.data_ptr()
    b.qcoef = qc.ctypes.data_as(C.POINTER(C.c_float))
    b.qnoise = None if qn is None else qn.data_ptr()
    b.when = int(when)
    return b, (mask, x0, qc, qn)


def make_sampler_options(img_shape, steps, blend=None, uncond=None, guidance_scale=1.0, noise_dropout=0.0, dropout_keep=None):
    """drm_sampler_options for a chain on `img_shape`: blend = (struct, keep-alive) of make_mask_blend or None; uncond [N,C,H,W]; dropout_keep
    [steps,N,C,H,W] 0 / 1 or None.  Returns (struct, keep-alive tuple)."""
    o = SamplerOptions()
    keep = [blend]
    if blend is not None:
        o.blend = C.pointer(blend[0])
    if uncond is not None:
        u = require_gpu_tensor(uncond, "unconditional_conditioning").float().contiguous()
        if tuple(u.shape) != tuple(img_shape):
            raise RuntimeError(f"unconditional_conditioning must be {tuple(img_shape)}")
        o.uncond = u.data_ptr()
        keep.append(u)
    o.guidance_scale = float(guidance_scale)
    o.noise_dropout = float(noise_dropout)
    if dropout_keep is not None:
        k = require_gpu_tensor(dropout_keep, "dropout_keep").float().contiguous()
        if tuple(k.shape) != (int(steps),) + tuple(img_shape):
            raise RuntimeError(f"dropout_keep must be [steps={steps}, N, C, H, W]")
        o.dropout_keep = k.data_ptr()
        keep.append(k)
    return o, tuple(keep)


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m drmnet_amd.build` (hipcc, gfx950). "
            "drmnet_amd has no CPU / PyTorch fallback for the hot path."
        )
    L = C.CDLL(LIB_PATH)
    vp, i32, i64p, fp = C.c_void_p, C.c_int, C.c_void_p, C.c_void_p
    L.drm_abi_version.restype = C.c_int
    L.drm_last_error.restype = C.c_char_p
    L.drm_unet_create.argtypes = [C.POINTER(UNetDesc), C.POINTER(vp)]
    L.drm_unet_destroy.argtypes = [vp]
    L.drm_unet_destroy.restype = None
    L.drm_unet_param_count.argtypes = [vp]
    L.drm_unet_param_info.argtypes = [vp, i32, C.c_char_p, i32, C.POINTER(C.c_int64), C.POINTER(C.c_int)]
    L.drm_unet_load_params.argtypes = [vp, C.POINTER(vp), i32, vp]
    L.drm_unet_load_params_set.argtypes = [vp, i32, C.POINTER(vp), i32, vp]
    L.drm_unet_use_set.argtypes = [vp, i32]
    L.drm_unet_workspace_bytes.argtypes = [vp, i32, i32, i32]
    L.drm_unet_workspace_bytes.restype = C.c_size_t
    L.drm_unet_forward.argtypes = [vp, fp, i32, fp, i32, vp, fp, i64p, fp, fp, i32, i32, i32, vp, C.c_size_t, vp]
    L.drm_linear_forward.argtypes = [fp, fp, fp, fp, i32, i32, i32, i32, i32, vp]
    L.drm_timestep_embedding.argtypes = [i64p, fp, i32, i32, vp]
    L.drm_op_norm_act_conv.argtypes = [fp, fp, fp, i32, fp, fp, i32, fp, fp, fp, i32, i32, i32, i32, i32, vp]
    L.drm_op_resblock.argtypes = [fp, i32, i32, fp, i32, fp, i32, C.POINTER(vp), i32, fp, i32, i32, i32, i32, vp]
    L.drm_op_attention_block.argtypes = [fp, C.POINTER(vp), fp, i32, i32, i32, i32, vp]
    L.drm_drmnet_create.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(DrmnetCfg), C.POINTER(vp)]
    L.drm_drmnet_destroy.argtypes = [vp]
    L.drm_drmnet_destroy.restype = None
    L.drm_drmnet_workspace_bytes.argtypes = [vp, i32, i32, i32]
    L.drm_drmnet_workspace_bytes.restype = C.c_size_t
    L.drm_drmnet_set_batch_parts.argtypes = [vp, i32]
    L.drm_drmnet_set_batch_parts.restype = i32
    L.drm_drmnet_set_batch_part_min.argtypes = [vp, i32]
    L.drm_drmnet_set_batch_part_min.restype = i32
    L.drm_drmnet_step.argtypes = [vp, fp, fp, vp, i32, i32, fp, C.c_uint64, fp, fp, vp, i32, i32, i32, vp, C.c_size_t, vp]
    L.drm_drmnet_sample.argtypes = [vp, fp, fp, fp, fp, C.c_uint64, i32, fp, fp, vp, C.POINTER(C.c_int32), i32, i32, i32, vp, C.c_size_t, vp]
    L.drm_sampler_workspace_bytes.argtypes = [vp, i32, i32, i32]
    L.drm_sampler_workspace_bytes.restype = C.c_size_t
    L.drm_ddim_sample.argtypes = [vp, fp, fp, C.POINTER(C.c_int64), C.POINTER(C.c_float), i32, i32, fp, C.c_uint64, i32, i32, i32, vp, C.c_size_t, vp]
    L.drm_ddim_sample_logged.argtypes = [vp, fp, fp, C.POINTER(C.c_int64), C.POINTER(C.c_float), i32, i32, fp, C.c_uint64, i32, fp, fp, i32, C.POINTER(C.c_int32),
                                         i32, i32, i32, vp, C.c_size_t, vp]
    L.drm_ddpm_sample.argtypes = [vp, fp, fp, fp, C.POINTER(C.c_float), i32, i32, fp, C.c_uint64, i32, i32, i32, vp, C.c_size_t, vp]
    L.drm_ddim_sample_ex.argtypes = [vp, fp, fp, C.POINTER(C.c_int64), C.POINTER(C.c_float), i32, i32, fp, C.c_uint64, C.POINTER(SamplerOptions), i32, fp, fp, i32,
                                     C.POINTER(C.c_int32), i32, i32, i32, vp, C.c_size_t, vp]
    L.drm_ddpm_sample_ex.argtypes = [vp, fp, fp, fp, C.POINTER(C.c_float), i32, i32, fp, C.c_uint64, C.POINTER(SamplerOptions), i32, i32, i32, vp, C.c_size_t, vp]
    L.drm_randn.argtypes = [fp, C.c_size_t, C.c_uint64, C.c_uint64, vp]
    u8p = vp
    L.drm_refmap_workspace_bytes.argtypes = [C.c_int64, i32, C.c_float]
    L.drm_refmap_workspace_bytes.restype = C.c_size_t
    L.drm_refmap_mask_make.argtypes = [fp, fp, C.c_int64, i32, i32, C.c_float, i32, fp, u8p, vp, C.c_size_t, vp]
    L.drm_erode_mask.argtypes = [u8p, i32, i32, i32, u8p, vp]
    L.drm_unet_set_precision.argtypes = [vp, i32]
    L.drm_set_op_precision.argtypes = [i32]
    L.drm_profile_enable.argtypes = [i32]
    L.drm_profile_enable.restype = None
    L.drm_profile_reset.restype = None
    L.drm_profile_collect.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.drm_set_graph_replay.argtypes = [i32]
    L.drm_graph_launches.restype = C.c_int64
    L.drm_map_chain.argtypes = [fp, fp, C.c_int64, i32, C.POINTER(C.c_int32), C.POINTER(C.c_float), i32, fp, fp, fp, vp]
    L.drm_masked_log_range.argtypes = [fp, fp, i32, i32, i32, fp, fp, vp]
    L.drm_luminance_scale.argtypes = [fp, i32, i32, C.c_float, fp, vp]
    L.drm_mirmap2envmap.argtypes = [fp, fp, fp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    L.drm_hdr2ldr.argtypes = [fp, u8p, i32, C.c_float, C.c_float, fp, vp]
    L.drm_resize.argtypes = [fp, fp, i32, i32, i32, i32, i32, i32, vp]
    L.drm_profile_variants.argtypes = [C.c_char_p, C.c_size_t]
    L.drm_profile_variants.restype = C.c_size_t
    if L.drm_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libdrmnet_hip.so ABI version {L.drm_abi_version()} != {ABI_VERSION}: rebuild with `python -m drmnet_amd.build`")
    _lib = L
    return L


def check(status: int) -> None:
    if status != 0:
        msg = lib().drm_last_error()
        raise RuntimeError(f"drmnet_hip error {status}: {msg.decode() if msg else '?'}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu_tensor(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (drmnet_amd has no CPU path); got device {t.device}")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


def ptr_array(tensors: Sequence[torch.Tensor]):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


class Workspace:
    """Caller-owned scratch, grown on demand and reused (torch allocator owns the memory)."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None

    def get(self, nbytes: int, device) -> torch.Tensor:
        if nbytes == 0:
            raise RuntimeError(f"workspace query failed: {lib().drm_last_error().decode()}")
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != torch.device(device):
            self.buf = None
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self.buf
