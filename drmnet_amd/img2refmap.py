"""Object image -> reflectance map (the step in front of the samplers), same call surface as the reference's
utils/img2refmap.py:6-37 `refmap_mask_make`, plus the mask erosion that scripts/estimate.py:43-50 does inline.

Device tensors in, device tensors out; the work runs in drmnet_amd/csrc/refmap.hip behind the C ABI
(drm_refmap_mask_make / drm_erode_mask).  No CPU path: CPU tensors are rejected.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

_ws = _lib.Workspace()


def refmap_mask_make(
    colors: torch.Tensor,  # [n, 3]
    normals: torch.Tensor,  # [n, 3]
    res: int,
    angle_threshold: Optional[float] = None,
    min_points: int = 0,
    refmap_batch_size: int = 512,  # the reference's memory knob; the HIP path bins pixels instead and ignores it
) -> Tuple[torch.Tensor, torch.Tensor]:
    if angle_threshold is None:
        raise TypeError("angle_threshold is required (the reference compares against it unconditionally)")
    if not colors.is_cuda or not normals.is_cuda:
        raise RuntimeError("refmap_mask_make runs on the GPU only: pass CUDA/HIP tensors")
    if colors.ndim != 2 or normals.ndim != 2 or normals.shape[1] != 3 or colors.shape[0] != normals.shape[0]:
        raise ValueError("colors must be [n, C] and normals [n, 3]")
    colors = colors.contiguous().float()
    normals = normals.contiguous().float()
    n, ch = colors.shape
    L = _lib.lib()
    refmap = torch.empty((res, res, ch), dtype=torch.float32, device=colors.device)
    refmask = torch.empty((res, res), dtype=torch.uint8, device=colors.device)
    with torch.cuda.device(colors.device):
        nbytes = L.drm_refmap_workspace_bytes(n, res, float(angle_threshold))
        ws = _ws.get(nbytes, colors.device)
        _lib.check(L.drm_refmap_mask_make(colors.data_ptr(), normals.data_ptr(), n, ch, res, float(angle_threshold), int(min_points),
                                          refmap.data_ptr(), refmask.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr(colors.device)))
    return refmap, refmask.bool()


def erode_mask(mask: torch.Tensor, erode_kernel_size: int = 5) -> torch.Tensor:
    """scripts/estimate.py:43-50 ("edge removing"): mask [H, W] bool -> mask with its inner rim removed."""
    if not mask.is_cuda:
        raise RuntimeError("erode_mask runs on the GPU only: pass a CUDA/HIP tensor")
    if erode_kernel_size <= 0:
        return mask
    m8 = mask.to(torch.uint8).contiguous()
    out = torch.empty_like(m8)
    with torch.cuda.device(mask.device):
        _lib.check(_lib.lib().drm_erode_mask(m8.data_ptr(), m8.shape[0], m8.shape[1], int(erode_kernel_size), out.data_ptr(),
                                             _lib.stream_ptr(mask.device)))
    return out.bool()
