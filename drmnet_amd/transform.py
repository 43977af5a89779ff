"""Mirror reflectance map -> lat-long environment map warp (reference utils/transform.py:106-144).

The warp grid depends only on the shapes, so it is built once per (mirror size, envmap size, device) and cached; the
resampling itself is a bilinear ``grid_sample`` with border padding (host glue after the samplers, not the hot path).
Geometry: an envmap direction d (zenith +y, left edge -z, azimuth clockwise) is seen in the orthographic mirror ball at the
pixel whose normal is the half vector n = normalize(d + view), view = +z; the mirror map is parametrised by
(theta, phi) of n about top = +y / tangent = view.
"""
from __future__ import annotations

import math
from functools import lru_cache
from typing import Tuple

import torch


@lru_cache(maxsize=16)
def _envmap_grid(oh: int, ow: int, device: str) -> torch.Tensor:
    dev = torch.device(device)
    theta = (torch.arange(oh, device=dev, dtype=torch.float32) + 0.5) * (math.pi / oh)
    phi = -(torch.arange(ow, device=dev, dtype=torch.float32) + 0.5) * (2 * math.pi / ow)  # reverse_azimuth=True
    th, ph = torch.meshgrid(theta, phi, indexing="ij")
    # direction with zenith (0,1,0), tangent (0,0,-1), binormal = zenith x tangent = (-1,0,0)
    st = torch.sin(th)
    d = torch.stack([-st * torch.sin(ph), torch.cos(th), -st * torch.cos(ph)], dim=-1)
    n = torch.nn.functional.normalize(d + torch.tensor([0.0, 0.0, 1.0], device=dev), dim=-1, eps=1e-12)
    # (theta, phi) of n about normal (0,1,0), tangent (0,0,1), binormal = (1,0,0)
    n_theta = torch.arccos(n[..., 1])
    n_phi = torch.arctan2(n[..., 0], n[..., 2])
    u = n_phi * (2 / math.pi)
    v = n_theta * (2 / math.pi) - 1
    return torch.stack([u, v], dim=-1)


def mirmap2envmap(mirmap: torch.Tensor, output_shape: Tuple[int, int], log_scale_interpolation: bool = False) -> torch.Tensor:
    """mirmap [B,C,H,W] -> envmap [B,C,OH,OW] (view = +z, top = +y; the only configuration the reference supports)."""
    oh, ow = output_shape
    grid = _envmap_grid(int(oh), int(ow), str(mirmap.device)).to(mirmap.dtype)
    src = torch.log(mirmap.clip(1e-7)) if log_scale_interpolation else mirmap
    env = torch.nn.functional.grid_sample(src, grid[None].expand(mirmap.size(0), -1, -1, -1), mode="bilinear", padding_mode="border",
                                          align_corners=False)
    return torch.exp(env) if log_scale_interpolation else env


def hdr2ldr(x, mask=None, alpha: float = 0.18, gamma: float = 2.2):
    """utils/tonemap.py:4-9 of the reference: scale so the geometric-mean luminance (over lit pixels) maps to ``alpha``,
    clip to [0,1], gamma.  x: [H,W,3] numpy array -> float array in [0,1]."""
    import numpy as np

    x = np.asarray(x)
    L = 0.212671 * x[:, :, 0] + 0.715160 * x[:, :, 1] + 0.072169 * x[:, :, 2]
    lit = L > 5e-5
    mask = np.logical_and(mask, lit) if mask is not None else lit
    assert mask.ndim == 2
    coeff = alpha / np.exp((np.log(L.clip(0) + 1e-7) * mask).sum() / mask.sum())
    return (x * coeff).clip(0, 1) ** (1 / gamma)
