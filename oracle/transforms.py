"""Oracle: the elementwise maps either side of the samplers, the envmap warp and the tone map, restated (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.  Follows the reference:
  BaseDataset.transform / rescale            dataset/basedataset.py:29-112
  DRMNet.get_input_for_predict (scaling)     models/drmnet.py:1017-1034
  mirmap2envmap / thetaphi2xyz / xyz2thetaphi   utils/transform.py:17-89,106-144
  DRMNet.r0toenvmap                          models/drmnet.py:931-941
  hdr2ldr                                    utils/tonemap.py:4-9
Pinned by tests/golden/transforms.npz (outputs of the reference's own functions; tools/make_golden.py make_transforms).
"""
from __future__ import annotations

import math

import numpy as np
import torch


def transform(x: torch.Tensor, func: str, mask=None, params=None):
    """BaseDataset.transform with dynamic_normalize=True whenever a mask is given -> (y, (log10min, log10max) or None)."""
    for name in func.split("_")[::-1]:  # f_g = f(g(x)): rightmost first
        if name.startswith("resize"):
            continue  # no-op at the stored size (the only case on the shipped path)
        if name == "log":
            x = torch.log10(x + 0.1) + 1
        elif name == "log10":
            x = torch.log10(x)
        elif name.startswith("lowerbound"):
            x = x.clip(float(name[10:]))
        elif name == "0p1tom1p1":
            x = x * 2 - 1
        elif name == "normalizedLogarithmic":
            if mask is not None:
                dims = (-1, -2, -3)
                top = (x * mask).amax(dim=dims, keepdim=True)
                params = (torch.log10((x * mask + (1 - mask.float()) * top).amin(dim=dims, keepdim=True)), torch.log10(top))
            x = (torch.log10(x) - params[0]) / (params[1] - params[0])
        else:
            raise NotImplementedError(name)
    return x, params


def rescale(x: torch.Tensor, func: str, clamp_before_exp=0.0, params=None):
    def p10(v):
        return torch.pow(10, v.clamp(max=clamp_before_exp) if clamp_before_exp else v)

    for name in func.split("_"):  # inverses, leftmost first
        if name.startswith("resize") or name.startswith("lowerbound"):
            continue
        if name == "log":
            x = p10(x - 1) - 0.1
        elif name == "log10":
            x = p10(x)
        elif name == "0p1tom1p1":
            x = (x + 1) / 2
        elif name == "normalizedLogarithmic":
            x = p10(x * (params[1] - params[0]) + params[0])
        else:
            raise NotImplementedError(name)
    return x


def luminance_scale(x: torch.Tensor, scaler: float) -> torch.Tensor:
    """models/drmnet.py:1020-1026 -> [B]."""
    L = 0.212671 * x[:, 0] + 0.715160 * x[:, 1] + 0.072169 * x[:, 2]
    lit = L > 0
    return scaler / torch.exp((torch.log(L.clip(1e-5)) * lit).sum(dim=(1, 2)) / lit.sum(dim=(1, 2)))


def envmap_grid(oh: int, ow: int) -> torch.Tensor:
    """Sample positions (u, v) in [-1, 1]^2 of the mirror map for every envmap texel: direction d(theta, phi) with zenith +y, left
    edge -z, azimuth reversed; half vector n = normalize(d + view), view = +z; (theta_n about +y, phi_n from +z towards +x)."""
    theta = (torch.arange(oh, dtype=torch.float32) + 0.5) * (math.pi / oh)
    phi = -(torch.arange(ow, dtype=torch.float32) + 0.5) * (2 * math.pi / ow)
    th, ph = torch.meshgrid(theta, phi, indexing="ij")
    st = torch.sin(th)
    d = torch.stack([-st * torch.sin(ph), torch.cos(th), -st * torch.cos(ph)], dim=-1)
    n = torch.nn.functional.normalize(d + torch.tensor([0.0, 0.0, 1.0]), dim=-1, eps=1e-12)
    return torch.stack([torch.arctan2(n[..., 0], n[..., 2]) * (2 / math.pi), torch.arccos(n[..., 1]) * (2 / math.pi) - 1], dim=-1)


def mirmap2envmap(mir: torch.Tensor, shape, log_scale_interpolation=False, basis=None, channels_last=False) -> torch.Tensor:
    if basis is not None:
        mir = mir / basis
    src = torch.log(mir.clip(1e-7)) if log_scale_interpolation else mir
    grid = envmap_grid(int(shape[0]), int(shape[1]))[None].expand(mir.shape[0], -1, -1, -1)
    env = torch.nn.functional.grid_sample(src, grid, mode="bilinear", padding_mode="border", align_corners=False)
    env = torch.exp(env) if log_scale_interpolation else env
    return env.permute(0, 2, 3, 1) if channels_last else env


def hdr2ldr(x: np.ndarray, mask=None, alpha=0.18, gamma=2.2) -> np.ndarray:
    L = 0.212671 * x[:, :, 0] + 0.715160 * x[:, :, 1] + 0.072169 * x[:, :, 2]
    lit = L > 5e-5 if mask is None else np.logical_and(mask, L > 5e-5)
    coeff = alpha / np.exp((np.log(L.clip(0) + 1e-7) * lit).sum() / lit.sum())
    return (x * coeff).clip(0, 1) ** (1 / gamma)
