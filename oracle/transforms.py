"""Oracle: the elementwise maps either side of the samplers, the envmap warp and the tone map, restated (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.  Follows the reference:
  BaseDataset.transform / rescale            dataset/basedataset.py:29-112
  DRMNet.get_input_for_predict (scaling)     models/drmnet.py:1017-1034
  mirmap2envmap / thetaphi2xyz / xyz2thetaphi   utils/transform.py:17-89,106-144
  DRMNet.r0toenvmap                          models/drmnet.py:931-941
  hdr2ldr                                    utils/tonemap.py:4-9
  BaseDataset "resize" / mask resize         dataset/basedataset.py:44-50, models/obsnet.py:691
Pinned by tests/golden/transforms.npz and resize.npz (outputs of the reference's own functions; tools/make_golden.py make_transforms /
make_resize).  The resize itself lives in third-party code the reference calls (torchvision 0.13.1 functional.resize -> torch 1.12.1
interpolate(antialias=True)): its published algorithm (ATen UpSampleKernel.cpp, _compute_indices_min_size_weights_aa) is restated in
`resize` below as explicit per-axis weight matrices.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def _aa_filter(mode: str, x: np.ndarray) -> np.ndarray:
    x = np.abs(x).astype(np.float32)
    if mode == "bilinear":
        return np.where(x < 1, 1 - x, 0).astype(np.float32)
    a = np.float32(-0.5)
    return np.where(x < 1, ((a + 2) * x - (a + 3)) * x * x + 1, np.where(x < 2, (((x - 5) * x + 8) * x - 4) * a, 0)).astype(np.float32)


def resize_matrix(in_len: int, out_len: int, mode: str) -> np.ndarray:
    """[out_len, in_len] fp32 weights of one axis: anti-aliased bilinear / bicubic (align_corners=False), or nearest (legacy floor rule)."""
    W = np.zeros((out_len, in_len), dtype=np.float32)
    scale = np.float32(in_len) / np.float32(out_len)
    if mode == "nearest":
        for i in range(out_len):
            W[i, min(int(np.floor(np.float32(i) * scale)), in_len - 1)] = 1
        return W
    half = np.float32(1.0 if mode == "bilinear" else 2.0)
    support = half * scale if scale >= 1 else half
    inv = np.float32(1) / scale if scale >= 1 else np.float32(1)
    for i in range(out_len):
        centre = scale * (np.float32(i) + np.float32(0.5))
        first = max(int(centre - support + np.float32(0.5)), 0)
        count = min(int(centre + support + np.float32(0.5)), in_len) - first
        w = _aa_filter(mode, (np.arange(count, dtype=np.float32) + np.float32(first) - centre + np.float32(0.5)) * inv)
        total = w.sum(dtype=np.float32)
        W[i, first:first + count] = w / total if total != 0 else w
    return W


def resize(x: torch.Tensor, size, mode: str = "bilinear") -> torch.Tensor:
    """torchvision's resize(x, size, mode, antialias=True) on the last two dims (W filtered first, then H), or nearest."""
    oh, ow = size
    Wh = torch.from_numpy(resize_matrix(x.shape[-2], oh, mode))
    Ww = torch.from_numpy(resize_matrix(x.shape[-1], ow, mode))
    return torch.matmul(Wh, torch.matmul(x.float(), Ww.T))


def transform(x: torch.Tensor, func: str, mask=None, params=None, size=None):
    """BaseDataset.transform with dynamic_normalize=True whenever a mask is given -> (y, (log10min, log10max) or None)."""
    for name in func.split("_")[::-1]:  # f_g = f(g(x)): rightmost first
        if name.startswith("resize"):
            if size is not None and tuple(x.shape[-2:]) != (size, size):
                x = resize(x, (size, size), "bilinear" if name == "resize" else name[6:].replace("-", "_").lower())
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
