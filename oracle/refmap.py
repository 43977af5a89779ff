"""CPU restatement (numpy) of the object-image -> reflectance-map gather that precedes the samplers.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline.

Follows utils/img2refmap.py:6-37 (refmap_mask_make), utils/transform.py:55-89 (xyz2thetaphi with normal = [0, 1, 0],
tangent = [-1, 0, 0]  =>  theta = acos(y), phi = atan2(z, -x)) and the mask erosion of scripts/estimate.py:43-50.
Pinned against tests/golden/refmap_sample_*.npz, generated from the reference on its own data/sample inputs
(tools/make_golden.py --only refmap).
"""
import numpy as np


def thetaphi_of_normals(normals: np.ndarray) -> np.ndarray:
    """float32 [n, 2]; the trigonometry is evaluated in float64 and rounded once (the reference evaluates it in float32 with
    <= 1 ulp functions: the two agree except where that library is not correctly rounded, which matters only for a
    normal that sits within 1 ulp of a texel border)."""
    n = np.asarray(normals, dtype=np.float32).astype(np.float64)
    theta = np.arccos(np.clip(n[:, 1], -1.0, 1.0))
    theta = np.where(np.abs(n[:, 1]) > 1.0, np.nan, theta)  # acos outside [-1, 1] is NaN in the reference as well
    phi = np.arctan2(n[:, 2], -n[:, 0])
    return np.stack([theta, phi], -1).astype(np.float32)


def refmap_mask_make(colors: np.ndarray, normals: np.ndarray, res: int, angle_threshold: float, min_points: int = 0):
    """Per texel (theta_i, phi_j) of the res x res half-sphere grid: among the pixels whose normal lies within
    angle_threshold in L-inf of (theta, phi), the one whose colour sum is the LOWER median (torch.nanmedian) gives the
    texel its colour; mask = a pixel was found.  Returns (refmap [res, res, C] float32, refmask [res, res] bool)."""
    colors = np.asarray(colors, dtype=np.float32)
    tp = thetaphi_of_normals(normals)
    step = np.float32(np.pi / res)
    centres = (np.arange(res, dtype=np.float32) + np.float32(0.5)) * step  # img2refmap.py:16-17
    thr = np.float32(angle_threshold)
    s = (colors[:, 0] + colors[:, 1]) + colors[:, 2] if colors.shape[1] == 3 else colors.sum(-1, dtype=np.float32)
    refmap = np.zeros((res * res, colors.shape[1]), np.float32)
    refmask = np.zeros(res * res, bool)
    # candidate rows / columns per pixel: |centre - angle| <= thr, evaluated in float32 exactly as the reference's `angles > thr`
    dth = np.abs(centres[:, None] - tp[None, :, 0])  # [res, n]
    dph = np.abs(centres[:, None] - tp[None, :, 1])
    in_t = ~(dth > thr)  # NaN angles compare False with `>` => "inside", as in the reference
    in_p = ~(dph > thr)
    valid = ~np.isnan(s)
    for i in range(res):
        rows = np.nonzero(in_t[i])[0]
        if rows.size == 0:
            continue
        sub = in_p[:, rows]  # [res, m]
        for j in np.nonzero(sub.any(1))[0]:
            idx = rows[sub[j]]
            if idx.size < min_points:
                continue
            idx = idx[valid[idx]]
            if idx.size == 0:
                continue
            order = np.argsort(s[idx], kind="stable")
            pick = idx[order[(idx.size - 1) // 2]]
            refmap[i * res + j] = colors[pick]
            refmask[i * res + j] = True
    return refmap.reshape(res, res, -1), refmask.reshape(res, res)


def erode_mask(mask: np.ndarray, k: int) -> np.ndarray:
    """scripts/estimate.py:43-50: drop mask pixels that have a non-mask pixel inside a disk footprint of diameter k
    (zero 'same' padding: the image border does not erode)."""
    if k <= 0:
        return mask.copy()
    ii = np.arange(k) + 0.5
    ker = np.sqrt((ii[:, None] - k / 2) ** 2 + (ii[None, :] - k / 2) ** 2) <= k / 2
    inv = ~mask
    H, W = mask.shape
    left = (k - 1) // 2
    pad = np.zeros((H + k - 1, W + k - 1), bool)
    pad[left : left + H, left : left + W] = inv
    hit = np.zeros((H, W), bool)
    for a in range(k):
        for b in range(k):
            if ker[a, b]:
                hit |= pad[a : a + H, b : b + W]
    return mask & ~hit
