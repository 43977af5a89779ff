"""GPU parity of the image -> reflectance-map gather (drm_refmap_mask_make, drm_erode_mask) against goldens recorded
from the reference on its data/sample inputs (bit-exact: colours are copied, masks are boolean) and against the oracle
on seeded synthetic pixels (overlapping thresholds, NaN colours, min_points)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD as GOLDEN_DIR, gold
from drmnet_amd import file_io
from oracle import refmap as orf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (no fallback)"
    return torch.device("cuda:0")


def sample_inputs():
    d = os.path.join(GOLDEN_DIR, "sample")
    img = file_io.load_exr(os.path.join(d, "image.exr"))
    nrm = np.load(os.path.join(d, "normal.npy"))
    m = file_io.load_png(os.path.join(d, "mask.png")) > 0
    return img, nrm, m & (np.linalg.norm(nrm, axis=-1) > 0.5)


def test_erode_mask_vs_reference(dev):
    from drmnet_amd.img2refmap import erode_mask

    g = gold("refmap_sample")
    _, _, mask0 = sample_inputs()
    assert np.array_equal(mask0, g["mask0"])
    out = erode_mask(torch.from_numpy(mask0).to(dev), 5).cpu().numpy()
    assert np.array_equal(out, g["mask_eroded"])
    for k in (1, 3, 4, 7):
        assert np.array_equal(erode_mask(torch.from_numpy(mask0).to(dev), k).cpu().numpy(), orf.erode_mask(mask0, k)), k


@pytest.mark.parametrize("tag", ["128", "16", "32wide"])
def test_refmap_mask_make_vs_reference(dev, tag):
    from drmnet_amd.img2refmap import refmap_mask_make

    g = gold("refmap_sample")
    img, nrm, _ = sample_inputs()
    mask = g["mask_eroded"]
    colors = torch.from_numpy(img[mask]).to(dev)
    normals = torch.from_numpy(nrm[mask]).to(dev)
    refmap, refmask = refmap_mask_make(colors, normals, res=int(g[f"res_{tag}"]), angle_threshold=float(g[f"thr_{tag}"]))
    assert refmask.dtype == torch.bool and tuple(refmap.shape) == g[f"refmap_{tag}"].shape
    assert np.array_equal(refmask.cpu().numpy(), g[f"refmask_{tag}"])
    diff = (refmap.cpu().numpy() != g[f"refmap_{tag}"]).any(-1).sum()
    print(f"refmap {tag}: {int(refmask.sum())} texels set, {int(diff)} differ")
    assert diff == 0  # colours are copies of input pixels: bit-exact


def test_refmap_edge_cases_vs_oracle(dev):
    from drmnet_amd.img2refmap import refmap_mask_make

    rng = np.random.default_rng(5)
    n = 20000
    v = rng.normal(size=(n, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    c = rng.random((n, 3)).astype(np.float32)
    c[::97] = np.nan  # NaN colours are skipped by nanmedian
    c[5::211, 1] = c[5::211, 0]  # ties in some sums
    for res, thr, mp in ((64, np.pi / 64 / 2, 0), (32, np.pi / 20, 0), (16, np.pi / 32, 40), (8, 3.0, 0)):
        ref_map, ref_mask = orf.refmap_mask_make(c, v, res, thr, min_points=mp)
        out_map, out_mask = refmap_mask_make(torch.from_numpy(c).to(dev), torch.from_numpy(v).to(dev), res, thr, min_points=mp)
        assert np.array_equal(out_mask.cpu().numpy(), ref_mask), (res, thr, mp)
        assert np.array_equal(out_map.cpu().numpy(), ref_map, equal_nan=True), (res, thr, mp)
    # empty input
    e_map, e_mask = refmap_mask_make(torch.zeros((0, 3), device=dev), torch.zeros((0, 3), device=dev), 8, 0.1)
    assert not e_mask.any() and (e_map == 0).all()
    with pytest.raises(RuntimeError):
        refmap_mask_make(torch.zeros((4, 3)), torch.zeros((4, 3)), 8, 0.1)
