"""CPU checks of the image-side plumbing: the refmap oracle against goldens recorded from the reference on its own
data/sample inputs, and the OpenEXR / PNG codecs (drmnet_amd/file_io.py) by round trip and by the sample file's invariants."""
import os

import numpy as np
import torch

from conftest import GOLD, gold
from drmnet_amd import file_io
from oracle import refmap as orf

SAMPLE = os.path.join(GOLD, "sample")


def _inputs():
    img = file_io.load_exr(os.path.join(SAMPLE, "image.exr"))
    nrm = np.load(os.path.join(SAMPLE, "normal.npy"))
    m = file_io.load_png(os.path.join(SAMPLE, "mask.png")) > 0
    return img, nrm, m & (np.linalg.norm(nrm, axis=-1) > 0.5)


def test_exr_reader_on_reference_sample(tmp_path):
    img = file_io.load_exr(os.path.join(SAMPLE, "image.exr"))
    assert img.shape == (256, 256, 3) and img.dtype == np.float32 and np.isfinite(img).all()
    assert 0 < img.min() and img.max() < 10
    t = file_io.load_exr(os.path.join(SAMPLE, "image.exr"), as_torch=True, channel_first=True)
    assert isinstance(t, torch.Tensor) and tuple(t.shape) == (3, 256, 256) and torch.equal(t[0], torch.from_numpy(img[..., 0]))
    # re-encoding the decoded image with the same layout (ZIP, FLOAT, B G R, 16-line blocks) reproduces the file size
    p = tmp_path / "again.exr"
    file_io.save_exr(p, img)
    assert abs(os.path.getsize(p) - os.path.getsize(os.path.join(SAMPLE, "image.exr"))) <= 64
    assert np.array_equal(file_io.load_exr(p), img)


def test_exr_round_trip_uncompressed_and_odd_sizes(tmp_path):
    rng = np.random.default_rng(0)
    for shape, comp in (((5, 7, 3), True), ((33, 18, 3), True), ((17, 4, 3), False)):
        a = rng.normal(size=shape).astype(np.float32) * 100
        a[0, 0] = [np.inf, 0.0, -0.0]
        p = tmp_path / f"t_{shape[0]}_{int(comp)}.exr"
        file_io.save_exr(p, a, compress=comp)
        assert np.array_equal(file_io.load_exr(p), a)
        assert np.array_equal(file_io.load_exr(p, channel_first=True), a.transpose(2, 0, 1))


def test_png_round_trip_and_reference_conventions(tmp_path):
    rng = np.random.default_rng(1)
    ldr = rng.random((9, 11, 3))
    mask = rng.random((9, 11)) > 0.5
    file_io.save_png(tmp_path / "a.png", ldr, mask=mask)
    back = file_io.load_png(tmp_path / "a.png")
    assert back.shape == (9, 11, 4) and np.abs(back[..., :3] - ldr).max() <= 0.5 / 255 + 1e-12
    assert np.array_equal(back[..., 3] > 0, mask)
    m = file_io.load_png(os.path.join(SAMPLE, "mask.png"))
    assert m.shape == (256, 256) and set(np.unique(m)) <= {0.0, 1.0}


def test_refmap_oracle_vs_reference_goldens():
    g = gold("refmap_sample")
    img, nrm, mask0 = _inputs()
    assert np.array_equal(mask0, g["mask0"])
    me = orf.erode_mask(mask0, 5)
    assert np.array_equal(me, g["mask_eroded"])
    for tag in ("128", "16", "32wide"):
        rm, mk = orf.refmap_mask_make(img[me], nrm[me], int(g[f"res_{tag}"]), float(g[f"thr_{tag}"]))
        assert np.array_equal(mk, g[f"refmask_{tag}"]), tag
        assert np.array_equal(rm, g[f"refmap_{tag}"]), tag
