"""Oracle (oracle/transforms.py) against outputs of the reference's own map / warp / tone-map functions
(tests/golden/transforms.npz, tools/make_golden.py make_transforms), plus the host-side compilation of the transform string."""
import numpy as np
import pytest
import torch

from conftest import gold, rel_l2
from oracle import transforms as ot

OBS_FUNC = "resize_0p1tom1p1_normalizedLogarithmic_lowerbound1e-6"
T = lambda a: torch.from_numpy(np.asarray(a))


def test_log_maps():
    g = gold("transforms")
    y, _ = ot.transform(T(g["log_x"]), "log")
    assert torch.allclose(y, T(g["log_y"]), rtol=1e-6, atol=1e-6)
    assert rel_l2(ot.rescale(T(g["log_net"]), "log", 20), g["log_rescaled"]) < 1e-6
    assert rel_l2(ot.rescale(T(g["log_net"]).clamp(max=30), "log", 0.0), g["log_rescaled_noclamp"]) < 1e-6


def test_normalized_logarithmic_maps():
    g = gold("transforms")
    y, (lo, hi) = ot.transform(T(g["nl_x"]), OBS_FUNC, mask=T(g["nl_mask"]))
    assert torch.equal(lo, T(g["nl_lo"])) and torch.equal(hi, T(g["nl_hi"]))
    assert torch.allclose(y, T(g["nl_y"]), rtol=1e-6, atol=1e-6)
    assert rel_l2(ot.rescale(T(g["nl_net"]), OBS_FUNC, 20, (lo, hi)), g["nl_rescaled"]) < 1e-6
    y3, (lo3, hi3) = ot.transform(T(g["nl_x"])[0], OBS_FUNC, mask=T(g["nl_mask"])[0])
    assert torch.equal(lo3, T(g["nl3_lo"])) and torch.allclose(y3, T(g["nl3_y"]), rtol=1e-6, atol=1e-6)


def test_exposure_scale_and_input_map():
    g = gold("transforms")
    x = T(g["gi_x"])[:3]
    s = ot.luminance_scale(x, float(g["gi_scaler"]))
    assert torch.allclose(s, T(g["gi_scale"]), rtol=1e-6)
    y, _ = ot.transform(x * s[:, None, None, None], "log")
    assert torch.allclose(y, T(g["gi_LrK"]), rtol=1e-5, atol=1e-6)


def test_envmap_warp_and_tonemap():
    g = gold("transforms")
    mir = T(g["mir"])
    assert rel_l2(ot.mirmap2envmap(mir, (16, 32)), g["env"]) < 1e-6
    assert rel_l2(ot.mirmap2envmap(mir, (16, 32), log_scale_interpolation=True), g["env_log"]) < 1e-6
    assert rel_l2(ot.mirmap2envmap(mir[:1], (10, 28)), g["env_odd"]) < 1e-6
    mir128 = torch.exp(torch.randn((1, 3, 128, 128), generator=torch.Generator().manual_seed(int(g["mir128_seed"]))) * 0.5)
    assert rel_l2(ot.mirmap2envmap(mir128, (128, 256)), g["env128"]) < 1e-6
    assert rel_l2(ot.mirmap2envmap(mir, (16, 32), basis=T(g["basis"]), channels_last=True), g["r0env"]) < 1e-6
    assert np.abs(ot.hdr2ldr(g["ldr_x"]) - g["ldr"]).max() < 1e-6
    assert np.abs(ot.hdr2ldr(g["ldr_x"], g["ldr_mask"]) - g["ldr_masked"]).max() < 1e-6
    assert np.abs(ot.hdr2ldr(g["ldr_x"], alpha=0.3, gamma=1.8) - g["ldr_a"]).max() < 1e-6


def test_transform_string_compiles_to_map_chains():
    """Host logic of drmnet_amd.dataset.BaseDataset: right-to-left forward chain, left-to-right inverse chain, clamp handling."""
    import math

    from drmnet_amd.dataset import BaseDataset

    ds = BaseDataset(16, OBS_FUNC, clamp_before_exp=20)
    # forward program = elementwise segments cut at each resize, which keeps its place in the chain (basedataset.py:29-35)
    assert ds._forward == [[("lowerbound", 1e-6), ("norm_log", 0.0), ("unit_to_signed", 0.0)], "bilinear", []]
    assert ds._inverse == [("signed_to_unit", 0.0), ("denorm_log", 0.0), ("exp10", 20.0)]
    assert BaseDataset(16, "log_resizeNEAREST")._forward == [[], "nearest", [("log_p1", 0.0)]]
    d2 = BaseDataset(128, "log", clamp_before_exp=0.0)
    assert d2._forward == [[("log_p1", 0.0)]] and d2._inverse == [("exp_m1", math.inf)]
    assert BaseDataset(128, "log", clamp_before_exp=False).clamp_before_exp == 10  # basedataset.py:25
    with pytest.raises(NotImplementedError):
        BaseDataset(16, "gamma2p2")
    with pytest.raises(RuntimeError):  # no CPU path
        d2.transform(torch.ones(1, 3, 128, 128))
    with pytest.raises(RuntimeError):
        ds.rescale(torch.ones(1, 3, 16, 16).cuda() if torch.cuda.is_available() else torch.ones(1, 3, 16, 16))
