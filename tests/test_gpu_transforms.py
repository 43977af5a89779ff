"""HIP boundary maps, envmap warp and tone map (csrc/transform.hip behind drm_map_chain / drm_masked_log_range /
drm_luminance_scale / drm_mirmap2envmap / drm_hdr2ldr) against outputs of the reference's own functions
(tests/golden/transforms.npz) and, at other sizes, against the oracle."""
import numpy as np
import pytest
import torch

from conftest import gold, rel_l2
from oracle import transforms as ot

pytestmark = pytest.mark.gpu
OBS_FUNC = "resize_0p1tom1p1_normalizedLogarithmic_lowerbound1e-6"
# log10 / pow / atan2 / acos of the device libm differ from the host's by an ulp or two: elementwise tolerance, not bit equality
RT, AT = 2e-6, 2e-6


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (no fallback)"
    return torch.device("cuda:0")


def T(a, dev):
    return torch.from_numpy(np.asarray(a)).to(dev)


def test_dataset_log_maps_vs_reference(dev):
    from drmnet_amd.dataset import BaseDataset

    g = gold("transforms")
    ds = BaseDataset(16, "log", clamp_before_exp=20)
    assert torch.allclose(ds.transform(T(g["log_x"], dev)).cpu(), torch.from_numpy(g["log_y"]), rtol=RT, atol=AT)
    assert rel_l2(ds.rescale(T(g["log_net"], dev)).cpu(), g["log_rescaled"]) < 1e-6
    assert rel_l2(BaseDataset(16, "log", clamp_before_exp=0.0).rescale(T(g["log_net"], dev).clamp(max=30)).cpu(), g["log_rescaled_noclamp"]) < 1e-6
    # round trip at the metric shape, ragged element count (not a multiple of 4 per image), NaN propagation of the lower bound
    x = torch.exp(torch.randn((5, 3, 128, 256), generator=torch.Generator().manual_seed(1)) * 1.5 - 2).to(dev)
    big = BaseDataset(128, "log", clamp_before_exp=20)
    assert torch.allclose(big.rescale(big.transform(x)), x, rtol=1e-5, atol=1e-6)
    r = torch.rand((3, 1, 7, 9), generator=torch.Generator().manual_seed(2)) + 0.1
    assert torch.allclose(BaseDataset(7, "log").transform(r.to(dev)).cpu(), torch.log10(r + 0.1) + 1, rtol=RT, atol=AT)
    lb = BaseDataset(4, "lowerbound0.5").transform(torch.tensor([[[[float("nan"), 0.2, 0.7, 1.0]] * 4]]).to(dev)).cpu()
    assert torch.isnan(lb[0, 0, 0, 0]) and lb[0, 0, 0, 1:].tolist() == [0.5, 0.699999988079071, 1.0]


def test_dataset_normalized_logarithmic_vs_reference(dev):
    from drmnet_amd.dataset import BaseDataset

    g = gold("transforms")
    ds = BaseDataset(16, OBS_FUNC, clamp_before_exp=20)
    y = ds.transform(T(g["nl_x"], dev), dynamic_normalize=True, mask=T(g["nl_mask"], dev))
    lo, hi = ds.Logarithmic_params
    assert tuple(lo.shape) == tuple(g["nl_lo"].shape)
    assert torch.allclose(lo.cpu(), torch.from_numpy(g["nl_lo"]), rtol=RT, atol=AT) and torch.allclose(hi.cpu(), torch.from_numpy(g["nl_hi"]), rtol=RT, atol=AT)
    assert torch.allclose(y.cpu(), torch.from_numpy(g["nl_y"]), rtol=1e-5, atol=1e-5)
    assert rel_l2(ds.rescale(T(g["nl_net"], dev)).cpu(), g["nl_rescaled"]) < 1e-5
    y3 = ds.transform(T(g["nl_x"], dev)[0], dynamic_normalize=True, mask=T(g["nl_mask"], dev)[0])
    assert tuple(ds.Logarithmic_params[0].shape) == tuple(g["nl3_lo"].shape)
    assert torch.allclose(y3.cpu(), torch.from_numpy(g["nl3_y"]), rtol=1e-5, atol=1e-5)
    with pytest.raises(RuntimeError):
        BaseDataset(16, OBS_FUNC).rescale(T(g["nl_net"], dev))  # no statistics recorded yet


def test_exposure_scale_vs_reference(dev):
    from drmnet_amd import ops

    g = gold("transforms")
    x = T(g["gi_x"], dev)[:3].contiguous()
    s = ops.luminance_scale(x, float(g["gi_scaler"]))
    assert torch.allclose(s.cpu(), torch.from_numpy(g["gi_scale"]), rtol=2e-6)
    y = ops.map_chain(x, [("img_mul", 0.0), ("log_p1", 0.0)], scale=s)
    assert torch.allclose(y.cpu(), torch.from_numpy(g["gi_LrK"]), rtol=1e-5, atol=2e-6)
    big = torch.exp(torch.randn((32, 3, 128, 256), generator=torch.Generator().manual_seed(3))).to(dev)
    assert torch.allclose(ops.luminance_scale(big, 0.12).cpu(), ot.luminance_scale(big.cpu(), 0.12), rtol=1e-5)
    assert torch.allclose(ops.map_chain(ops.map_chain(big, [("img_mul", 0.0)], scale=s.new_full((32,), 3.0)), [("img_div", 0.0), ("clip0", 0.0)],
                                        scale=s.new_full((32,), 3.0)), big, rtol=1e-6)


def test_envmap_warp_and_tonemap_vs_reference(dev):
    from drmnet_amd import ops
    from drmnet_amd.transform import hdr2ldr, mirmap2envmap

    g = gold("transforms")
    mir = T(g["mir"], dev)
    # Stated tolerance 1e-5 rel-L2: the sample position goes through sin / cos / atan2 / acos (device libm vs the host's, an ulp or
    # two apart) and is then scaled by W/2 pixels; on these white-noise test images (O(1) change per pixel) that is a few 1e-6.
    WARP_TOL = 1e-5
    assert rel_l2(mirmap2envmap(mir, (16, 32)).cpu(), g["env"]) < WARP_TOL
    assert rel_l2(mirmap2envmap(mir, (16, 32), log_scale_interpolation=True).cpu(), g["env_log"]) < WARP_TOL
    assert rel_l2(mirmap2envmap(mir[:1].contiguous(), (10, 28)).cpu(), g["env_odd"]) < WARP_TOL
    mir128 = torch.exp(torch.randn((1, 3, 128, 128), generator=torch.Generator().manual_seed(int(g["mir128_seed"]))) * 0.5)
    assert rel_l2(mirmap2envmap(mir128.to(dev), (128, 256)).cpu(), g["env128"]) < WARP_TOL
    assert rel_l2(ops.mirmap2envmap(mir, (16, 32), basis=T(g["basis"], dev), channels_last=True).cpu(), g["r0env"]) < WARP_TOL
    smooth = torch.linspace(0.5, 2.0, 128)[None, None, :, None] * torch.linspace(1.0, 3.0, 128)[None, None, None, :] * torch.ones(2, 3, 1, 1)
    assert rel_l2(mirmap2envmap(smooth.contiguous().to(dev), (128, 256)).cpu(), ot.mirmap2envmap(smooth, (128, 256))) < 1e-6  # a smooth map: 1e-6
    assert np.abs(hdr2ldr(g["ldr_x"]) - g["ldr"]).max() < 2e-6
    assert np.abs(hdr2ldr(g["ldr_x"], g["ldr_mask"]) - g["ldr_masked"]).max() < 2e-6
    assert np.abs(hdr2ldr(g["ldr_x"], alpha=0.3, gamma=1.8) - g["ldr_a"]).max() < 2e-6
    with pytest.raises(NotImplementedError):
        mirmap2envmap(mir, (16, 32), reverse_azimuth=False)


def test_resize_keeps_its_position_in_the_chain(dev):
    """dataset/basedataset.py:29-35 applies the maps right to left INCLUDING resize: "log_resize" = log(resize(x)), "resize_log" =
    resize(log(x)) -- different numbers on an image that actually changes size."""
    from drmnet_amd.dataset import BaseDataset

    x = torch.exp(torch.randn((2, 3, 32, 32), generator=torch.Generator().manual_seed(8))).to(dev)
    rs = lambda t: torch.nn.functional.interpolate(t.reshape(-1, 1, 32, 32), size=(16, 16), mode="bilinear", antialias=True, align_corners=False).reshape(2, 3, 16, 16)
    lg = lambda t: torch.log10(t + 0.1) + 1
    a = BaseDataset(16, "log_resize").transform(x)
    b = BaseDataset(16, "resize_log").transform(x)
    assert tuple(a.shape) == (2, 3, 16, 16)
    assert torch.allclose(a, lg(rs(x)), rtol=1e-5, atol=2e-6) and torch.allclose(b, rs(lg(x)), rtol=1e-5, atol=2e-6)
    assert not torch.allclose(a, b, atol=1e-3)
