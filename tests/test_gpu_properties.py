"""Size-independent properties of the hot path on the GPU: the complete 1000-step ancestral chain (device loop == the same chain driven
step by step through the per-step drop-in), and exact power-of-two homogeneity of the split-precision convs on un-normalised inputs
(what the per-image range guard promises: scaling an image by 2^k changes nothing but the exponent of the result)."""
import pytest
import torch

from drmnet_amd import ops

from test_gpu_samplers import tiny_obsnet  # noqa: E402  (same tiny ObsNet the reference traces were recorded with)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_full_1000_step_ancestral_chain_device_loop_vs_host_driven_steps(dev, precision):
    """ObsNetDiffusion.p_sample_loop over the WHOLE schedule (models/obsnet.py:500-564, T = 1000; BASELINE configs[1]'s loop length) on the
    device against 1000 calls of the per-step drop-in p_sample (ddpm.py:1120-1167) with the same injected noise."""
    m = tiny_obsnet(dev).set_precision(precision)
    g = torch.Generator().manual_seed(11)
    B, T = 2, 1000
    cond = torch.randn((B, 3, 16, 16), generator=g).to(dev)
    x_T = torch.randn((B, 3, 16, 16), generator=g).to(dev)
    noise = torch.randn((T, B, 3, 16, 16), generator=g).to(dev)
    pred_x0, inter = m.p_sample_loop(cond, (B, 3, 16, 16), return_intermediates=True, x_T=x_T, verbose=False, noise=noise)
    img = inter["x_inter"][-1]
    x = x_T.clone()
    x0 = None
    for i, t in enumerate(reversed(range(T))):
        x, x0 = m.p_sample(x, cond, torch.full((B,), t, dtype=torch.long, device=dev), clip_denoised=False, return_x0=True, noise=noise[i])
    assert torch.isfinite(img).all() and torch.isfinite(x).all()
    e_img, e_x0 = rel_l2(img.cpu(), x.cpu()), rel_l2(pred_x0.cpu(), x0.cpu())
    print(f"1000-step chain ({precision}): img {e_img:.2e} pred_x0 {e_x0:.2e}")
    # measured: 0 (identical bits) -- the drop-in step and the device loop run the same kernels in the same order
    assert e_img < 1e-6 and e_x0 < 1e-6


@pytest.mark.parametrize("k", [1, 3])
def test_split_conv_is_exactly_homogeneous_in_powers_of_two(dev, k):
    """f16x3 conv on a raw (un-normalised) input: y(2^j x) == 2^j y(x) bit for bit, also with a different j per image -- the range guard's
    per-image 2^k absorbs the factor exactly, so the staged fp16 hi / lo operands are identical."""
    ops.set_precision("f16x3")
    try:
        g = torch.Generator().manual_seed(5)
        n, cin, cout, h, w = 3, 64, 64, 16, 16
        x = torch.randn((n, cin, h, w), generator=g).to(dev)
        wt = (torch.randn((cout, cin, k, k), generator=g) / (k * cin**0.5)).to(dev)
        y = ops.norm_act_conv(x, wt)
        for js in ((7, 7, 7), (-9, 0, 12), (20, -20, 5)):
            f = torch.tensor([2.0**j for j in js], device=dev).view(n, 1, 1, 1)
            ys = ops.norm_act_conv(x * f, wt)
            assert torch.equal(ys, y * f), js
    finally:
        ops.set_precision("fp32")
