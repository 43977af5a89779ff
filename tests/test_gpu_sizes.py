"""Every size the reference accepts, not only the shipped 128x128 / 128x256 (VERDICT r02 "missing" 4): UNetModel / EncoderUNetModel are
fully convolutional (openaimodel.py:731-768), so any H, W divisible by 2^(levels-1) must run.  Maps that are not a whole number of
the kernels' pixel tiles take masked edge tiles (csrc/conv.hip, csrc/conv_split2.hip RAG instantiations).

* full-width IllNet / RefNet / ObsNet at 64x64, 96x160, 192x192, 32x64, 16x48 against outputs recorded from the reference
  (tests/golden/full_*_sizes.npz, tools/make_golden.py --only full_sizes), fp32 and the fp32-accurate split mode;
* the tiny networks at awkward sizes (4x4 ... 36x68: 1x1, 3x5, 11x5, 9x17 deepest maps) against the CPU oracle.
"""
import pytest
import torch

from conftest import gold, rel_l2
from drmnet_amd import synth
from oracle import unet as ou
from test_gpu_nets import build, full_inputs

pytestmark = pytest.mark.gpu
NET_TOL = 2e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (no fallback)"
    return torch.device("cuda:0")


@pytest.mark.parametrize("name,cfg,kind", [("illnet", ou.ILLNET_CFG, "unet"), ("refnet", ou.REFNET_CFG, "encoder"), ("obsnet", ou.OBSNET_CFG, "unet")])
def test_full_width_nets_at_other_sizes_vs_reference(dev, name, cfg, kind):
    gd = gold(f"full_{name}_sizes")
    m = build(cfg, kind, int(gd["seed"]), dev)
    for key in sorted(k for k in gd if k.startswith("out_")):
        n, h, w = (int(v) for v in key[4:].split("x"))
        xc, t_emb = full_inputs(n, h, w)
        t = torch.from_numpy(gd["t"])[:n].to(dev)
        for precision in ("fp32", "f16x3"):
            m.set_precision(precision)
            out = m(xc.to(dev), t_emb=t_emb.to(dev)) if name == "illnet" else m(xc.to(dev), t)
            e = rel_l2(out.cpu(), gd[key])
            print(f"{name} {n}x{h}x{w} ({precision}): {e:.2e}")
            assert tuple(out.shape) == tuple(gd[key].shape) and e < NET_TOL, (key, precision, e)
    del m
    torch.cuda.empty_cache()


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "f16"])
@pytest.mark.parametrize("n,h,w", [(3, 4, 4), (2, 12, 20), (1, 44, 20), (5, 24, 40), (2, 36, 68)])
def test_tiny_nets_at_awkward_sizes_vs_oracle(dev, precision, n, h, w):
    g = torch.Generator().manual_seed(100 * h + w)
    t = torch.randint(0, 1000, (n,), generator=g)
    tol = 5e-3 if precision == "f16" else NET_TOL
    if (h, w) == (4, 4):
        # a 1x1 deepest map leaves GroupNorm32 two values per group: var + eps is ill-conditioned and the reference arithmetic itself moves
        # by 6e-4 between 1 and 8 host threads (summation order) -- this size only checks that the shape runs and stays close
        tol = 5e-3 if precision != "f16" else None  # (the reduced-precision mode's 1e-3 is amplified ~100x there: finite-ness only)
    # the tiny U-Net has 3 levels (multiples of 4: deepest maps 1x1, 3x5, 11x5, 6x10, 9x17), the tiny encoder 2 (it gets the half sizes)
    for cfg, kind, seed in ((ou.TINY_UNET_CFG, "unet", 21), (ou.TINY_ENC_CFG, "encoder", 22)):
        x = torch.randn((n, 6, h, w) if kind == "unet" else (n, 6, h // 2, w // 2), generator=g)
        m = build(cfg, kind, seed, dev).set_precision(precision)
        P = synth.synth_state_dict(ou.param_manifest(cfg, kind), seed)
        topo = ou.build_topology(cfg, kind)
        ref = ou.unet_forward(P, topo, x, timesteps=t) if kind == "unet" else ou.encoder_forward(P, topo, x, t)
        out = m(x.to(dev), timesteps=t.to(dev)) if kind == "unet" else m(x.to(dev), t.to(dev))
        e = rel_l2(out.cpu(), ref)
        assert torch.isfinite(out).all() and (tol is None or e < tol), (kind, precision, (n, h, w), e)
