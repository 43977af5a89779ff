"""DRM_PREC_BF16: BASELINE configs[2] as written ("DRMNet DDIM 50-step, batch=256, bf16"): bf16 operands on v_mfma_f32_32x32x16_bf16, fp32 accumulate, on
the same kernels as the fp16 mode.  Reduced precision (8 significant bits): held to 3e-2 per network against the reference and compared with the f16
mode on the same inputs (bf16 must sit within an order of magnitude above it: three mantissa bits fewer -- a mis-packed operand or a wrong MFMA type
shows as an O(1) error)."""
import pytest
import torch

from conftest import NET_TOL, gold, rel_l2
from oracle import unet as ou
from test_gpu_nets import build, full_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (no fallback)"
    return torch.device("cuda:0")


@pytest.mark.parametrize("name,cfg,kind", [("illnet", ou.ILLNET_CFG, "unet"), ("refnet", ou.REFNET_CFG, "encoder"), ("obsnet", ou.OBSNET_CFG, "unet")])
def test_full_width_nets_in_bf16(dev, name, cfg, kind):
    errs = {}
    m = None
    for n, h, w in ((2, 128, 128), (1, 128, 256)):
        gd = gold(f"full_{name}_{h}x{w}")
        if m is None:
            m = build(cfg, kind, int(gd["seed"]), dev)
        xc, t_emb = full_inputs(n, h, w)
        t = torch.from_numpy(gd["t"]).to(dev)
        for precision in ("f16", "bf16"):
            m.set_precision(precision)
            out = m(xc.to(dev), t_emb=t_emb.to(dev)) if name == "illnet" else m(xc.to(dev), t)
            errs[(precision, h, w)] = rel_l2(out.cpu(), gd["out"])
        print(f"{name} {n}x{h}x{w}: f16 {errs[('f16', h, w)]:.2e}, bf16 {errs[('bf16', h, w)]:.2e}")
        assert errs[("bf16", h, w)] < NET_TOL["bf16"] and errs[("f16", h, w)] < NET_TOL["f16"]
        assert errs[("bf16", h, w)] < 30 * errs[("f16", h, w)]
    # sizes that are not a whole number of tiles (ragged instantiations) and the small maps
    gs = gold(f"full_{name}_sizes")
    m.set_precision("bf16")
    for key in sorted(k for k in gs if k.startswith("out_")):
        n, h, w = (int(v) for v in key[4:].split("x"))
        xc, t_emb = full_inputs(n, h, w)
        t = torch.from_numpy(gs["t"])[:n].to(dev)
        out = m(xc.to(dev), t_emb=t_emb.to(dev)) if name == "illnet" else m(xc.to(dev), t)
        e = rel_l2(out.cpu(), gs[key])
        print(f"{name} {key[4:]} (bf16): {e:.2e}")
        assert e < NET_TOL["bf16"]
    del m
    torch.cuda.empty_cache()


def test_ddim_chain_in_bf16_vs_reference_trace(dev):
    """the 50-step DDIM chain (tiny net, recorded from the reference) in bf16: the sampler damps per-step rounding noise"""
    from drmnet_amd.ddim import DDIMSampler
    from test_gpu_samplers import tiny_obsnet

    g = gold("ddim_trace_eta1")
    m = tiny_obsnet(dev).set_precision("bf16")
    cond, x_T, noise = (torch.from_numpy(g[k]).to(dev) for k in ("cond", "x_T", "noise"))
    x, _ = DDIMSampler(m).sample(50, cond.shape[0], (3, 16, 16), cond, eta=1.0, x_T=x_T, verbose=False, noise=noise)
    e = rel_l2(x.cpu(), g["x"])
    print(f"ddim 50 steps (bf16): {e:.2e}")
    assert e < NET_TOL["bf16"]
