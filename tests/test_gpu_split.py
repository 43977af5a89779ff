"""GPU parity of the split-precision path ("f16x3": fp16 hi/lo x 3 MFMA, fp32 accumulate) against the same oracle and
reference goldens as the fp32 path, at the SAME tolerances: it is an fp32-accurate evaluation, not a reduced-precision one."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import gold, rel_l2
from drmnet_amd import synth
from oracle import unet as ou
from test_gpu_nets import build, full_inputs
from test_gpu_ops import attn_manifest, block_inputs, g, resblock_manifest

pytestmark = pytest.mark.gpu
OP_TOL = 1e-5
NET_TOL = 2e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (no fallback)"
    return torch.device("cuda:0")


@pytest.fixture()
def split_ops():
    from drmnet_amd import ops

    ops.set_precision("f16x3")
    yield ops
    ops.set_precision("fp32")


@pytest.mark.parametrize(
    "n,cin,cout,h,w,k,norm,silu,emb,res,scale",
    [
        (2, 128, 128, 16, 16, 3, True, True, True, False, 1.0),
        (3, 256, 128, 8, 8, 3, True, True, False, True, 1.0),
        (5, 384, 64, 4, 8, 3, True, True, False, False, 1.0),
        (9, 768, 768, 4, 4, 3, True, True, True, True, 1.0),
        (2, 128, 3, 16, 32, 3, True, True, False, False, 1.0),
        (2, 256, 128, 16, 16, 1, False, False, False, False, 1.0),
        (2, 512, 1536, 8, 8, 1, True, False, False, False, 1.0),
        (1, 32, 32, 8, 8, 3, True, True, True, True, 1.0),
        (1, 128, 128, 128, 256, 3, True, True, True, False, 1.0),
        (2, 128, 128, 16, 16, 1, False, False, False, False, 300.0),   # large raw activations (residual stream), no norm
        (2, 128, 128, 16, 16, 3, False, False, False, False, 1e-3),    # tiny raw activations: lo halves in the fp16 subnormal range
        (2, 128, 128, 16, 16, 3, True, True, False, False, 1e4),       # weights scaled far from O(1): exercised by the 2^k pre-scaling
        # range guard of the un-normalised inputs (skip_connection / proj_out / stem: openaimodel.py:241, :314, :534): beyond the
        # fp16 limit 65504 and deep inside fp16's subnormal range -- staged through a per-image power of two, must stay fp32-accurate
        (2, 128, 128, 16, 16, 1, False, False, False, True, 1e5),
        (2, 128, 128, 16, 16, 1, False, False, False, False, 1e-6),
        (3, 256, 128, 8, 16, 3, False, False, True, False, 1e5),
        (2, 6, 128, 16, 32, 3, False, False, False, False, 1e-6),      # the stem's shape (Cin = 6)
        (2, 128, 128, 16, 16, 1, False, False, False, False, 3e7),
    ],
)
def test_split_norm_act_conv(dev, split_ops, n, cin, cout, h, w, k, norm, silu, emb, res, scale):
    gen = g(100 + cin + cout + h)
    x = torch.randn((n, cin, h, w), generator=gen) * 1.5 + 0.3
    wt = torch.randn((cout, cin, k, k), generator=gen) / math.sqrt(cin * k * k)
    if norm:
        wt = wt * scale  # weight-scale case
    else:
        x = x * scale  # activation-scale cases
    b = torch.randn((cout,), generator=gen) * 0.1
    gamma = 1 + 0.1 * torch.randn((cin,), generator=gen) if norm else None
    beta = 0.1 * torch.randn((cin,), generator=gen) if norm else None
    e = torch.randn((n, cout), generator=gen) if emb else None
    r = torch.randn((n, cout, h, w), generator=gen) if res else None
    a = x
    if norm:
        a = F.group_norm(a, 32, gamma, beta, 1e-5)
    if silu:
        a = ou.silu(a)
    ref = F.conv2d(a.double(), wt.double(), b.double(), padding=k // 2)
    if emb:
        ref = ref + e[:, :, None, None]
    if res:
        ref = ref + r
    to = lambda t: None if t is None else t.to(dev)
    out = split_ops.norm_act_conv(to(x), to(wt), to(b), to(gamma), to(beta), silu, to(e), to(r)).cpu()
    err = rel_l2(out, ref)
    print(f"split conv {cin}->{cout} {h}x{w} k{k} scale {scale:g}: rel_l2 {err:.2e}")
    assert err < OP_TOL


@pytest.mark.parametrize("cin,cout,hw", [(256, 128, 16), (128, 128, 16), (1536, 768, 4)])
def test_split_resblock_vs_reference_golden(dev, split_ops, cin, cout, hw):
    gd = gold(f"resblock_{cin}_{cout}_{hw}")
    x, emb = block_inputs(cin, cout, hw, hw, int(gd["n"]))
    P = synth.synth_state_dict(resblock_manifest(cin, cout), int(gd["seed"]))
    out = split_ops.resblock([p.to(dev) for p in P.values()], x.to(dev), emb.to(dev)).cpu()
    err = rel_l2(out, gd["out"])
    print(f"split resblock {cin}->{cout}@{hw}: rel_l2 {err:.2e}")
    assert err < OP_TOL


@pytest.mark.parametrize("ch,h,w", [(512, 16, 16), (768, 4, 8)])
def test_split_attention_block_vs_reference_golden(dev, split_ops, ch, h, w):
    gd = gold(f"attnblock_{ch}_{h}x{w}")
    x, _ = block_inputs(ch, ch, h, w, int(gd["n"]))
    P = synth.synth_state_dict(attn_manifest(ch), int(gd["seed"]))
    out = split_ops.attention_block([p.to(dev) for p in P.values()], x.to(dev)).cpu()
    err = rel_l2(out, gd["out"])
    print(f"split attention {ch}@{h}x{w}: rel_l2 {err:.2e}")
    assert err < OP_TOL


@pytest.mark.parametrize("name,cfg,kind", [("illnet", ou.ILLNET_CFG, "unet"), ("refnet", ou.REFNET_CFG, "encoder"), ("obsnet", ou.OBSNET_CFG, "unet")])
def test_split_full_width_nets_vs_reference_golden(dev, name, cfg, kind):
    m = None
    for n, h, w in ((2, 128, 128), (1, 128, 256)):
        gd = gold(f"full_{name}_{h}x{w}")
        if m is None:
            m = build(cfg, kind, int(gd["seed"]), dev).set_precision("f16x3")
        xc, t_emb = full_inputs(n, h, w)
        t = torch.from_numpy(gd["t"]).to(dev)
        out = m(xc.to(dev), t_emb=t_emb.to(dev)) if name == "illnet" else m(xc.to(dev), t)
        e = rel_l2(out.cpu(), gd["out"])
        print(f"split {name} {n}x{h}x{w}: rel_l2 {e:.2e}")
        assert e < NET_TOL
    # switching back re-packs the weights and reproduces the fp32 path
    m.set_precision("fp32")
    out32 = m(xc.to(dev), t_emb=t_emb.to(dev)) if name == "illnet" else m(xc.to(dev), t)
    assert rel_l2(out32.cpu(), gd["out"]) < NET_TOL
    del m
    torch.cuda.empty_cache()


def test_range_guard_is_per_image_and_reaches_the_blocks(dev, split_ops):
    """One batch holding a 1e5-scale image, an O(1) image and a 1e-6-scale image: every row keeps fp32-level accuracy (the staging
    factor is per image, gn.hip act_pow2_scale_kernel).  Then the two block-level users: a ResBlock whose raw input feeds the
    1x1 skip_connection at 1e4 x the usual magnitude, and an AttentionBlock whose value rows (hence proj_out's input) are large."""
    gen = g(4242)
    x = torch.randn((3, 128, 16, 16), generator=gen) * torch.tensor([1e5, 1.0, 1e-6]).view(3, 1, 1, 1)
    wt = torch.randn((128, 128, 1, 1), generator=gen) / math.sqrt(128)
    b = torch.zeros(128)
    ref = F.conv2d(x.double(), wt.double())
    out = split_ops.norm_act_conv(x.to(dev), wt.to(dev), b.to(dev)).cpu()
    for r in range(3):
        e = rel_l2(out[r], ref[r])
        print(f"per-image guard row {r}: {e:.2e}")
        assert e < OP_TOL
    # ResBlock with a skip_connection on a large raw input (fp64 reference through the oracle's functional form)
    cin, cout, hw = 256, 128, 16
    xb, emb = block_inputs(cin, cout, hw, hw, 2)
    xb = xb * 1e4
    P = synth.synth_state_dict(resblock_manifest(cin, cout), 31)
    D = {k: v.double() for k, v in P.items()}
    sil = lambda t: t * torch.sigmoid(t)
    hh = F.conv2d(sil(F.group_norm(xb.double(), 32, D["in_layers.0.weight"], D["in_layers.0.bias"], 1e-5)), D["in_layers.2.weight"], D["in_layers.2.bias"], padding=1)
    hh = hh + F.linear(sil(emb.double()), D["emb_layers.1.weight"], D["emb_layers.1.bias"])[:, :, None, None]
    hh = F.conv2d(sil(F.group_norm(hh, 32, D["out_layers.0.weight"], D["out_layers.0.bias"], 1e-5)), D["out_layers.3.weight"], D["out_layers.3.bias"], padding=1)
    want = F.conv2d(xb.double(), D["skip_connection.weight"], D["skip_connection.bias"]) + hh
    got = split_ops.resblock([p.to(dev) for p in P.values()], xb.to(dev), emb.to(dev)).cpu()
    e = rel_l2(got, want)
    print(f"resblock, raw input x 1e4: {e:.2e}")
    assert e < OP_TOL
    # AttentionBlock with the value third of qkv blown up 3e4 x (|att| ~ 1e5 > 65504)
    ch, h, w = 512, 16, 16
    xa, _ = block_inputs(ch, ch, h, w, 2)
    Pa = synth.synth_state_dict(attn_manifest(ch), 32)
    Pa["qkv.weight"][2 * ch:] *= 3e4
    Pa["qkv.bias"][2 * ch:] *= 3e4
    Pa["proj_out.weight"] *= 1e-4
    A = {k: v.double() for k, v in Pa.items()}
    xf = xa.double().reshape(2, ch, h * w)
    qkv = F.conv1d(F.group_norm(xf, 32, A["norm.weight"], A["norm.bias"], 1e-5), A["qkv.weight"], A["qkv.bias"])
    q, k_, v_ = qkv.split(ch, dim=1)
    pw = torch.softmax(torch.einsum("bct,bcs->bts", q, k_) / math.sqrt(ch), dim=-1)
    want = (xf + F.conv1d(torch.einsum("bts,bcs->bct", pw, v_), A["proj_out.weight"], A["proj_out.bias"])).reshape(2, ch, h, w)
    got = split_ops.attention_block([p.to(dev) for p in Pa.values()], xa.to(dev)).cpu()
    e = rel_l2(got - xa, want - xa.double())  # the residual x is added exactly: compare the attention branch itself
    print(f"attention, value rows x 3e4: {e:.2e}")
    assert e < 1e-4


F16_NET_TOL = 5e-3  # DRM_PREC_F16 is a reduced-precision mode (one fp16 MFMA per product); its tolerance is its own


@pytest.mark.parametrize("name,cfg,kind", [("illnet", ou.ILLNET_CFG, "unet"), ("refnet", ou.REFNET_CFG, "encoder"), ("obsnet", ou.OBSNET_CFG, "unet")])
def test_plain_f16_mode_full_width_nets(dev, name, cfg, kind):
    """BASELINE configs[2]'s reduced-precision sampling: same kernels with the lo planes and the two cross products left out.
    The stated tolerance is 5e-3 rel-L2 against the fp32 reference goldens (measured 3e-4 .. 1.5e-3)."""
    gd = gold(f"full_{name}_128x128")
    m = build(cfg, kind, int(gd["seed"]), dev).set_precision("f16")
    xc, t_emb = full_inputs(2, 128, 128)
    t = torch.from_numpy(gd["t"]).to(dev)
    out = m(xc.to(dev), t_emb=t_emb.to(dev)) if name == "illnet" else m(xc.to(dev), t)
    e = rel_l2(out.cpu(), gd["out"])
    print(f"plain f16 {name} 2x128x128: rel_l2 {e:.2e}")
    assert 1e-6 < e < F16_NET_TOL  # reduced precision must be visible (guards against silently running the split path)
    del m
    torch.cuda.empty_cache()
