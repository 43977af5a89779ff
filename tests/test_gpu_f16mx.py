"""The f16mx arithmetic mode (DRM_PREC_F16MX): the GroupNorm-fed 3x3 convs of the res blocks evaluate a*b as fp16 hi*hi plus BOTH cross terms in one
block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, e4m3 operands, power-of-two block factors); everything else runs as in f16x3.  The cross
terms are 2^-11 of a product and carry an e4m3 rounding, so the mode sits between f16x3 (~1e-6) and f16 (~1e-3): these tests hold it to the hot
path's contract -- 1e-4 rel-L2 against the reference (BASELINE.json north_star) -- on every network, on the sampler loops and on the whole chain,
against the same recorded reference outputs the other modes are tested on, and to 2e-5 on a single res block against exact fp32.
"""
import numpy as np
import pytest
import torch

from conftest import NET_TOL, gold, rel_l2
from drmnet_amd import ops, synth
from oracle import unet as ou
from test_gpu_configs34 import chain_draws, full_chain_models, full_drmnet, sample_object, shape_heads
from test_gpu_nets import build, full_inputs

pytestmark = pytest.mark.gpu
CONTRACT = 1e-4  # north-star tolerance (rel-L2 against the reference)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (no fallback)"
    return torch.device("cuda:0")


def res_manifest(cin, cout):
    m = [("in_layers.0.weight", (cin,)), ("in_layers.0.bias", (cin,)), ("in_layers.2.weight", (cout, cin, 3, 3)), ("in_layers.2.bias", (cout,)),
         ("emb_layers.1.weight", (cout, 512)), ("emb_layers.1.bias", (cout,)), ("out_layers.0.weight", (cout,)), ("out_layers.0.bias", (cout,)),
         ("out_layers.3.weight", (cout, cout, 3, 3)), ("out_layers.3.bias", (cout,))]
    if cin != cout:
        m += [("skip_connection.weight", (cout, cin, 1, 1)), ("skip_connection.bias", (cout,))]
    return m


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 128, 128, 128, 256), (2, 256, 128, 64, 128), (4, 384, 384, 32, 64), (8, 640, 640, 8, 16), (2, 768, 768, 4, 8),
                                            (3, 256, 384, 12, 20), (1, 512, 512, 16, 16), (2, 128, 128, 24, 40), (1, 64, 96, 7, 9)])
def test_res_block_vs_exact_fp32(dev, n, cin, cout, h, w):
    """every tile family of the 3x3 kernel (wide, 192-wide, narrow, split-K, ragged) in the f16mx form"""
    g = torch.Generator().manual_seed(h * 1000 + w + cin)
    x = torch.randn((n, cin, h, w), generator=g).to(dev)
    emb = torch.randn((n, 512), generator=g).to(dev)
    P = [p.to(dev) for p in synth.synth_state_dict(res_manifest(cin, cout), 3).values()]
    try:
        ops.set_precision("fp32")
        ref = ops.resblock(P, x, emb).clone()
        ops.set_precision("f16mx")
        out = ops.resblock(P, x, emb)
        e = rel_l2(out.cpu(), ref.cpu())
        again = ops.resblock(P, x, emb)
    finally:
        ops.set_precision("fp32")
    print(f"res block N={n} {cin}->{cout} @{h}x{w} f16mx vs fp32: {e:.2e}")
    assert torch.isfinite(out).all() and e < 2e-5
    assert rel_l2(again.cpu(), out.cpu()) < 2e-6  # (run to run: only the fp64 statistics atomics reorder)


def test_activations_beyond_the_fp8_range_stay_finite(dev):
    """e4m3 has no infinity and v_cvt_scalef32_pk_fp8_f32 makes NaN above 448: the staging clamps at +-3584 first.  A GroupNorm weight of 500
    drives the activations far past that -- the output must stay finite and close to fp32 where fp32 itself is meaningful."""
    n, c, h, w = 2, 128, 16, 16
    g = torch.Generator().manual_seed(5)
    x = torch.randn((n, c, h, w), generator=g).to(dev)
    emb = torch.randn((n, 512), generator=g).to(dev)
    P = [p.to(dev) for p in synth.synth_state_dict(res_manifest(c, c), 3).values()]
    P[0] = P[0] * 0 + 500.0  # in_layers GroupNorm weight: |GN output| up to ~2000, SiLU keeps the positive half
    try:
        ops.set_precision("fp32")
        ref = ops.resblock(P, x, emb).clone()
        ops.set_precision("f16mx")
        out = ops.resblock(P, x, emb)
    finally:
        ops.set_precision("fp32")
    assert torch.isfinite(out).all()
    assert rel_l2(out.cpu(), ref.cpu()) < 2e-5  # (values below the 3584 clamp: the clamp is not reached at 500 x ~4 sigma)
    P[0] = P[0] * 0 + 5000.0  # now the clamp IS reached: finite, and within the clamp's reach of fp32 (a saturating input, not a parity case)
    try:
        ops.set_precision("f16mx")
        out = ops.resblock(P, x, emb)
    finally:
        ops.set_precision("fp32")
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("name,cfg,kind", [("illnet", ou.ILLNET_CFG, "unet"), ("refnet", ou.REFNET_CFG, "encoder"), ("obsnet", ou.OBSNET_CFG, "unet")])
def test_full_width_nets_vs_reference(dev, name, cfg, kind):
    gd = gold(f"full_{name}_sizes")
    m = build(cfg, kind, int(gd["seed"]), dev).set_precision("f16mx")
    worst = 0.0
    for key in sorted(k for k in gd if k.startswith("out_")):
        n, h, w = (int(v) for v in key[4:].split("x"))
        xc, t_emb = full_inputs(n, h, w)
        t = torch.from_numpy(gd["t"])[:n].to(dev)
        out = m(xc.to(dev), t_emb=t_emb.to(dev)) if name == "illnet" else m(xc.to(dev), t)
        e = rel_l2(out.cpu(), gd[key])
        worst = max(worst, e)
        print(f"{name} {n}x{h}x{w} (f16mx): {e:.2e}")
        assert tuple(out.shape) == tuple(gd[key].shape) and e < CONTRACT, (key, e)
    for n, h, w in ((2, 128, 128), (1, 128, 256)):  # the shipped shapes (config shape; BASELINE configs[1]'s metric shape), all three networks
        g2 = gold(f"full_{name}_{h}x{w}")
        xc, t_emb = full_inputs(n, h, w)
        out = m(xc.to(dev), t_emb=t_emb.to(dev)) if name == "illnet" else m(xc.to(dev), torch.from_numpy(g2["t"]).to(dev))
        e = rel_l2(out.cpu(), g2["out"])
        worst = max(worst, e)
        print(f"{name} {tuple(out.shape)} (f16mx): {e:.2e}")
        assert e < CONTRACT
    assert worst < NET_TOL["f16mx"], worst  # 5e-5: half the contract on every whole network
    del m
    torch.cuda.empty_cache()


def test_full_width_p_sample_loop_vs_reference_trace(dev):
    g = gold("drmnet_loop_full")
    T, B = int(g["max_timesteps"]), int(g["B"])
    m = shape_heads(full_drmnet(dev, "f16mx", max_timesteps=T, epsilon=float(g["epsilon"]), gamma=float(g["gamma"]), delta=float(g["delta"])), g)
    LrK = synth.synth_refmaps(B, 128, 128, int(g["input_seed"]))
    gen = torch.Generator().manual_seed(int(g["gen_seed"]))
    n0 = torch.randn(LrK.shape, generator=gen)
    sn = torch.randn((T,) + tuple(LrK.shape), generator=gen)
    LrK, n0, sn = LrK.to(dev), n0.to(dev), sn.to(dev)
    Lr0, zK, K = m.p_sample_loop(LrK, [LrK], [LrK], verbose=False, noise0=n0, step_noise=sn)
    e = rel_l2(Lr0.cpu(), g["Lr0"])
    print(f"full-width DRMNet loop (f16mx): K = {K.tolist()} (reference {g['K'].tolist()}), Lr0 rel-L2 {e:.2e}")
    assert K.tolist() == g["K"].tolist()  # the per-sample early exits fall on the reference's steps
    assert e < CONTRACT
    assert np.allclose(zK.cpu().numpy(), g["zK"], atol=1e-4, equal_nan=True)
    del m
    torch.cuda.empty_cache()


def test_full_width_estimate_chain(dev):
    """scripts/estimate.py end to end (50 DDIM steps of ObsNet + K DRMNet steps) in f16mx against the chain recorded from the reference"""
    from drmnet_amd.estimate import estimate

    g = gold("estimate_chain_full")
    drm, obs = full_chain_models(g, dev, "f16mx")
    img, nrm, mask = sample_object(dev)
    x_T, noise, noise0, step_noise = (t.to(dev) for t in chain_draws(g))
    stages = {}
    hooks = {"cond_noise": torch.from_numpy(g["cond"]).to(dev), "x_T": x_T, "noise": noise, "noise0": noise0, "step_noise": step_noise, "stages": stages}
    Lr0, zK = estimate(drm, obs, img, nrm, mask, hooks=hooks)
    env = drm.r0toenvmap(Lr0[None], (drm.image_size, drm.image_size * 2))[0]
    e = {k: rel_l2(stages[k].cpu(), g[k]) for k in ("cond", "inpaint", "LrK")}
    e["Lr0"] = rel_l2(Lr0.cpu(), g["Lr0"])
    e["envmap"] = rel_l2(env.cpu(), g["envmap"])
    print("full-width estimate chain (f16mx):", {k: f"{v:.2e}" for k, v in e.items()}, "steps", drm.last_steps, "zK", zK.tolist())
    assert drm.last_steps == int(g["K"][0])
    assert max(e["inpaint"], e["LrK"], e["Lr0"], e["envmap"]) < CONTRACT
    del drm, obs
    torch.cuda.empty_cache()


def test_switching_between_the_split_modes_repacks_the_weights(dev):
    """f16x3 and f16mx keep different weight images (lo planes: fp16 vs e4m3): a handle that changes mode must re-pack"""
    gd = gold("full_illnet_128x128")
    m = build(ou.ILLNET_CFG, "unet", int(gd["seed"]), dev)
    xc, t_emb = full_inputs(2, 128, 128)
    x1, t1 = xc[:1].contiguous().to(dev), t_emb[:1].contiguous().to(dev)
    for precision, tol in (("f16mx", CONTRACT), ("f16x3", 2e-5), ("f16mx", CONTRACT), ("fp32", 2e-5)):
        out = m.set_precision(precision)(x1, t_emb=t1)
        e = rel_l2(out[0].cpu(), gd["out"][0])
        print(f"{precision}: {e:.2e}")
        assert e < tol, (precision, e)
    del m
    torch.cuda.empty_cache()
