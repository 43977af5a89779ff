"""GPU checks of BASELINE configs[3] and configs[4] at full width, and a full-width reverse loop against the reference
(VERDICT r02 "untested configs" / "full-width loop parity"):

* configs[3] per-GPU shard (2048 refmaps over 8 GPUs = 256 per GPU @3x128x256): IllNet and RefNet forwards at B = 256 against the
  reference goldens (probed rows), and the whole DRMNet reverse step at B = 256 -- rows against the CPU oracle's step on the
  same inputs and against each other;
* the reference's own p_sample_loop (models/drmnet.py:782-847) on the SHIPPED networks (configs/drmnet/eval_drmnet.yaml,
  128x128, B = 3; rows converging after 2 and 3 steps and one that never does): tests/golden/drmnet_loop_full.npz recorded by
  tools/make_golden.py --only drmnet_loop_full; device loop and host-driven loop (intermediates) against the trace;
* configs[4] (scripts/estimate.py:29-102 with the shipped full-width networks): the single-image chain against
  tests/golden/estimate_chain_full.npz (recorded from the reference's statements on data/sample), and estimate_batch at B = 8
  with early exit on: row 0 is the golden object, the other rows start from different x_T and must equal their own single runs.
"""
import os

import numpy as np
import pytest
import torch

from conftest import CONTRACT, GOLD, NET_TOL as MODE_TOL, gold, rel_l2
from drmnet_amd import synth
from oracle import samplers as osamp
from oracle import unet as ou
from test_gpu_nets import build, full_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(GOLD))
NET_TOL = 2e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (no fallback)"
    return torch.device("cuda:0")


def full_drmnet(dev, precision="f16x3", **over):
    """configs/drmnet/eval_drmnet.yaml as shipped (IllNet 237.8 M + RefNet 33.4 M parameters by the synth rule)."""
    from drmnet_amd.config import instantiate_from_config, load_config

    dcfg = load_config(os.path.join(ROOT, "configs/drmnet/eval_drmnet.yaml"))
    mp = dcfg["model"]["params"]
    mp.pop("ckpt_path", None)
    mp.update(use_ema=False, **over)
    m = instantiate_from_config(dcfg["model"])
    synth.load_synth(m.illnet_model.diffusion_model, synth.SEED_ILLNET)
    synth.load_synth(m.refnet_model.diffusion_model, synth.SEED_REFNET)
    m.illnet_model.z_emb_layer.load_state_dict(synth.synth_state_dict(
        [(k, tuple(v.shape)) for k, v in m.illnet_model.z_emb_layer.state_dict().items()], synth.SEED_ZEMB))
    m.ds = instantiate_from_config(dcfg["data"]["params"]["predict"])
    return m.to(dev).set_precision(precision)


def shape_heads(m, g):
    """The head edits tools/make_golden.py applied to the reference networks (values stored in the fixture)."""
    isd = m.illnet_model.diffusion_model.state_dict()
    isd["out.2.weight"] = isd["out.2.weight"] * float(g["ill_out_scale"])
    isd["out.2.bias"] = isd["out.2.bias"] * float(g["ill_out_scale"])
    m.illnet_model.diffusion_model.load_state_dict(isd)
    sd = m.refnet_model.diffusion_model.state_dict()
    sd["out.3.weight"] = sd["out.3.weight"] * float(g["head_w_scale"])
    sd["out.3.bias"] = torch.from_numpy(g["head_bias"]).to(sd["out.3.bias"])
    m.refnet_model.diffusion_model.load_state_dict(sd)
    return m


# --------------------------------------------------------------------------------------------- configs[3]: 256 refmaps per GPU

_STEP_ORACLE = {}


@pytest.mark.parametrize("B,precision", [(256, "f16x3"), (256, "f16mx"), (128, "f16mx")])
def test_illnet_and_refnet_batch256_vs_reference_golden(dev, B, precision):
    """B = 256: the per-GPU shard of configs[3]; B = 128: the north-star's batch 1024 over 8 GPUs"""
    xc, t_emb = full_inputs(1, 128, 256)
    xb = xc.repeat(B, 1, 1, 1).to(dev)
    xb[1::2] = xb[1::2].flip(-1)  # odd rows see a different (valid) input: a row mix-up cannot cancel out
    for name, cfg, kind in (("illnet", ou.ILLNET_CFG, "unet"), ("refnet", ou.REFNET_CFG, "encoder")):
        gd = gold(f"full_{name}_128x256")
        m = build(cfg, kind, int(gd["seed"]), dev).set_precision(precision)
        if name == "illnet":
            out = m(xb, t_emb=t_emb.repeat(B, 1).to(dev))
        else:
            out = m(xb, torch.from_numpy(gd["t"]).to(dev).repeat(B))
        assert out.shape[0] == B and torch.isfinite(out).all()
        for r in (0, 2, B // 2, B - 2):
            e = rel_l2(out[r].cpu(), gd["out"][0])
            print(f"{name} B={B} ({precision}) row {r}: {e:.2e}")
            assert e < MODE_TOL[precision], (name, r, e)
        assert rel_l2(out[1].cpu(), out[B - 1].cpu()) < 1e-6
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(5)).to(dev)
        if name == "illnet":
            outp = m(xb[perm], t_emb=t_emb.repeat(B, 1).to(dev))
        else:
            outp = m(xb[perm], torch.from_numpy(gd["t"]).to(dev).repeat(B))
        assert rel_l2(outp.cpu(), out[perm].cpu()) < 1e-6
        del m, out, outp
        torch.cuda.empty_cache()


@pytest.mark.parametrize("B,precision", [(256, "f16x3"), (256, "f16mx"), (128, "f16mx")])
def test_drmnet_step_batch256_vs_oracle_rows(dev, B, precision):
    """One whole reverse step (RefNet -> BRDF schedule -> z-MLP -> IllNet -> update, models/drmnet.py:796-839) of 256 refmaps
    @3x128x256: rows 0 / 1 / 254 / 255 against the CPU oracle's step on the same two distinct inputs, K and zk included."""
    m = full_drmnet(dev, precision, max_timesteps=2, epsilon=0.01, gamma=0.9)
    H, W = 128, 256
    base = synth.synth_refmaps(2, H, W, 77)
    LrK = base.repeat(B // 2, 1, 1, 1).to(dev)  # even rows = base[0], odd rows = base[1]
    g = torch.Generator().manual_seed(9)
    n0 = torch.randn((2, 3, H, W), generator=g)
    sn = torch.randn((2, 2, 3, H, W), generator=g)
    Lr0, zK, K, inter = m.p_sample_loop(LrK, [LrK], [LrK], return_intermediates=True, verbose=False, log_every_k=1,
                                        noise0=n0.repeat(B // 2, 1, 1, 1).to(dev), step_noise=sn.repeat(1, B // 2, 1, 1, 1).to(dev))
    dev_Lr0, zK_dev, K_dev = m.p_sample_loop(LrK, [LrK], [LrK], verbose=False, noise0=n0.repeat(B // 2, 1, 1, 1).to(dev),
                                             step_noise=sn.repeat(1, B // 2, 1, 1, 1).to(dev))
    assert torch.isfinite(Lr0).all() and K.tolist() == K_dev.tolist()
    assert rel_l2(dev_Lr0.cpu(), Lr0.cpu()) < 1e-5
    # the oracle on the two distinct inputs (CPU, fp32)
    Pu = synth.synth_state_dict(ou.param_manifest(ou.ILLNET_CFG, "unet"), synth.SEED_ILLNET)
    Pe = synth.synth_state_dict(ou.param_manifest(ou.REFNET_CFG, "encoder"), synth.SEED_REFNET)
    Pz = synth.synth_state_dict(ou.zemb_manifest(6, 128), synth.SEED_ZEMB)
    tu, te = ou.build_topology(ou.ILLNET_CFG, "unet"), ou.build_topology(ou.REFNET_CFG, "encoder")
    if "ref" not in _STEP_ORACLE:  # (the same two inputs for every (B, precision): the CPU oracle runs once per session)
        _STEP_ORACLE["ref"] = osamp.drmnet_sample(lambda xc, t: ou.encoder_forward(Pe, te, xc, t),
                                                  lambda xc, dz: ou.unet_forward(Pu, tu, xc, t_emb=ou.z_embed(Pz, dz)),
                                                  base, n0, sn, torch.tensor(m._z0.tolist()), 0.9, 0.01, float(m.delta), 2)
    ref = _STEP_ORACLE["ref"]
    assert K[:2].tolist() == ref[2].tolist()
    for r in (0, 1, B - 2, B - 1):
        e = rel_l2(Lr0[r].cpu(), ref[0][r % 2])
        print(f"DRMNet 2 steps, B = {B} @128x256 ({precision}), row {r}: rel-L2 vs oracle {e:.2e}")
        assert e < 1e-4 and K[r] == ref[2][r % 2]
    assert rel_l2(Lr0[0].cpu(), Lr0[B - 2].cpu()) < 1e-6 and rel_l2(Lr0[1].cpu(), Lr0[B - 1].cpu()) < 1e-6
    zk_last = inter["zk_inter"][-1]
    assert torch.allclose(zk_last[0], zk_last[B - 2], atol=1e-6)
    del m
    torch.cuda.empty_cache()


# --------------------------------------------------------------------------------------------- full-width loop vs the reference


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "f16mx"])
def test_full_width_p_sample_loop_vs_reference_trace(dev, precision):
    g = gold("drmnet_loop_full")
    T, B = int(g["max_timesteps"]), int(g["B"])
    m = shape_heads(full_drmnet(dev, precision, max_timesteps=T, epsilon=float(g["epsilon"]), gamma=float(g["gamma"]), delta=float(g["delta"])), g)
    LrK = synth.synth_refmaps(B, 128, 128, int(g["input_seed"]))
    gen = torch.Generator().manual_seed(int(g["gen_seed"]))
    n0 = torch.randn(LrK.shape, generator=gen)
    sn = torch.randn((T,) + tuple(LrK.shape), generator=gen)
    assert abs(synth.checksum(LrK) - float(g["LrK_sum"])) < 1e-6 * max(1.0, abs(float(g["LrK_sum"])))
    assert abs(synth.checksum(sn) - float(g["noise_sum"])) < 1e-6 * max(1.0, abs(float(g["noise_sum"])))
    LrK, n0, sn = LrK.to(dev), n0.to(dev), sn.to(dev)
    Lr0, zK, K = m.p_sample_loop(LrK, [LrK], [LrK], verbose=False, noise0=n0, step_noise=sn)
    e = rel_l2(Lr0.cpu(), g["Lr0"])
    print(f"full-width DRMNet loop ({precision}): K = {K.tolist()} (reference {g['K'].tolist()}), Lr0 rel-L2 {e:.2e}")
    assert K.tolist() == g["K"].tolist() and sorted(set(K.tolist())) == [2, 3, 5]  # early, mid-loop, never
    tol, ztol = (2e-5, 1e-5) if precision != "f16mx" else (MODE_TOL["f16mx"], 1e-4)
    assert e < tol
    assert np.allclose(zK.cpu().numpy(), g["zK"], atol=ztol, equal_nan=True) and np.isnan(g["zK"]).any()
    # host-driven loop with intermediates: every logged step of the reference trace
    Lr0h, zKh, Kh, inter = m.p_sample_loop(LrK, [LrK], [LrK], return_intermediates=True, verbose=False, log_every_k=1, noise0=n0, step_noise=sn)
    assert Kh.tolist() == g["K"].tolist() and rel_l2(Lr0h.cpu(), g["Lr0"]) < tol
    steps = torch.stack(inter["Lrk_inter"][1:])[:, :, :, ::4, ::4].cpu()
    assert tuple(steps.shape) == tuple(g["Lrk_steps"].shape)
    for i in range(steps.shape[0]):
        assert rel_l2(steps[i], g["Lrk_steps"][i]) < tol, i
        assert np.allclose(inter["zk_inter"][i].cpu().numpy(), g["zk_steps"][i], atol=ztol, equal_nan=True), i
    del m
    torch.cuda.empty_cache()


# --------------------------------------------------------------------------------------------- configs[4]: the whole chain, full width


def full_chain_models(g, dev, precision):
    from drmnet_amd.config import instantiate_from_config, load_config

    drm = shape_heads(full_drmnet(dev, precision, max_timesteps=int(g["max_timesteps"]), epsilon=float(g["epsilon"]), gamma=float(g["gamma"]),
                                  delta=float(g["delta"])), g)
    ocfg = load_config(os.path.join(ROOT, "configs/obsnet/eval_obsnet.yaml"))
    op = ocfg["model"]["params"]
    op.pop("ckpt_path", None)
    op.update(use_ema=False, linear_start=float(g["obs_linear_start"]), linear_end=float(g["obs_linear_end"]))
    obs = instantiate_from_config(ocfg["model"])
    synth.load_synth(obs.model.diffusion_model, synth.SEED_OBSNET)
    osd = obs.model.diffusion_model.state_dict()
    osd["out.2.weight"] = osd["out.2.weight"] * float(g["obs_out_scale"])
    osd["out.2.bias"] = osd["out.2.bias"] * float(g["obs_out_scale"])
    obs.model.diffusion_model.load_state_dict(osd)
    obs.ds = instantiate_from_config(ocfg["data"]["params"]["predict"])
    return drm, obs.to(dev).set_precision(precision)


def sample_object(dev):
    from drmnet_amd import file_io

    d = os.path.join(GOLD, "sample")
    img = file_io.load_exr(os.path.join(d, "image.exr"), as_torch=True).to(dev)
    nrm = torch.from_numpy(np.load(os.path.join(d, "normal.npy"))).to(dev)
    mask = torch.logical_and(file_io.load_png(os.path.join(d, "mask.png"), as_torch=True).to(dev) > 0, torch.linalg.norm(nrm, dim=-1) > 0.5)
    return img, nrm, mask


def chain_draws(g, res=128):
    """The reference run's draws, regenerated (tools/make_golden.py: torch CPU generator, seed 77, in this order)."""
    gen = torch.Generator().manual_seed(int(g["gen_seed"]))
    T = int(g["max_timesteps"])
    x_T = torch.randn((1, 3, res, res), generator=gen)
    noise = torch.randn((50, 1, 3, res, res), generator=gen)
    noise0 = torch.randn((1, 3, res, res), generator=gen)
    step_noise = torch.randn((T, 1, 3, res, res), generator=gen)
    return x_T, noise, noise0, step_noise


@pytest.mark.parametrize("precision,B", [("f16x3", 8), ("f16mx", 8), ("f16mx", 64)])
def test_full_width_estimate_chain_and_batch8(dev, precision, B):
    """B = 64 is configs[4]'s per-GPU shard (batch 512 object images over 8 GPUs), in the arithmetic bench.py runs it in"""
    from drmnet_amd.estimate import estimate, estimate_batch

    g = gold("estimate_chain_full")
    drm, obs = full_chain_models(g, dev, precision)
    assert drm.ds.size == 128 and obs.ddim_steps == 50
    img, nrm, mask = sample_object(dev)
    x_T, noise, noise0, step_noise = (t.to(dev) for t in chain_draws(g))
    stages = {}
    hooks = {"cond_noise": torch.from_numpy(g["cond"]).to(dev), "x_T": x_T, "noise": noise, "noise0": noise0, "step_noise": step_noise, "stages": stages}
    Lr0, zK = estimate(drm, obs, img, nrm, mask, hooks=hooks)
    env = drm.r0toenvmap(Lr0[None], (drm.image_size, drm.image_size * 2))[0]
    assert np.array_equal(stages["refmask"].cpu().numpy(), g["refmask"])
    e = {k: rel_l2(stages[k].cpu(), g[k]) for k in ("cond", "inpaint", "LrK")}
    e["Lr0"] = rel_l2(Lr0.cpu(), g["Lr0"])
    e["envmap"] = rel_l2(env.cpu(), g["envmap"])
    print(f"full-width estimate chain ({precision}):", {k: f"{v:.2e}" for k, v in e.items()}, "steps", drm.last_steps, "zK", zK.tolist())
    assert e["cond"] < 1e-6 and max(e["inpaint"], e["LrK"], e["Lr0"], e["envmap"]) < 1e-4  # north-star tolerance, end to end (50 + K steps)
    assert drm.last_steps == int(g["K"][0]) and np.allclose(zK.cpu().numpy(), g["zK"][0], atol=1e-5 if precision != "f16mx" else 1e-4)

    # configs[4]: a batch of objects through estimate_batch with early exit on.  Row 0 = the golden object with the golden draws; rows
    # 1.. = the same object started from other x_T / step noise (different inpaintings -> different BRDF trajectories).
    gen = torch.Generator().manual_seed(4242)
    T = int(g["max_timesteps"])
    bx_T = torch.cat([x_T.cpu(), torch.randn((B - 1, 3, 128, 128), generator=gen)]).to(dev)
    bnoise = torch.cat([noise.cpu(), torch.randn((50, B - 1, 3, 128, 128), generator=gen)], dim=1).to(dev)
    bnoise0 = torch.cat([noise0.cpu(), torch.randn((B - 1, 3, 128, 128), generator=gen)]).to(dev)
    bstep = torch.cat([step_noise.cpu(), torch.randn((T, B - 1, 3, 128, 128), generator=gen)], dim=1).to(dev)
    bh = {"cond_noise": hooks["cond_noise"].repeat(B, 1, 1, 1), "x_T": bx_T, "noise": bnoise, "noise0": bnoise0, "step_noise": bstep}
    Lr0_b, zK_b, K_b = estimate_batch(drm, obs, img[None].repeat(B, 1, 1, 1), nrm[None].repeat(B, 1, 1, 1), mask[None].repeat(B, 1, 1), hooks=bh)
    print(f"estimate_batch B = {B} full width ({precision}): K =", K_b.tolist())
    assert torch.isfinite(Lr0_b).all() and rel_l2(Lr0_b[0].cpu(), g["Lr0"]) < 1e-4 and int(K_b[0]) == int(g["K"][0])
    for r in (3, B - 1):  # a row of the batch == that object alone
        h1 = {"cond_noise": hooks["cond_noise"], "x_T": bx_T[r:r + 1], "noise": bnoise[:, r:r + 1], "noise0": bnoise0[r:r + 1], "step_noise": bstep[:, r:r + 1]}
        Lr0_1, zK_1 = estimate(drm, obs, img, nrm, mask, hooks=h1)
        assert rel_l2(Lr0_b[r].cpu(), Lr0_1.cpu()) < 1e-5 and drm.last_steps == int(K_b[r])
        assert np.allclose(zK_b[r].cpu().numpy(), zK_1.cpu().numpy(), atol=1e-5, equal_nan=True)
    del drm, obs
    torch.cuda.empty_cache()


def test_precision_switches_on_one_handle_at_batch1(dev):
    """the statistics pool (which also holds the split-K ticket counters of the split modes at small batches) is sized per
    (N, H, W, precision): fp32 first, then the split modes, then fp32 again on the SAME handle and shape."""
    gd = gold("full_illnet_128x128")
    m = build(ou.ILLNET_CFG, "unet", int(gd["seed"]), dev)
    xc, t_emb = full_inputs(2, 128, 128)
    x1, t1 = xc[:1].contiguous().to(dev), t_emb[:1].contiguous().to(dev)
    for precision, tol in (("fp32", 2e-5), ("f16x3", 2e-5), ("f16", 5e-3), ("fp32", 2e-5), ("f16x3", 2e-5)):
        out = m.set_precision(precision)(x1, t_emb=t1)
        e = rel_l2(out[0].cpu(), gd["out"][0])
        assert e < tol, (precision, e)
    del m
    torch.cuda.empty_cache()
