"""CPU: the oracle against the round-4 fixtures recorded from the reference (tools/make_golden.py --only long_chains stress ema_ckpt resize):

* the two loop lengths the shipped configs run -- DRMNet's 150-step reverse process (models/drmnet.py:782-847) with rows that leave after
  3 ... 148 steps and one that never converges, and the whole 1000-step ancestral chain (ldm/models/diffusion/ddpm.py:1120-1167);
* full-width networks with heavy-tailed weights and GroupNorm gains x 10 (the weight-stress case of the split arithmetic modes);
* checkpoints written by the reference's own modules with use_ema=True (live weights, LitEma shadows moved by LitEma.forward), sampled
  inside and outside `with model.ema_scope()`;
* BaseDataset's resize at sizes != input (torchvision's anti-aliased resize) and the nearest mask resize.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, gold, rel_l2
from drmnet_amd import synth
from oracle import samplers as osamp
from oracle import transforms as ot
from oracle import unet as ou

TOL = 2e-5


def long_drm_inputs(g):
    B, T = int(g["B"]), int(g["T"])
    LrK = synth.synth_refmaps(B, 16, 16, int(g["input_seed"]))
    gen = torch.Generator().manual_seed(int(g["gen_seed"]))
    n0 = torch.randn(LrK.shape, generator=gen)
    sn = torch.randn((T,) + tuple(LrK.shape), generator=gen)
    assert synth.checksum(LrK) == pytest.approx(float(g["LrK_sum"]), rel=1e-6) and synth.checksum(sn) == pytest.approx(float(g["noise_sum"]), rel=1e-12)
    return LrK, n0, sn


def long_obs_inputs(g):
    B, T = int(g["B"]), int(g["T"])
    gen = torch.Generator().manual_seed(int(g["gen_seed"]))
    cond = synth.synth_refmaps(B, 16, 16, int(g["input_seed"])) * 2 - 1
    x_T = torch.randn((B, 3, 16, 16), generator=gen)
    noise = torch.randn((T, B, 3, 16, 16), generator=gen)
    assert synth.checksum(noise) == pytest.approx(float(g["noise_sum"]), rel=1e-12)
    return cond, x_T, noise


def test_drmnet_150_step_loop():
    g = gold("drmnet_loop_150")
    Pu = synth.synth_state_dict(ou.param_manifest(ou.TINY_UNET_CFG, "unet"), 21)
    Pe = synth.synth_state_dict(ou.param_manifest(ou.TINY_ENC_CFG, "encoder"), 22)
    Pz = synth.synth_state_dict(ou.zemb_manifest(6, 32), synth.SEED_ZEMB)
    Pu["out.2.weight"] = Pu["out.2.weight"] * float(g["ill_out_scale"])
    Pu["out.2.bias"] = Pu["out.2.bias"] * float(g["ill_out_scale"])
    Pe["out.3.weight"] = Pe["out.3.weight"] * float(g["head_w_scale"])
    Pe["out.3.bias"] = torch.from_numpy(g["head_bias"])
    tu, te = ou.build_topology(ou.TINY_UNET_CFG, "unet"), ou.build_topology(ou.TINY_ENC_CFG, "encoder")
    LrK, n0, sn = long_drm_inputs(g)
    trace = []
    Lr0, zK, K = osamp.drmnet_sample(lambda xc, t: ou.encoder_forward(Pe, te, xc, t), lambda xc, dz: ou.unet_forward(Pu, tu, xc, t_emb=ou.z_embed(Pz, dz)),
                                     LrK, n0, sn, torch.from_numpy(g["z0"]), float(g["gamma"]), float(g["epsilon"]), float(g["delta"]), int(g["T"]), trace=trace)
    assert K.tolist() == g["K"].tolist() and max(K.tolist()) == 150 and min(K.tolist()) < 10 and sorted(K.tolist())[-3] > 100
    assert np.array_equal(np.isnan(zK.numpy()), np.isnan(g["zK"])) and np.isnan(g["zK"]).any()
    assert np.allclose(np.nan_to_num(zK.numpy()), np.nan_to_num(g["zK"]), atol=1e-5)
    assert rel_l2(Lr0, g["Lr0"]) < 1e-4
    assert rel_l2(trace[100]["Lr_k"][trace[100]["idx"]], torch.from_numpy(g["Lrk_steps"][10])[trace[100]["idx"]]) < 1e-4  # logged every 10 steps


def test_ancestral_1000_step_chain():
    g = gold("ddpm_trace_1000")
    Pu = synth.synth_state_dict(ou.param_manifest(ou.TINY_UNET_CFG, "unet"), 21)
    tu = ou.build_topology(ou.TINY_UNET_CFG, "unet")
    S = osamp.ddpm_schedule(int(g["T"]), float(g["linear_start"]), float(g["linear_end"]))
    cond, x_T, noise = long_obs_inputs(g)
    pred_x0, img, imgs = osamp.ddpm_sample(lambda xc, t: ou.unet_forward(Pu, tu, xc, timesteps=t), cond, x_T, noise, S)
    assert len(imgs) == 1000
    assert rel_l2(img, g["x"]) < 1e-4 and rel_l2(pred_x0, g["pred_x0"]) < 1e-4
    # the reference logs after i = 999 and after every i % 100 == 0: x_inter[1 + k] is the state after i = 1000 - 100 k (k >= 1)
    assert rel_l2(imgs[0], g["x_inter"][0]) < TOL and rel_l2(imgs[99], g["x_inter"][1]) < 1e-4 and rel_l2(imgs[499], g["x_inter"][5]) < 1e-4


@pytest.mark.parametrize("gain", [3, 10])
@pytest.mark.parametrize("name,cfg,kind", [("illnet", ou.ILLNET_CFG, "unet"), ("refnet", ou.REFNET_CFG, "encoder"), ("obsnet", ou.OBSNET_CFG, "unet")])
def test_stress_weights_full_width(name, cfg, kind, gain):
    """gain 3: the oracle reproduces the reference at the usual fp32 re-association bar.  gain 10: two fp32 evaluations of these networks
    differ by 1e-5 ... 1e-2 (the fixture's fp32 reference sits that far from the same network in fp64), so the oracle is held to the fp64
    answer it generated (exactly) and to the fp32 reference only within that noise."""
    from test_oracle_golden import full_inputs

    g = gold(f"stress{gain}_{name}")
    P = synth.synth_state_dict(ou.param_manifest(cfg, kind), int(g["seed"]), rule=f"stress:{gain}")
    assert synth.checksum(torch.cat([v.flatten() for v in P.values()])) == pytest.approx(float(g["wsum"]), rel=1e-9)
    topo = ou.build_topology(cfg, kind)
    n, h, w = 2, 32, 64
    xc, t_emb = full_inputs(n, h, w)
    t = torch.from_numpy(g["t"])[:n]
    fwd = lambda P, x, te: ou.unet_forward(P, topo, x, t_emb=te) if name == "illnet" else (ou.encoder_forward(P, topo, x, t) if kind == "encoder" else ou.unet_forward(P, topo, x, timesteps=t))
    out = fwd(P, xc, t_emb)
    ref32, ref64 = g[f"out_{n}x{h}x{w}"], g[f"out64_{n}x{h}x{w}"]
    r64 = rel_l2(ref32, ref64)
    if gain == 3:
        assert r64 < 5e-6 and rel_l2(out, ref32) < TOL
    else:
        assert rel_l2(out, ref32) < max(20 * r64, 2e-4)
        with ou.working_dtype(torch.float64):
            out64 = fwd({k: v.double() for k, v in P.items()}, xc.double(), t_emb.double())
        assert rel_l2(out64, ref64) < 1e-9


def ckpt_params(sd, prefix, ema_prefix, keys, ema):
    """live: sd[prefix + key]; EMA: sd[ema_prefix + (key with the dots removed)] (ldm/modules/ema.py:16-21)"""
    return {k: (sd[ema_prefix + k.replace(".", "")] if ema else sd[prefix + k]) for k in keys}


def test_reference_written_drmnet_checkpoint_and_ema_scope():
    g = gold("ema_drmnet")
    sd = torch.load(os.path.join(GOLD, "drmnet_tiny_ema.ckpt"), map_location="cpu", weights_only=True)["state_dict"]
    assert int(sd["illnet_model_ema.num_updates"]) == int(g["num_updates"]) == 3 and float(sd["illnet_model_ema.decay"]) == pytest.approx(0.9999)
    ku = ["diffusion_model." + k for k, _ in ou.param_manifest(ou.TINY_UNET_CFG, "unet")]
    ke = ["diffusion_model." + k for k, _ in ou.param_manifest(ou.TINY_ENC_CFG, "encoder")]
    kz = [k for k, _ in ou.zemb_manifest(6, 32)]
    T, B = int(g["T"]), int(g["B"])
    LrK = synth.synth_refmaps(B, 16, 32, int(g["input_seed"]))
    gen = torch.Generator().manual_seed(int(g["gen_seed"]))
    n0 = torch.randn(LrK.shape, generator=gen)
    sn = torch.randn((T,) + tuple(LrK.shape), generator=gen)
    tu, te = ou.build_topology(ou.TINY_UNET_CFG, "unet"), ou.build_topology(ou.TINY_ENC_CFG, "encoder")
    outs = {}
    for which, ema in (("live", False), ("ema", True)):
        Pu = {k[len("diffusion_model."):]: v for k, v in ckpt_params(sd, "illnet_model.", "illnet_model_ema.", ku, ema).items()}
        Pe = {k[len("diffusion_model."):]: v for k, v in ckpt_params(sd, "refnet_model.", "refnet_model_ema.", ke, ema).items()}
        Pz = ckpt_params(sd, "illnet_model.", "illnet_model_ema.", kz, ema)
        Lr0, zK, K = osamp.drmnet_sample(lambda xc, t: ou.encoder_forward(Pe, te, xc, t), lambda xc, dz: ou.unet_forward(Pu, tu, xc, t_emb=ou.z_embed(Pz, dz)),
                                         LrK, n0, sn, sd["z0"], float(g["gamma"]), float(g["epsilon"]), float(g["delta"]), T)
        assert K.tolist() == g["K"].tolist()
        assert rel_l2(Lr0, g[f"Lr0_{which}"]) < TOL, which
        outs[which] = Lr0
        te_in = torch.randn((B, 32), generator=torch.Generator().manual_seed(int(g["temb_seed"])))
        assert rel_l2(ou.unet_forward(Pu, tu, torch.cat([LrK, LrK], 1), t_emb=te_in), g[f"illnet_{which}"]) < TOL
    assert rel_l2(outs["live"], outs["ema"]) > 1e-3  # the shadows really moved away from the live weights


def test_reference_written_obsnet_checkpoint_and_ema_scope():
    g = gold("ema_obsnet")
    sd = torch.load(os.path.join(GOLD, "obsnet_tiny_ema.ckpt"), map_location="cpu", weights_only=True)["state_dict"]
    keys = ["diffusion_model." + k for k, _ in ou.param_manifest(ou.TINY_UNET_CFG, "unet")]
    B = int(g["B"])
    gen = torch.Generator().manual_seed(int(g["gen_seed"]))
    cond = synth.synth_refmaps(B, 16, 16, int(g["input_seed"])) * 2 - 1
    x_T = torch.randn((B, 3, 16, 16), generator=gen)
    noise = torch.randn((50, B, 3, 16, 16), generator=gen)
    tu = ou.build_topology(ou.TINY_UNET_CFG, "unet")
    S = osamp.ddpm_schedule(1000, 1e-4, 0.09)
    assert torch.equal(S["alphas_cumprod"].float(), sd["alphas_cumprod"])  # the schedule buffers travel in the reference's checkpoint too
    d = osamp.ddim_schedule(S["alphas_cumprod"], 1000, 50, 1.0)
    for which, ema in (("live", False), ("ema", True)):
        P = {k[len("diffusion_model."):]: v for k, v in ckpt_params(sd, "model.", "model_ema.", keys, ema).items()}
        x, _ = osamp.ddim_sample(lambda xc, t: ou.unet_forward(P, tu, xc, timesteps=t), cond, x_T, noise, d, num_steps=int(g["steps"]))
        assert rel_l2(x, g[f"x_{which}"]) < TOL, which
        if ema:
            eps = ou.unet_forward(P, tu, torch.cat([x_T, cond], 1), timesteps=torch.full((B,), 981, dtype=torch.long))
            assert rel_l2(eps, g["eps_ema"]) < TOL


def test_resize_restatement_vs_reference():
    g = gold("resize")
    hdr, rect, big, mask = (torch.from_numpy(g[k]) for k in ("hdr", "rect", "big", "mask"))
    assert rel_l2(ot.resize(hdr, (16, 16)), g["resize_only"]) < 1e-6
    assert rel_l2(ot.transform(hdr, "log_resize", size=16)[0], g["log_of_resized"]) < 1e-6
    assert rel_l2(ot.transform(hdr, "resize_log", size=16)[0], g["resized_log"]) < 1e-6
    assert rel_l2(ot.resize(rect, (16, 16)), g["rect_16"]) < 1e-6 and rel_l2(ot.resize(rect, (24, 24)), g["rect_24"]) < 1e-6
    assert rel_l2(ot.resize(big, (48, 48)), g["big_48"]) < 1e-6
    assert rel_l2(ot.resize(hdr, (16, 16), "bicubic"), g["bicubic_16"]) < 1e-6
    assert torch.equal(ot.resize(hdr, (16, 16), "nearest"), torch.from_numpy(g["nearest_16"]))
    for key, src, size in (("mask_16", mask, (16, 16)), ("mask_64", mask, (64, 64)), ("mask_rect", mask[:, :, :24, :], (16, 16))):
        assert torch.equal(ot.resize(src, size, "nearest"), torch.from_numpy(g[key])), key
