#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE's own Python on CPU in this container.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py [--only NAME ...]

Needs /root/reference (read-only).  Writes only data (npz/json) -- no reference
source, bytecode or text is stored.  Weights are the seeded rule of
drmnet_amd/synth.py, loaded into the reference modules by state_dict order; the
fixtures keep (seed, checksums), not the weights.  All noise is injected by
patching the reference's RNG call sites (torch.randn_like in models/drmnet.py:797,823;
noise_like in ddim.py:256 / ddpm.py:1156) so traces are reproducible.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import refharness as rh  # noqa: E402
from drmnet_amd import synth  # noqa: E402
from oracle import unet as ou  # only for the CFG dicts (restated from the YAMLs)  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def save(name, **arrs):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()})
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.0f} KiB)")


def gen(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return g


def load_rule(module, seed):
    synth.load_synth(module, seed)
    return synth.checksum(torch.cat([v.flatten() for v in module.state_dict().values()]))


# ----------------------------------------------------------------------------- manifests


def make_manifests(oa):
    out = {}
    for name, cfg, cls in (
        ("illnet", ou.ILLNET_CFG, oa.UNetModel),
        ("refnet", ou.REFNET_CFG, oa.EncoderUNetModel),
        ("obsnet", ou.OBSNET_CFG, oa.UNetModel),
        ("tiny_unet", ou.TINY_UNET_CFG, oa.UNetModel),
        ("tiny_enc", ou.TINY_ENC_CFG, oa.EncoderUNetModel),
    ):
        m = cls(**cfg)
        out[name] = [[k, list(v.shape)] for k, v in m.state_dict().items()]
        print(f"  {name}: {len(out[name])} tensors, {sum(v.numel() for v in m.state_dict().values()) / 1e6:.2f} M params")
    DRM, OBS, _, _ = rh.ref_classes()
    cfg = rh.load_yaml_params("configs/drmnet/eval_drmnet.yaml")["model"]["params"]
    cfg.pop("ckpt_path")
    out["drmnet_model"] = [[k, list(v.shape)] for k, v in DRM(**cfg).state_dict().items()]
    cfg = rh.load_yaml_params("configs/obsnet/eval_obsnet.yaml")["model"]["params"]
    cfg.pop("ckpt_path")
    out["obsnet_model"] = [[k, list(v.shape)] for k, v in OBS(**cfg).state_dict().items()]
    with open(os.path.join(GOLD, "manifests.json"), "w") as f:
        json.dump(out, f)
    print("  wrote manifests.json")


# ----------------------------------------------------------------------------- primitives / schedules


def make_primitives(oa):
    from ldm.modules.diffusionmodules.util import timestep_embedding

    t = torch.tensor([0, 1, 21, 149, 981, 999], dtype=torch.long)
    save("timestep_embedding", t=t, emb=timestep_embedding(t, 128))

    DRM, OBS, DDIM, _ = rh.ref_classes()
    cfg = rh.load_yaml_params("configs/obsnet/eval_obsnet.yaml")["model"]["params"]
    cfg.pop("ckpt_path")
    cfg["unet_config"] = {"target": cfg["unet_config"]["target"], "params": dict(ou.TINY_UNET_CFG)}
    m = OBS(**cfg).eval()
    names = [
        "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
        "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
        "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
    ]
    save("ddpm_schedule", **{n: getattr(m, n) for n in names})
    for eta in (0.0, 1.0):
        s = DDIM(m)
        s.make_schedule(ddim_num_steps=50, ddim_eta=eta, verbose=False)
        # the five fp32 scalars p_sample_ddim derives per index (ddim.py:243-258)
        coef = np.zeros((50, 5), dtype=np.float32)
        for i in range(50):
            a_t = torch.full((1,), s.ddim_alphas[i])
            a_prev = torch.full((1,), s.ddim_alphas_prev[i])
            sigma = torch.full((1,), s.ddim_sigmas[i])
            s1m = torch.full((1,), s.ddim_sqrt_one_minus_alphas[i])
            coef[i] = [a_t.sqrt().item(), s1m.item(), a_prev.sqrt().item(), (1.0 - a_prev - sigma**2).sqrt().item(), sigma.item()]
        save(f"ddim_schedule_eta{int(eta)}", timesteps=s.ddim_timesteps, coef=coef,
             alphas=np.asarray(s.ddim_alphas, dtype=np.float64), alphas_prev=np.asarray(s.ddim_alphas_prev, dtype=np.float64),
             sigmas=np.asarray(s.ddim_sigmas, dtype=np.float64))


def make_brdf_schedule():
    DRM, _, _, _ = rh.ref_classes()
    cfg = rh.load_yaml_params("configs/drmnet/eval_drmnet.yaml")["model"]["params"]
    cfg.pop("ckpt_path")
    cfg["illnet_config"] = {"target": cfg["illnet_config"]["target"], "params": dict(ou.TINY_UNET_CFG)}
    cfg["refnet_config"] = {"target": cfg["refnet_config"]["target"], "params": dict(ou.TINY_ENC_CFG)}
    m = DRM(**cfg).eval()
    g = gen(77)
    z_out = torch.rand((16, 6), generator=g) * 1.6 - 0.3
    z_out[0] = m.z0  # exactly converged row (distance == 0 branch)
    z_out[1] = m.z0 + 1e-3
    arrs = {"z_out": z_out, "z0": m.z0, "gamma": m.gamma, "epsilon": m.epsilon}
    for i in (0, 1, 7, 50, 90, 149):
        zk, zK = m.get_brdf_out(z_out, reversed_k=i)
        arrs[f"zk_{i}"] = zk
        arrs[f"conv_{i}"] = m.check_convergence(zk)
        arrs["zK"] = zK
    save("brdf_schedule", **arrs)


# ----------------------------------------------------------------------------- U-Net forwards


def make_tiny_nets(oa):
    g = gen(11)
    for tag, (h, w) in (("16x16", (16, 16)), ("16x32", (16, 32))):
        x = torch.randn((2, 6, h, w), generator=g)
        t = torch.tensor([3, 977], dtype=torch.long)
        t_emb = torch.randn((2, 32), generator=g)
        u = oa.UNetModel(**ou.TINY_UNET_CFG).eval()
        cs = load_rule(u, 21)
        with torch.no_grad():
            save(f"tiny_unet_{tag}", x=x, t=t, t_emb=t_emb, out_t=u(x, timesteps=t), out_temb=u(x, t_emb=t_emb), seed=21, wsum=cs)
        e = oa.EncoderUNetModel(**ou.TINY_ENC_CFG).eval()
        cs = load_rule(e, 22)
        with torch.no_grad():
            save(f"tiny_enc_{tag}", x=x, t=t, out=e(x, t), seed=22, wsum=cs)


def block_inputs(kind, a, b, h, w, n):
    """Regenerable inputs for the single-block fixtures (tests rebuild them from the same seed)."""
    g = gen(1000 + a + 7 * b + 13 * h + 17 * w)
    emb = torch.randn((n, 512), generator=g)
    x = torch.randn((n, a, h, w), generator=g)
    return x, emb


def make_blocks(oa):
    """Full-width single blocks from the reference classes (weights by rule; inputs regenerable; only outputs stored)."""
    for cin, cout, hw, n in ((256, 128, 16, 2), (128, 128, 16, 2), (1536, 768, 4, 2)):
        rb = oa.ResBlock(cin, 512, 0.0, out_channels=cout).eval()
        cs = load_rule(rb, 31)
        x, emb = block_inputs("res", cin, cout, hw, hw, n)
        with torch.no_grad():
            save(f"resblock_{cin}_{cout}_{hw}", out=rb(x, emb), seed=31, wsum=cs, xsum=synth.checksum(x), n=n)
    for ch, h, w, n in ((512, 16, 16, 2), (384, 32, 32, 1), (768, 4, 8, 2)):
        ab = oa.AttentionBlock(ch, num_heads=1, num_head_channels=-1).eval()
        cs = load_rule(ab, 32)
        x, _ = block_inputs("attn", ch, ch, h, w, n)
        with torch.no_grad():
            save(f"attnblock_{ch}_{h}x{w}", out=ab(x), seed=32, wsum=cs, xsum=synth.checksum(x), n=n)


def full_inputs(n, h, w):
    x = synth.synth_refmaps(n, h, w, synth.SEED_INPUT)
    g = gen(synth.SEED_INPUT + 1)
    xk = x + 0.025 * torch.randn(x.shape, generator=g)
    t_emb = torch.randn((n, 128), generator=g)
    return torch.cat([xk, x], dim=1).contiguous(), t_emb


def make_full_nets(oa):
    for name, cfg, cls, seed in (
        ("illnet", ou.ILLNET_CFG, oa.UNetModel, synth.SEED_ILLNET),
        ("refnet", ou.REFNET_CFG, oa.EncoderUNetModel, synth.SEED_REFNET),
        ("obsnet", ou.OBSNET_CFG, oa.UNetModel, synth.SEED_OBSNET),
    ):
        m = cls(**cfg).eval()
        cs = load_rule(m, seed)
        for n, h, w in ((2, 128, 128), (1, 128, 256)):
            xc, t_emb = full_inputs(n, h, w)
            t = torch.tensor([7, 981][:n], dtype=torch.long)
            t0 = time.time()
            with torch.no_grad():
                if name == "illnet":
                    out = m(xc, t_emb=t_emb)
                else:
                    out = m(xc, t)
            print(f"  {name} {n}x{h}x{w}: {time.time() - t0:.1f}s  out std {out.std():.4f} absmax {out.abs().max():.3f}")
            save(f"full_{name}_{h}x{w}", out=out, t=t, seed=seed, wsum=cs, xsum=synth.checksum(xc), tembsum=synth.checksum(t_emb))


FULL_ROWS = (0, 13, 31)


def full_rows_inputs(B=32, h=128, w=256):
    """32 DISTINCT rows at the metric shape: refmaps, noised copies, per-row embeddings and per-row timesteps (regenerable: seeds 4242 / 4243)."""
    x = synth.synth_refmaps(B, h, w, 4242)
    g = gen(4243)
    xk = x + 0.025 * torch.randn(x.shape, generator=g)
    t_emb = torch.randn((B, 128), generator=g) * (0.25 + torch.arange(B, dtype=torch.float32) / 16.0)[:, None]
    t = (torch.arange(B, dtype=torch.long) * 31 + 7) % 1000
    return x, xk, t_emb, t


def make_full_rows(oa):
    """BASELINE configs[1]'s batch with 32 distinct rows, embeddings and timesteps: the reference's outputs for rows 0, 13 and 31 (evaluated as a batch
    of three: the reference has no cross-row term) of IllNet (t_emb per row), RefNet and ObsNet (timestep per row) at full width, 3x128x256, and of one
    DRMNet reverse step (DRMNet.p_mean_variance, models/drmnet.py:752-770, reversed_k = 3) on those rows.  Outputs stored at every second pixel."""
    x, xk, t_emb, t = full_rows_inputs()
    rows = list(FULL_ROWS)
    xc = torch.cat([xk, x], dim=1)[rows].contiguous()
    out = dict(rows=np.asarray(rows), xsum=synth.checksum(x), xksum=synth.checksum(xk), tembsum=synth.checksum(t_emb), t=t)
    for name, cfg, cls, seed in (("illnet", ou.ILLNET_CFG, oa.UNetModel, synth.SEED_ILLNET), ("refnet", ou.REFNET_CFG, oa.EncoderUNetModel, synth.SEED_REFNET),
                                 ("obsnet", ou.OBSNET_CFG, oa.UNetModel, synth.SEED_OBSNET)):
        m = cls(**cfg).eval()
        load_rule(m, seed)
        t0 = time.time()
        with torch.no_grad():
            y = m(xc, t_emb=t_emb[rows]) if name == "illnet" else m(xc, t[rows])
        print(f"  {name} rows {rows}: {time.time() - t0:.1f}s  out std {y.std():.4f}")
        out[name] = y if name == "refnet" else y[:, :, ::2, ::2]
        del m
    m = full_drmnet(gamma=0.9, epsilon=0.01, max_timesteps=150)
    with torch.no_grad():
        mean, delta, z_out = m.p_mean_variance(xk[rows], [x[rows]], [x[rows]], reversed_k=3)
    out.update(step_mean=mean[:, :, ::2, ::2], step_z_out=z_out, step_k=3, step_delta=delta)
    save("full_rows", **out)


def make_full_sizes(oa):
    """UNetModel / EncoderUNetModel are fully convolutional (openaimodel.py:731-768): any H, W divisible by 2^(levels-1).
    Full-width outputs at sizes other than the shipped 128x128 / 128x256 (a different ds.size)."""
    for name, cfg, cls, seed, sizes in (
        ("illnet", ou.ILLNET_CFG, oa.UNetModel, synth.SEED_ILLNET, ((1, 64, 64), (1, 96, 160), (1, 192, 192), (2, 32, 64))),
        ("refnet", ou.REFNET_CFG, oa.EncoderUNetModel, synth.SEED_REFNET, ((1, 64, 64), (1, 96, 160), (2, 16, 48))),
        ("obsnet", ou.OBSNET_CFG, oa.UNetModel, synth.SEED_OBSNET, ((1, 64, 64), (1, 96, 160), (2, 16, 48))),
    ):
        m = cls(**cfg).eval()
        cs = load_rule(m, seed)
        arrs = {}
        for n, h, w in sizes:
            xc, t_emb = full_inputs(n, h, w)
            t = torch.tensor([7, 981][:n], dtype=torch.long)
            with torch.no_grad():
                out = m(xc, t_emb=t_emb) if name == "illnet" else m(xc, t)
            print(f"  {name} {n}x{h}x{w}: out std {out.std():.4f}")
            arrs[f"out_{n}x{h}x{w}"] = out
        save(f"full_{name}_sizes", seed=seed, wsum=cs, t=torch.tensor([7, 981]), **arrs)


# ----------------------------------------------------------------------------- samplers


def tiny_drmnet(gamma, epsilon, max_timesteps, delta=0.025):
    DRM, _, _, _ = rh.ref_classes()
    cfg = rh.load_yaml_params("configs/drmnet/eval_drmnet.yaml")["model"]["params"]
    cfg.pop("ckpt_path")
    cfg["illnet_config"] = {"target": cfg["illnet_config"]["target"], "params": dict(ou.TINY_UNET_CFG)}
    cfg["refnet_config"] = {"target": cfg["refnet_config"]["target"], "params": dict(ou.TINY_ENC_CFG)}
    cfg.update(image_size=16, gamma=gamma, epsilon=epsilon, max_timesteps=max_timesteps, delta=delta, use_ema=False)
    m = DRM(**cfg).eval()
    synth.load_synth(m.illnet_model.diffusion_model, 21)
    synth.load_synth(m.refnet_model.diffusion_model, 22)
    zsd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in m.illnet_model.z_emb_layer.state_dict().items()], synth.SEED_ZEMB)
    m.illnet_model.z_emb_layer.load_state_dict(zsd)
    return m


FULL_CHAIN_T, FULL_CHAIN_EPS = 8, 0.70
FULL_LOOP_EPS, FULL_LOOP_ILL_SCALE = 0.55, 0.02


def full_drmnet(gamma, epsilon, max_timesteps, delta=0.025):
    """The shipped configs/drmnet/eval_drmnet.yaml networks (IllNet 237.8 M, RefNet 33.3 M parameters by the synth rule)."""
    DRM, _, _, _ = rh.ref_classes()
    cfg = rh.load_yaml_params("configs/drmnet/eval_drmnet.yaml")["model"]["params"]
    cfg.pop("ckpt_path")
    cfg.update(gamma=gamma, epsilon=epsilon, max_timesteps=max_timesteps, delta=delta, use_ema=False)
    m = DRM(**cfg).eval()
    synth.load_synth(m.illnet_model.diffusion_model, synth.SEED_ILLNET)
    synth.load_synth(m.refnet_model.diffusion_model, synth.SEED_REFNET)
    zsd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in m.illnet_model.z_emb_layer.state_dict().items()], synth.SEED_ZEMB)
    m.illnet_model.z_emb_layer.load_state_dict(zsd)
    return m


def make_drmnet_loop(full=False):
    import models.drmnet as refdrm

    cases = (("a", (16, 16), 0.9, 0.78, 17, 10.0), ("b", (16, 32), 0.92, 0.715, 15, 6.0))
    if full:  # full-width p_sample_loop (models/drmnet.py:782-847) at the config shape: rows converging mid-loop, at the last step, never
        cases = (("full", (128, 128), 0.9, FULL_LOOP_EPS, 5, 3.0),)
    for tag, (h, w), gamma, eps, T, wscale in cases:
        m = (full_drmnet if full else tiny_drmnet)(gamma=gamma, epsilon=eps, max_timesteps=T)
        # spread the per-sample convergence step: amplify the RefNet head and centre it near z0
        head = m.refnet_model.diffusion_model.out[3]
        head_bias = torch.tensor([0.95, 0.9, 0.97, 0.92, 0.05, 0.9])
        B = 3 if full else 5
        LrK = synth.synth_refmaps(B, h, w, 99)
        if full:
            # random-weight full-width nets: damp the IllNet head (a trained denoiser makes small residual updates; undamped, Lr_k
            # grows by O(1) per step and every z saturates at the clamp) and centre the RefNet head so that the rows' z_out start
            # 0.73 away from z0, spread by the (amplified) weight part: K = [never, 2, 3] at epsilon = FULL_LOOP_EPS
            with torch.no_grad():
                o = m.illnet_model.diffusion_model.out[2]
                o.weight.mul_(FULL_LOOP_ILL_SCALE)
                o.bias.mul_(FULL_LOOP_ILL_SCALE)
                wpart = m.apply_model(m.refnet_model, LrK, 0, [LrK]) - head.bias
                head_bias = m.z0 + torch.tensor([-0.3, -0.3, -0.3, -0.3, 0.3, -0.3]) - wscale * wpart.mean(0)
        with torch.no_grad():
            head.weight.mul_(wscale)
            head.bias.copy_(head_bias)
        g = gen(41)
        noise0 = torch.randn(LrK.shape, generator=g)
        step_noise = torch.randn((T,) + tuple(LrK.shape), generator=g)
        # make one row converge immediately: bias RefNet head so z_out ~ z0 is impossible per-row; instead rely on spread.
        state = {"call": 0, "active": torch.ones(B, dtype=torch.bool), "nc_idx": None, "step": 0}
        orig_randn_like = torch.randn_like
        orig_check = m.check_convergence

        def check(zk):
            conv = orig_check(zk)
            idx = torch.where(state["active"])[0]
            state["nc_idx"] = idx[~conv]
            state["active"][idx[conv]] = False
            return conv

        def randn_like(t, **kw):
            if state["call"] == 0:
                state["call"] += 1
                return noise0.clone()
            out = step_noise[state["step"]][state["nc_idx"]]
            assert out.shape == t.shape, (out.shape, t.shape)
            state["step"] += 1
            state["call"] += 1
            return out

        m.check_convergence = check
        refdrm.torch.randn_like = randn_like
        try:
            Lr0, zK, K, inter = m.p_sample_loop(LrK, [LrK], [LrK], return_intermediates=True, verbose=False, log_every_k=1)
        finally:
            refdrm.torch.randn_like = orig_randn_like
        print(f"  drmnet loop {tag}: K = {K.tolist()}  zK nan rows = {torch.isnan(zK).any(dim=1).tolist()}")
        if full:
            dist = torch.stack([(z - m.z0).norm(dim=-1) for z in inter["zk_inter"]])
            print("  per-step |zk - z0| (rows):", dist.tolist())
            # inputs and draws are regenerable (synth_refmaps(B, h, w, 99); torch CPU generator seed 41: noise0, step_noise[T])
            save("drmnet_loop_full", Lr0=Lr0, zK=zK, K=K, z0=m.z0, gamma=gamma, epsilon=eps, delta=0.025, max_timesteps=T, head_w_scale=wscale,
                 head_bias=head_bias, ill_out_scale=FULL_LOOP_ILL_SCALE, gen_seed=41, input_seed=99, B=B, Lrk_steps=torch.stack(inter["Lrk_inter"][1:])[:, :, :, ::4, ::4],
                 zk_steps=torch.stack(inter["zk_inter"]), LrK_sum=synth.checksum(LrK), noise_sum=synth.checksum(step_noise))
            continue
        save(f"drmnet_loop_{tag}", LrK=LrK, noise0=noise0, step_noise=step_noise, Lr0=Lr0, zK=zK, K=K, z0=m.z0,
             gamma=gamma, epsilon=eps, delta=0.025, max_timesteps=T, head_w_scale=wscale, head_bias=head_bias,
             Lrk_steps=torch.stack(inter["Lrk_inter"][1:]), zk_steps=torch.stack(inter["zk_inter"]))


def tiny_obsnet():
    _, OBS, DDIM, _ = rh.ref_classes()
    cfg = rh.load_yaml_params("configs/obsnet/eval_obsnet.yaml")["model"]["params"]
    cfg.pop("ckpt_path")
    cfg["unet_config"] = {"target": cfg["unet_config"]["target"], "params": dict(ou.TINY_UNET_CFG)}
    cfg.update(image_size=16, use_ema=False)
    m = OBS(**cfg).eval()
    synth.load_synth(m.model.diffusion_model, 21)
    return m, DDIM


def make_obsnet_samplers():
    import ldm.models.diffusion.ddim as refddim
    import ldm.models.diffusion.ddpm as refddpm

    m, DDIM = tiny_obsnet()
    B, h, w = 3, 16, 16
    g = gen(51)
    cond = synth.synth_refmaps(B, h, w, 98) * 2 - 1
    x_T = torch.randn((B, 3, h, w), generator=g)
    noise = torch.randn((50, B, 3, h, w), generator=g)
    ctr = {"i": 0}

    def noise_like(shape, device, repeat=False):
        out = noise[ctr["i"]]
        ctr["i"] += 1
        assert tuple(out.shape) == tuple(shape)
        return out

    o1, o2 = refddim.noise_like, refddpm.noise_like
    refddim.noise_like = noise_like
    refddpm.noise_like = noise_like
    try:
        for eta in (1.0, 0.0):
            ctr["i"] = 0
            x, inter = DDIM(m).sample(50, B, (3, h, w), cond, eta=eta, x_T=x_T, verbose=False, log_every_t=1)
            save(f"ddim_trace_eta{int(eta)}", cond=cond, x_T=x_T, noise=noise, x=x, x_inter=torch.stack(inter["x_inter"][1:]))
        ctr["i"] = 0
        pred_x0, inter = m.p_sample_loop(cond, (B, 3, h, w), return_intermediates=True, x_T=x_T, verbose=False, start_T=6, log_every_t=1)
        save("ddpm_trace", cond=cond, x_T=x_T, noise=noise[:6], pred_x0=pred_x0, x_inter=torch.stack(inter["x_inter"][1:]))
    finally:
        refddim.noise_like, refddpm.noise_like = o1, o2


def make_sampler_masks():
    """The samplers' mask / x0 / temperature arguments, from the reference's own loops on the tiny ObsNet (B = 3 @16x16):
    DDIMSampler.sample(mask=, x0=, temperature=0.7) over 50 steps (ddim.py:175-178, :255), ObsNetDiffusion.p_sample_loop(mask=, x0=) over 6 steps
    (blend BEFORE p_sample, x0 / q_sample(x0, t - 1): models/obsnet.py:545-547), LatentDiffusion.p_sample_loop(mask=, x0=) over 6 steps (blend AFTER
    p_sample, q_sample(x0, t): ddpm.py:1300-1302), and two p_sample calls with temperature = 0.7 (ddpm.py:1157).  q_sample's draws are injected."""
    import ldm.models.diffusion.ddim as refddim
    import ldm.models.diffusion.ddpm as refddpm

    m, DDIM = tiny_obsnet()
    B, h, w = 3, 16, 16
    g = gen(77)
    cond = synth.synth_refmaps(B, h, w, 98) * 2 - 1
    x0 = synth.synth_refmaps(B, h, w, 99) * 2 - 1
    x_T = torch.randn((B, 3, h, w), generator=g)
    noise = torch.randn((50, B, 3, h, w), generator=g)
    qnoise = torch.randn((50, B, 3, h, w), generator=g)
    mask1 = (torch.rand((B, 1, h, w), generator=g) > 0.5).float()      # one channel, broadcast
    mask3 = torch.rand((B, 3, h, w), generator=g)                       # per-channel soft mask
    ctr = {"i": 0, "q": 0}

    def noise_like(shape, device, repeat=False):
        out = noise[ctr["i"]]
        ctr["i"] += 1
        assert tuple(out.shape) == tuple(shape)
        return out

    q_orig = m.q_sample

    def q_sample(x_start, t, noise=None):
        out = q_orig(x_start, t, noise=qnoise[ctr["q"]])
        ctr["q"] += 1
        return out

    o1, o2 = refddim.noise_like, refddpm.noise_like
    refddim.noise_like = noise_like
    refddpm.noise_like = noise_like
    m.q_sample = q_sample
    out = dict(cond=cond, x0=x0, x_T=x_T, noise=noise, qnoise=qnoise, mask1=mask1, mask3=mask3, temperature=0.7)
    try:
        for tag, mk in (("m1", mask1), ("m3", mask3)):
            ctr.update(i=0, q=0)
            x, inter = DDIM(m).sample(50, B, (3, h, w), cond, eta=1.0, x_T=x_T, verbose=False, log_every_t=1, mask=mk, x0=x0, temperature=0.7)
            out[f"ddim_{tag}_x"] = x
            out[f"ddim_{tag}_x_inter"] = torch.stack(inter["x_inter"][1:])
        ctr.update(i=0, q=0)
        pred_x0, inter = m.p_sample_loop(cond, (B, 3, h, w), return_intermediates=True, x_T=x_T, verbose=False, start_T=6, log_every_t=1, mask=mask1, x0=x0)
        out.update(obs_pred_x0=pred_x0, obs_x_inter=torch.stack(inter["x_inter"][1:]))
        ctr.update(i=0, q=0)
        img, inter = refddpm.LatentDiffusion.p_sample_loop(m, cond, (B, 3, h, w), return_intermediates=True, x_T=x_T, verbose=False, start_T=6, log_every_t=1,
                                                           mask=mask3, x0=x0)
        out.update(ldm_x=img, ldm_x_inter=torch.stack(inter[1:]))
        ctr.update(i=0, q=0)
        xs, img = [], x_T
        for t in (5, 4, 0):
            img = refddpm.LatentDiffusion.p_sample(m, img, cond, torch.full((B,), t, dtype=torch.long), clip_denoised=False, temperature=0.7)
            xs.append(img)
        out.update(temp_x=torch.stack(xs), temp_t=np.asarray([5, 4, 0]))
    finally:
        refddim.noise_like, refddpm.noise_like = o1, o2
        del m.q_sample
    save("sampler_masks", **out)


def make_sampler_guidance():
    """Classifier-free guidance and noise_dropout, from the reference's own DDIM loop on the tiny ObsNet (B = 3 @16x16, eta = 1, 50 steps):
    DDIMSampler.sample(unconditional_guidance_scale=3.0, unconditional_conditioning=<another refmap batch>) (ddim.py:225-232) and
    DDIMSampler.sample(noise_dropout=0.3) with F.dropout's keep masks injected (ddim.py:256-257); LatentDiffusion.p_sample(noise_dropout=0.3) at three
    timesteps (ddpm.py:1158-1159)."""
    import ldm.models.diffusion.ddim as refddim
    import ldm.models.diffusion.ddpm as refddpm

    m, DDIM = tiny_obsnet()
    B, h, w = 3, 16, 16
    g = gen(83)
    cond = synth.synth_refmaps(B, h, w, 98) * 2 - 1
    ucond = synth.synth_refmaps(B, h, w, 97) * 2 - 1
    x_T = torch.randn((B, 3, h, w), generator=g)
    noise = torch.randn((50, B, 3, h, w), generator=g)
    p = 0.3
    keep = (torch.rand((50, B, 3, h, w), generator=g) >= p).float()
    ctr = {"i": 0, "d": 0}

    def noise_like(shape, device, repeat=False):
        out = noise[ctr["i"]]
        ctr["i"] += 1
        return out

    F = torch.nn.functional
    orig_dropout = F.dropout

    def dropout(x, p=0.5, training=True, inplace=False):
        if not training:  # (the network's own nn.Dropout layers in eval mode)
            return x
        out = x * keep[ctr["d"]] / (1.0 - p)
        ctr["d"] += 1
        return out

    o1, o2 = refddim.noise_like, refddpm.noise_like
    refddim.noise_like = noise_like
    refddpm.noise_like = noise_like
    F.dropout = dropout
    out = dict(cond=cond, ucond=ucond, x_T=x_T, noise=noise, keep=keep, p=p, scale=3.0)
    try:
        ctr.update(i=0, d=0)
        x, inter = DDIM(m).sample(50, B, (3, h, w), cond, eta=1.0, x_T=x_T, verbose=False, log_every_t=1, unconditional_guidance_scale=3.0,
                                  unconditional_conditioning=ucond)
        out.update(cfg_x=x, cfg_first=inter["x_inter"][1])
        ctr.update(i=0, d=0)
        x, inter = DDIM(m).sample(50, B, (3, h, w), cond, eta=1.0, x_T=x_T, verbose=False, log_every_t=1, noise_dropout=p)
        out.update(drop_x=x, drop_first=inter["x_inter"][1], drop_calls=ctr["d"])
        ctr.update(i=0, d=0)
        xs, img = [], x_T
        for t in (5, 4, 0):
            img = refddpm.LatentDiffusion.p_sample(m, img, cond, torch.full((B,), t, dtype=torch.long), clip_denoised=False, noise_dropout=p)
            xs.append(img)
        out.update(ddpm_drop_x=torch.stack(xs), ddpm_drop_t=np.asarray([5, 4, 0]))
    finally:
        refddim.noise_like, refddpm.noise_like = o1, o2
        F.dropout = orig_dropout
    print(f"  guidance / dropout: F.dropout calls in the 50-step chain {out['drop_calls']}")
    save("sampler_guidance", **out)


def make_ddim_variants():
    """DDIMSampler.ddim_sampling called directly with the schedule argument DDIMSampler.sample never passes (ddim.py:156-158): `timesteps` (a subset of
    the 50-step DDIM schedule: timesteps = 30 -> its first 29 entries), eta = 1.  Tiny ObsNet, B = 3 @16x16, draws injected."""
    import ldm.models.diffusion.ddim as refddim

    m, DDIM = tiny_obsnet()
    B, h, w = 3, 16, 16
    g = gen(61)
    cond = synth.synth_refmaps(B, h, w, 98) * 2 - 1
    x_T = torch.randn((B, 3, h, w), generator=g)
    noise = torch.randn((50, B, 3, h, w), generator=g)
    ctr = {"i": 0}

    def noise_like(shape, device, repeat=False):
        out = noise[ctr["i"]]
        ctr["i"] += 1
        return out

    o1 = refddim.noise_like
    refddim.noise_like = noise_like
    out = dict(cond=cond, x_T=x_T, noise=noise)
    try:
        s = DDIM(m)
        s.make_schedule(ddim_num_steps=50, ddim_eta=1.0, verbose=False)
        ctr["i"] = 0
        x, inter = s.ddim_sampling(cond, (B, 3, h, w), x_T=x_T, timesteps=30, log_every_t=1, verbose=False)
        out.update(subset_x=x, subset_n=len(inter["x_inter"]) - 1, subset_first=inter["x_inter"][1])
        try:  # ddim_use_original_steps cannot run in the reference (ddim.py:242 reads a sampler buffer from the model): recorded as such
            s.ddim_sampling(cond, (B, 3, h, w), x_T=x_T, ddim_use_original_steps=True, timesteps=12, log_every_t=1, verbose=False)
            out["orig_runs"] = 1
        except AttributeError:
            out["orig_runs"] = 0
    finally:
        refddim.noise_like = o1
    print(f"  ddim variants: subset steps {out['subset_n']}, original-steps path runs in the reference: {bool(out['orig_runs'])}")
    save("ddim_variants", **out)


def make_full_samplers():
    """Full-width ObsNet (configs/obsnet/eval_obsnet.yaml, 147.6 M parameters by the synth rule) at the metric shape 3x128x256:
    the first two DDIM steps (eta = 1; ddim.py:206-259 p_sample_ddim at index 49, 48) and the first two ancestral steps
    (ddpm.py:1120-1167 p_sample at t = 999, 998) from a given x_T with the per-step draws injected.  Only inputs that cannot be
    regenerated (none: everything is seeded) and the reference's outputs are stored."""
    import ldm.models.diffusion.ddim as refddim
    import ldm.models.diffusion.ddpm as refddpm

    _, OBS, DDIM, _ = rh.ref_classes()
    cfg = rh.load_yaml_params("configs/obsnet/eval_obsnet.yaml")["model"]["params"]
    cfg.pop("ckpt_path")
    cfg.update(use_ema=False)
    m = OBS(**cfg).eval()
    cs = load_rule(m.model.diffusion_model, synth.SEED_OBSNET)
    B, h, w = 1, 128, 256
    g = gen(61)
    cond = synth.synth_refmaps(B, h, w, synth.SEED_INPUT) * 2 - 1
    x_T = torch.randn((B, 3, h, w), generator=g)
    noise = torch.randn((2, B, 3, h, w), generator=g)
    ctr = {"i": 0}

    def noise_like(shape, device, repeat=False):
        out = noise[ctr["i"]]
        ctr["i"] += 1
        assert tuple(out.shape) == tuple(shape)
        return out

    o1, o2 = refddim.noise_like, refddpm.noise_like
    refddim.noise_like = noise_like
    refddpm.noise_like = noise_like
    try:
        s = DDIM(m)
        s.make_schedule(ddim_num_steps=50, ddim_eta=1.0, verbose=False)
        x = x_T
        xs, p0 = [], []
        t0 = time.time()
        with torch.no_grad():
            for index in (49, 48):
                ts = torch.full((B,), int(s.ddim_timesteps[index]), dtype=torch.long)
                x, pred = s.p_sample_ddim(x, cond, ts, index=index)
                xs.append(x)
                p0.append(pred)
        ctr["i"] = 0
        y = x_T
        ys, q0 = [], []
        with torch.no_grad():
            for t in (999, 998):
                y, x0 = m.p_sample(y, cond, torch.full((B,), t, dtype=torch.long), clip_denoised=m.clip_denoised, return_x0=True)
                ys.append(y)
                q0.append(x0)
        print(f"  full-width sampler steps: {time.time() - t0:.1f}s  ddim |x| {float(xs[-1].abs().max()):.3e}  ddpm |x| {float(ys[-1].abs().max()):.3e}")
    finally:
        refddim.noise_like, refddpm.noise_like = o1, o2
    save("full_obsnet_sampler_steps", ddim_x=torch.stack(xs), ddim_pred_x0=torch.stack(p0), ddpm_x=torch.stack(ys), ddpm_pred_x0=torch.stack(q0),
         seed=synth.SEED_OBSNET, wsum=cs, gen_seed=61, cond_sum=synth.checksum(cond), xT_sum=synth.checksum(x_T))


def make_transforms():
    """The elementwise maps either side of the samplers and the envmap warp / tone map after them, from the reference's own
    functions: BaseDataset.transform / rescale for both shipped transform_func strings (dataset/basedataset.py:29-112),
    DRMNet.get_input_for_predict's exposure scaling (models/drmnet.py:1017-1034), mirmap2envmap (utils/transform.py:106-144),
    DRMNet.r0toenvmap (models/drmnet.py:931-941) and hdr2ldr (utils/tonemap.py:4-9).  Inputs are seeded and stored (small)."""
    rh.install_stubs()
    from dataset.basedataset import BaseDataset
    from utils.tonemap import hdr2ldr
    from utils.transform import mirmap2envmap

    g = gen(71)
    out = {}
    # --- "log" (DRMNet dataset): HDR radiance -> network space and back, with the clamp of the shipped config and without
    hdr = torch.exp(torch.randn((3, 3, 16, 24), generator=g) * 1.5 - 2.0)
    hdr[0, 0, 0, :4] = 0.0
    net = torch.randn((3, 3, 16, 24), generator=g) * 1.5
    net[1, 1, 2, :3] = torch.tensor([25.0, 21.5, -30.0])  # above / below the clamp_before_exp = 20 of the shipped config
    ds = BaseDataset(size=16, transform_func="log", clamp_before_exp=20)
    out.update(log_x=hdr, log_y=ds.transform(hdr), log_net=net, log_rescaled=ds.rescale(net))
    out["log_rescaled_noclamp"] = BaseDataset(size=16, transform_func="log", clamp_before_exp=0.0).rescale(net.clamp(max=30))
    # --- ObsNet's string: lower bound, per-image masked log normalisation, [0,1] -> [-1,1]; resize is a no-op at the stored size
    ds2 = BaseDataset(size=16, transform_func="resize_0p1tom1p1_normalizedLogarithmic_lowerbound1e-6", clamp_before_exp=20)
    x2 = torch.exp(torch.randn((3, 3, 16, 16), generator=g) * 2.0 - 3.0)
    x2[0, :, :2] = 0.0  # below the lower bound
    m2 = (torch.rand((3, 1, 16, 16), generator=g) > 0.4).float()
    y2 = ds2.transform(x2, dynamic_normalize=True, mask=m2)
    lo, hi = ds2.Logarithmic_params
    net2 = torch.rand((3, 3, 16, 16), generator=g) * 2.4 - 1.2
    out.update(nl_x=x2, nl_mask=m2, nl_y=y2, nl_lo=lo, nl_hi=hi, nl_net=net2, nl_rescaled=ds2.rescale(net2))
    x3 = x2[0]  # 3-D input: one image, statistics over all three dims
    y3 = ds2.transform(x3, dynamic_normalize=True, mask=m2[0])
    out.update(nl3_y=y3, nl3_lo=ds2.Logarithmic_params[0], nl3_hi=ds2.Logarithmic_params[1])
    # --- exposure normalisation of get_input_for_predict
    drm = tiny_drmnet(gamma=0.9, epsilon=1.0, max_timesteps=3)
    drm.ds = BaseDataset(size=16, transform_func="log", clamp_before_exp=20)
    lrk = torch.exp(torch.randn((4, 3, 16, 16), generator=g) * 1.2 - 1.0)
    lrk[2, :, 5:9, 5:9] = 0.0  # unlit pixels are left out of the geometric mean
    LrK, _, illc, refc, tag = drm.get_input_for_predict({"LrK": lrk, "tag": ["a", "b", "c", "d"]}, bs=3)
    out.update(gi_x=lrk, gi_LrK=LrK, gi_scale=drm.normalizing_scale, gi_scaler=drm.refmap_input_scaler)
    # --- mirror map -> envmap, r0toenvmap, tone map
    mir = torch.exp(torch.randn((2, 3, 16, 16), generator=g) * 0.8)
    out.update(mir=mir, env=mirmap2envmap(mir, (16, 32)), env_log=mirmap2envmap(mir, (16, 32), log_scale_interpolation=True),
               env_odd=mirmap2envmap(mir[:1], (10, 28)))
    mir128 = torch.exp(torch.randn((1, 3, 128, 128), generator=gen(72)) * 0.5)
    out.update(env128=mirmap2envmap(mir128, (128, 256)), mir128_seed=72)
    basis = torch.rand((3, 16, 16), generator=g) + 0.5
    drm.basis_r0 = basis
    out.update(basis=basis, r0env=drm.r0toenvmap(mir, (16, 32)))
    envh = out["env"][0].permute(1, 2, 0).clone().numpy()
    envh[:2] *= 1e-6  # unlit rows (L <= 5e-5) are left out of the mean
    msk = (torch.rand((16, 32), generator=g) > 0.3).numpy()
    out.update(ldr_x=envh, ldr=hdr2ldr(envh), ldr_mask=msk, ldr_masked=hdr2ldr(envh, msk), ldr_a=hdr2ldr(envh, alpha=0.3, gamma=1.8))
    save("transforms", **out)


def make_refmap():
    """refmap_mask_make (utils/img2refmap.py:6-37) and the mask erosion of scripts/estimate.py:43-50 on the reference's own
    data/sample inputs.  The three sample files are DATA and are copied next to the fixtures (tests/golden/sample/); the EXR
    is decoded with drmnet_amd/file_io.py (no OpenCV in this image) -- the decoded array is what the reference function gets."""
    import shutil

    rh.install_stubs()
    from utils.img2refmap import refmap_mask_make  # the reference function

    sys.path.insert(0, ROOT)
    from drmnet_amd import file_io

    src = "/root/reference/data/sample"
    dst = os.path.join(GOLD, "sample")
    os.makedirs(dst, exist_ok=True)
    for f in ("image.exr", "normal.npy", "mask.png"):
        shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
        os.chmod(os.path.join(dst, f), 0o644)
    img = file_io.load_exr(os.path.join(dst, "image.exr"), as_torch=True)
    normal = torch.from_numpy(np.load(os.path.join(dst, "normal.npy")))
    normal_mask = torch.linalg.norm(normal, dim=-1) > 0.5
    input_mask = file_io.load_png(os.path.join(dst, "mask.png"), as_torch=True)
    mask0 = torch.logical_and(input_mask, normal_mask)  # estimate.py:128-136
    # estimate.py:43-50, verbatim semantics on CPU
    k = 5
    inv_mask = ~mask0
    kernel = torch.stack(torch.meshgrid(*torch.arange(k).expand(2, -1), indexing="ij"))
    kernel = kernel + 0.5
    kernel = torch.linalg.norm(kernel - k / 2, axis=0) <= k / 2
    kernel = kernel[None, None].float()
    inv_mask = torch.nn.functional.conv2d(inv_mask[None, None].float(), kernel, padding="same").bool()[0, 0]
    mask = torch.logical_and(mask0, ~inv_mask)
    cases = {"128": (128, np.pi / 128 / 2), "16": (16, np.pi / 16 / 2), "32wide": (32, np.pi / 40)}
    out = {"mask0": mask0.numpy(), "mask_eroded": mask.numpy()}
    for tag, (res, thr) in cases.items():
        refmap, refmask = refmap_mask_make(img[mask], normal[mask], res=res, angle_threshold=thr)
        out[f"refmap_{tag}"] = refmap.numpy()
        out[f"refmask_{tag}"] = refmask.numpy()
        out[f"res_{tag}"] = np.int64(res)
        out[f"thr_{tag}"] = np.float64(thr)
        print(f"  refmap {tag}: {int(refmask.sum())} texels of {res * res} set, n = {int(mask.sum())} pixels")
    save("refmap_sample", **out)


def make_estimate_chain(full=False):
    """scripts/estimate.py:29-107 + :138-145 end to end on data/sample with 16x16 tiny networks (weights by rule, no
    checkpoint): erosion -> refmap_mask_make -> ObsNet cond transform -> DDIM-50 (eta = 1, noise injected) -> rescale ->
    DRMNet input transform -> DRMNet reverse loop (noise injected) -> rescale / clip / un-normalise -> r0toenvmap -> hdr2ldr.
    estimate() itself hard-codes .cuda() / cuda.synchronize(); its statements are executed here in the same order on CPU."""
    import ldm.models.diffusion.ddim as refddim
    import models.drmnet as refdrm
    from dataset.basedataset import BaseDataset
    from utils.img2refmap import refmap_mask_make
    from utils.tonemap import hdr2ldr

    sys.path.insert(0, ROOT)
    from drmnet_amd import file_io

    res = 128 if full else 16
    # Random-weight networks are not denoisers: with the production noise schedule (alpha_bar_T ~ 1e-20) the DDIM iterate
    # explodes to the rescale clamp (1e20) and the rest of the chain is inf/NaN.  The chain fixture therefore uses a gentle
    # schedule and a damped ObsNet head so every stage stays finite and O(1); the schedules proper are pinned elsewhere.
    obs_sched = dict(linear_start=1e-5, linear_end=2e-4)
    obs_out_scale = 0.05
    _, OBS, DDIM, _ = rh.ref_classes()
    ocfg_m = rh.load_yaml_params("configs/obsnet/eval_obsnet.yaml")["model"]["params"]
    ocfg_m.pop("ckpt_path")
    if not full:
        ocfg_m["unet_config"] = {"target": ocfg_m["unet_config"]["target"], "params": dict(ou.TINY_UNET_CFG)}
    ocfg_m.update(image_size=res, use_ema=False, **obs_sched)
    obs = OBS(**ocfg_m).eval()
    synth.load_synth(obs.model.diffusion_model, synth.SEED_OBSNET if full else 21)
    with torch.no_grad():
        obs.model.diffusion_model.out[2].weight.mul_(obs_out_scale)
        obs.model.diffusion_model.out[2].bias.mul_(obs_out_scale)
    # full width (the shipped eval configs, 128x128 refmaps): T and epsilon chosen so that the sample converges mid-loop
    T, epsilon = (FULL_CHAIN_T, FULL_CHAIN_EPS) if full else (17, 1.0)
    drm = full_drmnet(gamma=0.9, epsilon=epsilon, max_timesteps=T) if full else tiny_drmnet(gamma=0.9, epsilon=1.0, max_timesteps=17)
    head = drm.refnet_model.diffusion_model.out[3]
    head_bias = torch.tensor([0.95, 0.9, 0.97, 0.92, 0.05, 0.9])
    ill_out_scale = 0.02 if full else 0.05  # damped IllNet head: the residual updates stay small, the map stays inside the rescale clamp
    head_w_scale = 3.0 if full else 10.0
    with torch.no_grad():
        head.weight.mul_(head_w_scale)
        head.bias.copy_(head_bias)
        drm.illnet_model.diffusion_model.out[2].weight.mul_(ill_out_scale)
        drm.illnet_model.diffusion_model.out[2].bias.mul_(ill_out_scale)
    ocfg = rh.load_yaml_params("configs/obsnet/eval_obsnet.yaml")["data"]["params"]["predict"]["params"]
    dcfg = rh.load_yaml_params("configs/drmnet/eval_drmnet.yaml")["data"]["params"]["predict"]["params"]
    obs.ds = BaseDataset(**dict(ocfg, size=res))
    drm.ds = BaseDataset(**dict(dcfg, size=res))

    d = os.path.join(GOLD, "sample")
    input_img = file_io.load_exr(os.path.join(d, "image.exr"), as_torch=True)
    input_normal = torch.from_numpy(np.load(os.path.join(d, "normal.npy")))
    normal_mask = torch.linalg.norm(input_normal, dim=-1) > 0.5
    mask = torch.logical_and(file_io.load_png(os.path.join(d, "mask.png"), as_torch=True), normal_mask)
    k = 5
    inv_mask = ~mask
    kernel = torch.stack(torch.meshgrid(*torch.arange(k).expand(2, -1), indexing="ij")) + 0.5
    kernel = (torch.linalg.norm(kernel - k / 2, axis=0) <= k / 2)[None, None].float()
    inv_mask = torch.nn.functional.conv2d(inv_mask[None, None].float(), kernel, padding="same").bool()[0, 0]
    mask = torch.logical_and(mask, ~inv_mask)
    refmap_est, refmask = refmap_mask_make(input_img[mask], input_normal[mask], res=res, angle_threshold=np.pi / 128 / 2)  # (the literal of scripts/estimate.py:57)
    batch = {"tag": ["sample"], "raw_refmap": refmap_est.permute(2, 0, 1)[None], "raw_refmask": refmask[None]}
    if full:
        torch.manual_seed(20261003)  # get_cond_for_predict fills the unobserved texels from torch's GLOBAL generator (models/obsnet.py:698): seeded, so that
        #                              re-running this step reproduces the committed fixture bit for bit (the tiny chain fixture stores its draw instead)
    with torch.no_grad():
        c, _, _ = obs.get_cond_for_predict(batch)
    g = gen(77)
    x_T = torch.randn((1, 3, res, res), generator=g)
    noise = torch.randn((50, 1, 3, res, res), generator=g)
    ctr = {"i": 0}

    def noise_like(shape, device, repeat=False):
        out = noise[ctr["i"]]
        ctr["i"] += 1
        return out

    o1 = refddim.noise_like
    refddim.noise_like = noise_like
    try:
        with torch.no_grad():
            samples, _ = DDIM(obs).sample(obs.ddim_steps, 1, (obs.channels, obs.image_size, obs.image_size), c, verbose=False,
                                          log_every_t=max(obs.log_every_t * obs.ddim_steps // obs.num_timesteps, 1), eta=obs.ddim_eta, x_T=x_T)
    finally:
        refddim.noise_like = o1
    inpaint_sample = obs.ds.rescale(obs.decode_first_stage(samples))[0]
    batch = {"tag": ["sample"], "LrK": inpaint_sample[None]}
    LrK, _, illnet_c, refnet_c, _ = drm.get_input_for_predict(batch)
    if full:  # centre the (random-weight) RefNet head 0.73 away from z0 on this input, as make_drmnet_loop(full=True) does
        with torch.no_grad():
            wpart = drm.apply_model(drm.refnet_model, LrK, 0, refnet_c) - head.bias
            head_bias = drm.z0 + torch.tensor([-0.3, -0.3, -0.3, -0.3, 0.3, -0.3]) - wpart[0]
            head.bias.copy_(head_bias)
    noise0 = torch.randn(LrK.shape, generator=g)
    step_noise = torch.randn((T,) + tuple(LrK.shape), generator=g)
    state = {"call": 0, "step": 0}
    orig = torch.randn_like

    def randn_like(t, **kw):
        if state["call"] == 0:
            state["call"] += 1
            return noise0.clone()
        out = step_noise[state["step"]]
        state["step"] += 1
        return out

    refdrm.torch.randn_like = randn_like
    orig_check = drm.check_convergence

    def check(zk):
        print("   |zk - z0| =", (zk - drm.z0).norm(dim=-1).tolist())
        return orig_check(zk)

    drm.check_convergence = check
    try:
        with torch.no_grad():
            samples2, zK_est, K = drm.p_sample_loop(LrK, illnet_c, refnet_c, verbose=False)
    finally:
        refdrm.torch.randn_like = orig
    Lr0_sample = drm.ds.rescale(drm.decode_first_stage(samples2))[0].clip(0)
    if drm.refmap_input_scaler is not None:
        Lr0_sample = Lr0_sample / drm.normalizing_scale[0]
    envmap = drm.r0toenvmap(Lr0_sample[None], (drm.image_size, drm.image_size * 2))[0]
    ldr = hdr2ldr(envmap.cpu().numpy())
    print(f"  chain: refmask {int(refmask.sum())}/{res * res}, K = {K.tolist()}, zK = {zK_est[0].tolist()}, env {tuple(envmap.shape)}")
    common = dict(K=K, zK=zK_est, Lr0=Lr0_sample, envmap=envmap, ldr=ldr, head_bias=head_bias, head_w_scale=head_w_scale, gamma=0.9, epsilon=epsilon,
                  max_timesteps=T, delta=0.025, obs_out_scale=obs_out_scale, ill_out_scale=ill_out_scale, obs_linear_start=obs_sched["linear_start"],
                  obs_linear_end=obs_sched["linear_end"])
    if full:
        # the draws are regenerable (torch CPU generator, seed 77, in the order x_T, noise[50], noise0, step_noise[T]; the cond's
        # noise fill is stored: it comes from torch's global generator inside get_cond_for_predict): only outputs are stored
        save("estimate_chain_full", refmask=refmask, cond=c, inpaint=inpaint_sample, LrK=LrK, gen_seed=77, **common)
    else:
        save("estimate_chain", refmap=refmap_est, refmask=refmask, cond=c, x_T=x_T, noise=noise, inpaint=inpaint_sample, LrK=LrK, noise0=noise0,
             step_noise=step_noise, **common)
    for k_, v_ in (("inpaint", inpaint_sample), ("Lr0", Lr0_sample), ("envmap", envmap)):
        assert torch.isfinite(v_).all(), k_
        print(f"  {k_}: min {float(v_.min()):.3e} max {float(v_.max()):.3e}")



# ----------------------------------------------------------------------------- round 4: long chains, weight stress, EMA checkpoints, resize

LONG_DRM = dict(T=150, gamma=0.97, epsilon=0.012, head_w_scale=25.0, ill_out_scale=0.02, head_off=0.05, B=6, gen_seed=43, input_seed=99)
LONG_OBS = dict(T=1000, linear_start=1e-4, linear_end=2e-3, B=2, gen_seed=53, input_seed=98)


def make_long_chains():
    """The two loop lengths the shipped configs run, end to end through the reference's own loops on 16x16 tiny networks:
    max_timesteps = 150 of DRMNet's reverse process (models/drmnet.py:782-847, configs/drmnet/eval_drmnet.yaml) with rows that leave
    after 3 ... 148 steps and one that never converges, and the whole T = 1000 ancestral chain of ObsNetDiffusion.p_sample_loop
    (models/obsnet.py:500-564 over ldm/models/diffusion/ddpm.py:1120-1167).  Inputs and draws are regenerable from the stored seeds
    (torch CPU generator); only the reference's outputs are stored.  Random-weight networks are not denoisers: the IllNet head is
    damped (values stored) and the ancestral chain runs a milder beta range than the shipped one (whose 1 / sqrt(abar_T) = 1e7
    growth needs a trained eps), with the head undamped."""
    import ldm.models.diffusion.ddpm as refddpm
    import models.drmnet as refdrm

    c = LONG_DRM
    T, B = c["T"], c["B"]
    m = tiny_drmnet(gamma=c["gamma"], epsilon=c["epsilon"], max_timesteps=T)
    head = m.refnet_model.diffusion_model.out[3]
    LrK = synth.synth_refmaps(B, 16, 16, c["input_seed"])
    with torch.no_grad():
        o = m.illnet_model.diffusion_model.out[2]
        o.weight.mul_(c["ill_out_scale"])
        o.bias.mul_(c["ill_out_scale"])
        wpart = m.apply_model(m.refnet_model, LrK, 0, [LrK]) - head.bias
        head_bias = m.z0 + c["head_off"] * torch.tensor([-1.0, -1, -1, -1, 1, -1]) - c["head_w_scale"] * wpart.mean(0)
        head.weight.mul_(c["head_w_scale"])
        head.bias.copy_(head_bias)
    g = gen(c["gen_seed"])
    noise0 = torch.randn(LrK.shape, generator=g)
    step_noise = torch.randn((T,) + tuple(LrK.shape), generator=g)
    state = {"call": 0, "active": torch.ones(B, dtype=torch.bool), "nc_idx": None, "step": 0, "margin": []}
    orig_randn_like, orig_check = torch.randn_like, m.check_convergence

    def check(zk):
        conv = orig_check(zk)
        idx = torch.where(state["active"])[0]
        state["nc_idx"] = idx[~conv]
        state["active"][idx[conv]] = False
        state["margin"].append(float(((zk - m.z0).norm(dim=-1) / m.epsilon - 1).abs().min()))
        return conv

    def randn_like(t, **kw):
        if state["call"] == 0:
            state["call"] += 1
            return noise0.clone()
        out = step_noise[state["step"]][state["nc_idx"]]
        assert out.shape == t.shape, (out.shape, t.shape)
        state["step"] += 1
        return out

    m.check_convergence = check
    refdrm.torch.randn_like = randn_like
    try:
        Lr0, zK, K, inter = m.p_sample_loop(LrK, [LrK], [LrK], return_intermediates=True, verbose=False, log_every_k=10)
    finally:
        refdrm.torch.randn_like = orig_randn_like
    print(f"  drmnet 150-step loop: K = {K.tolist()}, zK nan rows = {torch.isnan(zK).any(dim=1).tolist()}, |Lr0| max {float(Lr0.abs().max()):.2f}, "
          f"closest |zk - z0| / epsilon to 1 over all steps: {min(state['margin']):.3e}")
    save("drmnet_loop_150", Lr0=Lr0, zK=zK, K=K, z0=m.z0, head_bias=head_bias, Lrk_steps=torch.stack(inter["Lrk_inter"][1:]),
         zk_steps=torch.stack(inter["zk_inter"]), LrK_sum=synth.checksum(LrK), noise_sum=synth.checksum(step_noise), delta=0.025,
         **{k: v for k, v in c.items()})

    c = LONG_OBS
    _, OBS, _, _ = rh.ref_classes()
    cfg = rh.load_yaml_params("configs/obsnet/eval_obsnet.yaml")["model"]["params"]
    cfg.pop("ckpt_path")
    cfg["unet_config"] = {"target": cfg["unet_config"]["target"], "params": dict(ou.TINY_UNET_CFG)}
    cfg.update(image_size=16, use_ema=False, linear_start=c["linear_start"], linear_end=c["linear_end"])
    assert cfg["timesteps"] == c["T"]
    obs = OBS(**cfg).eval()
    synth.load_synth(obs.model.diffusion_model, 21)
    B = c["B"]
    g = gen(c["gen_seed"])
    cond = synth.synth_refmaps(B, 16, 16, c["input_seed"]) * 2 - 1
    x_T = torch.randn((B, 3, 16, 16), generator=g)
    noise = torch.randn((c["T"], B, 3, 16, 16), generator=g)
    ctr = {"i": 0}

    def noise_like(shape, device, repeat=False):
        out = noise[ctr["i"]]
        ctr["i"] += 1
        assert tuple(out.shape) == tuple(shape)
        return out

    o2 = refddpm.noise_like
    refddpm.noise_like = noise_like
    try:
        pred_x0, inter = obs.p_sample_loop(cond, (B, 3, 16, 16), return_intermediates=True, x_T=x_T, verbose=False, log_every_t=100)
    finally:
        refddpm.noise_like = o2
    assert ctr["i"] == c["T"]
    print(f"  ancestral 1000-step chain: |x| max along the chain {[round(float(x.abs().max()), 2) for x in inter['x_inter']]}, abar_T {float(obs.alphas_cumprod[-1]):.3f}")
    save("ddpm_trace_1000", x=inter["x_inter"][-1], pred_x0=pred_x0, x_inter=torch.stack(inter["x_inter"][1:]), cond_sum=synth.checksum(cond),
         noise_sum=synth.checksum(noise), **{k: v for k, v in c.items()})


def make_stress():
    """Weight stress for the split arithmetic modes: the three shipped networks at full width with heavy-tailed
    weights (Student-t, 4 degrees of freedom) and GroupNorm gains x 3 and x 10 (drmnet_amd/synth.py rule="stress:<gain>"), forwards of
    the reference's own modules at 64x64 (and the metric shape's aspect at 32x64)."""
    _, _, _, oa = rh.ref_classes()
    for gain, (name, cfg, cls, seed) in ((g_, c_) for g_ in (3, 10) for c_ in (
            ("illnet", ou.ILLNET_CFG, oa.UNetModel, synth.SEED_ILLNET), ("refnet", ou.REFNET_CFG, oa.EncoderUNetModel, synth.SEED_REFNET),
            ("obsnet", ou.OBSNET_CFG, oa.UNetModel, synth.SEED_OBSNET))):
        m = cls(**cfg).eval()
        synth.load_synth(m, seed + 100, rule=f"stress:{gain}")
        cs = synth.checksum(torch.cat([v.flatten() for v in m.state_dict().values()]))
        arrs = {}
        for n, h, w in ((1, 64, 64), (2, 32, 64)):
            xc, t_emb = full_inputs(n, h, w)
            t = torch.tensor([7, 981][:n], dtype=torch.long)
            with torch.no_grad():
                out = m(xc, t_emb=t_emb) if name == "illnet" else m(xc, t)
                # the same arithmetic in fp64 (the oracle, which every other fixture pins to the reference, carried in double): under these
                # weights the fp32 forward itself carries rounding noise of 1e-5 .. 1e-4 -- two fp32 evaluations with different summation
                # orders differ by that much -- so the fixture also holds the fp64 answer, the yardstick for "how far is fp32 from the
                # truth" next to "how far is the HIP path".  (The reference's modules cast to fp32 inside GroupNorm32 and cannot run in fp64.)
                kind = "encoder" if name == "refnet" else "unet"
                P64 = {k: v.double() for k, v in m.state_dict().items()}
                topo = ou.build_topology(cfg, kind)
                with ou.working_dtype(torch.float64):
                    out64 = (ou.unet_forward(P64, topo, xc.double(), t_emb=t_emb.double()) if name == "illnet" else
                             ou.encoder_forward(P64, topo, xc.double(), t) if kind == "encoder" else ou.unet_forward(P64, topo, xc.double(), timesteps=t))
            e = float((out.double() - out64).norm() / out64.norm())
            print(f"  stress x{gain} {name} {n}x{h}x{w}: out std {out.std():.4f} absmax {out.abs().max():.3f}  fp32 vs fp64 reference: {e:.2e}")
            arrs[f"out_{n}x{h}x{w}"] = out
            arrs[f"out64_{n}x{h}x{w}"] = out64
        save(f"stress{gain}_{name}", seed=seed + 100, gain=gain, wsum=cs, t=torch.tensor([7, 981]), **arrs)


def make_ema_ckpt():
    """a15: checkpoints WRITTEN BY THE REFERENCE and sampled under the reference's ema_scope.  Both models are built by the
    reference's own classes with use_ema=True (16x16 tiny networks), their live weights set by the synth rule, the LitEma shadows
    (ldm/modules/ema.py:5-44) initialised from them and then moved by the reference's own decay update -- LitEma.forward called
    three times on live weights perturbed in between, so decay follows num_updates (ema.py:24-44) -- and the whole state_dict
    (live parameters, dot-less shadow buffers, decay, num_updates, schedule buffers) is written with torch.save({"state_dict": ...})
    to tests/golden/*.ckpt (data: tensors only).  Recorded: p_sample_loop / sample_log outputs of the reference inside and outside
    `with model.ema_scope():` (models/drmnet.py:242-258, ldm/models/diffusion/ddpm.py:189-202)."""
    import ldm.models.diffusion.ddim as refddim
    import models.drmnet as refdrm

    DRM, OBS, DDIM, _ = rh.ref_classes()

    def drift(module, ema, seed):
        """three EMA updates of the reference, the live weights moving in between"""
        g = gen(seed)
        for _ in range(3):
            with torch.no_grad():
                for p in module.parameters():
                    p.add_(0.05 * p.abs().mean() * torch.randn(p.shape, generator=g))
            ema(module)

    # ---- DRMNet
    cfg = rh.load_yaml_params("configs/drmnet/eval_drmnet.yaml")["model"]["params"]
    cfg.pop("ckpt_path")
    cfg["illnet_config"] = {"target": cfg["illnet_config"]["target"], "params": dict(ou.TINY_UNET_CFG)}
    cfg["refnet_config"] = {"target": cfg["refnet_config"]["target"], "params": dict(ou.TINY_ENC_CFG)}
    T = 6
    cfg.update(image_size=16, gamma=0.9, epsilon=1e-3, max_timesteps=T, delta=0.025, use_ema=True)
    m = DRM(**cfg).eval()
    synth.load_synth(m.illnet_model.diffusion_model, 21)
    synth.load_synth(m.refnet_model.diffusion_model, 22)
    m.illnet_model.z_emb_layer.load_state_dict(synth.synth_state_dict([(k, tuple(v.shape)) for k, v in m.illnet_model.z_emb_layer.state_dict().items()], synth.SEED_ZEMB))
    for wrapper, ema in ((m.illnet_model, m.illnet_model_ema), (m.refnet_model, m.refnet_model_ema)):
        for name, p in wrapper.named_parameters():  # shadows start from the (synth) live weights, as at the start of training
            getattr(ema, ema.m_name2s_name[name]).copy_(p.detach())
    drift(m.illnet_model, m.illnet_model_ema, 301)
    drift(m.refnet_model, m.refnet_model_ema, 302)
    path = os.path.join(GOLD, "drmnet_tiny_ema.ckpt")
    torch.save({"state_dict": m.state_dict()}, path)
    print(f"  wrote drmnet_tiny_ema.ckpt ({os.path.getsize(path) / 1024:.0f} KiB), num_updates {int(m.illnet_model_ema.num_updates)}")
    B = 3
    LrK = synth.synth_refmaps(B, 16, 32, 5)
    g = gen(9)
    noise0 = torch.randn(LrK.shape, generator=g)
    step_noise = torch.randn((T,) + tuple(LrK.shape), generator=g)
    orig_randn_like = torch.randn_like

    def run_loop():
        state = {"call": 0, "step": 0}

        def randn_like(t, **kw):  # epsilon = 1e-3: no row converges within T steps, every draw is the full batch
            if state["call"] == 0:
                state["call"] += 1
                return noise0.clone()
            out = step_noise[state["step"]]
            assert out.shape == t.shape
            state["step"] += 1
            return out

        refdrm.torch.randn_like = randn_like
        try:
            return m.p_sample_loop(LrK, [LrK], [LrK], verbose=False)
        finally:
            refdrm.torch.randn_like = orig_randn_like

    live = run_loop()
    with m.ema_scope("golden"):
        inside = run_loop()
        x = torch.cat([LrK, LrK], 1)
        te = torch.randn((B, 32), generator=gen(10))
        with torch.no_grad():
            ill_ema = m.illnet_model.diffusion_model(x, t_emb=te)
    after = run_loop()
    assert torch.equal(after[0], live[0]) and not torch.equal(inside[0], live[0]) and live[2].tolist() == [T] * B
    with torch.no_grad():
        ill_live = m.illnet_model.diffusion_model(x, t_emb=te)
    save("ema_drmnet", Lr0_live=live[0], Lr0_ema=inside[0], K=live[2], illnet_live=ill_live, illnet_ema=ill_ema, T=T, B=B, gamma=0.9, epsilon=1e-3,
         delta=0.025, gen_seed=9, temb_seed=10, input_seed=5, num_updates=int(m.illnet_model_ema.num_updates), decay=float(m.illnet_model_ema.decay))

    # ---- ObsNet
    ocfg = rh.load_yaml_params("configs/obsnet/eval_obsnet.yaml")["model"]["params"]
    ocfg.pop("ckpt_path")
    ocfg["unet_config"] = {"target": ocfg["unet_config"]["target"], "params": dict(ou.TINY_UNET_CFG)}
    ocfg.update(image_size=16, use_ema=True)
    obs = OBS(**ocfg).eval()
    synth.load_synth(obs.model.diffusion_model, 21)
    for name, p in obs.model.named_parameters():
        getattr(obs.model_ema, obs.model_ema.m_name2s_name[name]).copy_(p.detach())
    drift(obs.model, obs.model_ema, 303)
    path = os.path.join(GOLD, "obsnet_tiny_ema.ckpt")
    torch.save({"state_dict": obs.state_dict()}, path)
    print(f"  wrote obsnet_tiny_ema.ckpt ({os.path.getsize(path) / 1024:.0f} KiB)")
    B = 2
    g = gen(3)
    cond = synth.synth_refmaps(B, 16, 16, 98) * 2 - 1
    x_T = torch.randn((B, 3, 16, 16), generator=g)
    noise = torch.randn((50, B, 3, 16, 16), generator=g)
    ctr = {"i": 0}

    def noise_like(shape, device, repeat=False):
        out = noise[ctr["i"]]
        ctr["i"] += 1
        return out

    o1 = refddim.noise_like
    refddim.noise_like = noise_like

    def run_ddim(steps=3):
        """the first `steps` of the 50-step eta = 1 DDIM schedule through the reference's p_sample_ddim (ddim.py:206-259), under whatever weights are live"""
        s = DDIM(obs)
        s.make_schedule(ddim_num_steps=50, ddim_eta=1.0, verbose=False)
        ctr["i"] = 0
        x = x_T
        with torch.no_grad():
            for index in range(49, 49 - steps, -1):
                ts = torch.full((B,), int(s.ddim_timesteps[index]), dtype=torch.long)
                x, _ = s.p_sample_ddim(x, cond, ts, index=index)
        return x

    try:
        live = run_ddim()
        with obs.ema_scope("golden"):
            inside = run_ddim()
            with torch.no_grad():
                eps_ema = obs.apply_model(x_T, torch.full((B,), 981, dtype=torch.long), cond)
        assert torch.equal(run_ddim(), live) and not torch.equal(inside, live)
    finally:
        refddim.noise_like = o1
    save("ema_obsnet", x_live=live, x_ema=inside, eps_ema=eps_ema, B=B, gen_seed=3, input_seed=98, steps=3)


def make_resize():
    """An actual resize in BaseDataset.transform (dataset/basedataset.py:29-50: torchvision.transforms.functional.resize to
    (size, size), antialias=True; tools/refharness.py restates torchvision's tensor path over torch.nn.functional.interpolate) and
    the nearest mask resize of get_cond_for_predict (models/obsnet.py:691), from the reference's own BaseDataset at sizes != input."""
    rh.install_stubs()
    from dataset.basedataset import BaseDataset

    g = gen(81)
    out = {}
    hdr = torch.exp(torch.randn((2, 3, 40, 40), generator=g) * 1.2 - 1.0)
    rect = torch.exp(torch.randn((3, 24, 56), generator=g) * 1.2 - 1.0)  # 3-D, non-square, non-integer scale
    big = torch.exp(torch.randn((1, 3, 128, 128), generator=g) * 1.0 - 1.0)
    out.update(hdr=hdr, rect=rect, big=big)
    out["resize_only"] = BaseDataset(size=16, transform_func="resize").transform(hdr)
    out["log_of_resized"] = BaseDataset(size=16, transform_func="log_resize").transform(hdr)       # log(resize(x))
    out["resized_log"] = BaseDataset(size=16, transform_func="resize_log").transform(hdr)          # resize(log(x))
    out["rect_16"] = BaseDataset(size=16, transform_func="resize").transform(rect)
    out["rect_24"] = BaseDataset(size=24, transform_func="resize").transform(rect)                  # H unchanged, W 56 -> 24
    out["big_48"] = BaseDataset(size=48, transform_func="resize").transform(big)                    # 128 -> 48 (scale 2.67)
    out["bicubic_16"] = BaseDataset(size=16, transform_func="resizeBICUBIC").transform(hdr)
    out["nearest_16"] = BaseDataset(size=16, transform_func="resizeNEAREST").transform(hdr)
    mask = (torch.rand((2, 1, 40, 40), generator=g) > 0.5).float()
    out.update(mask=mask, mask_16=torch.nn.functional.interpolate(mask, size=(16, 16)), mask_64=torch.nn.functional.interpolate(mask, size=(64, 64)),
               mask_rect=torch.nn.functional.interpolate(mask[:, :, :24, :], size=(16, 16)))
    save("resize", **out)

STEPS = {
    "refmap": lambda oa: make_refmap(),
    "estimate_chain": lambda oa: make_estimate_chain(),
    "manifests": lambda oa: make_manifests(oa),
    "primitives": lambda oa: make_primitives(oa),
    "brdf": lambda oa: make_brdf_schedule(),
    "tiny": lambda oa: make_tiny_nets(oa),
    "blocks": lambda oa: make_blocks(oa),
    "drmnet_loop": lambda oa: make_drmnet_loop(),
    "drmnet_loop_full": lambda oa: make_drmnet_loop(full=True),
    "estimate_chain_full": lambda oa: make_estimate_chain(full=True),
    "full_sizes": lambda oa: make_full_sizes(oa),
    "full_rows": lambda oa: make_full_rows(oa),
    "obsnet_samplers": lambda oa: make_obsnet_samplers(),
    "sampler_masks": lambda oa: make_sampler_masks(),
    "ddim_variants": lambda oa: make_ddim_variants(),
    "sampler_guidance": lambda oa: make_sampler_guidance(),
    "full": lambda oa: make_full_nets(oa),
    "full_samplers": lambda oa: make_full_samplers(),
    "transforms": lambda oa: make_transforms(),
    "long_chains": lambda oa: make_long_chains(),
    "stress": lambda oa: make_stress(),
    "ema_ckpt": lambda oa: make_ema_ckpt(),
    "resize": lambda oa: make_resize(),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    _, _, _, oa = rh.ref_classes()
    for name, fn in STEPS.items():
        if args.only and name not in args.only:
            continue
        print(f"[{name}]")
        fn(oa)
    meta = {"torch": torch.__version__, "numpy": np.__version__, "reference": "kyotovision-public/DRMNet @ /root/reference (2025-02-23)"}
    with open(os.path.join(GOLD, "META.json"), "w") as f:
        json.dump(meta, f)


if __name__ == "__main__":
    main()
