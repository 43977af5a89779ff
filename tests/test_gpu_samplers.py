"""GPU parity of the sampler loops (device loops behind the C ABI, driven through the reference-named host classes)
against traces recorded from the reference (tests/golden/*_trace / drmnet_loop_*), with injected noise."""
import numpy as np
import pytest
import torch

from conftest import ACCURATE_MODES, CONTRACT, gold, rel_l2
from drmnet_amd import synth
from oracle import unet as ou

pytestmark = pytest.mark.gpu

UNET_T = {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": dict(ou.TINY_UNET_CFG)}
ENC_T = {"target": "ldm.modules.diffusionmodules.openaimodel.EncoderUNetModel", "params": dict(ou.TINY_ENC_CFG)}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (no fallback)"
    return torch.device("cuda:0")


def tiny_drmnet(g, dev):
    from drmnet_amd.drmnet import DRMNet

    m = DRMNet(illnet_config=UNET_T, refnet_config=ENC_T, renderer_config=None, max_timesteps=int(g["max_timesteps"]), image_size=16,
               concat_mode=True, use_ema=False, gamma=float(g["gamma"]), epsilon=float(g["epsilon"]), delta=float(g["delta"]),
               z0=[1, 1, 1, 1, 0, 1], brdf_param_names=["a"] * 6)
    synth.load_synth(m.illnet_model.diffusion_model, 21)
    synth.load_synth(m.refnet_model.diffusion_model, 22)
    m.illnet_model.z_emb_layer.load_state_dict(synth.synth_state_dict(
        [(k, tuple(v.shape)) for k, v in m.illnet_model.z_emb_layer.state_dict().items()], synth.SEED_ZEMB))
    sd = m.refnet_model.diffusion_model.state_dict()
    sd["out.3.weight"] = sd["out.3.weight"] * float(g["head_w_scale"])
    sd["out.3.bias"] = torch.from_numpy(g["head_bias"])
    m.refnet_model.diffusion_model.load_state_dict(sd)
    return m.to(dev)


@pytest.mark.parametrize("precision", ACCURATE_MODES)
@pytest.mark.parametrize("tag", ["a", "b"])
def test_drmnet_p_sample_loop_vs_reference_trace(dev, tag, precision):
    g = gold(f"drmnet_loop_{tag}")
    m = tiny_drmnet(g, dev).set_precision(precision)
    LrK = torch.from_numpy(g["LrK"]).to(dev)
    n0 = torch.from_numpy(g["noise0"]).to(dev)
    sn = torch.from_numpy(g["step_noise"]).to(dev)
    Lr0, zK, K = m.p_sample_loop(LrK, [LrK], [LrK], verbose=False, noise0=n0, step_noise=sn)
    print(f"drmnet loop {tag} ({precision}): K={K.tolist()} steps={m.last_steps} rel_l2={rel_l2(Lr0.cpu(), g['Lr0']):.2e}")
    assert K.cpu().tolist() == g["K"].tolist()
    assert np.array_equal(np.isnan(zK.cpu().numpy()), np.isnan(g["zK"]))
    assert np.allclose(np.nan_to_num(zK.cpu().numpy()), np.nan_to_num(g["zK"]), atol=1e-5 if precision != "f16mx" else 1e-4)
    assert rel_l2(Lr0.cpu(), g["Lr0"]) < 1e-4
    assert m.last_steps == int(g["K"].max())
    # host-driven variant (per-step C-ABI entry point) returns the reference's intermediates layout
    Lr0b, zKb, Kb, inter = m.p_sample_loop(LrK, [LrK], [LrK], return_intermediates=True, verbose=False, log_every_k=1, noise0=n0, step_noise=sn)
    assert Kb.cpu().tolist() == g["K"].tolist()
    assert rel_l2(Lr0b.cpu(), g["Lr0"]) < 1e-4
    zk_steps = torch.stack(inter["zk_inter"]).cpu().numpy()
    assert zk_steps.shape == g["zk_steps"].shape
    assert np.allclose(np.nan_to_num(zk_steps), np.nan_to_num(g["zk_steps"]), atol=2e-5 if precision != "f16mx" else 1e-4)
    Lrk_steps = torch.stack(inter["Lrk_inter"][1:]).cpu()
    assert rel_l2(Lrk_steps, g["Lrk_steps"]) < 1e-4


def test_drmnet_loop_without_early_exit_and_philox(dev):
    g = gold("drmnet_loop_a")
    m = tiny_drmnet(g, dev)
    LrK = torch.from_numpy(g["LrK"]).to(dev)
    a = m.p_sample_loop(LrK, [LrK], [LrK], verbose=False, seed=7, early_exit=False)
    assert m.last_steps == int(g["max_timesteps"])
    b = m.p_sample_loop(LrK, [LrK], [LrK], verbose=False, seed=7, early_exit=False)
    c = m.p_sample_loop(LrK, [LrK], [LrK], verbose=False, seed=8, early_exit=False)
    assert torch.equal(a[0], b[0]) and not torch.equal(a[0], c[0])
    assert torch.isfinite(a[0]).all()
    assert a[2].tolist() == [int(g["max_timesteps"])] * LrK.shape[0] and torch.isnan(a[1]).all()


@pytest.mark.parametrize("parts", [2, 3])
def test_drmnet_step_over_forked_batch_parts_equals_the_single_stream_step(dev, parts, monkeypatch):
    """drm_drmnet_set_batch_parts: the row ranges of a step on internal streams (forked from / joined into the caller's stream) give what the
    single-stream step gives -- through the reference trace (rows leave the loop at different steps: the ranges shrink and collapse to one), the
    host-driven per-step entry point, and a batch that is not a multiple of the parts.  The parts engage from 64 rows each in the product;
    drm_drmnet_set_batch_part_min lets a tiny batch through them."""
    g = gold("drmnet_loop_b")
    LrK = torch.from_numpy(g["LrK"]).to(dev)
    n0 = torch.from_numpy(g["noise0"]).to(dev)
    sn = torch.from_numpy(g["step_noise"]).to(dev)
    ref = tiny_drmnet(g, dev).set_precision("f16x3")
    Lr0_1, zK_1, K_1 = ref.p_sample_loop(LrK, [LrK], [LrK], verbose=False, noise0=n0, step_noise=sn)
    m = tiny_drmnet(g, dev).set_precision("f16x3").set_batch_parts(parts, min_rows=1)
    Lr0, zK, K = m.p_sample_loop(LrK, [LrK], [LrK], verbose=False, noise0=n0, step_noise=sn)
    assert K.cpu().tolist() == g["K"].tolist() == K_1.cpu().tolist()
    # (a range of 1-2 rows takes other tile shapes / split-K forms than the whole batch: equal to f16x3 rounding, not bitwise)
    assert rel_l2(Lr0.cpu(), g["Lr0"]) < 1e-4 and rel_l2(Lr0.cpu(), Lr0_1.cpu()) < 2e-5
    assert np.allclose(np.nan_to_num(zK.cpu().numpy()), np.nan_to_num(zK_1.cpu().numpy()), atol=1e-5)
    Lr0b, _, Kb, _ = m.p_sample_loop(LrK, [LrK], [LrK], return_intermediates=True, verbose=False, log_every_k=1, noise0=n0, step_noise=sn)
    assert Kb.cpu().tolist() == g["K"].tolist() and rel_l2(Lr0b.cpu(), Lr0_1.cpu()) < 2e-5
    # identity rows (no row list), Philox noise, 7 rows over the parts: same as the single-stream sampler with the same seed
    x7 = synth.synth_refmaps(7, 16, 16, synth.SEED_INPUT).to(dev)
    # (the first steps are compared: this tiny random pair of networks amplifies a last-bit difference step by step)
    a = m.p_sample_loop(x7, [x7], [x7], return_intermediates=True, verbose=False, log_every_k=1, seed=11, early_exit=False)
    b = ref.p_sample_loop(x7, [x7], [x7], return_intermediates=True, verbose=False, log_every_k=1, seed=11, early_exit=False)
    errs = [rel_l2(p.cpu(), q.cpu()) for p, q in zip(a[3]["Lrk_inter"], b[3]["Lrk_inter"])]
    print("forked vs single-stream, 7 identity rows, per step:", " ".join(f"{e:.1e}" for e in errs))
    assert max(errs[:3]) < 5e-6 and torch.isfinite(a[0]).all()


def tiny_obsnet(dev):
    from drmnet_amd.obsnet import ObsNetDiffusion

    m = ObsNetDiffusion(unet_config=UNET_T, linear_start=1e-4, linear_end=0.09, log_every_t=2000, timesteps=1000, first_stage_key="LrK",
                        cond_stage_key="raw_refmap", padding_mode="noise", image_size=16, channels=3, concat_mode=True, ddim_steps=50,
                        clip_denoised=False, masked_loss=False, use_ema=False)
    synth.load_synth(m.model.diffusion_model, 21)
    return m.to(dev)


@pytest.mark.parametrize("precision", ACCURATE_MODES)
@pytest.mark.parametrize("eta", [0, 1])
def test_ddim_sample_vs_reference_trace(dev, eta, precision):
    from drmnet_amd.ddim import DDIMSampler

    g = gold(f"ddim_trace_eta{eta}")
    m = tiny_obsnet(dev).set_precision(precision)
    cond, x_T, noise = (torch.from_numpy(g[k]).to(dev) for k in ("cond", "x_T", "noise"))
    s = DDIMSampler(m)
    x1, _ = s.sample(50, cond.shape[0], (3, 16, 16), cond, eta=float(eta), x_T=x_T, verbose=False, noise=noise, num_steps=1)
    e1 = rel_l2(x1.cpu(), g["x_inter"][0])
    x, inter = s.sample(50, cond.shape[0], (3, 16, 16), cond, eta=float(eta), x_T=x_T, verbose=False, noise=noise)
    e = rel_l2(x.cpu(), g["x"])
    print(f"ddim eta={eta} ({precision}): first step {e1:.2e}, 50 steps {e:.2e}")
    assert (e1 < 2e-5 and e < 1e-5) if precision != "f16mx" else (e1 < CONTRACT and e < CONTRACT)  # 50 steps: observed 6.6e-7 (fp32) .. 2e-6
    assert torch.equal(inter["x_inter"][0], x_T)
    # ObsNetDiffusion.sample_log(ddim=True) is the estimate.py entry point (scripts/estimate.py:72-79)
    y, _ = m.sample_log(cond=cond, batch_size=cond.shape[0], ddim=True, ddim_steps=50, eta=float(eta), x_T=x_T, noise=noise)
    assert torch.equal(y, x)


@pytest.mark.parametrize("log_every_t", [1, 10, 100])
def test_ddim_intermediates_are_logged_as_the_reference_logs_them(dev, log_every_t):
    """ddim.py:171-204: (img, pred_x0) are appended after the step of `index` when index % log_every_t == 0 or at the first step; x_T leads both lists.
    The fixture holds the reference's per-step img record (log_every_t = 1)."""
    from drmnet_amd.ddim import DDIMSampler

    g = gold("ddim_trace_eta1")
    m = tiny_obsnet(dev).set_precision("f16x3")
    cond, x_T, noise = (torch.from_numpy(g[k]).to(dev) for k in ("cond", "x_T", "noise"))
    s = DDIMSampler(m)
    x, inter = s.sample(50, cond.shape[0], (3, 16, 16), cond, eta=1.0, x_T=x_T, verbose=False, noise=noise, log_every_t=log_every_t)
    want = [j for j in range(50) if (49 - j) % log_every_t == 0 or j == 0]  # iteration numbers the reference logs
    assert len(inter["x_inter"]) == len(inter["pred_x0"]) == 1 + len(want)
    assert torch.equal(inter["x_inter"][0], x_T) and torch.equal(inter["pred_x0"][0], x_T)
    for k, j in enumerate(want):
        assert rel_l2(inter["x_inter"][1 + k].cpu(), g["x_inter"][j]) < 2e-5, (log_every_t, j)
        assert bool(torch.isfinite(inter["pred_x0"][1 + k]).all())
    assert torch.equal(inter["x_inter"][-1], x)  # index 0 is always logged: the last entry is the returned sample
    # the logged chain is the unlogged chain
    x2, _ = s.sample(50, cond.shape[0], (3, 16, 16), cond, eta=1.0, x_T=x_T, verbose=False, noise=noise, log_every_t=0)
    assert torch.equal(x2, x)


@pytest.mark.parametrize("precision", ACCURATE_MODES)
def test_ddpm_ancestral_vs_reference_trace(dev, precision):
    g = gold("ddpm_trace")
    m = tiny_obsnet(dev).set_precision(precision)
    cond, x_T, noise = (torch.from_numpy(g[k]).to(dev) for k in ("cond", "x_T", "noise"))
    pred_x0, inter = m.p_sample_loop(cond, tuple(x_T.shape), return_intermediates=True, x_T=x_T, verbose=False, start_T=6, noise=noise)
    e_img = rel_l2(inter["x_inter"][-1].cpu(), g["x_inter"][-1])
    e_x0 = rel_l2(pred_x0.cpu(), g["pred_x0"])
    print(f"ddpm 6 steps ({precision}): img {e_img:.2e} pred_x0 {e_x0:.2e}")
    assert (e_img < 1e-5 and e_x0 < 1e-5) if precision != "f16mx" else (e_img < CONTRACT and e_x0 < CONTRACT)


@pytest.mark.parametrize("precision", ["f16x3", "f16mx"])
def test_sampler_mask_x0_temperature_vs_reference_traces(dev, precision):
    """[r6, VERDICT r5 item 9] the samplers' `mask` / `x0` / `temperature` arguments on the device loops (mask_blend_kernel + the sigma column), against
    the reference's own three loops: DDIMSampler.sample(mask=, x0=, temperature=0.7) (ddim.py:175-178, :255; 50 steps, one-channel binary and
    three-channel soft masks), ObsNetDiffusion.p_sample_loop(mask=, x0=) (blend before p_sample at t - 1, models/obsnet.py:545-547) and
    LatentDiffusion.p_sample_loop(mask=, x0=) (blend after p_sample at t, ddpm.py:1300-1302); graph replay == eager; Philox q-noise runs."""
    from drmnet_amd import ops
    from drmnet_amd.ddim import DDIMSampler
    from drmnet_amd.obsnet import LatentDiffusion

    g = gold("sampler_masks")
    m = tiny_obsnet(dev).set_precision(precision)
    T = lambda k: torch.from_numpy(g[k]).to(dev)
    cond, x_T, x0, noise, qnoise = T("cond"), T("x_T"), T("x0"), T("noise"), T("qnoise")
    temp = float(g["temperature"])
    tol = 2e-5 if precision != "f16mx" else CONTRACT
    s = DDIMSampler(m)
    for tag in ("m1", "m3"):
        mk = T("mask1" if tag == "m1" else "mask3")
        x, inter = s.sample(50, 3, (3, 16, 16), cond, eta=1.0, x_T=x_T, verbose=False, noise=noise, mask=mk, x0=x0, temperature=temp, mask_noise=qnoise, log_every_t=1)
        e1, e = rel_l2(inter["x_inter"][1].cpu(), g[f"ddim_{tag}_x_inter"][0]), rel_l2(x.cpu(), g[f"ddim_{tag}_x"])
        print(f"ddim mask {tag} ({precision}): first step {e1:.2e}, 50 steps {e:.2e}")
        assert e1 < tol and e < tol and len(inter["x_inter"]) == 51
        x_nolog, _ = s.sample(50, 3, (3, 16, 16), cond, eta=1.0, x_T=x_T, verbose=False, noise=noise, mask=mk, x0=x0, temperature=temp, mask_noise=qnoise, log_every_t=0)
        assert torch.equal(x_nolog, x)
    pred_x0, inter = m.p_sample_loop(cond, tuple(x_T.shape), return_intermediates=True, x_T=x_T, verbose=False, start_T=6, noise=noise[:6], mask=T("mask1"), x0=x0,
                                     mask_noise=qnoise[:6])
    e_img, e_x0 = rel_l2(inter["x_inter"][-1].cpu(), g["obs_x_inter"][-1]), rel_l2(pred_x0.cpu(), g["obs_pred_x0"])
    img = LatentDiffusion.p_sample_loop(m, cond, tuple(x_T.shape), x_T=x_T, verbose=False, start_T=6, noise=noise[:6], mask=T("mask3"), x0=x0, mask_noise=qnoise[:6])
    e_ldm = rel_l2(img.cpu(), g["ldm_x"])
    print(f"ddpm masks ({precision}): obsnet form img {e_img:.2e} pred_x0 {e_x0:.2e}; ldm form {e_ldm:.2e}")
    assert e_img < tol and e_x0 < tol and e_ldm < tol
    # temperature alone on the ancestral loop == the noise tensor scaled by it (same kernel path, the sigma column carries the factor)
    a = m.p_sample_loop(cond, tuple(x_T.shape), x_T=x_T, verbose=False, start_T=6, noise=noise[:6], temperature=temp)
    b = m.p_sample_loop(cond, tuple(x_T.shape), x_T=x_T, verbose=False, start_T=6, noise=noise[:6] * temp)
    assert rel_l2(a.cpu(), b.cpu()) < 1e-6
    # graph replay of the masked chain is the eager chain, bit for bit
    ops.set_graph_replay(True)
    try:
        xg, _ = s.sample(50, 3, (3, 16, 16), cond, eta=1.0, x_T=x_T, verbose=False, noise=noise, mask=T("mask3"), x0=x0, temperature=temp, mask_noise=qnoise, log_every_t=0)
    finally:
        ops.set_graph_replay(False)
    assert torch.equal(xg, x)
    # Philox q-noise: runs, finite, known region pulled towards x0 (binary mask, last blend at t = ddim_timesteps[0] = 1: almost x0 itself before the last step)
    xp, _ = s.sample(50, 3, (3, 16, 16), cond, eta=0.0, x_T=x_T, verbose=False, seed=5, mask=T("mask1"), x0=x0, log_every_t=0)
    assert torch.isfinite(xp).all()
    with pytest.raises(ValueError):
        s.sample(50, 3, (3, 16, 16), cond, eta=0.0, x_T=x_T, verbose=False, seed=5, mask=T("mask1"))
    with pytest.raises(NotImplementedError):
        s.sample(50, 3, (3, 16, 16), cond, eta=0.0, x_T=x_T, verbose=False, seed=5, score_corrector=object())


@pytest.mark.parametrize("precision", ["f16x3", "f16mx"])
def test_guidance_and_noise_dropout_vs_reference_traces(dev, precision):
    """[r6] classifier-free guidance (p_sample_ddim, ddim.py:225-232: a second forward per step on the unconditional conditioning, e = e_u + s (e_c - e_u))
    and noise_dropout (ddim.py:256-257; ddpm.py:1158-1159: F.dropout on the step noise, keep masks injected) on the device loops against the reference's
    own 50-step chains; graph replay == eager; Philox dropout runs and drops about p of the noise."""
    from drmnet_amd import ops
    from drmnet_amd.ddim import DDIMSampler

    g = gold("sampler_guidance")
    m = tiny_obsnet(dev).set_precision(precision)
    T = lambda k: torch.from_numpy(g[k]).to(dev)
    cond, ucond, x_T, noise, keep = T("cond"), T("ucond"), T("x_T"), T("noise"), T("keep")
    p, scale = float(g["p"]), float(g["scale"])
    s = DDIMSampler(m)
    kw = dict(eta=1.0, x_T=x_T, verbose=False, noise=noise)
    tol = (2e-5, 2e-5) if precision != "f16mx" else (CONTRACT, CONTRACT)  # (first step, 50 steps; observed 1.7e-6 / 3.5e-5 at guidance scale 3)
    x, inter = s.sample(50, 3, (3, 16, 16), cond, log_every_t=1, unconditional_guidance_scale=scale, unconditional_conditioning=ucond, **kw)
    e1, e = rel_l2(inter["x_inter"][1].cpu(), g["cfg_first"]), rel_l2(x.cpu(), g["cfg_x"])
    print(f"ddim guidance scale {scale} ({precision}): first step {e1:.2e}, 50 steps {e:.2e}")
    assert e1 < tol[0] and e < tol[1]
    # scale 1 or no unconditional conditioning: the plain chain (the reference's own condition, ddim.py:225)
    plain, _ = s.sample(50, 3, (3, 16, 16), cond, log_every_t=0, **kw)
    same, _ = s.sample(50, 3, (3, 16, 16), cond, log_every_t=0, unconditional_guidance_scale=1.0, unconditional_conditioning=ucond, **kw)
    assert torch.equal(same, plain) and rel_l2(plain.cpu(), g["cfg_x"]) > 1e-2
    xd, inter = s.sample(50, 3, (3, 16, 16), cond, log_every_t=1, noise_dropout=p, dropout_keep=keep, **kw)
    e1, e = rel_l2(inter["x_inter"][1].cpu(), g["drop_first"]), rel_l2(xd.cpu(), g["drop_x"])
    print(f"ddim noise_dropout {p} ({precision}): first step {e1:.2e}, 50 steps {e:.2e}")
    assert e1 < tol[0] and e < (2e-5 if precision != "f16mx" else CONTRACT)
    # ancestral loop: dropout with all-ones keep masks and p = 0.5 == the noise doubled (same kernel path, factor 1 / (1 - p))
    a = m.p_sample_loop(cond, tuple(x_T.shape), x_T=x_T, verbose=False, start_T=6, noise=noise[:6], noise_dropout=0.5, dropout_keep=torch.ones_like(noise[:6]))
    b = m.p_sample_loop(cond, tuple(x_T.shape), x_T=x_T, verbose=False, start_T=6, noise=noise[:6] * 2.0)
    assert rel_l2(a.cpu(), b.cpu()) < 1e-6
    ops.set_graph_replay(True)
    try:
        xg, _ = s.sample(50, 3, (3, 16, 16), cond, log_every_t=0, unconditional_guidance_scale=scale, unconditional_conditioning=ucond, **kw)
        xdg, _ = s.sample(50, 3, (3, 16, 16), cond, log_every_t=0, noise_dropout=p, dropout_keep=keep, **kw)
    finally:
        ops.set_graph_replay(False)
    assert torch.equal(xg, x) and torch.equal(xdg, xd)
    # Philox keep masks: eta = 1 chain with p = 0.3 differs from the plain chain and stays finite; p out of range is rejected
    xp, _ = s.sample(50, 3, (3, 16, 16), cond, eta=1.0, x_T=x_T, verbose=False, seed=5, noise_dropout=p, log_every_t=0)
    xq, _ = s.sample(50, 3, (3, 16, 16), cond, eta=1.0, x_T=x_T, verbose=False, seed=5, log_every_t=0)
    assert torch.isfinite(xp).all() and rel_l2(xp.cpu(), xq.cpu()) > 1e-3
    with pytest.raises(ValueError):
        s.sample(50, 3, (3, 16, 16), cond, eta=1.0, x_T=x_T, verbose=False, seed=5, noise_dropout=1.0)


def test_ddim_timesteps_subset_vs_reference_trace(dev):
    """[r6] DDIMSampler.ddim_sampling(timesteps=30) (ddim.py:156-158: the first 29 entries of the 50-step schedule) against the reference's trace;
    ddim_use_original_steps raises here as it does in the reference (recorded in the fixture)."""
    from drmnet_amd.ddim import DDIMSampler

    g = gold("ddim_variants")
    m = tiny_obsnet(dev).set_precision("f16x3")
    cond, x_T, noise = (torch.from_numpy(g[k]).to(dev) for k in ("cond", "x_T", "noise"))
    s = DDIMSampler(m)
    s.make_schedule(50, ddim_eta=1.0, verbose=False)
    x, inter = s.ddim_sampling(cond, tuple(x_T.shape), x_T=x_T, noise=noise, timesteps=30, log_every_t=1, verbose=False)
    assert len(inter["x_inter"]) - 1 == int(g["subset_n"]) == 29
    e1, e = rel_l2(inter["x_inter"][1].cpu(), g["subset_first"]), rel_l2(x.cpu(), g["subset_x"])
    print(f"ddim timesteps=30 (29 steps): first {e1:.2e}, final {e:.2e}")
    assert e1 < 2e-5 and e < 2e-5
    assert int(g["orig_runs"]) == 0
    with pytest.raises(NotImplementedError):
        s.ddim_sampling(cond, tuple(x_T.shape), x_T=x_T, noise=noise, ddim_use_original_steps=True, verbose=False)
    with pytest.raises(ValueError):
        s.ddim_sampling(cond, tuple(x_T.shape), x_T=x_T, noise=noise, timesteps=1, verbose=False)


def test_step_dropins_match_reference_named_methods(dev):
    """DRMNet.forward / p_mean_variance and LatentDiffusion.apply_model / p_sample keep the reference's per-step semantics."""
    from oracle import samplers as osamp

    g = gold("drmnet_loop_a")
    m = tiny_drmnet(g, dev)
    LrK = torch.from_numpy(g["LrK"]).to(dev)
    mean, delta, z_out = m.p_mean_variance(LrK, [LrK], [LrK], reversed_k=3)
    Pu = synth.synth_state_dict(ou.param_manifest(ou.TINY_UNET_CFG, "unet"), 21)
    Pe = synth.synth_state_dict(ou.param_manifest(ou.TINY_ENC_CFG, "encoder"), 22)
    Pe["out.3.weight"] = Pe["out.3.weight"] * float(g["head_w_scale"])
    Pe["out.3.bias"] = torch.from_numpy(g["head_bias"])
    Pz = synth.synth_state_dict(ou.zemb_manifest(6, 32), synth.SEED_ZEMB)
    tu, te = ou.build_topology(ou.TINY_UNET_CFG, "unet"), ou.build_topology(ou.TINY_ENC_CFG, "encoder")
    x = torch.from_numpy(g["LrK"])
    xc = torch.cat([x, x], 1)
    z_ref = ou.encoder_forward(Pe, te, xc, torch.full((x.shape[0],), 3, dtype=torch.long))
    zk, _ = osamp.brdf_schedule(z_ref, torch.from_numpy(g["z0"]), float(g["gamma"]), 3)
    out_ref = x + ou.unet_forward(Pu, tu, xc, t_emb=ou.z_embed(Pz, zk - torch.from_numpy(g["z0"])))
    assert delta == float(g["delta"])
    assert rel_l2(z_out.cpu(), z_ref) < 2e-5 and rel_l2(mean.cpu(), out_ref) < 2e-5


@pytest.mark.parametrize("precision", ["f16x3", "f16mx"])
def test_graph_replayed_chain_equals_eager_chain(dev, precision):
    """drm_ddim_sample / drm_ddpm_sample: step 1 eager, step 2 captured, the rest replayed (per-step scalars in a device table)
    must be bit-identical to the all-eager chain, with injected noise and with the Philox stream."""
    from drmnet_amd import ops
    from drmnet_amd.ddim import DDIMSampler

    g = gold("ddim_trace_eta1")
    m = tiny_obsnet(dev).set_precision(precision)
    cond, x_T, noise = (torch.from_numpy(g[k]).to(dev) for k in ("cond", "x_T", "noise"))
    s = DDIMSampler(m)
    run = lambda **kw: s.sample(50, cond.shape[0], (3, 16, 16), cond, eta=1.0, x_T=x_T, verbose=False, **kw)[0]
    try:
        ops.set_graph_replay(False)
        n0 = ops.graph_launches()
        eager, eager_p = run(noise=noise), run(seed=11)
        assert ops.graph_launches() == n0
        ops.set_graph_replay(True)
        graphed, graphed_p = run(noise=noise), run(seed=11)
        assert ops.graph_launches() == n0 + 2 * 49
        assert torch.equal(graphed, eager) and torch.equal(graphed_p, eager_p)
        assert rel_l2(graphed.cpu(), g["x"]) < 2e-4  # and still the reference's trace
        d_e = (ops.set_graph_replay(False), m.p_sample_loop(cond, tuple(x_T.shape), x_T=x_T, verbose=False, start_T=7, seed=5))[1]
        d_g = (ops.set_graph_replay(True), m.p_sample_loop(cond, tuple(x_T.shape), x_T=x_T, verbose=False, start_T=7, seed=5))[1]
        assert torch.equal(d_e, d_g)
    finally:
        ops.set_graph_replay(False)
