"""GPU parity on the round-4 fixtures recorded from the reference (tools/make_golden.py --only long_chains stress ema_ckpt resize), every case
in every accurate arithmetic mode incl. bench.py's default (f16mx):

* DRMNet's 150-step reverse process (models/drmnet.py:782-847; rows leaving after 3 ... 148 steps, one never) and the whole 1000-step
  ancestral chain (ldm/models/diffusion/ddpm.py:1120-1167), device loops behind the C ABI;
* the three shipped networks at full width with heavy-tailed weights and GroupNorm gains x 3 / x 10;
* checkpoints WRITTEN BY THE REFERENCE (use_ema=True; LitEma shadows moved by the reference's own LitEma.forward): init_from_ckpt ->
  ema_scope -> sampling, against what the reference produced inside and outside its own ema_scope;
* BaseDataset's resize at sizes != input and the nearest mask resize (drm_resize).
"""
import os

import numpy as np
import pytest
import torch

from conftest import ACCURATE_MODES, CONTRACT, GOLD, NET_TOL, gold, rel_l2
from drmnet_amd import ops, synth
from oracle import unet as ou
from test_gpu_nets import build, full_inputs
from test_round4_cpu import long_drm_inputs, long_obs_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(GOLD))
UNET_T = {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": dict(ou.TINY_UNET_CFG)}
ENC_T = {"target": "ldm.modules.diffusionmodules.openaimodel.EncoderUNetModel", "params": dict(ou.TINY_ENC_CFG)}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (no fallback)"
    return torch.device("cuda:0")


# --------------------------------------------------------------------------------------------- long chains


@pytest.mark.parametrize("precision", ACCURATE_MODES)
def test_drmnet_150_step_loop_vs_reference(dev, precision):
    from drmnet_amd.drmnet import DRMNet

    g = gold("drmnet_loop_150")
    T = int(g["T"])
    m = DRMNet(illnet_config=UNET_T, refnet_config=ENC_T, renderer_config=None, max_timesteps=T, image_size=16, concat_mode=True, use_ema=False,
               gamma=float(g["gamma"]), epsilon=float(g["epsilon"]), delta=float(g["delta"]), z0=[1, 1, 1, 1, 0, 1], brdf_param_names=["a"] * 6)
    synth.load_synth(m.illnet_model.diffusion_model, 21)
    synth.load_synth(m.refnet_model.diffusion_model, 22)
    m.illnet_model.z_emb_layer.load_state_dict(synth.synth_state_dict(
        [(k, tuple(v.shape)) for k, v in m.illnet_model.z_emb_layer.state_dict().items()], synth.SEED_ZEMB))
    isd = m.illnet_model.diffusion_model.state_dict()
    isd["out.2.weight"] = isd["out.2.weight"] * float(g["ill_out_scale"])
    isd["out.2.bias"] = isd["out.2.bias"] * float(g["ill_out_scale"])
    m.illnet_model.diffusion_model.load_state_dict(isd)
    sd = m.refnet_model.diffusion_model.state_dict()
    sd["out.3.weight"] = sd["out.3.weight"] * float(g["head_w_scale"])
    sd["out.3.bias"] = torch.from_numpy(g["head_bias"])
    m.refnet_model.diffusion_model.load_state_dict(sd)
    m = m.to(dev).set_precision(precision)
    LrK, n0, sn = (t.to(dev) for t in long_drm_inputs(g))
    Lr0, zK, K = m.p_sample_loop(LrK, [LrK], [LrK], verbose=False, noise0=n0, step_noise=sn)
    e = rel_l2(Lr0.cpu(), g["Lr0"])
    print(f"DRMNet 150-step loop ({precision}): K = {K.tolist()} (reference {g['K'].tolist()}), Lr0 rel-L2 {e:.2e}")
    assert K.tolist() == g["K"].tolist() and m.last_steps == T
    assert np.array_equal(np.isnan(zK.cpu().numpy()), np.isnan(g["zK"]))
    # (zK is the RefNet output through a head amplified x 25 by the fixture, which spreads the rows' exits over the 150 steps: its absolute
    # error is 25 x that of the plain network)
    assert np.allclose(np.nan_to_num(zK.cpu().numpy()), np.nan_to_num(g["zK"]), atol=(1e-5 if precision != "f16mx" else 1e-4) * 5)
    assert e < (2e-5 if precision != "f16mx" else CONTRACT)
    # host-driven loop, every tenth step of the reference's log
    Lr0h, zKh, Kh, inter = m.p_sample_loop(LrK, [LrK], [LrK], return_intermediates=True, verbose=False, log_every_k=10, noise0=n0, step_noise=sn)
    assert Kh.tolist() == g["K"].tolist()
    steps = torch.stack(inter["Lrk_inter"][1:]).cpu()
    assert tuple(steps.shape) == tuple(g["Lrk_steps"].shape)
    worst = max(rel_l2(steps[i], g["Lrk_steps"][i]) for i in range(steps.shape[0]))
    print(f"  worst logged step: {worst:.2e}")
    assert worst < (2e-5 if precision != "f16mx" else CONTRACT)


@pytest.mark.parametrize("precision", ACCURATE_MODES)
def test_ancestral_1000_step_chain_vs_reference(dev, precision):
    from drmnet_amd.obsnet import ObsNetDiffusion

    g = gold("ddpm_trace_1000")
    m = ObsNetDiffusion(unet_config=UNET_T, linear_start=float(g["linear_start"]), linear_end=float(g["linear_end"]), log_every_t=2000, timesteps=int(g["T"]),
                        first_stage_key="LrK", cond_stage_key="raw_refmap", padding_mode="noise", image_size=16, channels=3, concat_mode=True, ddim_steps=None,
                        clip_denoised=False, masked_loss=False, use_ema=False)
    synth.load_synth(m.model.diffusion_model, 21)
    m = m.to(dev).set_precision(precision)
    cond, x_T, noise = (t.to(dev) for t in long_obs_inputs(g))
    pred_x0, inter = m.p_sample_loop(cond, tuple(x_T.shape), return_intermediates=True, x_T=x_T, verbose=False, noise=noise)
    e_x, e_0 = rel_l2(inter["x_inter"][-1].cpu(), g["x"]), rel_l2(pred_x0.cpu(), g["pred_x0"])
    print(f"ancestral 1000-step chain ({precision}): x_0 {e_x:.2e}, pred_x0 {e_0:.2e}")
    assert e_x < (2e-5 if precision != "f16mx" else CONTRACT) and e_0 < (2e-5 if precision != "f16mx" else CONTRACT)


# --------------------------------------------------------------------------------------------- weight stress


@pytest.mark.parametrize("precision", ACCURATE_MODES)
@pytest.mark.parametrize("gain", [3, 10])
@pytest.mark.parametrize("name,cfg,kind", [("illnet", ou.ILLNET_CFG, "unet"), ("refnet", ou.REFNET_CFG, "encoder"), ("obsnet", ou.OBSNET_CFG, "unet")])
def test_stress_weights_vs_reference(dev, name, cfg, kind, gain, precision):
    """gain 3: within the contract of the fp32 reference (which is itself within 3e-6 of fp64).  gain 10: the fp32 reference is 4e-6 ... 3e-2
    from the same network in fp64 (attention logits x 100: every fp32 evaluation is a different draw of that noise -- the exact-fp32 mode lands
    at 2.4 x the reference's distance on IllNet 64x64, 5.8 x on ObsNet): the HIP path must be no further from fp64 than 10 x the reference is, or inside the
    contract; what the case guards is a split mode falling OUT of that band (a range or scaling failure under large gains)."""
    from drmnet_amd.unet import EncoderUNetModel, UNetModel

    g = gold(f"stress{gain}_{name}")
    m = (UNetModel if kind == "unet" else EncoderUNetModel)(**cfg)
    synth.load_synth(m, int(g["seed"]), rule=f"stress:{gain}")
    # (the t-variates go through a division and a square root per weight: host CPUs differ in the last bit of some of them -- observed 1.5e-9
    # on the checksum between the build container and a GPU box -- which moves an output by ~1e-7, far below every bar here)
    assert synth.checksum(torch.cat([v.flatten() for v in m.state_dict().values()])) == pytest.approx(float(g["wsum"]), rel=1e-7)
    m = m.to(dev).set_precision(precision)
    for n, h, w in ((1, 64, 64), (2, 32, 64)):
        xc, t_emb = full_inputs(n, h, w)
        t = torch.from_numpy(g["t"])[:n].to(dev)
        out = m(xc.to(dev), t_emb=t_emb.to(dev)) if name == "illnet" else m(xc.to(dev), t)
        ref32, ref64 = g[f"out_{n}x{h}x{w}"], g[f"out64_{n}x{h}x{w}"]
        e32, e64, r64 = rel_l2(out.cpu(), ref32), rel_l2(out.cpu(), ref64), rel_l2(ref32, ref64)
        print(f"stress x{gain} {name} {n}x{h}x{w} ({precision}): vs fp32 reference {e32:.2e}, vs fp64 {e64:.2e} (the fp32 reference vs fp64: {r64:.2e})")
        assert torch.isfinite(out).all()
        if gain == 3:
            assert e32 < (NET_TOL[precision] if precision != "f16mx" else CONTRACT)
        else:
            # (noise-level comparison: exact fp32 lands at 2.4 x r64 on IllNet, 5.8 x on ObsNet 64x64.  f16mx carries ~20 x the rounding noise of
            # exact fp32 -- four-bit cross terms -- and these networks amplify it like any other: 9e-4 on RefNet 2x32x64 where fp32 sits at 4e-5.  It is
            # NOT expected to hold the contract under such gains; what is asserted is that it stays a noise effect (no range / saturation failure), and
            # test_auto_precision_measures_the_loaded_weights below asserts that the default "auto" mode sends such weights to f16x3)
            assert e64 < (max(100 * r64, CONTRACT) if precision == "f16mx" else max(10 * r64, 2e-5))
    del m
    torch.cuda.empty_cache()


# --------------------------------------------------------------------------------------------- reference-written checkpoints + EMA


@pytest.mark.parametrize("precision", ACCURATE_MODES)
def test_reference_written_drmnet_ckpt_under_ema_scope(dev, precision):
    from drmnet_amd.drmnet import DRMNet

    g = gold("ema_drmnet")
    T, B = int(g["T"]), int(g["B"])
    m = DRMNet(illnet_config=UNET_T, refnet_config=ENC_T, max_timesteps=T, image_size=16, concat_mode=True, use_ema=True, gamma=float(g["gamma"]),
               epsilon=float(g["epsilon"]), delta=float(g["delta"]), z0=[1, 1, 1, 1, 0, 1], brdf_param_names=["p"] * 6,
               ckpt_path=os.path.join(GOLD, "drmnet_tiny_ema.ckpt"))  # init_from_ckpt inside the constructor, as the reference's YAML path does
    assert int(m.illnet_model_ema.num_updates) == int(g["num_updates"]) and float(m.illnet_model_ema.decay) == pytest.approx(float(g["decay"]))
    m = m.to(dev).set_precision(precision)
    LrK = synth.synth_refmaps(B, 16, 32, int(g["input_seed"])).to(dev)
    gen = torch.Generator().manual_seed(int(g["gen_seed"]))
    n0 = torch.randn(LrK.shape, generator=gen).to(dev)
    sn = torch.randn((T,) + tuple(LrK.shape), generator=gen).to(dev)
    te = torch.randn((B, 32), generator=torch.Generator().manual_seed(int(g["temb_seed"]))).to(dev)
    x = torch.cat([LrK, LrK], 1)
    run = lambda: m.p_sample_loop(LrK, [LrK], [LrK], verbose=False, noise0=n0, step_noise=sn)
    tol = 2e-5 if precision != "f16mx" else CONTRACT
    live = run()
    with m.ema_scope("test"):
        inside = run()
        ill_ema = m.illnet_model.diffusion_model(x, t_emb=te)
    after = run()
    e_live, e_ema = rel_l2(live[0].cpu(), g["Lr0_live"]), rel_l2(inside[0].cpu(), g["Lr0_ema"])
    print(f"reference-written DRMNet ckpt ({precision}): live {e_live:.2e}, under ema_scope {e_ema:.2e}; live vs ema {rel_l2(g['Lr0_live'], g['Lr0_ema']):.2e}")
    assert live[2].tolist() == g["K"].tolist() == inside[2].tolist()
    assert e_live < tol and e_ema < tol and torch.equal(after[0], live[0])
    assert rel_l2(ill_ema.cpu(), g["illnet_ema"]) < tol and rel_l2(m.illnet_model.diffusion_model(x, t_emb=te).cpu(), g["illnet_live"]) < tol
    assert rel_l2(g["Lr0_live"], g["Lr0_ema"]) > 100 * tol  # the two weight sets are far apart on this scale


@pytest.mark.parametrize("precision", ACCURATE_MODES)
def test_reference_written_obsnet_ckpt_under_ema_scope(dev, precision):
    from drmnet_amd.obsnet import ObsNetDiffusion

    g = gold("ema_obsnet")
    B = int(g["B"])
    m = ObsNetDiffusion(unet_config=UNET_T, linear_start=1e-4, linear_end=0.09, log_every_t=2000, timesteps=1000, first_stage_key="LrK",
                        cond_stage_key="raw_refmap", padding_mode="noise", image_size=16, channels=3, concat_mode=True, ddim_steps=50, clip_denoised=False,
                        masked_loss=False, use_ema=True, ckpt_path=os.path.join(GOLD, "obsnet_tiny_ema.ckpt"))
    m = m.to(dev).set_precision(precision)
    gen = torch.Generator().manual_seed(int(g["gen_seed"]))
    cond = (synth.synth_refmaps(B, 16, 16, int(g["input_seed"])) * 2 - 1).to(dev)
    x_T = torch.randn((B, 3, 16, 16), generator=gen).to(dev)
    noise = torch.randn((50, B, 3, 16, 16), generator=gen).to(dev)
    kw = dict(cond=cond, batch_size=B, ddim=True, ddim_steps=50, eta=1.0, x_T=x_T, noise=noise, num_steps=int(g["steps"]))
    tol = 2e-5 if precision != "f16mx" else CONTRACT
    live = m.sample_log(**kw)[0]
    with m.ema_scope("Plotting"):
        inside = m.sample_log(**kw)[0]
        eps = m.apply_model(x_T, torch.full((B,), 981, dtype=torch.long, device=dev), [cond])
    e_live, e_ema = rel_l2(live.cpu(), g["x_live"]), rel_l2(inside.cpu(), g["x_ema"])
    print(f"reference-written ObsNet ckpt ({precision}): live {e_live:.2e}, under ema_scope {e_ema:.2e}")
    assert e_live < tol and e_ema < tol and rel_l2(eps.cpu(), g["eps_ema"]) < tol
    assert torch.equal(m.sample_log(**kw)[0], live)


# --------------------------------------------------------------------------------------------- resize


def test_resize_vs_reference(dev):
    from drmnet_amd.dataset import BaseDataset

    g = gold("resize")
    hdr, rect, big, mask = (torch.from_numpy(g[k]).to(dev) for k in ("hdr", "rect", "big", "mask"))
    tr = lambda size, func, x: BaseDataset(size=size, transform_func=func).transform(x).cpu()
    errs = {
        "resize_only": rel_l2(tr(16, "resize", hdr), g["resize_only"]),
        "log_of_resized": rel_l2(tr(16, "log_resize", hdr), g["log_of_resized"]),
        "resized_log": rel_l2(tr(16, "resize_log", hdr), g["resized_log"]),
        "rect_16": rel_l2(tr(16, "resize", rect), g["rect_16"]),
        "rect_24": rel_l2(tr(24, "resize", rect), g["rect_24"]),
        "big_48": rel_l2(tr(48, "resize", big), g["big_48"]),
        "bicubic_16": rel_l2(tr(16, "resizeBICUBIC", hdr), g["bicubic_16"]),
    }
    print("resize:", {k: f"{v:.1e}" for k, v in errs.items()})
    assert max(errs.values()) < 1e-6
    assert tuple(tr(16, "resize", rect).shape) == (3, 16, 16)
    assert torch.equal(tr(16, "resizeNEAREST", hdr), torch.from_numpy(g["nearest_16"]))
    for key, src, size in (("mask_16", mask, (16, 16)), ("mask_64", mask, (64, 64)), ("mask_rect", mask[:, :, :24, :].contiguous(), (16, 16))):
        assert torch.equal(ops.resize(src, size, "nearest").cpu(), torch.from_numpy(g[key])), key
    with pytest.raises(NotImplementedError):
        ops.resize(hdr, (16, 16), "lanczos")
    with pytest.raises(RuntimeError):
        ops.resize(hdr.cpu(), (16, 16))
    with pytest.raises(RuntimeError):  # beyond the kernel's tap budget: an argument error, not a silent truncation
        ops.resize(torch.ones((1, 1, 16, 4096), device=dev), (16, 16), "bicubic")


# --------------------------------------------------------------------------------------------- "auto": f16mx only where it measurably holds


@pytest.mark.parametrize("rule,expect", [("normal", "f16mx"), ("stress:10", "f16x3")])
def test_auto_precision_measures_the_loaded_weights(dev, rule, expect):
    """set_precision("auto") (bench.py's default): one seeded probe forward in f16x3 and one in f16mx on the weights actually loaded; f16mx is kept
    only if the two agree to 5e-5.  The synthetic weights of the benches and fixtures pass (RefNet ~3e-6); the same network with GroupNorm gains x 10
    amplifies rounding noise forty-fold (f16mx 9e-4 against the reference, f16x3 1e-5) and must be sent to f16x3; new weights are re-measured."""
    from drmnet_amd.unet import EncoderUNetModel

    g = gold("stress10_refnet")
    m = EncoderUNetModel(**ou.REFNET_CFG)
    synth.load_synth(m, int(g["seed"]), rule=rule)
    m = m.to(dev).set_precision("auto")
    assert m.auto_report is None  # nothing measured before the first forward
    xc, _ = full_inputs(2, 32, 64)
    out = m(xc.to(dev), torch.from_numpy(g["t"])[:2].to(dev))
    rep = m.auto_report
    print(f"auto precision on RefNet with {rule} weights: {rep}")
    assert rep["chosen"] == expect == m.precision and (rep["rel_l2_f16mx_vs_f16x3"] <= rep["tolerance"]) == (expect == "f16mx")
    if rule != "normal":
        e = rel_l2(out.cpu(), g["out64_2x32x64"])
        assert e < 1e-4  # (in f16x3 the stressed network is back inside the contract: 1.4e-5)
        synth.load_synth(m, synth.SEED_REFNET)  # other weights -> measured again on the next forward
        m(xc.to(dev), torch.from_numpy(g["t"])[:2].to(dev))
        assert m.auto_report["chosen"] == "f16mx" == m.precision
    assert m.set_precision("f16x3").auto_report is None  # an explicit mode leaves auto


def test_auto_precision_probe_rows_and_cache(dev):
    """[r5] The per-network probe is a batch (two inputs x three timesteps, worst row decides) and its reports are kept per (weight set, weight
    signature): toggling between the live weights and an EMA shadow does not repeat a measurement."""
    from drmnet_amd.unet import EncoderUNetModel

    m = EncoderUNetModel(**ou.TINY_ENC_CFG)
    synth.load_synth(m, 5)
    m = m.to(dev).set_precision("auto")
    rep = m.calibrate_precision()
    assert len(rep["rows"]) == 6 and rep["rel_l2_f16mx_vs_f16x3"] == pytest.approx(max(rep["rows"]), rel=1e-3)
    cache = m.__dict__["_auto"]["cache"]
    assert len(cache) == 1
    ema = [p.detach().clone() * 1.01 for p in m.param_tensors()]
    m.use_weights("ema", ema)
    rep_e = m.calibrate_precision()
    assert len(cache) == 2 and rep_e is not rep
    m.use_weights("live")
    assert m.calibrate_precision() is rep  # the stored report, no new measurement
    m.use_weights("ema", ema)
    assert m.calibrate_precision() is rep_e and len(cache) == 2


def test_auto_precision_chain_probe_overrules_the_single_forward_probe(dev):
    """[r5, VERDICT r4 item 6] A per-network probe compares one forward; the samplers apply ~100 of them to their own output.  In auto mode DRMNet runs
    eight reverse steps (ObsNet: eight DDIM steps) in the chosen modes and in f16x3 and keeps f16mx only if the chains agree to half the contract.
    Weights that pass the single-forward probe but fail the chain probe (here: the chain tolerance put below what f16mx can deliver) must land on
    f16x3 -- both networks -- while the per-network reports still show that each one passed on its own."""
    from drmnet_amd.drmnet import DRMNet
    from drmnet_amd.obsnet import ObsNetDiffusion

    unet_t = {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": dict(ou.TINY_UNET_CFG)}
    enc_t = {"target": "ldm.modules.diffusionmodules.openaimodel.EncoderUNetModel", "params": dict(ou.TINY_ENC_CFG)}

    def build():
        m = DRMNet(illnet_config=unet_t, refnet_config=enc_t, max_timesteps=12, image_size=16, concat_mode=True, use_ema=False, gamma=0.9, epsilon=0.01, delta=0.025,
                   z0=[1, 1, 1, 1, 0, 1], brdf_param_names=["p"] * 6)
        synth.load_synth(m.illnet_model.diffusion_model, 21)
        synth.load_synth(m.refnet_model.diffusion_model, 22)
        zman = [(k, tuple(v.shape)) for k, v in m.illnet_model.z_emb_layer.state_dict().items()]
        m.illnet_model.z_emb_layer.load_state_dict(synth.synth_state_dict(zman, synth.SEED_ZEMB))
        return m.to(dev)

    def auto(m, chain_tol):  # (the tiny networks amplify rounding noise more than the shipped ones: 1e-4 per forward in f16mx -- the test is about the mechanism,
        m.AUTO_CHAIN_TOLERANCE = chain_tol  #  so the per-network bar is put where they pass it and the chain bar decides)
        m.set_precision("auto")
        for net in (m.illnet_model.diffusion_model, m.refnet_model.diffusion_model):
            net.set_precision_auto(tolerance=1e-3)
        return m

    m = auto(build(), 1e-1)
    rep = m.calibrate_precision()
    ill, ref = m.illnet_model.diffusion_model, m.refnet_model.diffusion_model
    print(f"chain probe (tiny DRMNet): {rep}; illnet {ill.auto_report}; refnet {ref.auto_report}")
    assert rep["kept"] and rep["rel_l2_chain_vs_f16x3"] <= rep["tolerance"] and ill.precision == ref.precision == "f16mx"
    LrK = synth.synth_refmaps(2, 16, 32, 5).to(dev)
    out_auto = m.p_sample_loop(LrK, [LrK], [LrK], verbose=False, seed=3, early_exit=False)[0]
    # [r6] the first batch the sampler sees is handed to the chain probe (rows of the caller, once per weight signature): that record stands from then on
    rep_d = m.auto_chain_report
    assert rep_d is not rep and rep_d["probe_source"] == "caller" and rep["probe_source"] == "synthetic" and rep_d["kept"]
    m.p_sample_loop(LrK, [LrK], [LrK], verbose=False, seed=4, early_exit=False)
    assert m.calibrate_precision() is rep_d  # measured once for these weights

    # the case the chain probe exists for, as it occurs: every network passes its single-forward probe at 1e-3 (1.2e-4 / 1.7e-5), and eight steps of the
    # undamped tiny sampler blow the f16mx-vs-f16x3 difference up to ~3e-3 -- the SAME bar applied to the chain sends both networks to f16x3
    m2 = auto(build(), 1e-3)
    rep2 = m2.calibrate_precision()
    assert rep2["rel_l2_chain_vs_f16x3"] > 1e-3 > max(m2.illnet_model.diffusion_model.auto_report["rel_l2_f16mx_vs_f16x3"], m2.refnet_model.diffusion_model.auto_report["rel_l2_f16mx_vs_f16x3"])
    ill2, ref2 = m2.illnet_model.diffusion_model, m2.refnet_model.diffusion_model
    assert not rep2["kept"] and ill2.precision == ref2.precision == "f16x3"
    assert ill2.auto_report["chosen"] == "f16x3" and "chain probe" in ill2.auto_report["overridden_by"]
    assert ill2.auto_report["rel_l2_f16mx_vs_f16x3"] <= ill2.auto_report["tolerance"]  # ... although the single forward had passed
    out_x3 = m2.p_sample_loop(LrK, [LrK], [LrK], verbose=False, seed=3, early_exit=False)[0]
    m3 = build().set_precision("f16x3")
    assert torch.equal(out_x3, m3.p_sample_loop(LrK, [LrK], [LrK], verbose=False, seed=3, early_exit=False)[0])  # it really runs in f16x3
    assert rel_l2(out_auto.cpu(), out_x3.cpu()) < 1e-1  # (tiny undamped networks, 12 steps in f16mx: see above)

    # ObsNet: the same mechanism on the first eight DDIM steps
    obs = ObsNetDiffusion(unet_config=unet_t, linear_start=1e-4, linear_end=0.09, log_every_t=2000, timesteps=1000, first_stage_key="LrK",
                          cond_stage_key="raw_refmap", padding_mode="noise", image_size=16, channels=3, concat_mode=True, ddim_steps=50,
                          clip_denoised=False, masked_loss=False, use_ema=False)
    synth.load_synth(obs.model.diffusion_model, 23)
    obs = obs.to(dev).set_precision("auto")
    obs.model.diffusion_model.set_precision_auto(tolerance=1e-3)
    obs._auto_chain["tolerance"] = 1e-3
    rep_o = obs.calibrate_precision()
    print(f"chain probe (tiny ObsNet): {rep_o}")
    assert rep_o is not None and rep_o["kept"] == (obs.model.diffusion_model.precision == "f16mx")
    obs._auto_chain["done"].clear()
    obs._auto_chain["tolerance"] = 1e-9
    obs.model.diffusion_model._set_mode("f16mx")
    obs._auto_chain_probe()
    assert obs.model.diffusion_model.precision == "f16x3" and not obs.auto_chain_report["kept"]


def test_auto_precision_gates_on_the_callers_batch(dev):
    """[r6, VERDICT r5 item 5b] The auto gate decides on the CALLER's data: p_sample_loop / ddim_sampling hand rows of their first batch to the chain
    probe (set_precision("auto", probe=...) / calibrate_precision(probe=...) take them explicitly), so the error that decides is the one measured on
    those rows, not on the seeded synthetic pair.  Shown on the tiny sampler: the caller-probe measures its own number (different rows, different
    size than the synthetic probe); a tolerance just below that number flips both networks to f16x3 (bit-identical to an f16x3 model from then on),
    a tolerance just above keeps f16mx; another batch of the same model does not re-measure; new data through calibrate_precision(probe=) does."""
    from drmnet_amd.drmnet import DRMNet
    from drmnet_amd.obsnet import ObsNetDiffusion

    unet_t = {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": dict(ou.TINY_UNET_CFG)}
    enc_t = {"target": "ldm.modules.diffusionmodules.openaimodel.EncoderUNetModel", "params": dict(ou.TINY_ENC_CFG)}

    def build(chain_tol, probe=None):
        m = DRMNet(illnet_config=unet_t, refnet_config=enc_t, max_timesteps=12, image_size=16, concat_mode=True, use_ema=False, gamma=0.9, epsilon=0.01, delta=0.025,
                   z0=[1, 1, 1, 1, 0, 1], brdf_param_names=["p"] * 6)
        synth.load_synth(m.illnet_model.diffusion_model, 21)
        synth.load_synth(m.refnet_model.diffusion_model, 22)
        zman = [(k, tuple(v.shape)) for k, v in m.illnet_model.z_emb_layer.state_dict().items()]
        m.illnet_model.z_emb_layer.load_state_dict(synth.synth_state_dict(zman, synth.SEED_ZEMB))
        m = m.to(dev)
        m.AUTO_CHAIN_TOLERANCE = chain_tol
        m.set_precision("auto", probe=probe)
        for net in (m.illnet_model.diffusion_model, m.refnet_model.diffusion_model):
            net.set_precision_auto(tolerance=1e-2)  # (per-network bar where the tiny networks pass it: the chain bar decides, as in the test above)
        return m

    LrK = synth.synth_refmaps(5, 16, 32, 5).to(dev)
    kw = dict(verbose=False, seed=3, early_exit=False)
    m = build(1.0)
    out = m.p_sample_loop(LrK, [LrK], [LrK], **kw)[0]
    rep = m.auto_chain_report
    e = rep["rel_l2_chain_vs_f16x3"]
    print(f"caller-batch chain probe (tiny DRMNet, rows 0 and 2 of a 5x3x16x32 batch): {rep}")
    assert rep["probe_source"] == "caller" and rep["probe"].startswith("2x3x16x32 rows of the caller") and rep["kept"] and 0 < e < 1.0
    rep_s = build(1.0).calibrate_precision()  # no data yet: the synthetic pair at 128 x 128 -- another measurement
    assert rep_s["probe_source"] == "synthetic" and rep_s["rel_l2_chain_vs_f16x3"] != e
    m.p_sample_loop(LrK.flip(0), [LrK.flip(0)], [LrK.flip(0)], **kw)
    assert m.auto_chain_report is rep  # the next batch re-measures nothing
    # the same rows against a bar just below / just above what they measure
    m_lo, m_hi = build(e * 0.5), build(e * 2.0)
    out_lo = m_lo.p_sample_loop(LrK, [LrK], [LrK], **kw)[0]
    out_hi = m_hi.p_sample_loop(LrK, [LrK], [LrK], **kw)[0]
    assert not m_lo.auto_chain_report["kept"] and m_lo.illnet_model.diffusion_model.precision == m_lo.refnet_model.diffusion_model.precision == "f16x3"
    assert "chain probe" in m_lo.illnet_model.diffusion_model.auto_report["overridden_by"]
    assert m_hi.auto_chain_report["kept"] and "f16mx" in (m_hi.illnet_model.diffusion_model.precision, m_hi.refnet_model.diffusion_model.precision)
    assert abs(m_lo.auto_chain_report["rel_l2_chain_vs_f16x3"] - e) <= 1e-3 * e and abs(m_hi.auto_chain_report["rel_l2_chain_vs_f16x3"] - e) <= 1e-3 * e
    x3 = build(1.0).set_precision("f16x3")
    assert torch.equal(out_lo, x3.p_sample_loop(LrK, [LrK], [LrK], **kw)[0])  # flipped: it really runs in f16x3
    assert torch.equal(out_hi, out)
    # explicit probe rows: set_precision("auto", probe=...) measures on them before any sampling; calibrate_precision(probe=...) re-measures on new ones
    hdr = LrK * 1.0e4  # (a 1e4-range HDR refmap batch)
    m_p = build(1.0, probe=hdr)
    rep_p = m_p.calibrate_precision()
    assert rep_p["probe_source"] == "caller" and rep_p["rel_l2_chain_vs_f16x3"] != e
    rep_q = m_p.calibrate_precision(probe=LrK)
    print(f"1e4-range HDR rows: {rep_p['rel_l2_chain_vs_f16x3']:.2e}; ordinary rows {rep_q['rel_l2_chain_vs_f16x3']:.2e}")
    assert rep_q is not rep_p and abs(rep_q["rel_l2_chain_vs_f16x3"] - e) <= 1e-3 * e

    # ObsNet: the conditioning of the first sampler call
    from drmnet_amd.ddim import DDIMSampler

    obs = ObsNetDiffusion(unet_config=unet_t, linear_start=1e-4, linear_end=0.09, log_every_t=2000, timesteps=1000, first_stage_key="LrK",
                          cond_stage_key="raw_refmap", padding_mode="noise", image_size=16, channels=3, concat_mode=True, ddim_steps=50,
                          clip_denoised=False, masked_loss=False, use_ema=False)
    synth.load_synth(obs.model.diffusion_model, 23)
    obs = obs.to(dev).set_precision("auto")
    obs.model.diffusion_model.set_precision_auto(tolerance=1e-2)
    obs._auto_chain["tolerance"] = 1.0
    smp = DDIMSampler(obs)
    smp.make_schedule(50, ddim_eta=1.0, verbose=False)
    x, _ = smp.ddim_sampling(LrK, tuple(LrK.shape), seed=2, log_every_t=0, verbose=False)
    rep_o = obs.auto_chain_report
    print(f"caller-batch chain probe (tiny ObsNet): {rep_o}")
    assert rep_o["probe_source"] == "caller" and rep_o["probe"].startswith("2x3x16x32 conditioning rows of the caller") and torch.isfinite(x).all()
    eo = rep_o["rel_l2_chain_vs_f16x3"]
    obs2 = ObsNetDiffusion(unet_config=unet_t, linear_start=1e-4, linear_end=0.09, log_every_t=2000, timesteps=1000, first_stage_key="LrK",
                           cond_stage_key="raw_refmap", padding_mode="noise", image_size=16, channels=3, concat_mode=True, ddim_steps=50,
                           clip_denoised=False, masked_loss=False, use_ema=False)
    synth.load_synth(obs2.model.diffusion_model, 23)
    obs2 = obs2.to(dev).set_precision("auto")
    obs2.model.diffusion_model.set_precision_auto(tolerance=1e-2)
    obs2._auto_chain["tolerance"] = eo * 0.5
    obs2.p_sample_loop(LrK, tuple(LrK.shape), verbose=False, start_T=3, seed=2)  # (the ancestral loop hands its conditioning over too)
    assert not obs2.auto_chain_report["kept"] and obs2.model.diffusion_model.precision == "f16x3" and obs2.auto_chain_report["probe_source"] == "caller"


# ------------------------------------------------------------------------------------------------------------------------------
# Round 4, second half: the sparse-launch forms (batch-1 step).  AttentionBlock (openaimodel.py:278-333 over QKVAttentionLegacy
# :365-381) on the short-sequence path -- qk_small_kernel, row softmax inside the P v GEMM, per-image range guard of q / k / v --
# and the GroupNorm finalisation folded into the consumer conv's prologue (launches of <= 4 images).
# ------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ch,h,w", [(640, 8, 8), (768, 4, 4), (512, 16, 16)])
@pytest.mark.parametrize("precision", ["f16x3", "f16mx"])
def test_short_sequence_attention_keeps_its_range_guard(dev, ch, h, w, precision):
    """T = 64, 16 and (sparse launch) 256 with the value rows 3e4 x the usual size (|v| ~ 1e5 > fp16's 65504) and q 30 x / k 1/30 x: the staging
    powers of two are per image and per operand on this path too.  fp64 oracle; rows of a batch == single runs."""
    from test_gpu_ops import attn_manifest, block_inputs

    xa, _ = block_inputs(ch, ch, h, w, 2)
    xa[1] *= 5.0
    P = synth.synth_state_dict(attn_manifest(ch), 32)
    w_, b_ = P["qkv.weight"].clone(), P["qkv.bias"].clone()
    for lo, f in ((0, 30.0), (ch, 1.0 / 30.0), (2 * ch, 3e4)):
        w_[lo:lo + ch] *= f
        b_[lo:lo + ch] *= f
    P["qkv.weight"], P["qkv.bias"] = w_, b_
    P["proj_out.weight"] = P["proj_out.weight"] * 1e-4
    with ou.working_dtype(torch.float64):
        ref = ou.attention_block({"ab." + k: v.double() for k, v in P.items()}, ou.Attn("ab", ch), xa.double())
    try:
        ops.set_precision(precision)
        got = ops.attention_block([p.to(dev) for p in P.values()], xa.to(dev)).cpu()
        one = ops.attention_block([p.to(dev) for p in P.values()], xa[1:2].contiguous().to(dev)).cpu()
    finally:
        ops.set_precision("fp32")
    e = rel_l2(got - xa, ref - xa.double())  # the residual x is added exactly: the attention branch itself
    print(f"short-sequence attention {ch} @{h}x{w} ({precision}), v x 3e4, q x 30, k / 30: {e:.2e}")
    assert torch.isfinite(got).all() and e < 1e-4
    assert rel_l2(one[0], got[1]) < 1e-6


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "f16mx"])
def test_sparse_and_dense_launch_forms_agree(dev, precision):
    """The same images through the sparse-launch forms (<= 4 images: GroupNorm tables finalised in the conv prologue, short-sequence attention at
    16x16, wave-per-feature emb linears) and inside a batch of 6 (gn_finalize launches, conv-pipeline attention): the tiny and the full-width
    IllNet, rows compared pairwise."""
    from test_gpu_nets import build, full_inputs

    for cfg, kind, shape in ((ou.TINY_UNET_CFG, "unet", (6, 6, 16, 32)), (ou.ILLNET_CFG, "unet", (6, 6, 64, 64))):
        m = build(cfg, kind, 5, dev)
        gen = torch.Generator().manual_seed(17)
        x = torch.randn(shape, generator=gen)
        te = torch.randn((shape[0], cfg["model_channels"]), generator=gen)
        try:
            m.set_precision(precision)
            dense = m(x.to(dev), t_emb=te.to(dev)).cpu()
            sparse = torch.cat([m(x[i:i + 2].contiguous().to(dev), t_emb=te[i:i + 2].contiguous().to(dev)).cpu() for i in (0, 2, 4)])
        finally:
            m.set_precision("fp32")
        e = rel_l2(sparse, dense)
        print(f"sparse vs dense launch forms, {'tiny' if cfg is ou.TINY_UNET_CFG else 'full-width'} IllNet ({precision}): {e:.2e}")
        # (f16mx: the two forms run other tile families, i.e. two realisations of the mode's ~2e-5 rounding noise; the exact modes differ by summation order only)
        assert torch.isfinite(dense).all() and e < (NET_TOL["f16mx"] if precision == "f16mx" else 2e-6)
