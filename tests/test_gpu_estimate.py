"""GPU parity of the whole single-image chain (drmnet_amd/estimate.py, the mirror of scripts/estimate.py) against a trace
recorded from the reference on its data/sample inputs with 16x16 tiny networks (tests/golden/estimate_chain.npz,
tools/make_golden.py --only estimate_chain): erosion -> refmap -> ObsNet cond -> DDIM-50 -> rescale -> DRMNet loop ->
rescale -> r0toenvmap -> hdr2ldr, with the reference's random draws injected."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, gold, rel_l2
from drmnet_amd import file_io
from drmnet_amd.dataset import BaseDataset
from drmnet_amd import synth
from drmnet_amd.config import instantiate_from_config, load_config
from oracle import unet as ou

ROOT = os.path.dirname(os.path.dirname(GOLD))


def tiny_models(g, dev):
    """The reference's eval configs (same YAML surface, configs/) with the networks swapped for the 16x16 tiny ones, exactly
    as tools/make_golden.py builds the reference side."""
    dcfg = load_config(os.path.join(ROOT, "configs/drmnet/eval_drmnet.yaml"))
    mp = dcfg["model"]["params"]
    mp.pop("ckpt_path", None)
    mp["illnet_config"] = {"target": mp["illnet_config"]["target"], "params": dict(ou.TINY_UNET_CFG)}
    mp["refnet_config"] = {"target": mp["refnet_config"]["target"], "params": dict(ou.TINY_ENC_CFG)}
    mp.update(image_size=16, gamma=float(g["gamma"]), epsilon=float(g["epsilon"]), max_timesteps=int(g["max_timesteps"]), delta=float(g["delta"]),
              use_ema=False)
    drm = instantiate_from_config(dcfg["model"])
    synth.load_synth(drm.illnet_model.diffusion_model, 21)
    synth.load_synth(drm.refnet_model.diffusion_model, 22)
    drm.illnet_model.z_emb_layer.load_state_dict(synth.synth_state_dict(
        [(k, tuple(v.shape)) for k, v in drm.illnet_model.z_emb_layer.state_dict().items()], synth.SEED_ZEMB))
    isd = drm.illnet_model.diffusion_model.state_dict()
    isd["out.2.weight"] = isd["out.2.weight"] * float(g["ill_out_scale"])
    isd["out.2.bias"] = isd["out.2.bias"] * float(g["ill_out_scale"])
    drm.illnet_model.diffusion_model.load_state_dict(isd)
    sd = drm.refnet_model.diffusion_model.state_dict()
    sd["out.3.weight"] = sd["out.3.weight"] * float(g["head_w_scale"])
    sd["out.3.bias"] = torch.from_numpy(g["head_bias"])
    drm.refnet_model.diffusion_model.load_state_dict(sd)
    drm.ds = instantiate_from_config({"target": dcfg["data"]["params"]["predict"]["target"],
                                      "params": dict(dcfg["data"]["params"]["predict"]["params"], size=16)})
    ocfg = load_config(os.path.join(ROOT, "configs/obsnet/eval_obsnet.yaml"))
    op = ocfg["model"]["params"]
    op.pop("ckpt_path", None)
    op["unet_config"] = {"target": op["unet_config"]["target"], "params": dict(ou.TINY_UNET_CFG)}
    op.update(image_size=16, use_ema=False, linear_start=float(g["obs_linear_start"]), linear_end=float(g["obs_linear_end"]))
    obs = instantiate_from_config(ocfg["model"])
    synth.load_synth(obs.model.diffusion_model, 21)
    osd = obs.model.diffusion_model.state_dict()
    osd["out.2.weight"] = osd["out.2.weight"] * float(g["obs_out_scale"])
    osd["out.2.bias"] = osd["out.2.bias"] * float(g["obs_out_scale"])
    obs.model.diffusion_model.load_state_dict(osd)
    obs.ds = instantiate_from_config({"target": ocfg["data"]["params"]["predict"]["target"],
                                      "params": dict(ocfg["data"]["params"]["predict"]["params"], size=16)})
    return drm.to(dev), obs.to(dev)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "f16mx"])
def test_estimate_chain_vs_reference_trace(precision):
    from drmnet_amd.estimate import estimate
    from drmnet_amd.transform import hdr2ldr

    assert torch.cuda.is_available(), "GPU tests need a GPU (no fallback)"
    dev = torch.device("cuda:0")
    g = gold("estimate_chain")
    drm, obs = tiny_models(g, dev)
    drm.set_precision(precision)
    obs.set_precision(precision)
    assert isinstance(drm.ds, BaseDataset) and obs.ddim_steps == 50

    d = os.path.join(GOLD, "sample")
    img = file_io.load_exr(os.path.join(d, "image.exr"), as_torch=True).to(dev)
    nrm = torch.from_numpy(np.load(os.path.join(d, "normal.npy"))).to(dev)
    mask = torch.logical_and(file_io.load_png(os.path.join(d, "mask.png"), as_torch=True).to(dev) > 0, torch.linalg.norm(nrm, dim=-1) > 0.5)
    stages = {}
    hooks = {"cond_noise": torch.from_numpy(g["cond"]).to(dev), "x_T": torch.from_numpy(g["x_T"]).to(dev), "noise": torch.from_numpy(g["noise"]).to(dev),
             "noise0": torch.from_numpy(g["noise0"]).to(dev), "step_noise": torch.from_numpy(g["step_noise"]).to(dev), "stages": stages}
    Lr0, zK = estimate(drm, obs, img, nrm, mask, hooks=hooks)
    env = drm.r0toenvmap(Lr0[None], (drm.image_size, drm.image_size * 2))[0]
    ldr = hdr2ldr(env.cpu().numpy())

    assert np.array_equal(stages["refmask"].cpu().numpy(), g["refmask"]) and np.array_equal(stages["refmap"].cpu().numpy(), g["refmap"])
    e = {k: rel_l2(stages[k].cpu(), g[k]) for k in ("cond", "inpaint", "LrK")}
    e["Lr0"] = rel_l2(Lr0.cpu(), g["Lr0"])
    e["envmap"] = rel_l2(env.cpu(), g["envmap"])
    print(f"estimate chain ({precision}):", {k: f"{v:.2e}" for k, v in e.items()}, "zK", zK.tolist(), "steps", drm.last_steps)
    assert np.isfinite(g["Lr0"]).all() and np.isfinite(g["envmap"]).all()
    assert e["cond"] < 1e-6

    assert e["inpaint"] < 1e-4 and e["LrK"] < 1e-4 and e["Lr0"] < 1e-4 and e["envmap"] < 1e-4  # the north-star tolerance, end to end
    log_err = np.abs(np.log10(Lr0.cpu().numpy() + 0.1) - np.log10(g["Lr0"] + 0.1)).max()
    assert log_err < 1e-4, log_err
    assert drm.last_steps == int(g["K"][0])
    assert np.allclose(zK.cpu().numpy(), g["zK"][0], atol=1e-5 if precision != "f16mx" else 1e-4, equal_nan=True)
    assert np.abs(ldr.astype(np.float64) - g["ldr"]).max() < 5e-3
