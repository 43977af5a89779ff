"""CPU checks of the host-side operator surface: YAML plugin loading, state_dict compatibility with the reference
checkpoints' key layout, schedule tables of the product code (bit-exact vs reference goldens)."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, gold
from drmnet_amd.config import instantiate_from_config, load_config


def model_cfg(rel):
    cfg = load_config(os.path.join(ROOT, rel))
    cfg["model"]["params"].pop("ckpt_path")
    return cfg


@pytest.fixture(scope="module")
def drmnet():
    return instantiate_from_config(model_cfg("configs/drmnet/eval_drmnet.yaml")["model"])


@pytest.fixture(scope="module")
def obsnet():
    return instantiate_from_config(model_cfg("configs/obsnet/eval_obsnet.yaml")["model"])


def test_drmnet_yaml_instantiates_and_state_dict_matches_reference(drmnet, manifests):
    from drmnet_amd.drmnet import DRMNet

    assert isinstance(drmnet, DRMNet)
    mine = [[k, list(v.shape)] for k, v in drmnet.state_dict().items()]
    assert mine == manifests["drmnet_model"]
    assert drmnet.max_timesteps == 150 and drmnet.gamma == 0.95 and drmnet.epsilon == 0.01 and drmnet.delta == 0.025
    assert drmnet.z0.tolist() == [1, 1, 1, 1, 0, 1] and drmnet.refmap_input_scaler == 0.12
    with pytest.raises(NotImplementedError):
        drmnet.p_sample(None, None, None, 0)  # stub in the reference too


def test_obsnet_yaml_instantiates_and_state_dict_matches_reference(obsnet, manifests):
    from drmnet_amd.obsnet import ObsNetDiffusion

    assert isinstance(obsnet, ObsNetDiffusion)
    mine = [[k, list(v.shape)] for k, v in obsnet.state_dict().items()]
    assert mine == manifests["obsnet_model"]
    assert obsnet.ddim_steps == 50 and obsnet.ddim_eta == 1.0 and obsnet.num_timesteps == 1000 and obsnet.clip_denoised is False


def test_dataset_plugin_loads_from_the_reference_yaml():
    from drmnet_amd.dataset import BaseDataset

    ds = instantiate_from_config(model_cfg("configs/drmnet/eval_drmnet.yaml")["data"]["params"]["predict"])
    assert isinstance(ds, BaseDataset) and ds.size == 128 and ds.transform_func_str == "log" and ds.clamp_before_exp == 20
    ds2 = instantiate_from_config(load_config(os.path.join(ROOT, "configs/obsnet/eval_obsnet.yaml"))["data"]["params"]["predict"])
    assert ds2.transform_func_str == "resize_0p1tom1p1_normalizedLogarithmic_lowerbound1e-6"
    with pytest.raises(RuntimeError):  # the maps are HIP kernels: CPU tensors fail loudly
        ds.transform(torch.rand(2, 3, 128, 128))


def test_ema_scope_swaps_parameters_and_selects_the_ema_weight_set(drmnet):
    """models/drmnet.py:242-258: inside the scope the module parameters hold the EMA values (as in the reference) AND both
    engines are pointed at their EMA weight image, whose source tensors are the shadow buffers themselves."""
    ill = drmnet.illnet_model.diffusion_model
    p = next(drmnet.illnet_model.parameters())
    before = p.detach().clone()
    first = next(iter(dict(drmnet.illnet_model.named_parameters())))
    shadow_t = getattr(drmnet.illnet_model_ema, drmnet.illnet_model_ema.m_name2s_name[first])
    shadow_t.add_(1.0)
    assert drmnet._weight_set == "live" and ill._active_set == "live"
    with drmnet.ema_scope():
        assert torch.equal(p, shadow_t)
        assert drmnet._weight_set == "ema" and ill._active_set == "ema" and drmnet.refnet_model.diffusion_model._active_set == "ema"
        src = ill._ema_source
        assert len(src) == len(ill._keys)
        assert src[0].data_ptr() == getattr(drmnet.illnet_model_ema, "diffusion_model" + ill._keys[0].replace(".", "")).data_ptr()
    assert torch.equal(p, before)
    assert drmnet._weight_set == "live" and ill._active_set == "live"
    shadow_t.sub_(1.0)
    with pytest.raises(ZeroDivisionError):  # the scope restores on exceptions too
        with drmnet.ema_scope():
            1 / 0
    assert drmnet._weight_set == "live" and torch.equal(p, before)


def synth_reference_checkpoint(model, path, seed):
    """A checkpoint in the reference's layout: {"state_dict": {...}} with the live parameters, the LitEma shadow
    buffers under their dot-less names and num_updates / decay, as pytorch_lightning writes it for models/drmnet.py / ddpm.py."""
    g = torch.Generator().manual_seed(seed)
    weights = {k for k, _ in model.named_parameters()}
    sd = {}
    for k, v in model.state_dict().items():
        is_shadow = "_ema." in k and not k.endswith((".decay", ".num_updates"))
        if k in weights or is_shadow:
            sd[k] = torch.randn(v.shape, generator=g) * (0.05 if is_shadow else 0.02)
        else:
            sd[k] = v.clone()  # schedule tables, z0, EMA bookkeeping: persistent buffers with their true values
    torch.save({"state_dict": sd, "epoch": 3, "global_step": 1234, "pytorch-lightning_version": "1.9.0"}, path)
    return sd


def test_init_from_ckpt_loads_reference_layout(tmp_path, drmnet, obsnet, capsys):
    for model, seed, ema_attr, ema_key in ((drmnet, 5, "illnet_model_ema", "illnet_model_ema.diffusion_modelinput_blocks00weight"),
                                           (obsnet, 6, "model_ema", "model_ema.diffusion_modelinput_blocks00weight")):
        path = str(tmp_path / f"m{seed}.ckpt")
        original = {k: v.clone() for k, v in model.state_dict().items()}
        sd = synth_reference_checkpoint(model, path, seed)
        assert ema_key in sd and any(k.endswith("_ema.num_updates") or k == "model_ema.num_updates" for k in sd)
        model.init_from_ckpt(path, ignore_keys=["first_stage_model"])
        out = capsys.readouterr().out
        assert "with 0 missing and 0 unexpected keys" in out
        got = model.state_dict()
        for k in (ema_key, [k for k in sd if k.endswith("input_blocks.0.0.weight")][0]):
            assert torch.equal(got[k], sd[k]), k
        # ignore_keys drops by prefix and reports what it dropped
        model.init_from_ckpt(path, ignore_keys=[ema_attr])
        out = capsys.readouterr().out
        assert f"Deleting key {ema_key} from state_dict." in out and " 0 unexpected keys" in out and "Missing Keys" in out
        model.load_state_dict(original)  # the fixtures are shared with the other tests of this module


def test_product_schedule_tables_bit_exact(obsnet):
    g = gold("ddpm_schedule")
    for k, v in g.items():
        assert np.array_equal(getattr(obsnet, k).numpy(), v), k
    from drmnet_amd.ddim import DDIMSampler

    for eta in (0, 1):
        gd = gold(f"ddim_schedule_eta{eta}")
        s = DDIMSampler(obsnet)
        s.make_schedule(50, ddim_eta=float(eta), verbose=False)
        assert np.array_equal(s.ddim_timesteps, gd["timesteps"])
        assert np.array_equal(s.ddim_coef, gd["coef"])


def test_brdf_schedule_host_math(drmnet):
    g = gold("brdf_schedule")
    z_out = torch.from_numpy(g["z_out"])
    for i in (0, 1, 7, 50, 90, 149):
        zk, zK = drmnet.get_brdf_out(z_out, reversed_k=i)
        assert torch.equal(zk, torch.from_numpy(g[f"zk_{i}"]))
        assert torch.equal(drmnet.check_convergence(zk), torch.from_numpy(g[f"conv_{i}"]))


def test_constructors_accept_training_keys_and_reject_unknown_ones():
    """The reference's YAMLs carry training-only keys (losses, monitors, caches: models/drmnet.py:79-240, ddpm.py:60-135): they are accepted and
    ignored; a key neither the inference path nor that list knows is an error, and unsupported VALUES of known keys raise loudly."""
    from drmnet_amd.drmnet import DRMNet
    from drmnet_amd.obsnet import ObsNetDiffusion

    dparams = dict(model_cfg("configs/drmnet/eval_drmnet.yaml")["model"]["params"])
    tiny_u = {"target": dparams["illnet_config"]["target"], "params": dict(image_size=16, in_channels=6, out_channels=3, model_channels=32,
              attention_resolutions=[2], num_res_blocks=1, channel_mult=[1, 2], num_heads=1, resblock_updown=False, conv_resample=False)}
    tiny_e = {"target": dparams["refnet_config"]["target"], "params": dict(tiny_u["params"], out_channels=6, pool="adaptive")}
    base = dict(dparams, illnet_config=tiny_u, refnet_config=tiny_e, use_ema=False)
    m = DRMNet(**dict(base, loss_type="l1", monitor="val/loss", l_refmap_weight=2.0, cache_refmap=True, sigma=0.3))
    assert not hasattr(m, "l_refmap_weight") and not hasattr(m, "loss_type")
    with pytest.raises(TypeError):
        DRMNet(**dict(base, definitely_not_a_reference_key=1))
    with pytest.raises(NotImplementedError):
        DRMNet(**dict(base, scale_by_std=True))
    oparams = dict(model_cfg("configs/obsnet/eval_obsnet.yaml")["model"]["params"], unet_config=tiny_u, use_ema=False, image_size=16)
    o = ObsNetDiffusion(**dict(oparams, original_elbo_weight=0.5, l_simple_weight=2.0, use_positional_encodings=True))
    assert o.ddim_steps == 50 and o.clip_denoised is False and not hasattr(o, "l_simple_weight")
    with pytest.raises(TypeError):
        ObsNetDiffusion(**dict(oparams, not_a_key=True))
    with pytest.raises(NotImplementedError):
        ObsNetDiffusion(**dict(oparams, parameterization="x0"))
    # reference-style positional calls (models/drmnet.py:79-85, ddpm.py:60-64, models/obsnet.py:38-44) and cond_stage_forward
    rest = {k: v for k, v in base.items() if k not in ("illnet_config", "refnet_config", "renderer_config", "max_timesteps", "ckpt_path")}
    mp = DRMNet(tiny_u, tiny_e, None, 17, **rest)
    assert mp.max_timesteps == 17 and mp.renderer is None
    with pytest.raises(NotImplementedError):
        DRMNet(**dict(base, cond_stage_forward="encode"))
    from drmnet_amd.obsnet import DDPM

    d = DDPM(tiny_u, 200, "linear", use_ema=False, conditioning_key="concat")
    assert d.num_timesteps == 200
    orest = {k: v for k, v in oparams.items() if k not in ("cond_stage_key", "padding_mode", "ckpt_path")}
    op = ObsNetDiffusion(None, None, None, "raw_refmap", "zeros", **orest)
    assert op.cond_stage_key == "raw_refmap" and op.padding_mode == "zeros"
