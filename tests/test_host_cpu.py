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


def test_dataset_plugin_and_transforms():
    ds = instantiate_from_config(model_cfg("configs/drmnet/eval_drmnet.yaml")["data"]["params"]["predict"])
    x = torch.rand(2, 3, 128, 128) * 3
    y = ds.transform(x)
    assert torch.allclose(y, torch.log10(x + 0.1) + 1)
    assert torch.allclose(ds.rescale(y), x, atol=1e-5)
    ds2 = instantiate_from_config(load_config(os.path.join(ROOT, "configs/obsnet/eval_obsnet.yaml"))["data"]["params"]["predict"])
    m = (torch.rand(2, 1, 128, 128) > 0.5).float()
    z = ds2.transform(x.clamp_min(1e-3), dynamic_normalize=True, mask=m)
    assert float((z * m).max()) <= 1.0 + 1e-5 and float((z * m + (1 - m)).min()) >= -1.0 - 1e-5
    assert torch.allclose(ds2.rescale(z), x.clamp_min(1e-3), rtol=1e-4, atol=1e-5)


def test_ema_scope_swaps_and_restores(drmnet):
    p = next(drmnet.illnet_model.parameters())
    before = p.detach().clone()
    shadow = drmnet.illnet_model_ema.state_dict()
    name = drmnet.illnet_model_ema.m_name2s_name[next(iter(dict(drmnet.illnet_model.named_parameters())))]
    shadow_t = dict(drmnet.illnet_model_ema.named_buffers())[name]
    shadow_t.add_(1.0)
    with drmnet.ema_scope():
        assert torch.equal(p, shadow_t)
    assert torch.equal(p, before)
    shadow_t.sub_(1.0)


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
