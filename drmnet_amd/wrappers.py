"""Conditioning wrappers, EMA bookkeeping and the identity first stage (host glue around the HIP U-Nets).

Mirrors, by name and signature:
  DiffusionWrapper          ldm/models/diffusion/ddpm.py:1517-1543   (concat / no conditioning only)
  ZEmbDiffusionWrapper      models/drmnet.py:31-75
  LitEma                    ldm/modules/ema.py:5-76                  (store / copy_to / restore; buffer naming)
  ema_weights / load_checkpoint   the bodies of ema_scope / init_from_ckpt shared by DRMNet and DDPM
  IdentityFirstStage        ldm/models/autoencoder.py:420-437
"""
from __future__ import annotations

from contextlib import contextmanager
from typing import List, Optional

import torch
import torch.nn as nn

from . import ops
from .config import instantiate_from_config


class IdentityFirstStage(nn.Module):
    def __init__(self, *args, vq_interface=False, **kwargs):
        super().__init__()
        self.vq_interface = vq_interface

    def encode(self, x, *args, **kwargs):
        return x

    def decode(self, x, *args, **kwargs):
        return x

    def quantize(self, x, *args, **kwargs):
        if self.vq_interface:
            return x, None, [None, None, None]
        return x

    def forward(self, x, *args, **kwargs):
        return x


class NullRenderer:
    """Stand-in for the Mitsuba renderers (training data / basis_r0 render; out of scope)."""

    def __init__(self, refmap_res: int = 128, **kwargs):
        self.image_size = (refmap_res, refmap_res)
        self.envmap_size = (refmap_res, refmap_res * 2)
        self.spp = kwargs.get("spp", 0)

    def rendering(self, *a, **k):
        raise NotImplementedError("Mitsuba rendering is out of scope of the MI355X hot path; supply basis_r0 explicitly")


class DiffusionWrapper(nn.Module):
    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        self.diffusion_model = instantiate_from_config(diff_model_config)
        self.conditioning_key = conditioning_key
        assert self.conditioning_key in [None, "concat", "crossattn", "hybrid", "adm"]
        if self.conditioning_key not in (None, "concat"):
            raise NotImplementedError(f"conditioning_key={conditioning_key!r}: only None / 'concat' are on the shipped path")

    @property
    def device(self):
        return next(self.parameters()).device

    def forward(self, x, t, c_concat: list = None, c_crossattn: list = None, rows=None):
        if self.conditioning_key is None:
            return self.diffusion_model(x, t)
        cond = c_concat[0] if len(c_concat) == 1 else torch.cat(c_concat, dim=1)
        return self.diffusion_model.forward_parts(x, cond, timesteps=t, rows=rows)  # th.cat([x]+c_concat) folded into the input pack


class ZEmbDiffusionWrapper(DiffusionWrapper):
    def __init__(self, model_config, conditioning_key, z_dim, emb_z: bool = True, emb_z_crossattn: bool = False):
        super().__init__(model_config, conditioning_key)
        if emb_z_crossattn:
            raise NotImplementedError("emb_z_crossattn is not on the shipped path")
        self.emb_z = emb_z
        self.emb_z_crossattn = emb_z_crossattn
        if emb_z:
            mc = self.diffusion_model.model_channels
            # parameter holders only (same state_dict keys as the reference nn.Sequential); evaluated by ops.linear
            self.z_emb_layer = nn.Sequential(
                nn.Linear(z_dim, mc // 2), nn.SiLU(), nn.Linear(mc // 2, mc // 2), nn.SiLU(), nn.Linear(mc // 2, mc), nn.SiLU()
            )
            for p in self.z_emb_layer.parameters():
                p.requires_grad_(False)

    def z_emb_param_names(self) -> List[str]:
        return [f"z_emb_layer.{i}.{w}" for i in (0, 2, 4) for w in ("weight", "bias")]

    def z_emb_params(self) -> List[torch.Tensor]:
        l = self.z_emb_layer
        return [l[0].weight, l[0].bias, l[2].weight, l[2].bias, l[4].weight, l[4].bias]

    def embed_z(self, z):
        w = self.z_emb_params()
        h = ops.linear(z, w[0], w[1], silu_out=True)
        h = ops.linear(h, w[2], w[3], silu_out=True)
        return ops.linear(h, w[4], w[5], silu_out=True)

    def forward(self, x, z_emb, c_concat: list = None, c_crossattn: list = None, rows=None):
        z_emb = self.embed_z(z_emb) if self.emb_z else None
        if z_emb is None:
            raise NotImplementedError("emb_z=False is not on the shipped path")
        if self.conditioning_key is None or self.conditioning_key == "none":
            return self.diffusion_model(x, t_emb=z_emb)
        cond = c_concat[0] if len(c_concat) == 1 else torch.cat(c_concat, dim=1)
        return self.diffusion_model.forward_parts(x, cond, t_emb=z_emb, rows=rows)


class LitEma(nn.Module):
    """EMA shadow of a module's parameters, stored as buffers named ``name.replace('.', '')`` so the reference's checkpoints
    (``illnet_model_ema.diffusion_modelinput_blocks00weight`` ...; ldm/modules/ema.py:16-21) load by key.  Inference half only:
    ``store`` / ``copy_to`` / ``restore`` (ema.py:46-76) and ``shadow_for``; the decay update (``forward``) is training-only."""

    def __init__(self, model, decay=0.9999, use_num_upates=True):
        super().__init__()
        if not 0.0 <= decay <= 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.register_buffer("decay", torch.tensor(decay, dtype=torch.float32))
        self.register_buffer("num_updates", torch.tensor(0 if use_num_upates else -1, dtype=torch.int))
        self.m_name2s_name = {name: name.replace(".", "") for name, _ in model.named_parameters()}
        for name, p in model.named_parameters():
            self.register_buffer(self.m_name2s_name[name], p.detach().clone())
        self.collected_params = []

    def forward(self, model):
        raise NotImplementedError("EMA decay update is training-only (out of scope)")

    def shadow_for(self, names) -> List[torch.Tensor]:
        """The shadow buffers of the given parameter names, in that order (what the engine packs as its "ema" weight set)."""
        return [getattr(self, self.m_name2s_name[n]) for n in names]

    @torch.no_grad()
    def copy_to(self, model):
        for name, p in model.named_parameters():
            p.data.copy_(getattr(self, self.m_name2s_name[name]))

    @torch.no_grad()
    def store(self, parameters):
        self.collected_params = [p.detach().clone() for p in parameters]

    @torch.no_grad()
    def restore(self, parameters):
        for saved, p in zip(self.collected_params, parameters):
            p.data.copy_(saved)


@contextmanager
def ema_weights(owner, pairs, context=None):
    """Body of both models' ``ema_scope`` (models/drmnet.py:242-258, ldm/models/diffusion/ddpm.py:189-202).

    For every (wrapper, LitEma) pair the module parameters are swapped to the shadow values and back exactly as the reference
    does (anyone reading ``parameters()`` inside the scope sees EMA values), and the wrapped U-Net's engine is pointed at its
    second packed weight image, built from the shadow buffers themselves: the engine never depends on noticing the in-place
    swap, and entering / leaving the scope re-packs nothing.  ``owner._weight_set`` tells the owner which set is live (it keys
    the DRMNet sampler handle, whose z-embedding weights follow the same switch)."""
    if not owner.use_ema:
        yield None
        return
    for wrapper, ema in pairs:
        ema.store(wrapper.parameters())
        ema.copy_to(wrapper)
        unet = wrapper.diffusion_model
        unet.use_weights("ema", ema.shadow_for("diffusion_model." + k for k in unet._keys))
    owner._weight_set = "ema"
    if context is not None:
        print(f"{context}: Switched to EMA weights")
    try:
        yield None
    finally:
        for wrapper, ema in pairs:
            ema.restore(wrapper.parameters())
            wrapper.diffusion_model.use_weights("live")
        owner._weight_set = "live"
        if context is not None:
            print(f"{context}: Restored training weights")


def load_checkpoint(module, path, ignore_keys=(), into=None, verbose=True):
    """``init_from_ckpt`` of both models (models/drmnet.py:260-277, ldm/models/diffusion/ddpm.py:204-231): a torch pickle, optionally
    wrapped as {"state_dict": ...}; keys starting with an ``ignore_keys`` prefix are dropped; loaded non-strictly into ``into``
    (default: the whole module) and the missing / unexpected keys are reported.  Key layout: SURVEY.md 5."""
    try:
        blob = torch.load(path, map_location="cpu", weights_only=True)
    except Exception:  # Lightning checkpoints carry non-tensor objects (callback state, hyper-parameters)
        blob = torch.load(path, map_location="cpu", weights_only=False)
    state = blob["state_dict"] if "state_dict" in blob else blob
    for key in [k for k in state if any(k.startswith(prefix) for prefix in ignore_keys)]:
        print("Deleting key {} from state_dict.".format(key))
        del state[key]
    missing, unexpected = (module if into is None else into).load_state_dict(state, strict=False)
    print(f"Restored from {path} with {len(missing)} missing and {len(unexpected)} unexpected keys")
    if verbose and missing:
        print(f"Missing Keys: {missing}")
    if verbose and unexpected:
        print(f"Unexpected Keys: {unexpected}")
    return missing, unexpected
