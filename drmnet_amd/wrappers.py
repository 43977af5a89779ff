"""Conditioning wrappers, EMA bookkeeping and the identity first stage (host glue around the HIP U-Nets).

Mirrors, by name and signature:
  DiffusionWrapper          ldm/models/diffusion/ddpm.py:1517-1543   (concat / no conditioning only)
  ZEmbDiffusionWrapper      models/drmnet.py:31-75
  LitEma                    ldm/modules/ema.py:5-76                  (store / copy_to / restore; buffer naming)
  IdentityFirstStage        ldm/models/autoencoder.py:420-437
"""
from __future__ import annotations

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
    """Shadow parameters as buffers named ``name.replace('.', '')`` (ema.py:16-21) so reference checkpoints load.
    Only the inference half (store / copy_to / restore) is implemented; the decay update is training-only."""

    def __init__(self, model, decay=0.9999, use_num_upates=True):
        super().__init__()
        if decay < 0.0 or decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.m_name2s_name = {}
        self.register_buffer("decay", torch.tensor(decay, dtype=torch.float32))
        self.register_buffer("num_updates", torch.tensor(0, dtype=torch.int) if use_num_upates else torch.tensor(-1, dtype=torch.int))
        for name, p in model.named_parameters():
            s_name = name.replace(".", "")
            self.m_name2s_name.update({name: s_name})
            self.register_buffer(s_name, p.clone().detach().data)
        self.collected_params = []

    def forward(self, model):
        raise NotImplementedError("EMA decay update is training-only (out of scope)")

    def copy_to(self, model):
        m_param = dict(model.named_parameters())
        shadow = dict(self.named_buffers())
        for key in m_param:
            m_param[key].data.copy_(shadow[self.m_name2s_name[key]].data)

    def store(self, parameters):
        self.collected_params = [param.clone() for param in parameters]

    def restore(self, parameters):
        for c_param, param in zip(self.collected_params, parameters):
            param.data.copy_(c_param.data)
