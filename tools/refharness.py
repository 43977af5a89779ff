"""Stub harness that lets the (Python) reference import and run on CPU in THIS container.

Only tools/make_golden.py uses it. It never travels to the GPU box as anything but
this source file; /root/reference is absent there and nothing at test/bench time
imports this module.

The reference cannot run unmodified here: pytorch_lightning,
omegaconf, mitsuba, cv2, torchvision and taming are not installed, and
DDIMSampler.register_buffer hard-codes "cuda" (ldm/models/diffusion/ddim.py:23-27).
We inject inert stand-ins for those *third-party imports* into sys.modules (no
reference source is copied or modified) and subclass two reference classes to skip
the Mitsuba render (models/drmnet.py:328-347) and the hard "cuda" device.
"""
from __future__ import annotations

import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


def _mod(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs() -> None:
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    class LightningModule(nn.Module):
        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    def rank_zero_only(fn):
        return fn

    class _Callback:
        pass

    pl = _mod("pytorch_lightning", LightningModule=LightningModule, Callback=_Callback)
    pl.utilities = _mod("pytorch_lightning.utilities", rank_zero_only=rank_zero_only)
    _mod("pytorch_lightning.utilities.distributed", rank_zero_only=rank_zero_only)

    class OmegaConf:
        @staticmethod
        def create(x):
            return x

    _mod("omegaconf", OmegaConf=OmegaConf)
    _mod("omegaconf.listconfig", ListConfig=list)
    _mod("mitsuba", variant=lambda: None, set_variant=lambda *_: None)
    _mod("cv2")
    tv = _mod("torchvision")
    tv.utils = _mod("torchvision.utils", make_grid=lambda *a, **k: None)

    class _IM:
        BILINEAR = "bilinear"
        BICUBIC = "bicubic"
        NEAREST = "nearest"

    def _resize(x, size, interpolation=None, antialias=True):
        """torchvision.transforms.functional.resize on a float tensor (torchvision 0.13.1, the reference's pin, functional_tensor.py
        resize): torch.nn.functional.interpolate(img[None] if 3-D, size, mode, align_corners=False for bilinear / bicubic, antialias
        for those two modes only) -- torchvision is absent from this image, its published tensor path is restated here so that the
        reference's BaseDataset can run a real resize for the fixtures (tools/make_golden.py --only resize)."""
        if tuple(x.shape[-2:]) == tuple(size):
            return x
        mode = str(interpolation).lower()
        lead = x.shape[:-2]
        flat = x.reshape(-1, 1, *x.shape[-2:]).float()
        if mode in ("bilinear", "bicubic"):
            out = torch.nn.functional.interpolate(flat, size=tuple(size), mode=mode, align_corners=False, antialias=bool(antialias))
        else:
            out = torch.nn.functional.interpolate(flat, size=tuple(size), mode=mode)
        return out.reshape(*lead, *size)

    tv.transforms = _mod(
        "torchvision.transforms",
        InterpolationMode=_IM,
        functional=_mod("torchvision.transforms.functional", resize=_resize),
    )
    _mod("taming")
    _mod("taming.modules")
    _mod("taming.modules.vqvae")
    _mod("taming.modules.vqvae.quantize", VectorQuantizer2=object)
    _mod("taming.modules.losses")
    _mod("taming.modules.losses.vqperceptual")


def load_yaml_params(relpath: str) -> dict:
    import yaml

    with open(f"{REFERENCE_ROOT}/{relpath}") as f:
        cfg = yaml.safe_load(f)
    return cfg


def ref_classes():
    """Returns (DRMNetNoRenderer, ObsNetDiffusion, CPUDDIM, openaimodel module)."""
    install_stubs()
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.modules.diffusionmodules import openaimodel
    from models.drmnet import DRMNet
    from models.obsnet import ObsNetDiffusion

    class DRMNetNoRenderer(DRMNet):
        def instantiate_brdf_model(self, config):
            self.renderer = None
            self.register_buffer("basis_r0", torch.ones(3, self.image_size, self.image_size), persistent=False)

    class CPUDDIM(DDIMSampler):
        def register_buffer(self, name, attr):
            setattr(self, name, attr)

    return DRMNetNoRenderer, ObsNetDiffusion, CPUDDIM, openaimodel
