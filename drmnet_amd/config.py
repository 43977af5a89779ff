"""YAML plugin mechanism of the reference, honoured verbatim.

The reference instantiates everything from ``{"target": "pkg.mod.Class", "params": {...}}`` nodes
(ldm/util.py:78-93; call sites scripts/estimate.py:121-125, ddpm.py:1518-1522, drmnet.py:194-196).
The shipped YAMLs under configs/ are accepted unchanged: their dotted ``target`` strings are mapped onto
this package's classes; unknown targets fall back to a normal import (so user plugins still work).
"""
from __future__ import annotations

import importlib
from typing import Any, Dict

TARGET_REMAP: Dict[str, str] = {
    "models.drmnet.DRMNet": "drmnet_amd.drmnet.DRMNet",
    "models.obsnet.ObsNetDiffusion": "drmnet_amd.obsnet.ObsNetDiffusion",
    "ldm.modules.diffusionmodules.openaimodel.UNetModel": "drmnet_amd.unet.UNetModel",
    "ldm.modules.diffusionmodules.openaimodel.EncoderUNetModel": "drmnet_amd.unet.EncoderUNetModel",
    "ldm.models.autoencoder.IdentityFirstStage": "drmnet_amd.wrappers.IdentityFirstStage",
    "dataset.basedataset.BaseDataset": "drmnet_amd.dataset.BaseDataset",
    # training-data renderers are out of scope (Mitsuba 3 / OptiX); DRMNet treats a None renderer as "basis_r0 supplied externally"
    "utils.mitsuba3_utils.MitsubaRefMapRenderer": "drmnet_amd.wrappers.NullRenderer",
    "utils.mitsuba3_utils.MitsubaOrthoRenderer": "drmnet_amd.wrappers.NullRenderer",
}


def get_obj_from_str(string: str, reload: bool = False):
    string = TARGET_REMAP.get(string, string)
    module, cls = string.rsplit(".", 1)
    mod = importlib.import_module(module)
    if reload:
        importlib.reload(mod)
    return getattr(mod, cls)


def _plain(x: Any) -> Any:
    """OmegaConf nodes -> plain containers (omegaconf is optional; the YAMLs are plain YAML)."""
    try:
        from omegaconf import OmegaConf  # type: ignore

        if OmegaConf.is_config(x):
            return OmegaConf.to_container(x, resolve=True)
    except Exception:
        pass
    return x


def instantiate_from_config(config):
    config = _plain(config)
    if "target" not in config:
        if config == "__is_first_stage__":
            return None
        elif config == "__is_unconditional__":
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**(config.get("params", dict()) or dict()))


def load_config(path: str) -> dict:
    """OmegaConf.load stand-in (scripts/estimate.py:120,123): the configs are plain YAML."""
    import yaml

    with open(path) as f:
        return yaml.safe_load(f)
