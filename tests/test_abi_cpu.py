"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/drmnet_hip.h declares,
and its parameter table equals the reference state_dict() layout (no compute calls)."""
import ctypes as C
import os
import re

import pytest
import torch

from conftest import ROOT
from drmnet_amd import _lib
from oracle import unet as ou

HEADER = os.path.join(ROOT, "include", "drmnet_hip.h")


def header_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(drm_[a-z0-9_]+)\s*\(", txt)))


def test_library_loads_and_exports_every_declared_symbol():
    L = _lib.lib()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/drmnet_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == syms
    assert L.drm_abi_version() == _lib.ABI_VERSION == 3


def table(cfg, kind):
    from drmnet_amd.unet import EncoderUNetModel, UNetModel

    cls = UNetModel if kind == "unet" else EncoderUNetModel
    m = cls(**cfg)
    return m, [[k, list(v.shape)] for k, v in m.state_dict().items()]


@pytest.mark.parametrize(
    "name,cfg,kind",
    [("illnet", ou.ILLNET_CFG, "unet"), ("refnet", ou.REFNET_CFG, "encoder"), ("obsnet", ou.OBSNET_CFG, "unet"),
     ("tiny_unet", ou.TINY_UNET_CFG, "unet"), ("tiny_enc", ou.TINY_ENC_CFG, "encoder")],
)
def test_engine_param_table_matches_reference_state_dict(manifests, name, cfg, kind):
    m, tab = table(cfg, kind)
    assert tab == manifests[name]
    assert m.workspace_bytes(2, 128, 128) > 0 if name in ("illnet", "refnet", "obsnet") else m.workspace_bytes(2, 16, 16) > 0


def test_unsupported_configs_are_rejected_loudly():
    from drmnet_amd.unet import UNetModel

    base = dict(ou.TINY_UNET_CFG)
    for bad in (dict(use_spatial_transformer=True, context_dim=8), dict(resblock_updown=True), dict(conv_resample=True),
                dict(use_scale_shift_norm=True), dict(num_heads=2), dict(use_fp16=True), dict(dims=3)):
        cfg = dict(base)
        cfg.update(bad)
        with pytest.raises(NotImplementedError):
            UNetModel(**cfg)


def test_errors_do_not_cross_the_abi_as_exceptions():
    L = _lib.lib()
    d = _lib.UNetDesc()
    d.kind = 7
    h = C.c_void_p()
    assert L.drm_unet_create(C.byref(d), C.byref(h)) != 0
    assert b"kind" in L.drm_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(1)


def test_cpu_tensors_fail_loudly():
    from drmnet_amd.unet import UNetModel

    m = UNetModel(**ou.TINY_UNET_CFG)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 6, 16, 16), timesteps=torch.zeros(1, dtype=torch.long))


def test_workspace_query_rejects_bad_shapes():
    from drmnet_amd.unet import UNetModel

    m = UNetModel(**ou.TINY_UNET_CFG)
    assert m.workspace_bytes(1, 10, 10) == 0  # 10x10 is not divisible down to a 4x4 map
