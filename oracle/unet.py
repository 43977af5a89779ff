"""Oracle: functional fp32 restatement of the reference U-Nets (TEST INFRASTRUCTURE).

Follows ldm/modules/diffusionmodules/openaimodel.py of the reference:
  UNetModel.__init__/forward          :452-713, :731-768
  EncoderUNetModel.__init__/forward   :777-953, :969-991
  ResBlock._forward                   :255-275   (use_scale_shift_norm=False, no up/down)
  AttentionBlock._forward             :325-333
  QKVAttentionLegacy.forward          :365-381   (single head)
  Downsample / Upsample (no conv)     :154-160 / :109-119
and ldm/modules/diffusionmodules/util.py:
  GroupNorm32 (32 groups, eps 1e-5)   :199-216
  timestep_embedding                  :151-171

Nothing here is an nn.Module: the network is a flat list of block descriptors
built from the config, and parameters are looked up by the reference's own
state_dict key names, so a reference checkpoint / synthetic manifest feeds it
directly.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

GN_GROUPS = 32
GN_EPS = 1e-5


@dataclass
class Res:
    prefix: str
    cin: int
    cout: int

    @property
    def has_skip(self) -> bool:
        return self.cin != self.cout


@dataclass
class Attn:
    prefix: str
    ch: int


@dataclass
class Down:
    ch: int


@dataclass
class Up:
    ch: int


@dataclass
class Topology:
    kind: str  # "unet" | "encoder"
    in_channels: int
    model_channels: int
    out_channels: int
    input_blocks: List[list] = field(default_factory=list)  # list of lists of layer descriptors
    middle: list = field(default_factory=list)
    output_blocks: List[list] = field(default_factory=list)
    final_ch: int = 0


def _check_supported(cfg: dict) -> None:
    """The shipped configs fix these; anything else is rejected loudly."""
    bad = []
    if cfg.get("use_spatial_transformer", False):
        bad.append("use_spatial_transformer")
    if cfg.get("resblock_updown", False):
        bad.append("resblock_updown")
    if cfg.get("conv_resample", True):
        bad.append("conv_resample (default True in the reference ctor; configs set False)")
    if cfg.get("use_scale_shift_norm", False):
        bad.append("use_scale_shift_norm")
    if cfg.get("num_heads", 1) != 1 or cfg.get("num_head_channels", -1) != -1:
        bad.append("num_heads != 1")
    if cfg.get("dims", 2) != 2:
        bad.append("dims != 2")
    if cfg.get("use_fp16", False):
        bad.append("use_fp16")
    if cfg.get("num_classes") is not None:
        bad.append("num_classes")
    if cfg.get("pool", "adaptive") != "adaptive":
        bad.append("pool != adaptive")
    if cfg.get("use_positional_embedded_attention", False):
        bad.append("use_positional_embedded_attention")
    if cfg.get("dropout", 0.0) not in (0, 0.0):
        bad.append("dropout != 0 (inference only)")
    if bad:
        raise NotImplementedError("unsupported U-Net params: " + ", ".join(bad))


def build_topology(cfg: dict, kind: str) -> Topology:
    """Enumerates blocks exactly as the reference constructors do
    (openaimodel.py:528-707 for UNetModel, :824-929 for EncoderUNetModel)."""
    _check_supported(cfg)
    mc = int(cfg["model_channels"])
    mult = [int(m) for m in cfg["channel_mult"]]
    nrb = int(cfg["num_res_blocks"])
    attn_res = set(int(a) for a in cfg["attention_resolutions"])
    topo = Topology(kind, int(cfg["in_channels"]), mc, int(cfg["out_channels"]))

    topo.input_blocks.append([("stem", "input_blocks.0.0")])
    chans = [mc]
    ch, ds, idx = mc, 1, 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            layers = [Res(f"input_blocks.{idx}.0", ch, m * mc)]
            ch = m * mc
            if ds in attn_res:
                layers.append(Attn(f"input_blocks.{idx}.1", ch))
            topo.input_blocks.append(layers)
            chans.append(ch)
            idx += 1
        if level != len(mult) - 1:
            topo.input_blocks.append([Down(ch)])
            chans.append(ch)
            idx += 1
            ds *= 2
    topo.middle = [Res("middle_block.0", ch, ch), Attn("middle_block.1", ch), Res("middle_block.2", ch, ch)]
    if kind == "encoder":
        topo.final_ch = ch
        return topo
    oidx = 0
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            ich = chans.pop()
            layers = [Res(f"output_blocks.{oidx}.0", ch + ich, mc * m)]
            ch = mc * m
            if ds in attn_res:
                layers.append(Attn(f"output_blocks.{oidx}.1", ch))
            if level and i == nrb:
                layers.append(Up(ch))
                ds //= 2
            topo.output_blocks.append(layers)
            oidx += 1
    topo.final_ch = ch
    return topo


def param_manifest(cfg: dict, kind: str) -> List[Tuple[str, Tuple[int, ...]]]:
    """Ordered (key, shape) list == reference ``state_dict()`` order of the U-Net module."""
    topo = build_topology(cfg, kind)
    mc, ted = topo.model_channels, topo.model_channels * 4
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def lin(p, o, i):
        out.append((p + ".weight", (o, i)))
        out.append((p + ".bias", (o,)))

    def conv(p, o, i, k):
        out.append((p + ".weight", (o, i, k, k)))
        out.append((p + ".bias", (o,)))

    def norm(p, c):
        out.append((p + ".weight", (c,)))
        out.append((p + ".bias", (c,)))

    def res(r: Res):
        norm(r.prefix + ".in_layers.0", r.cin)
        conv(r.prefix + ".in_layers.2", r.cout, r.cin, 3)
        lin(r.prefix + ".emb_layers.1", r.cout, ted)
        norm(r.prefix + ".out_layers.0", r.cout)
        conv(r.prefix + ".out_layers.3", r.cout, r.cout, 3)
        if r.has_skip:
            conv(r.prefix + ".skip_connection", r.cout, r.cin, 1)

    def attn(a: Attn):
        norm(a.prefix + ".norm", a.ch)
        out.append((a.prefix + ".qkv.weight", (3 * a.ch, a.ch, 1)))
        out.append((a.prefix + ".qkv.bias", (3 * a.ch,)))
        out.append((a.prefix + ".proj_out.weight", (a.ch, a.ch, 1)))
        out.append((a.prefix + ".proj_out.bias", (a.ch,)))

    def layers(ls):
        for l in ls:
            if isinstance(l, Res):
                res(l)
            elif isinstance(l, Attn):
                attn(l)
            elif isinstance(l, tuple):
                conv(l[1], mc, topo.in_channels, 3)

    lin("time_embed.0", ted, mc)
    lin("time_embed.2", ted, ted)
    for b in topo.input_blocks:
        layers(b)
    layers(topo.middle)
    for b in topo.output_blocks:
        layers(b)
    norm("out.0", topo.final_ch)
    if kind == "unet":
        conv("out.2", topo.out_channels, mc, 3)
    else:
        conv("out.3", topo.out_channels, topo.final_ch, 1)
    return out


# ----------------------------------------------------------------------------- primitives

# Working precision.  fp32 restates the reference (GroupNorm32 / the softmax cast to fp32, util.py:214-216, openaimodel.py:376).  fp64 is
# the same arithmetic carried in double -- the yardstick of the weight-stress fixtures (tests/golden/stress_*.npz: `out64_*`), where two
# fp32 evaluations of one network differ by up to 1e-4 and "distance from the fp32 reference" stops being a measure of correctness.
_DT = torch.float32


class working_dtype:
    """with working_dtype(torch.float64): ... -- the forwards below run in that precision (parameters must be given in it)"""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global _DT
        self.prev, _DT = _DT, self.dtype

    def __exit__(self, *exc):
        global _DT
        _DT = self.prev



def timestep_embedding(timesteps: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """util.py:151-171 -- cat(cos, sin), f_j = exp(-ln(max_period) * j / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb.to(_DT)


def group_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """util.py:214-216 GroupNorm32: fp32, 32 groups, eps 1e-5, affine."""
    return F.group_norm(x.to(_DT), GN_GROUPS, w, b, GN_EPS)


def silu(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(x)


def res_block(P: Dict[str, torch.Tensor], r: Res, x: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    """openaimodel.py:255-275."""
    p = r.prefix
    h = silu(group_norm(x, P[p + ".in_layers.0.weight"], P[p + ".in_layers.0.bias"]))
    h = F.conv2d(h, P[p + ".in_layers.2.weight"], P[p + ".in_layers.2.bias"], padding=1)
    e = F.linear(silu(emb), P[p + ".emb_layers.1.weight"], P[p + ".emb_layers.1.bias"])
    h = h + e[:, :, None, None]
    h = silu(group_norm(h, P[p + ".out_layers.0.weight"], P[p + ".out_layers.0.bias"]))
    h = F.conv2d(h, P[p + ".out_layers.3.weight"], P[p + ".out_layers.3.bias"], padding=1)
    if r.has_skip:
        x = F.conv2d(x, P[p + ".skip_connection.weight"], P[p + ".skip_connection.bias"])
    return x + h


def attention_block(P: Dict[str, torch.Tensor], a: Attn, x: torch.Tensor) -> torch.Tensor:
    """openaimodel.py:325-333 + :365-381 (one head: q,k,v are channel thirds)."""
    p = a.prefix
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, hh * ww)
    n = F.group_norm(xf.to(_DT), GN_GROUPS, P[p + ".norm.weight"], P[p + ".norm.bias"], GN_EPS)
    qkv = F.conv1d(n, P[p + ".qkv.weight"], P[p + ".qkv.bias"])
    q, k, v = qkv.split(c, dim=1)
    s = 1.0 / math.sqrt(math.sqrt(c))
    w = torch.einsum("bct,bcs->bts", q * s, k * s)
    w = torch.softmax(w.to(_DT), dim=-1)
    o = torch.einsum("bts,bcs->bct", w, v)
    o = F.conv1d(o, P[p + ".proj_out.weight"], P[p + ".proj_out.bias"])
    return (xf + o).reshape(b, c, hh, ww)


def _run_layers(P, layers, h, emb):
    for l in layers:
        if isinstance(l, Res):
            h = res_block(P, l, h, emb)
        elif isinstance(l, Attn):
            h = attention_block(P, l, h)
        elif isinstance(l, Down):
            h = F.avg_pool2d(h, 2, 2)
        elif isinstance(l, Up):
            h = F.interpolate(h, scale_factor=2, mode="nearest")
        else:  # stem
            h = F.conv2d(h, P[l[1] + ".weight"], P[l[1] + ".bias"], padding=1)
    return h


def time_embed(P, t_emb: torch.Tensor) -> torch.Tensor:
    """openaimodel.py:521-526,750."""
    e = F.linear(t_emb, P["time_embed.0.weight"], P["time_embed.0.bias"])
    return F.linear(silu(e), P["time_embed.2.weight"], P["time_embed.2.bias"])


@torch.no_grad()
def unet_forward(
    P: Dict[str, torch.Tensor],
    topo: Topology,
    x: torch.Tensor,
    timesteps: Optional[torch.Tensor] = None,
    t_emb: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """UNetModel.forward, openaimodel.py:731-768. Exactly one of timesteps / t_emb."""
    if (timesteps is None) == (t_emb is None):
        raise ValueError("timesteps and t_emb cannot be specified at the same time")
    if t_emb is None:
        t_emb = timestep_embedding(timesteps, topo.model_channels)
    emb = time_embed(P, t_emb)
    hs = []
    h = x.to(_DT)
    for b in topo.input_blocks:
        h = _run_layers(P, b, h, emb)
        hs.append(h)
    h = _run_layers(P, topo.middle, h, emb)
    for b in topo.output_blocks:
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_layers(P, b, h, emb)
    h = silu(group_norm(h, P["out.0.weight"], P["out.0.bias"]))
    return F.conv2d(h, P["out.2.weight"], P["out.2.bias"], padding=1)


@torch.no_grad()
def encoder_forward(P: Dict[str, torch.Tensor], topo: Topology, x: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
    """EncoderUNetModel.forward, openaimodel.py:969-991, pool='adaptive' head :922-929."""
    emb = time_embed(P, timestep_embedding(timesteps, topo.model_channels))
    h = x.to(_DT)
    for b in topo.input_blocks:
        h = _run_layers(P, b, h, emb)
    h = _run_layers(P, topo.middle, h, emb)
    h = silu(group_norm(h, P["out.0.weight"], P["out.0.bias"]))
    h = h.mean(dim=(2, 3), keepdim=True)
    h = F.conv2d(h, P["out.3.weight"], P["out.3.bias"])
    return h.flatten(1)


def z_embed(Pz: Dict[str, torch.Tensor], delta_z: torch.Tensor) -> torch.Tensor:
    """ZEmbDiffusionWrapper.z_emb_layer, models/drmnet.py:38-45: three Linear, SiLU after EACH."""
    h = silu(F.linear(delta_z, Pz["z_emb_layer.0.weight"], Pz["z_emb_layer.0.bias"]))
    h = silu(F.linear(h, Pz["z_emb_layer.2.weight"], Pz["z_emb_layer.2.bias"]))
    return silu(F.linear(h, Pz["z_emb_layer.4.weight"], Pz["z_emb_layer.4.bias"]))


def zemb_manifest(z_dim: int, model_channels: int) -> List[Tuple[str, Tuple[int, ...]]]:
    h = model_channels // 2
    return [
        ("z_emb_layer.0.weight", (h, z_dim)),
        ("z_emb_layer.0.bias", (h,)),
        ("z_emb_layer.2.weight", (h, h)),
        ("z_emb_layer.2.bias", (h,)),
        ("z_emb_layer.4.weight", (model_channels, h)),
        ("z_emb_layer.4.bias", (model_channels,)),
    ]


# ----------------------------------------------------------------------------- the three shipped nets

ILLNET_CFG = dict(  # configs/drmnet/eval_drmnet.yaml:35-48
    image_size=128, in_channels=6, out_channels=3, model_channels=128, attention_resolutions=[8, 16, 32],
    num_res_blocks=2, dropout=0.0, channel_mult=[1, 2, 3, 4, 5, 6], num_heads=1, resblock_updown=False, conv_resample=False,
)
REFNET_CFG = dict(  # configs/drmnet/eval_drmnet.yaml:50-65
    image_size=128, in_channels=6, model_channels=128, out_channels=6, num_res_blocks=2, attention_resolutions=[8, 16],
    dropout=0.0, channel_mult=[1, 1, 2, 3, 4], conv_resample=False, resblock_updown=False, num_heads=1,
    use_scale_shift_norm=False, pool="adaptive",
)
OBSNET_CFG = dict(  # configs/obsnet/eval_obsnet.yaml:23-35
    image_size=128, in_channels=6, out_channels=3, model_channels=128, attention_resolutions=[4, 8, 16],
    num_res_blocks=2, channel_mult=[1, 2, 3, 4, 5], num_heads=1, resblock_updown=False, conv_resample=False,
)
# reduced-width nets used by fast fixtures (same code paths: skip 1x1, attention, down/up, concat)
TINY_UNET_CFG = dict(
    image_size=16, in_channels=6, out_channels=3, model_channels=32, attention_resolutions=[2, 4],
    num_res_blocks=1, channel_mult=[1, 2, 2], num_heads=1, resblock_updown=False, conv_resample=False,
)
TINY_ENC_CFG = dict(
    image_size=16, in_channels=6, out_channels=6, model_channels=32, attention_resolutions=[2],
    num_res_blocks=1, channel_mult=[1, 2], num_heads=1, resblock_updown=False, conv_resample=False, pool="adaptive",
)
