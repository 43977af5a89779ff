#!/usr/bin/env python3
"""Accuracy of the f16mx mode (fp16 hi*hi + block-scaled fp8 cross terms on the GroupNorm-fed 3x3 convs) next to f16x3 and f16:
res blocks against the exact-fp32 mode, the full-width networks against the outputs recorded from the reference (tests/golden)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from conftest import gold, rel_l2  # noqa: E402
from drmnet_amd import ops, synth  # noqa: E402
from oracle import unet as ou  # noqa: E402
from race_screen_cases import res_manifest  # noqa: E402
from test_gpu_nets import build, full_inputs  # noqa: E402

dev = torch.device("cuda:0")
MODES = ("f16x3", "f16mx", "f16")
print("res blocks vs exact fp32 (rel-L2):")
for n, cin, cout, h, w in [(4, 128, 128, 128, 256), (4, 256, 128, 64, 128), (4, 384, 384, 32, 64), (8, 640, 640, 8, 16), (2, 768, 768, 4, 8), (3, 256, 384, 12, 20)]:
    g = torch.Generator().manual_seed(h * 1000 + w + cin)
    x = torch.randn((n, cin, h, w), generator=g).to(dev)
    emb = torch.randn((n, 512), generator=g).to(dev)
    P = [p.to(dev) for p in synth.synth_state_dict(res_manifest(cin, cout), 3).values()]
    ops.set_precision("fp32")
    ref = ops.resblock(P, x, emb).clone()
    row = []
    for mode in MODES:
        ops.set_precision(mode)
        out = ops.resblock(P, x, emb)
        row.append(f"{mode} {rel_l2(out.cpu(), ref.cpu()):.2e}")
    print(f"  N={n} {cin}->{cout} @{h}x{w}: " + "   ".join(row), flush=True)

print("full-width networks vs the reference's outputs (rel-L2):")
for name, cfg, kind in (("illnet", ou.ILLNET_CFG, "unet"), ("refnet", ou.REFNET_CFG, "encoder"), ("obsnet", ou.OBSNET_CFG, "unet")):
    gd = gold(f"full_{name}_sizes")
    m = build(cfg, kind, int(gd["seed"]), dev)
    for key in sorted(k for k in gd if k.startswith("out_")):
        n, h, w = (int(v) for v in key[4:].split("x"))
        xc, t_emb = full_inputs(n, h, w)
        t = torch.from_numpy(gd["t"])[:n].to(dev)
        row = []
        for mode in MODES:
            m.set_precision(mode)
            out = m(xc.to(dev), t_emb=t_emb.to(dev)) if name == "illnet" else m(xc.to(dev), t)
            row.append(f"{mode} {rel_l2(out.cpu(), gd[key]):.2e}")
        print(f"  {name} {n}x{h}x{w}: " + "   ".join(row), flush=True)
    del m
    torch.cuda.empty_cache()
