#!/usr/bin/env python3
"""Times the attention core (library profiler kind 2: everything between the qkv conv and proj_out) of an AttentionBlock at ObsNet's ds = 4 level:
    python tools/attn_bench.py [N] [H] [W] [precision]      default 32 32 64 f16x3  (T = 2048 keys, C = 384: the 3x128x256 metric shape)
Prints ms per block (core and whole block), algorithmic TFLOP/s (4 N T^2 C) and the effective GB/s of q, k, v in + o out."""
import ctypes as C
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from drmnet_amd import _lib, ops, synth  # noqa: E402

n, h, w = (int(v) for v in (sys.argv[1:4] + ["32", "32", "64"][len(sys.argv[1:4]):]))
prec = sys.argv[4] if len(sys.argv) > 4 else "f16x3"
ch = 384
dev = torch.device("cuda:0")
man = [("norm.weight", (ch,)), ("norm.bias", (ch,)), ("qkv.weight", (3 * ch, ch, 1)), ("qkv.bias", (3 * ch,)), ("proj_out.weight", (ch, ch, 1)), ("proj_out.bias", (ch,))]
P = [p.to(dev) for p in synth.synth_state_dict(man, 11).values()]
x = torch.randn((n, ch, h, w), generator=torch.Generator().manual_seed(1)).to(dev)
ops.set_precision(prec)
L = _lib.lib()
for _ in range(3):
    ops.attention_block(P, x)
torch.cuda.synchronize()
L.drm_profile_reset()
L.drm_profile_enable(1)
reps = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    out = ops.attention_block(P, x)
e1.record()
torch.cuda.synchronize()
L.drm_profile_enable(0)
K = 5
ms, fl, by, cnt = (C.c_double * K)(), (C.c_double * K)(), (C.c_double * K)(), (C.c_int64 * K)()
_lib.check(L.drm_profile_collect(ms, fl, by, cnt))
T = h * w
core = ms[2] / reps
print(f"attention block N={n} C={ch} T={T} ({prec}): core {core:.3f} ms  ({fl[2] / ms[2] / 1e9:.1f} TF algorithmic, {by[2] / ms[2] / 1e6:.0f} GB/s of q/k/v/o), "
      f"whole block {e0.elapsed_time(e1) / reps:.3f} ms, finite {bool(torch.isfinite(out).all())}")
