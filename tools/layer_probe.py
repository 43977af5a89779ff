#!/usr/bin/env python3
"""Times ResBlocks through the op-level ABI (NHWC inside, realistic epilogues)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drmnet_amd import _lib, ops, synth
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
zero = len(sys.argv) > 2 and sys.argv[2] == "zero"  # all-zero operands: same instruction stream, (almost) no datapath toggling -> what DVFS gives back
ops.set_precision(prec)
L = _lib.lib(); dev = torch.device("cuda:0")
def man(cin, cout):
    m = [("in_layers.0.weight", (cin,)), ("in_layers.0.bias", (cin,)), ("in_layers.2.weight", (cout, cin, 3, 3)), ("in_layers.2.bias", (cout,)),
         ("emb_layers.1.weight", (cout, 512)), ("emb_layers.1.bias", (cout,)), ("out_layers.0.weight", (cout,)), ("out_layers.0.bias", (cout,)),
         ("out_layers.3.weight", (cout, cout, 3, 3)), ("out_layers.3.bias", (cout,))]
    if cin != cout: m += [("skip_connection.weight", (cout, cin, 1, 1)), ("skip_connection.bias", (cout,))]
    return m
SHAPES = [(32, 128, 128, 128, 256), (32, 256, 256, 64, 128)]
if os.environ.get("LAYER_SHAPES"):  # "n,cin,cout,h,w;..." (e.g. the batch-1 step's levels: 1,128,128,128,128;1,256,256,64,64)
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["LAYER_SHAPES"].split(";")]
for (n, cin, cout, h, w) in SHAPES:
    P = [p.to(dev) for p in synth.synth_state_dict(man(cin, cout), 1).values()]
    g = torch.Generator().manual_seed(0)
    x = torch.randn((n, cin, h, w), generator=g).to(dev); emb = torch.randn((n, 512), generator=g).to(dev)
    if zero:
        P = [torch.zeros_like(p) for p in P]; x = torch.zeros_like(x); emb = torch.zeros_like(emb)
    ops.resblock(P, x, emb); torch.cuda.synchronize()
    L.drm_profile_reset(); L.drm_profile_enable(1)
    for _ in range(3): ops.resblock(P, x, emb)
    torch.cuda.synchronize(); L.drm_profile_enable(0)
    K = 5; ms, fl, by, cnt = (C.c_double*K)(), (C.c_double*K)(), (C.c_double*K)(), (C.c_int64*K)()
    L.drm_profile_collect(ms, fl, by, cnt)
    print(f"{prec}{' ZERO operands' if zero else ''} resblock {cin}->{cout} @{h}x{w}: conv3x3 {ms[0]/cnt[0]:.3f} ms/launch ({fl[0]/ms[0]/1e9:.0f} TF)"
          + (f"   conv1x1 (skip) {ms[1]/cnt[1]:.4f} ms/launch ({fl[1]/ms[1]/1e9:.0f} TF)" if cnt[1] else ""), flush=True)
    del x, P; torch.cuda.empty_cache()
