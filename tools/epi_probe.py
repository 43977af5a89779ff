#!/usr/bin/env python3
"""Fixed per-tile cost of the fused 3x3 conv through the op-level entry (drm_op_norm_act_conv): one launch (profiler kind 0) of Cin -> Cout
at a map size for the epilogue variants (plain / + emb / + residual) and a sweep over Cin; the intercept of time(Cin) is what a tile pays
besides its K loop.  NOTE: the op-level entry stores NCHW without fused statistics -- the networks' own NHWC + statistics epilogue is
measured by tools/epi_cost.sh on whole ResBlocks instead."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drmnet_amd import _lib, ops
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
ops.set_precision(prec)
L = _lib.lib(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def t_conv(n, cin, cout, h, w, emb=False, res=False, gn=True, reps=4):
    x = torch.randn((n, cin, h, w), generator=g).to(dev)
    wt = (torch.randn((cout, cin, 3, 3), generator=g) / (3 * cin ** 0.5)).to(dev); b = torch.randn((cout,), generator=g).to(dev)
    ga = torch.ones(cin, device=dev) if gn else None; be = torch.zeros(cin, device=dev) if gn else None
    e = torch.randn((n, cout), generator=g).to(dev) if emb else None
    r = torch.randn((n, cout, h, w), generator=g).to(dev) if res else None
    ops.norm_act_conv(x, wt, b, ga, be, gn, e, r); torch.cuda.synchronize()
    L.drm_profile_reset(); L.drm_profile_enable(1)
    for _ in range(reps): ops.norm_act_conv(x, wt, b, ga, be, gn, e, r)
    torch.cuda.synchronize(); L.drm_profile_enable(0)
    K = 5; ms, fl, by, cnt = (C.c_double*K)(), (C.c_double*K)(), (C.c_double*K)(), (C.c_int64*K)()
    L.drm_profile_collect(ms, fl, by, cnt)
    return ms[0] / cnt[0], fl[0] / ms[0] / 1e9
for (h, w, cout, cins) in [(128, 256, 128, (128, 256, 384)), (64, 128, 256, (256, 384, 512))]:
    for cin in cins:
        t, tf = t_conv(32, cin, cout, h, w)
        print(f"{prec} {cin}->{cout} @{h}x{w} plain: {t:.4f} ms {tf:.0f} TF", flush=True)
    cin = cins[0]
    for name, kw in (("emb", dict(emb=True)), ("res", dict(res=True)), ("emb+res", dict(emb=True, res=True)), ("raw input", dict(gn=False))):
        t, tf = t_conv(32, cin, cout, h, w, **kw)
        print(f"{prec} {cin}->{cout} @{h}x{w} {name}: {t:.4f} ms {tf:.0f} TF", flush=True)
