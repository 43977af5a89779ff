#!/usr/bin/env python3
"""Times single conv layers through the op-level ABI with the library profiler (kernel time only).
usage: python tools/conv_probe.py [precision]."""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drmnet_amd import _lib, ops

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
ops.set_precision(prec)
L = _lib.lib()
dev = torch.device("cuda:0")
shapes = [(32, 128, 128, 128, 256, 3), (32, 256, 256, 64, 128, 3), (32, 384, 384, 32, 64, 3), (32, 512, 512, 16, 32, 3),
          (32, 768, 768, 4, 8, 3), (32, 256, 128, 128, 256, 1), (32, 384, 128, 128, 256, 3)]
if len(sys.argv) > 2:
    shapes = shapes[: int(sys.argv[2])]
for (n, cin, cout, h, w, k) in shapes:
    g = torch.Generator().manual_seed(0)
    x = torch.randn((n, cin, h, w), generator=g).to(dev)
    wt = (torch.randn((cout, cin, k, k), generator=g) / math.sqrt(cin * k * k)).to(dev)
    b = torch.zeros(cout, device=dev)
    gamma = torch.ones(cin, device=dev); beta = torch.zeros(cin, device=dev)
    ops.norm_act_conv(x, wt, b, gamma, beta, True)
    torch.cuda.synchronize()
    L.drm_profile_reset(); L.drm_profile_enable(1)
    for _ in range(3):
        ops.norm_act_conv(x, wt, b, gamma, beta, True)
    torch.cuda.synchronize(); L.drm_profile_enable(0)
    K = 5
    ms, fl, by, cnt = (C.c_double * K)(), (C.c_double * K)(), (C.c_double * K)(), (C.c_int64 * K)()
    L.drm_profile_collect(ms, fl, by, cnt)
    i = 0 if k == 3 else 1
    print(f"{prec} conv{k}x{k} {cin}->{cout} @{h}x{w} B={n}: {ms[i]/cnt[i]:.3f} ms  {fl[i]/ms[i]/1e9:.1f} TF", flush=True)
    del x, wt
    torch.cuda.empty_cache()
