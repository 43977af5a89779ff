#!/usr/bin/env python3
"""BASELINE configs[2] shape: ObsNet DDIM chain (eta = 1, Philox noise) at 3x128x256, whole chain timed, graph replay on / off.
usage: python tools/ddim_chain.py <batch> [steps] [precision]   (DRM_GRAPH=0 switches the hipGraph replay off)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from drmnet_amd import ops, synth
from drmnet_amd.ddim import DDIMSampler

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
prec = sys.argv[3] if len(sys.argv) > 3 else "f16x3"
dev = torch.device("cuda:0")
ops.set_graph_replay(os.environ.get("DRM_GRAPH", "1") != "0")
m = bench.build_models("obsnet", dev, prec)
x = synth.synth_refmaps(B, 128, 256, synth.SEED_INPUT).to(dev)
xT = torch.randn(x.shape, generator=torch.Generator().manual_seed(6)).to(dev)
s = DDIMSampler(m)
s.make_schedule(50, ddim_eta=1.0, verbose=False)
s.ddim_sampling(x, tuple(x.shape), x_T=xT, num_steps=4, seed=1)  # warm-up (packs the weights, sizes the workspace)
torch.cuda.synchronize()
n0 = ops.graph_launches()
t0 = time.perf_counter()
out, _ = s.ddim_sampling(x, tuple(x.shape), x_T=xT, num_steps=steps, seed=1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"ObsNet DDIM {prec} B={B} {steps} steps: {dt / steps * 1e3:.3f} ms/step  {B * steps / dt:.1f} denoise steps/s  graph launches {ops.graph_launches() - n0}  finite {bool(torch.isfinite(out).all())}")
