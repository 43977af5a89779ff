#!/usr/bin/env python3
"""Race / flakiness screen of the hot kernels (cdna_hip_programming.md: "an early read passes reference checks whenever the DMA happens to land
first -- place reads by the vmcnt / barrier count, never by clean runs"; so clean runs are necessary, not sufficient -- this script is the
necessary part, run over many repetitions and shapes).

Every case runs REPS times on the same inputs; run k must equal run 0 up to the only order-dependent arithmetic left on the path (the fp64
GroupNorm statistics atomics: 2^-53 relative on the sums, i.e. ~1e-7 on an fp32 output after a normalisation) and every output must be finite.
A torn tile, a stale LDS-DMA read or a lost split-K slab shows up as an O(1) difference on some repetition.

    python tools/race_screen.py [reps]          (GPU box; exits 1 on the first mismatch)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from drmnet_amd import ops, synth  # noqa: E402

TOL = 2e-6

from race_screen_cases import attn_manifest, res_manifest  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


CASES = [  # (kind, batch, cin, cout, h, w): persistent wide tiles, 192-wide tiles, narrow tiles + split-K, small-map reduction, ragged tiles,
    # the attention paths: conv pipeline (T = 512), single-kernel form (attn_flash.hip: C = 384, T = 2048 / 1024; XCD-grouped and not), small-T GEMMs
    ("res", 32, 128, 128, 128, 256), ("res", 32, 256, 128, 128, 256), ("res", 32, 384, 384, 32, 64), ("res", 32, 640, 640, 8, 16),
    ("res", 32, 768, 768, 4, 8), ("res", 1, 512, 512, 16, 16), ("res", 1, 768, 768, 4, 4), ("res", 2, 1408, 640, 8, 8), ("res", 3, 256, 384, 12, 20),
    ("res", 2, 128, 128, 24, 40), ("attn", 32, 512, 512, 16, 32), ("attn", 8, 384, 384, 32, 64), ("attn", 32, 384, 384, 32, 64), ("attn", 5, 384, 384, 32, 32),
    ("attn", 32, 768, 768, 4, 8), ("attn", 1, 640, 640, 8, 8),
]
# fused split-K finish with eight splits per tile on DIFFERENT XCDs (42 output tiles: the splits of a tile are 42 workgroup ids apart, 42 % 8 != 0) --
# the hand-off the sc1 accesses carry without fences: screened at many repetitions by tests/test_gpu_race.py
SPLITK_XCD = [("res", 1, 1344, 672, 16, 16), ("res", 2, 1344, 672, 16, 16)]
QUICK = [("res", 32, 128, 128, 128, 256), ("res", 32, 768, 768, 4, 8), ("res", 1, 512, 512, 16, 16), ("res", 3, 256, 384, 12, 20), ("attn", 8, 384, 384, 32, 64),
         ("attn", 5, 384, 384, 32, 32), ("attn", 32, 512, 512, 16, 32)]


def run_screen(reps, cases=CASES, precisions=("f16mx", "f16x3", "fp32", "f16", "bf16"), verbose=True):
    """-> number of cases whose repetitions differ by more than TOL (or go non-finite)"""
    dev = torch.device("cuda:0")
    bad = 0
    try:
        for precision in precisions:
            ops.set_precision(precision)
            for kind, n, cin, cout, h, w in cases:
                if precision not in ("f16mx", "f16x3") and n == 32 and h >= 128:
                    continue  # (the big shapes once, in the headline mode)
                g = torch.Generator().manual_seed(h * 1000 + w + cin)
                x = torch.randn((n, cin, h, w), generator=g).to(dev)
                if kind == "res":
                    P = [p.to(dev) for p in synth.synth_state_dict(res_manifest(cin, cout), 3).values()]
                    emb = torch.randn((n, 512), generator=g).to(dev)
                    run = lambda: ops.resblock(P, x, emb)
                else:
                    P = [p.to(dev) for p in synth.synth_state_dict(attn_manifest(cin), 4).values()]
                    run = lambda: ops.attention_block(P, x)
                ref = run().clone()
                worst = 0.0
                for _ in range(reps):
                    out = run()
                    if not bool(torch.isfinite(out).all()):
                        worst = float("inf")
                        break
                    worst = max(worst, rel(out, ref))
                flag = "" if worst < TOL else "   <-- MISMATCH"
                if flag:
                    bad += 1
                if verbose:
                    print(f"{precision:6s} {kind:4s} N={n:<3d} {cin:4d}->{cout:<4d} @{h}x{w:<4d}: max rel diff over {reps} repetitions {worst:.2e}{flag}", flush=True)
                del x, P, ref
                torch.cuda.empty_cache()
    finally:
        ops.set_precision("fp32")
    return bad


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    n_bad = run_screen(reps)
    print("race screen:", "FAILED" if n_bad else "clean", f"({n_bad} case(s) over tolerance {TOL:g})")
    sys.exit(1 if n_bad else 0)
