#!/usr/bin/env python3
"""Headline benchmark: U-Net denoise steps/sec of DRMNet's reverse process on synthetic 3x128x256 refmaps.

    python bench.py --gpus N --steps K --warmup W
    N > 1 works both ways: launched by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or as the plain command above, in
    which case this process spawns the N ranks itself as fresh child processes BEFORE it touches the GPU and relays rank 0's line.

One "step" = one pass of the hot path over one batch: a full DRMNet reverse step for `--batch` refmaps per GPU
(RefNet forward -> BRDF schedule -> z-embedding MLP -> IllNet forward -> fused state update, models/drmnet.py:809-825 of
the reference), all rows active, fp32, inputs resident in HBM.  value = sample-steps/s summed over all GPUs (weak scaling:
every rank owns its own `--batch` refmaps; the path has no collective -- samples are independent).

The JSON line carries
  roofline     : dominant kernel = fused GroupNorm+SiLU+conv3x3 implicit GEMM (conv.hip); achieved = algorithmic conv3x3
                 FLOPs of its launches / their summed duration, both measured with HIP events on the launch stream inside the
                 timed region (library profiler, include/drmnet_hip.h drm_profile_*); peak = the dense MFMA peak of the mode;
                 traffic = HBM bytes per launch of the dominant instantiation from two `rocprofv3 --pmc` child runs of this same
                 command (FETCH_SIZE, WRITE_SIZE: separate passes), made after the timed region (live_traffic below).
  cpu_baseline : the CPU oracle (oracle/, a port of the reference's arithmetic) timed on this box's host cores on a bounded
                 sample of the same workload (rank 0, N = 1 only).
Other workloads (--workload illnet | refnet | obsnet | obsnet_ddim) time a single network / sampler for the DESIGN.md tables.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
F16_MFMA_PEAK_TFLOPS = 2516.0  # dense f16/bf16 MFMA peak (v_mfma_f32_32x32x16_f16); the split path issues 3 MFMA FLOPs per algorithmic FLOP
# algorithmic matmul GFLOP per sample per forward (measured from the reference modules)
GFLOP = {"illnet": {(128, 128): 202.66, (128, 256): 406.76}, "refnet": {(128, 128): 34.83, (128, 256): 70.10},
         "obsnet": {(128, 128): 215.34, (128, 256): 448.24}}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="refmaps per GPU (BASELINE config[1]: 32)")
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--workload", default="drmnet_step", choices=["drmnet_step", "illnet", "refnet", "obsnet", "obsnet_ddim", "obsnet_ddim_chain", "estimate_chain"])
    ap.add_argument("--precision", default="auto", choices=["auto", "fp32", "f16x3", "f16", "f16mx", "bf16"],
                    help="conv arithmetic: auto (default) = f16mx per network only where a seeded probe forward on the loaded weights agrees with f16x3 to 5e-5 "
                         "(half the contract), else f16x3 -- the choice and the measured figure are in the line's `precision_auto` object; "
                         "f16mx = fp32 operands split into fp16 hi+lo; hi*hi on the f16 MFMA, both cross terms of the "
                         "GroupNorm-fed 3x3 convs in one block-scaled fp8 MFMA (2.4e-5 .. 4e-5 rel-L2 per network against the reference, 1e-4 contract: "
                         "tests/test_gpu_f16mx.py); f16x3 = all three products on the f16 MFMA (~2e-6: passes the SAME tolerances as fp32, "
                         "tests/test_gpu_split.py); fp32 = v_mfma_f32_32x32x2_f32; f16 = reduced precision (~1e-3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="skip the two rocprofv3 --pmc child runs that measure roofline.traffic in this run (the committed profile is imported instead, hash-gated)")
    ap.add_argument("--no-strict-fp32", action="store_true", help="skip the short exact-fp32 pass that follows the headline measurement")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads (B = 128 / B = 1 steps, ObsNet DDIM chain at B = 256, full chain) that follow it")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the batch-1 parity forward passes (keeps a rocprofv3 trace of this command to the timed workload's launches)")
    return ap.parse_args()


def build_models(workload, dev, precision="fp32"):
    from drmnet_amd import synth
    from drmnet_amd.config import instantiate_from_config, load_config

    if workload == "estimate_chain":  # both models + their dataset transforms (scripts/estimate.py:120-125)
        drm = build_models("drmnet_step", dev, precision)
        obs = build_models("obsnet", dev, precision)
        drm.ds = instantiate_from_config(load_config(os.path.join(ROOT, "configs/drmnet/eval_drmnet.yaml"))["data"]["params"]["predict"])
        obs.ds = instantiate_from_config(load_config(os.path.join(ROOT, "configs/obsnet/eval_obsnet.yaml"))["data"]["params"]["predict"])
        return drm, obs
    if workload in ("drmnet_step", "illnet", "refnet"):
        cfg = load_config(os.path.join(ROOT, "configs/drmnet/eval_drmnet.yaml"))["model"]
        cfg["params"].pop("ckpt_path")
        cfg["params"]["use_ema"] = False  # no checkpoint offline: EMA == live weights, skip the second copy
        m = instantiate_from_config(cfg)
        synth.load_synth(m.illnet_model.diffusion_model, synth.SEED_ILLNET)
        synth.load_synth(m.refnet_model.diffusion_model, synth.SEED_REFNET)
        m.illnet_model.z_emb_layer.load_state_dict(synth.synth_state_dict(
            [(k, tuple(v.shape)) for k, v in m.illnet_model.z_emb_layer.state_dict().items()], synth.SEED_ZEMB))
        m.set_precision(precision)
        return m.to(dev)
    cfg = load_config(os.path.join(ROOT, "configs/obsnet/eval_obsnet.yaml"))["model"]
    cfg["params"].pop("ckpt_path")
    cfg["params"]["use_ema"] = False
    m = instantiate_from_config(cfg)
    synth.load_synth(m.model.diffusion_model, synth.SEED_OBSNET)
    m.set_precision(precision)
    return m.to(dev)


def chain_inputs(B, dev, radius=128):
    """B synthetic objects for the full chain: a shaded, textured unit sphere seen by an orthographic camera
    (normals as utils/transform.py:147-167), 256x256 HDR image + normals + mask each."""
    import numpy as np
    from drmnet_amd import synth

    lin = np.linspace(-radius + 0.5, radius - 0.5, 2 * radius)
    xx, yy = np.meshgrid(lin, lin[::-1])
    zsq = radius ** 2 - (xx ** 2 + yy ** 2)
    nrm = np.stack([xx, yy, np.sqrt(np.clip(zsq, 0, None))], -1).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    nrm[zsq < 0] = 0
    normals = torch.from_numpy(nrm).to(dev).expand(B, -1, -1, -1).contiguous()
    masks = torch.linalg.norm(normals, dim=-1) > 0.5
    g = torch.Generator().manual_seed(synth.SEED_INPUT)
    imgs = (torch.exp(torch.randn((B, 2 * radius, 2 * radius, 3), generator=g) * 0.5 - 2.0)).to(dev)
    imgs = imgs * (0.2 + normals[..., 2:3].clamp_min(0))
    return imgs, normals, masks


def make_step(args, model, dev):
    """Returns (callable running ONE step on the current stream, algorithmic GFLOP per sample-step, description)."""
    from drmnet_amd import _lib, synth

    B, H, W = args.batch, args.height, args.width
    if args.workload == "estimate_chain":
        # BASELINE configs[4] / SURVEY 8d config 5: object image + normals + mask -> refmap -> ObsNet DDIM-50 -> DRMNet loop
        # (150 steps, early exit off so the work is countable) -> Lr0, at the config shape (128x128 refmaps from 256x256 images)
        from drmnet_amd.estimate import estimate_batch

        drm, obs = model
        res = drm.ds.size
        imgs, normals, masks = chain_inputs(B, dev)
        state = {"n": 0}

        def step():
            estimate_batch(drm, obs, imgs, normals, masks, early_exit=False, seed=100 + state["n"])
            state["n"] += 1

        gf = obs.ddim_steps * GFLOP["obsnet"].get((res, res), 0) + drm.max_timesteps * (GFLOP["illnet"].get((res, res), 0) + GFLOP["refnet"].get((res, res), 0))
        return step, gf, (f"full chain per object image: erosion + refmap_mask_make + ObsNet DDIM-{obs.ddim_steps} + DRMNet loop "
                          f"({drm.max_timesteps} steps, early exit off) at {res}x{res}; value counts object images")
    x = synth.synth_refmaps(B, H, W, synth.SEED_INPUT).to(dev)
    L = _lib.lib()
    key = (H, W)
    if args.workload == "drmnet_step":
        h = model._engine()
        ws = model._ws.get(int(L.drm_drmnet_workspace_bytes(h, B, H, W)), dev)
        Lr_k = x.clone()
        state = {"i": 0}

        def step():
            i = state["i"] % model.max_timesteps
            _lib.check(L.drm_drmnet_step(h, Lr_k.data_ptr(), x.data_ptr(), None, B, i, None, 1234, None, None, None, B, H, W,
                                         ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev)))
            state["i"] += 1

        gf = GFLOP["illnet"].get(key, 0) + GFLOP["refnet"].get(key, 0)
        step.state = Lr_k  # the tensor every step updates in place: main() checks it stays finite
        step.counter = state  # {"i": steps done}: the reverse-step index and the Philox key of the next step
        return step, gf, "DRMNet reverse step = RefNet + z-MLP + IllNet + update (all rows active)"
    if args.workload in ("illnet", "refnet", "obsnet"):
        unet = {"illnet": lambda: model.illnet_model.diffusion_model, "refnet": lambda: model.refnet_model.diffusion_model,
                "obsnet": lambda: model.model.diffusion_model}[args.workload]()
        gen = torch.Generator().manual_seed(5)
        t_emb = torch.randn((B, 128), generator=gen).to(dev)
        t = torch.full((B,), 500, dtype=torch.long, device=dev)
        xk = (x + 0.025 * torch.randn(x.shape, generator=gen).to(dev)).contiguous()

        def step():
            if args.workload == "illnet":
                unet.forward_parts(xk, x, t_emb=t_emb)
            elif args.workload == "refnet":
                unet.forward_parts(xk, x, t)
            else:
                unet.forward_parts(xk, x, timesteps=t)

        return step, GFLOP[args.workload].get(key, 0), f"{args.workload} U-Net forward"
    # obsnet_ddim: one DDIM step (U-Net + fused update, Philox noise); obsnet_ddim_chain: the whole 50-step chain with the
    # captured-step hipGraph replay (BASELINE configs[2]: DDIM 50-step, batch 256, reduced precision, graph) -- one bench step = one chain
    from drmnet_amd import ops
    from drmnet_amd.ddim import DDIMSampler

    s = DDIMSampler(model)
    s.make_schedule(50, ddim_eta=1.0, verbose=False)
    xT = torch.randn(x.shape, generator=torch.Generator().manual_seed(6)).to(dev)
    if args.workload == "obsnet_ddim_chain":
        ops.set_graph_replay(True)

        def chain():
            out, _ = s.ddim_sampling(x, tuple(x.shape), x_T=xT, seed=1)
            chain.state = out

        chain.units = 50  # denoise steps per bench step and sample
        return chain, GFLOP["obsnet"].get(key, 0), "ObsNet DDIM chain, 50 steps (eta = 1, Philox noise), step 2 captured as a hipGraph and replayed"

    def step():
        s.ddim_sampling(x, tuple(x.shape), x_T=xT, num_steps=1, seed=1)

    return step, GFLOP["obsnet"].get(key, 0), "ObsNet DDIM step (eta=1)"


def parity_check(model, dev, precision):
    """The networks that were just timed, in the precision they were timed in, against outputs recorded from the REFERENCE
    (tests/golden/full_*_128x256.npz: same seeded weights and inputs; data files, not the oracle).  Keeps the bench line honest
    about its arithmetic: the split-precision default must sit at fp32-level error, far inside the 1e-4 contract."""
    import numpy as np
    from drmnet_amd import synth

    res = {"tolerance_rel_l2": 1e-4, "cases": {}}
    try:
        x = synth.synth_refmaps(1, 128, 256, synth.SEED_INPUT)
        gen = torch.Generator().manual_seed(synth.SEED_INPUT + 1)
        xk = x + 0.025 * torch.randn(x.shape, generator=gen)
        t_emb = torch.randn((1, 128), generator=gen)
        xc = torch.cat([xk, x], dim=1).contiguous().to(dev)
        for name, net, kw in (("illnet", model.illnet_model.diffusion_model, "t_emb"), ("refnet", model.refnet_model.diffusion_model, "t")):
            g = np.load(os.path.join(ROOT, "tests", "golden", f"full_{name}_128x256.npz"))
            out = net(xc, t_emb=t_emb.to(dev)) if kw == "t_emb" else net(xc, torch.from_numpy(g["t"]).to(dev))
            ref = torch.from_numpy(g["out"]).double()
            res["cases"][f"{name} 1x3x128x256 vs reference output"] = float(((out.cpu().double() - ref).norm() / ref.norm()).item())
        res["max_rel_l2"] = max(res["cases"].values())
        res["within_tolerance"] = bool(res["max_rel_l2"] < res["tolerance_rel_l2"])
        res["precision"] = precision
    except (OSError, KeyError) as e:  # fixtures not shipped with this copy
        res["error"] = str(e)
    return res


def obsnet_parity(obs, dev) -> dict:
    """The third shipped network against its reference output (tests/golden/full_obsnet_128x256.npz), in the mode `obs` runs in."""
    import numpy as np
    from drmnet_amd import synth

    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "full_obsnet_128x256.npz"))
        x = synth.synth_refmaps(1, 128, 256, synth.SEED_INPUT)
        gen = torch.Generator().manual_seed(synth.SEED_INPUT + 1)
        xk = x + 0.025 * torch.randn(x.shape, generator=gen)
        xc = torch.cat([xk, x], dim=1).contiguous().to(dev)
        out = obs.model.diffusion_model(xc, torch.from_numpy(g["t"]).to(dev))
        ref = torch.from_numpy(g["out"]).double()
        return {"obsnet 1x3x128x256 vs reference output": float(((out.cpu().double() - ref).norm() / ref.norm()).item())}
    except (OSError, KeyError) as e:
        return {"obsnet error": str(e)}


def cpu_baseline(args):
    """Oracle (CPU port of the reference arithmetic) on a bounded sample: batch 1 of the same shape, a few steps."""
    from drmnet_amd import synth
    from oracle import samplers as osamp
    from oracle import unet as ou

    H, W = args.height, args.width
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 32))  # more intra-op threads than that only slow oneDNN down at batch 1
    torch.set_num_threads(threads)
    Pi = synth.synth_state_dict(ou.param_manifest(ou.ILLNET_CFG, "unet"), synth.SEED_ILLNET)
    Pr = synth.synth_state_dict(ou.param_manifest(ou.REFNET_CFG, "encoder"), synth.SEED_REFNET)
    Pz = synth.synth_state_dict(ou.zemb_manifest(6, 128), synth.SEED_ZEMB)
    ti, tr = ou.build_topology(ou.ILLNET_CFG, "unet"), ou.build_topology(ou.REFNET_CFG, "encoder")
    x = synth.synth_refmaps(1, H, W, synth.SEED_INPUT)
    z0 = torch.tensor([1.0, 1, 1, 1, 0, 1])
    Lr = x.clone()

    def one(i):
        xc = torch.cat([Lr, x], 1)
        z = ou.encoder_forward(Pr, tr, xc, torch.full((1,), i, dtype=torch.long))
        zk, _ = osamp.brdf_schedule(z, z0, 0.95, i)
        return Lr + ou.unet_forward(Pi, ti, xc, t_emb=ou.z_embed(Pz, zk - z0))

    tw = time.time()
    one(0)  # warm-up (also bounds the sample: a slow host gets fewer timed steps)
    tw = time.time() - tw
    budget = 15.0
    n, t0 = 0, time.time()
    while n < 1 or (n < 12 and (time.time() - t0) + tw < budget):
        one(n)
        n += 1
    dt = time.time() - t0
    return {"value": round(n / dt, 4), "unit": "U-Net denoise steps/sec", "cores": threads, "kind": "port",
            "sample": f"{n} DRMNet reverse steps (RefNet+IllNet, fp32) of 1 refmap 3x{H}x{W}, oracle/ on host CPU, {dt:.1f}s"}

DOMINANT_VARIANTS = {"f16x3": "void drm::conv_split2_kernel<9, 16, 16, 4, 2, 2, 2, 2, 3, 3, false, false, false>",
                     "f16mx": "void drm::conv_split2_kernel<9, 16, 16, 4, 2, 2, 2, 2, 3, 2, false, false, false>"}


def profile_variants(L) -> dict:
    """name -> {kind, launches, ms, flops, bytes}: the library profiler's per-instantiation totals (drm_profile_variants), as of the last collect"""
    need = L.drm_profile_variants(None, 0)
    buf = C.create_string_buffer(int(need) + 16)
    L.drm_profile_variants(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, kind, n, ms, fl, by = line.split("\t")
        out[name] = {"kind": int(kind), "launches": int(n), "ms": float(ms), "flops": float(fl), "bytes": float(by)}
    return out


def kernel_source_hash() -> str:
    import hashlib

    with open(os.path.join(ROOT, "drmnet_amd", "csrc", "conv_split2.hip"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def imported_traffic(applicable: bool, precision: str = "f16x3") -> dict:
    import glob

    out = {"traffic": None}
    DOMINANT_VARIANT = DOMINANT_VARIANTS.get(precision)
    if not applicable or DOMINANT_VARIANT is None:
        return out
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic*.json")), reverse=True):
        try:
            with open(path) as f:
                prof = json.load(f)
            k = prof["kernels"][DOMINANT_VARIANT]
        except (OSError, KeyError, ValueError):
            continue
        rel = os.path.relpath(path, ROOT)
        if prof.get("conv_split2_sha16") != kernel_source_hash():
            out["traffic_note"] = f"{rel} was measured on a different conv_split2.hip (sha {prof.get('conv_split2_sha16')}): not imported"
            return out
        out["traffic"] = k["hbm_bytes_per_launch_corrected"]
        out["traffic_kernel"] = DOMINANT_VARIANT
        out["traffic_source"] = f"imported from {rel} (2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes, per launch of {DOMINANT_VARIANT}; same kernel source, sha {prof['conv_split2_sha16']})"
        return out
    return out


def under_profiler() -> bool:
    return any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")


CHILD_ENV_DROP = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "DRM_BENCH_DIST",
                  "TORCHELASTIC_RUN_ID", "DRM_LIB_PATH_CHILD")


def pmc_child(args, counters, timeout_s: int = 300):
    """One child run of this script under `rocprofv3 --pmc <counters> --kernel-trace` (a fresh process in its own session: never an exec, this
    process holds the GPU; the whole process group is killed on a timeout; rank / rendezvous variables are not inherited).  Returns
    {kernel name: {"n": dispatches, "duration_ns": sum, counter: sum, ...}} or None."""
    import glob
    import shutil
    import signal
    import sqlite3
    import subprocess
    import tempfile

    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp) or under_profiler():
        return None
    env = {k: v for k, v in os.environ.items() if k not in CHILD_ENV_DROP}
    env["TMPDIR"] = "/tmp"
    try:
        with tempfile.TemporaryDirectory(prefix="drm_pmc_", dir="/tmp") as td:
            cmd = [rp, "--pmc", *counters, "--kernel-trace", "-d", td, "--", sys.executable, os.path.abspath(__file__), "--precision", args.precision,
                   "--batch", str(args.batch), "--height", str(args.height), "--width", str(args.width), "--steps", "2", "--warmup", "1",
                   "--no-cpu-baseline", "--no-profile", "--no-parity-check", "--no-strict-fp32", "--no-secondary", "--no-live-traffic"]
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)  # rocprofv3 AND the python it started
                except ProcessLookupError:
                    pass
                pr.wait()
                return None
            dbs = sorted(glob.glob(os.path.join(td, "**", "*_results.db"), recursive=True))
            if rc != 0 or not dbs:
                return None
            con = sqlite3.connect(dbs[-1])
            cols = [c[1] for c in con.execute("pragma table_info(counters_collection)")]
            kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
            dcol = ", duration" if "duration" in cols else ", 0"
            per, seen = {}, set()
            for name, d, cname, v, dur in con.execute(f"select {kcol}, dispatch_id, counter_name, value{dcol} from counters_collection"):
                k = per.setdefault(name.split("(")[0], {"n": 0, "duration_ns": 0.0})
                k[cname] = k.get(cname, 0.0) + v
                if d not in seen:
                    seen.add(d)
                    k["n"] += 1
                    k["duration_ns"] += dur or 0.0
            con.close()
            return per
    except Exception:  # noqa: BLE001 -- any failure of the side measurement falls back to the committed profile
        return None


FAMILY_1X1 = ("conv_split2_kernel<1,",)
FAMILY_ATTN = ("attn_", "softmax_rows", "pack_attn", "qk_small", "pv_small", "bgemm64s")


def live_traffic(args, dominant: str, timeout_s: int = 300) -> dict:
    """roofline.traffic measured IN this run: two child runs of this same script under `rocprofv3 --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` (separate passes, --kernel-trace only: MI355X_MICROARCH.md HBM section), three steps each; per launch of `dominant`:
    2 x FETCH_SIZE + WRITE_SIZE (KB; gfx950 reports half of wide streaming reads).  [r5] The same passes give the per-step traffic of the 1x1 conv
    family and of the attention core's kernels (roofline_conv1x1 / roofline_attention), and a third child run (SQ_VALU_MFMA_BUSY_CYCLES,
    GRBM_GUI_ACTIVE) the dominant variant's MFMA-busy fraction and the clock the chip held.  Returns {} when rocprofv3 is not there, a child fails or
    its database cannot be read -- the committed profile is imported then."""
    fe, wr = pmc_child(args, ["FETCH_SIZE"], timeout_s), None
    if fe is not None:
        wr = pmc_child(args, ["WRITE_SIZE"], timeout_s)
    if fe is None or wr is None or dominant not in fe or dominant not in wr:
        return {}
    f_dom, w_dom = fe[dominant], wr[dominant]
    f_kb, w_kb = f_dom["FETCH_SIZE"] / f_dom["n"], w_dom["WRITE_SIZE"] / w_dom["n"]
    out = {"traffic": int((2.0 * f_kb + w_kb) * 1024), "traffic_kernel": dominant,
           "traffic_source": (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child runs of this command (separate passes, --kernel-trace only; "
                              f"{f_dom['n']} / {w_dom['n']} launches of the dominant variant), 2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes; raw per launch: "
                              f"FETCH_SIZE {f_kb:.1f} KB, WRITE_SIZE {w_kb:.1f} KB")}
    steps_in_child = 3.0  # 1 warm-up + 2 timed steps, nothing else (every appended workload is switched off in the child)
    for key, pats in (("family_traffic_conv1x1", FAMILY_1X1), ("family_traffic_attention", FAMILY_ATTN)):
        f_sum = sum(v.get("FETCH_SIZE", 0.0) for k, v in fe.items() if any(p in k for p in pats))
        w_sum = sum(v.get("WRITE_SIZE", 0.0) for k, v in wr.items() if any(p in k for p in pats))
        n = sum(v["n"] for k, v in fe.items() if any(p in k for p in pats))
        if n:
            out[key] = {"bytes_per_step": int((2.0 * f_sum + w_sum) * 1024 / steps_in_child), "launches_per_step": round(n / steps_in_child, 1)}
    sq = pmc_child(args, ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"], timeout_s)
    if sq is not None and dominant in sq and sq[dominant].get("GRBM_GUI_ACTIVE"):
        d = sq[dominant]
        gui = d["GRBM_GUI_ACTIVE"] / d["n"]  # summed over the 8 XCDs
        out["mfma_busy"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["n"] / (1024.0 * gui / 8.0), 3)
        if d["duration_ns"]:
            out["clock_ghz"] = round(gui / 8.0 / (d["duration_ns"] / d["n"]), 3)
        out["mfma_busy_source"] = (f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE child run of this command ({d['n']} launches of the dominant variant): "
                                   "busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); clock = GRBM_GUI_ACTIVE / 8 / launch duration (MI355X_MICROARCH.md, DVFS give-back)")
    return out


def rank_aggregate(dt_local: float, units_local: float, dist=None, device=None):
    """The contract's aggregation: time = MAX over ranks of the barrier-bracketed loop, work = SUM over ranks of the units each
    rank processed; value = work / time.  `dist` = an initialised torch.distributed module (nccl on GPUs, gloo in the CPU test)
    or None for a single process.  Returns (time_s, total_units)."""
    if dist is None:
        return dt_local, units_local
    t = torch.tensor([dt_local], dtype=torch.float64, device=device)
    u = torch.tensor([units_local], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def self_launch(args, command=None, have=None, poll_s=0.2) -> int:
    """`python bench.py --gpus N` with no launcher around it: start one fresh child process per GPU (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set as torch.distributed.run would), wait for them, return the worst exit code.  Runs before this
    process has made any HIP call (torch.cuda.device_count() does not initialise the runtime): a process that has touched the GPU
    must never be replaced or forked on this pool.  Rank 0's stdout (the JSON line) is inherited; other ranks' stdout is dropped.
    A rank that exits non-zero ends the job: the surviving ranks (parked at their next barrier, where rank 0 would otherwise wait
    out the collective's timeout) are terminated by PID, so a failed rank yields rc != 0 and NO JSON line -- rank 0 prints it only
    after the last barrier.  `command` / `have` are test hooks (the child command line; the visible GPU count)."""
    import socket
    import subprocess

    n = args.gpus
    have = torch.cuda.device_count() if have is None else have
    if have < n:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    command = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] if command is None else list(command)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(command, env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc, live = 0, list(procs)
    while live:
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            rc = max(rc, abs(code))
        if rc != 0 and live:  # a rank failed: the others can only hang at a barrier
            for pr in live:
                pr.terminate()
            for pr in live:
                try:
                    pr.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    pr.kill()
                    pr.wait()
            print(f"bench.py: a rank exited with code {rc}; the remaining {len(live)} rank(s) were terminated, no result line", file=sys.stderr)
            break
        if live:
            time.sleep(poll_s)
    return rc


def model_nets(model) -> dict:
    if isinstance(model, tuple):
        return {"illnet": model[0].illnet_model.diffusion_model, "refnet": model[0].refnet_model.diffusion_model, "obsnet": model[1].model.diffusion_model}
    if hasattr(model, "illnet_model"):
        return {"illnet": model.illnet_model.diffusion_model, "refnet": model.refnet_model.diffusion_model}
    return {"obsnet": model.model.diffusion_model} if hasattr(model, "model") else {}


def restore_modes(model, chosen: dict) -> None:
    """Every network back to the mode the headline ran it in (auto mode may have chosen differently per network: ADVICE r4)."""
    for k, net in model_nets(model).items():
        if k in chosen and net.precision != chosen[k]:
            net._set_mode(chosen[k])


def timed_state_check(args, model, dev, initial, i0: int, final) -> dict:
    """the state the TIMED loop produced, against the same steps from the same state in f16x3 (the three-product split mode every
    test holds to the fp32 tolerances).  Same reverse-step indices, same Philox keys; rel-L2 of the final states must stay inside the 1e-4 contract."""
    for net in model_nets(model).values():
        net._set_mode("f16x3")
    step, _, _ = make_step(args, model, dev)
    step.state.copy_(initial)
    step.counter["i"] = i0
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    a, b = final.double().flatten(1), step.state.double().flatten(1)
    rows = ((a - b).norm(dim=1) / b.norm(dim=1).clamp_min(1e-300))
    return {"timed_state_rel_l2": float(((a - b).norm() / b.norm()).item()), "worst_row": float(rows.max().item()), "steps": args.steps, "against": "f16x3",
            "tolerance": 1e-4, "note": "the final state of the timed loop vs the same steps (indices, Philox keys) from the same initial state in f16x3"}


def strict_fp32_pass(args, model, dev, L, _lib, precision="fp32"):
    """The same workload in exact-fp32 arithmetic (v_mfma_f32_32x32x2_f32) -- or, with precision="f16x3", in the three-product split mode --
    a short pass run AFTER the headline measurement so the driver's record carries every accurate mode's number next to the headline's."""
    model.set_precision(precision)
    step, gflop, _ = make_step(args, model, dev)
    for _ in range(2):
        step()
    torch.cuda.synchronize(dev)
    L.drm_profile_reset()
    L.drm_profile_enable(2)
    n = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        step()
    e1.record()
    torch.cuda.synchronize(dev)
    dt = e0.elapsed_time(e1) * 1e-3  # device time between two events on the launch stream
    L.drm_profile_enable(0)
    K0 = 5
    tm, tf, tb, tn = (C.c_double * K0)(), (C.c_double * K0)(), (C.c_double * K0)(), (C.c_int64 * K0)()
    _lib.check(L.drm_profile_collect(tm, tf, tb, tn))
    ach = tf[0] / (tm[0] * 1e-3) / 1e12 if tm[0] > 0 else None
    if precision == "f16x3":
        return {"value": round(args.batch * n / dt, 3), "unit": "denoise steps/sec (samples x steps / s)", "steps": n, "ms_per_step": round(dt / n * 1e3, 3),
                "dtype": "f32 via split f16x3 MFMA (fp16 hi+lo operands, 3 MFMAs per product, fp32 accumulate; ~2e-6 rel-L2 against the reference)",
                "roofline": None if ach is None else {"bound": "mfma", "kernel": "conv_split2_kernel<9,...,TERMS=3>", "achieved": round(ach, 2), "peak": F16_MFMA_PEAK_TFLOPS,
                                                      "unit": "TFLOP/s", "frac": round(ach / F16_MFMA_PEAK_TFLOPS, 4), "executed_frac_of_f16_peak": round(3 * ach / F16_MFMA_PEAK_TFLOPS, 4),
                                                      "avg_launch_ms": round(tm[0] / max(tn[0], 1), 4)}}
    return {"value": round(args.batch * n / dt, 3), "unit": "denoise steps/sec (samples x steps / s)", "steps": n, "ms_per_step": round(dt / n * 1e3, 3),
            "dtype": "f32 (v_mfma_f32_32x32x2_f32, exact fp32 products)",
            "roofline": None if ach is None else {"bound": "mfma", "kernel": "conv_split2_kernel<9,...,TERMS=0> (the same LDS-DMA pipeline kernel on fp32 operands, v_mfma_f32_32x32x2_f32)", "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                                                  "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "avg_launch_ms": round(tm[0] / max(tn[0], 1), 4)}}


def family_rooflines(step, steps, gflop, batch, dt_timed, precision):
    """Per-family kernel breakdown and the three matrix-family roofline objects of `step` from an instrumented pass of `steps` steps (the library's
    launch profiler: HIP events on the launch stream around every launch, algorithmic FLOPs / bytes from the launch arguments)."""
    import ctypes as C

    from drmnet_amd import _lib

    L = _lib.lib()
    L.drm_profile_reset()
    L.drm_profile_enable(1)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    L.drm_profile_enable(0)
    K = 5
    ms, fl, by, n = (C.c_double * K)(), (C.c_double * K)(), (C.c_double * K)(), (C.c_int64 * K)()
    _lib.check(L.drm_profile_collect(ms, fl, by, n))
    variants = {k_: v_ for k_, v_ in profile_variants(L).items() if v_["kind"] == 0}  # (3x3 family only)
    names = ["conv3x3_gn_silu_igemm", "conv1x1_igemm", "attention_core", "gn_channel_moments", "other"]
    peak = F16_MFMA_PEAK_TFLOPS if precision != "fp32" else FP32_MFMA_PEAK_TFLOPS
    res = {"kernel_breakdown": {names[k]: {"ms_per_step": round(ms[k] / steps, 3), "launches_per_step": round(n[k] / steps, 1),
                                           "tflops": round(fl[k] / ms[k] / 1e9, 2) if ms[k] > 0 else None,
                                           "algorithmic_GBps": round(by[k] / ms[k] / 1e6, 1) if ms[k] > 0 else None} for k in range(K) if n[k] > 0},
           "kernel_breakdown_note": f"instrumented pass of {steps} steps (single stream), not the timed one; timed step {dt_timed * 1e3:.2f} ms",
           "achieved_tflops": round(batch / dt_timed * gflop / 1e3, 2)}
    if n[0] > 0 and ms[0] > 0:
        ach = fl[0] / (ms[0] * 1e-3) / 1e12
        res["roofline"] = {"bound": "mfma", "kernel": "conv_split2_kernel<9,...> (3x3 family)", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                           "frac": round(ach / peak, 4), "launches_per_step": round(n[0] / steps, 1), "avg_launch_ms": round(ms[0] / n[0], 4),
                           "ms_per_step": round(ms[0] / steps, 3), "traffic": None}
        if variants:
            dom = max(variants.items(), key=lambda kv: kv[1]["ms"])
            dv = dom[1]
            dach = dv["flops"] / (dv["ms"] * 1e-3) / 1e12
            res["roofline"]["dominant_variant"] = {"kernel": dom[0], "launches_per_step": round(dv["launches"] / steps, 1), "avg_launch_ms": round(dv["ms"] / dv["launches"], 4),
                                                   "achieved": round(dach, 2), "frac": round(dach / peak, 4)}
    if n[1] > 0 and ms[1] > 0:
        gbs = by[1] / (ms[1] * 1e-3) / 1e9
        res["roofline_conv1x1"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                                   "tflops": round(fl[1] / (ms[1] * 1e-3) / 1e12, 2), "ms_per_step": round(ms[1] / steps, 3), "launches_per_step": round(n[1] / steps, 1), "traffic": None}
    if n[2] > 0 and ms[2] > 0:
        tf_ = fl[2] / (ms[2] * 1e-3) / 1e12
        res["roofline_attention"] = {"bound": "mfma", "achieved": round(tf_, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf_ / peak, 4),
                                     "ms_per_step": round(ms[2] / steps, 3), "launches_per_step": round(n[2] / steps, 1), "traffic": None}
    return res


def secondary_pass(args, model, dev):
    """The rest of BASELINE's metric and configs in the same driver line, each a short device-timed run AFTER the
    headline measurement: the DRMNet step at 128 refmaps per GPU (north-star: batch 1024 over 8 GPUs) and at batch 1 @128x128 (the
    reference's own use, scripts/estimate.py), the ObsNet DDIM-50 chain at batch 256 (configs[2]; fp32-accurate split mode and the
    reduced-precision f16 mode, eager and hipGraph replay), and the full chain object image -> Lr0 (configs[4], metric part 2:
    "full-chain samples/sec") at 64 objects per GPU (batch 512 over 8 GPUs) with per-stage device time."""
    import copy

    from drmnet_amd import ops
    from drmnet_amd.config import instantiate_from_config, load_config
    from drmnet_amd.ddim import DDIMSampler
    from drmnet_amd.estimate import estimate_batch
    from drmnet_amd import synth

    def timed(fn, n, warm=1):
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) * 1e-3 / n

    out = {"note": "device time between two events on the launch stream; same seeded synthetic weights and inputs as the headline"}
    unit = "denoise steps/sec (samples x steps / s)"
    for name, (b, h, w, n) in {"drmnet_step_b128_3x128x256": (128, 128, 256, 4), "drmnet_step_b1_3x128x128": (1, 128, 128, 30)}.items():
        a = copy.copy(args)
        a.batch, a.height, a.width = b, h, w
        step, gf, _ = make_step(a, model, dev)
        dt = timed(step, n, warm=2)
        out[name] = {"value": round(b / dt, 2), "unit": unit, "ms_per_step": round(dt * 1e3, 3), "steps": n, "precision": args.precision}
        if b == 128 and not args.no_profile:
            # the north-star's per-GPU batch with its own rooflines: a second, UNTIMED pass with every kernel family
            # instrumented (HIP events around every launch on the launch stream; the profiled pass runs the rows on one stream -- the forked
            # row ranges of the timed pass cannot be bracketed by events of one stream)
            try:
                out[name].update(family_rooflines(step, 2, gf, b, dt, args.precision))
            except Exception as e:  # noqa: BLE001
                out[name]["rooflines_error"] = f"{type(e).__name__}: {e}"
    torch.cuda.empty_cache()

    # ---- ObsNet DDIM-50 chain, batch 256 @3x128x256 (BASELINE configs[2])
    acc = args.precision  # the accurate mode the headline ran in
    obs = build_models("obsnet", dev, getattr(args, "precision_requested", acc))
    if getattr(args, "precision_requested", acc) == "auto":  # --precision auto: ObsNet measures itself too (network probe + DDIM chain probe)
        obs.calibrate_precision()
        out["obsnet_precision_auto"] = dict(obs.model.diffusion_model.auto_report or {}, chain=obs.auto_chain_report)
        acc = obs.model.diffusion_model.precision
        obs._auto_chain = None  # (the explicit modes below are not to be second-guessed)
    out["obsnet_parity"] = obsnet_parity(obs, dev)
    x = synth.synth_refmaps(256, 128, 256, synth.SEED_INPUT).to(dev)
    xT = torch.randn(x.shape, generator=torch.Generator().manual_seed(6)).to(dev)
    chains = {}
    for prec in ((acc, "f16x3", "f16", "bf16") if acc != "f16x3" else ("f16x3", "f16", "bf16")):
        obs.set_precision(prec)
        smp = DDIMSampler(obs)
        smp.make_schedule(50, ddim_eta=1.0, verbose=False)
        for graph in (False, True):
            ops.set_graph_replay(graph)
            res = {}

            def chain():
                res["x"], _ = smp.ddim_sampling(x, tuple(x.shape), x_T=xT, seed=1)

            dt = timed(chain, 1, warm=0 if chains else 1)
            chains[f"{prec}_{'graph' if graph else 'eager'}"] = {"value": round(256 * 50 / dt, 1), "unit": unit, "s_per_chain": round(dt, 3),
                                                                  "finite": bool(torch.isfinite(res["x"]).all().item())}
    ops.set_graph_replay(False)
    # the literal wording of configs[1] ("reverse DDPM 1000-step, batch 32"): the only 1000-step schedule of the reference is ObsNet's ancestral
    # sampler: 100 of its 1000 steps through the device loop (the per-step cost does not depend on t), fp32-accurate mode
    try:
        obs.set_precision(acc)
        x32, xT32 = x[:32].contiguous(), xT[:32].contiguous()
        res = {}

        def ddpm():
            res["x"] = obs.p_sample_loop(x32, tuple(x32.shape), x_T=xT32, verbose=False, start_T=100, seed=3)

        dt = timed(ddpm, 1, warm=1)
        out["obsnet_ancestral_ddpm_b32_3x128x256"] = {"value": round(32 * 100 / dt, 1), "unit": unit, "s_per_100_steps": round(dt, 3), "steps_timed": 100,
                                                     "of_schedule": 1000, "finite": bool(torch.isfinite(res["x"]).all().item()), "precision": acc}
    except Exception as e:  # noqa: BLE001
        out["obsnet_ancestral_ddpm_b32_3x128x256"] = {"error": f"{type(e).__name__}: {e}"}
    chains["note"] = ("f16mx / f16x3 = the two accurate split modes (1e-4 contract; cross terms on the block-scaled fp8 MFMA / on the f16 MFMA); bf16 = configs[2] AS WRITTEN "
                      "(bf16 operands on v_mfma_f32_32x32x16_bf16, fp32 accumulate; 3e-2 tolerance, tests/test_gpu_bf16.py); f16 = the same kernels on fp16 operands "
                      "(three more mantissa bits at the same speed; 5e-3 tolerance, tests/test_gpu_configs.py)")
    out["obsnet_ddim50_chain_b256_3x128x256"] = chains
    del x, xT
    torch.cuda.empty_cache()

    # ---- full chain (BASELINE configs[4]; metric part 2): 64 object images per GPU (batch 512 over 8 GPUs), 256x256 -> 128x128 refmaps
    obs.set_precision(acc)
    obs.ds = instantiate_from_config(load_config(os.path.join(ROOT, "configs/obsnet/eval_obsnet.yaml"))["data"]["params"]["predict"])
    model.ds = instantiate_from_config(load_config(os.path.join(ROOT, "configs/drmnet/eval_drmnet.yaml"))["data"]["params"]["predict"])
    B = 64
    imgs, normals, masks = chain_inputs(B, dev)
    fc = {"objects_per_gpu": B, "refmap": "3x128x128 from 256x256 object images", "ddim_steps": obs.ddim_steps, "max_timesteps": model.max_timesteps}
    for tag, ee in (("early_exit_off", False), ("natural_early_exit", True)):
        estimate_batch(model, obs, imgs, normals, masks, early_exit=ee, seed=7)  # warm-up (workspace, kernel attributes)
        tm = {}
        t0 = time.perf_counter()
        _, _, K = estimate_batch(model, obs, imgs, normals, masks, early_exit=ee, seed=8, hooks={"timing": tm})
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        steps = int(K.sum().item()) if ee else B * model.max_timesteps
        fc[tag] = {"value": round(B / dt, 3), "unit": "object images/sec", "s_per_batch": round(dt, 3), "stage_ms": {k: round(v, 1) for k, v in tm.items()},
                   "drmnet_sample_steps": steps, "K_min_max": [int(K.min().item()), int(K.max().item())]}
    # ---- the reference's own use: scripts/estimate.py on ONE object image (data/sample: 256x256 EXR + normals + mask), full-width networks
    try:
        import numpy as np
        from drmnet_amd import file_io
        from drmnet_amd.estimate import estimate

        d = os.path.join(ROOT, "tests", "golden", "sample")
        img = file_io.load_exr(os.path.join(d, "image.exr"), as_torch=True).to(dev)
        nrm = torch.from_numpy(np.load(os.path.join(d, "normal.npy"))).to(dev)
        msk = torch.logical_and(file_io.load_png(os.path.join(d, "mask.png"), as_torch=True).to(dev) > 0, torch.linalg.norm(nrm, dim=-1) > 0.5)
        estimate(model, obs, img, nrm, msk)  # warm-up
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        estimate(model, obs, img, nrm, msk)
        torch.cuda.synchronize(dev)
        dt1 = time.perf_counter() - t0
        out["estimate_single_image"] = {"value": round(dt1, 3), "unit": "s per object image (wall, host included)", "drmnet_steps": int(model.last_steps),
                                        "ddim_steps": obs.ddim_steps, "note": "estimate() on tests/golden/sample (the reference's data/sample), 128x128 refmap, batch 1, "
                                        "natural early exit of the synthetic RefNet"}
    except (OSError, KeyError) as e:  # sample files not shipped with this copy
        out["estimate_single_image"] = {"error": str(e)}
    fc["note"] = ("value = whole-call wall time (host included). Random-weight networks do not converge the way trained ones do: the natural-exit row shows the "
                  "early-exit machinery at whatever K the synthetic RefNet produces; the early-exit-off row is the countable one (150 steps per object)")
    out["full_chain"] = fc
    return out


def full_chain_all_ranks(args, model, dev, dist):
    """Metric part 2 at N GPUs ("full-chain samples/sec @1/8 GPU"): every rank runs the whole chain (object images -> refmaps -> ObsNet DDIM-50
    -> DRMNet loop, early exit off so the work is countable) on its own 64 synthetic objects -- batches shard by object, no collective on the
    path -- bracketed by barriers; time = MAX over ranks, work = SUM (rank_aggregate).  Runs after the headline measurement, on N > 1 only
    (at N = 1 the `secondary.full_chain` object carries it with per-stage times).  A rank that fails still meets the others at every barrier."""
    from drmnet_amd.config import instantiate_from_config, load_config
    from drmnet_amd.estimate import estimate_batch

    B, err, run = 64, None, None  # configs[4]: batch 512 over 8 GPUs
    try:
        obs = build_models("obsnet", dev, args.precision)
        obs.ds = instantiate_from_config(load_config(os.path.join(ROOT, "configs/obsnet/eval_obsnet.yaml"))["data"]["params"]["predict"])
        model.ds = instantiate_from_config(load_config(os.path.join(ROOT, "configs/drmnet/eval_drmnet.yaml"))["data"]["params"]["predict"])
        imgs, normals, masks = chain_inputs(B, dev)
        run = lambda seed: estimate_batch(model, obs, imgs, normals, masks, early_exit=False, seed=seed)
        run(7)  # warm-up
        torch.cuda.synchronize(dev)
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    try:
        if err is None:
            run(8)
            torch.cuda.synchronize(dev)
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    if dist is not None:
        dist.barrier()
    dt, total = rank_aggregate(time.perf_counter() - t0, float(B) if err is None else 0.0, dist, dev)
    bad, _ = rank_aggregate(0.0, 0.0 if err is None else 1.0, dist, dev) if dist is not None else (0.0, 0.0 if err is None else 1.0)
    out = {"value": round(total / dt, 3) if dt > 0 else None, "unit": "object images/sec (all ranks)", "objects_per_gpu": B, "s_per_batch": round(dt, 3),
           "ddim_steps": 50, "max_timesteps": int(model.max_timesteps), "early_exit": False}
    if err is not None or _ > 0:
        out["error"] = err or f"{int(_)} other rank(s) failed"
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            sys.exit(self_launch(args))  # no launcher: spawn the ranks ourselves (before any HIP call in this process)
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: drmnet_amd has no CPU path", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    json_fd = None  # where the one JSON line goes when fd 1 had to be pointed away from stdout
    if world > 1 or os.environ.get("DRM_BENCH_DIST") == "1":  # (DRM_BENCH_DIST=1: exercise the RCCL path with a single rank)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on STDOUT when its communicator comes up; stdout carries the one JSON line only, so the
        # communicator is created (init + a first collective) with fd 1 pointed at stderr
        # -- and stays there for the whole run (the banner was also seen AFTER the JSON line, printed from a later collective / at teardown):
        # the JSON line is written to the saved descriptor of the real stdout
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist.barrier()
        torch.cuda.synchronize(dev)

    from drmnet_amd import _lib

    L = _lib.lib()
    model = build_models(args.workload, dev, args.precision)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # --precision auto is resolved HERE, before any warm-up or timed step (with --warmup 0 the probes would otherwise run inside the timed
    # region and the line would be labelled with the pre-calibration mode): every network measures f16mx against f16x3 on its loaded weights (probe
    # batch: unet.py), then the model's chain probe (eight reverse / DDIM steps) has the last word
    requested_precision = args.precision
    auto_report = None
    nets = model_nets(model)
    if args.precision == "auto":
        for mdl in (model if isinstance(model, tuple) else (model,)):
            mdl.calibrate_precision()
        auto_report = {k: v.auto_report for k, v in nets.items()}
        for mdl, tag in zip(model if isinstance(model, tuple) else (model,), ("drmnet_chain", "obsnet_chain") if isinstance(model, tuple) else
                            (("drmnet_chain",) if hasattr(model, "illnet_model") else ("obsnet_chain",))):
            auto_report[tag] = mdl.auto_chain_report
        chosen = {k: v.precision for k, v in nets.items()}
        # the arithmetic the line is labelled with = that of the network carrying the FLOPs (IllNet / ObsNet)
        args.precision = chosen.get("illnet") or chosen.get("obsnet") or "f16x3"
        args.precision_requested = requested_precision
        auto_report["chosen"] = chosen
        if dist is not None:  # the probes are seeded, so every rank must have come to the same choice
            code = torch.tensor([sum((1 << i) for i, k in enumerate(sorted(chosen)) if chosen[k] == "f16mx")], dtype=torch.int64, device=dev)
            lo, hi = code.clone(), code.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            if int(lo.item()) != int(hi.item()):
                print(f"bench.py: ranks disagree on the auto precision choice (codes {int(lo.item())} .. {int(hi.item())})", file=sys.stderr)
                sys.exit(4)
    chosen_modes = {k: v.precision for k, v in nets.items()}
    step, gflop, desc = make_step(args, model, dev)
    for _ in range(args.warmup):
        step()
    barrier()
    # the state the timed loop starts from (re-run in f16x3 afterwards: timed_state_check)
    state0 = getattr(step, "state", None)
    state0 = None if state0 is None or not hasattr(step, "counter") else (state0.clone(), int(step.counter["i"]))
    profile = not args.no_profile
    if profile:  # the timed region instruments the dominant kernel family only (HIP events around the fused 3x3 convs)
        L.drm_profile_reset()
        L.drm_profile_enable(2)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    # non-finite operands would toggle fewer datapath bits (and raise the clock): the timed state must still be a number
    state = getattr(step, "state", None)
    state_finite = None if state is None else bool(torch.isfinite(state).all().item())
    if state_finite is False:
        print("bench.py: the sampler state went non-finite inside the timed region; the measurement is invalid", file=sys.stderr)
        sys.exit(3)
    state_final = None if state0 is None else state.clone()
    timed = None
    timed_variants = {}
    if profile:
        L.drm_profile_enable(0)
        K0 = 5
        tm, tf, tb, tn = (C.c_double * K0)(), (C.c_double * K0)(), (C.c_double * K0)(), (C.c_int64 * K0)()
        _lib.check(L.drm_profile_collect(tm, tf, tb, tn))
        timed = (tm[0], tf[0], tb[0], int(tn[0]))
        timed_variants = profile_variants(L)  # per instantiation of the 3x3 family, inside the timed region
        # per-family breakdown: a second, UNTIMED pass of the same steps with every family instrumented
        L.drm_profile_reset()
        L.drm_profile_enable(1)
        for _ in range(args.steps):
            step()
        barrier()
        L.drm_profile_enable(0)
    dt, total_units = rank_aggregate(dt, float(args.batch * args.steps * getattr(step, "units", 1)), dist, dev)

    roofline = None
    breakdown = None
    extra = {}
    if profile:
        K = 5
        ms, fl, by, n = (C.c_double * K)(), (C.c_double * K)(), (C.c_double * K)(), (C.c_int64 * K)()
        _lib.check(L.drm_profile_collect(ms, fl, by, n))
        if timed is not None and timed[3] > 0:  # the conv3x3 / roofline figures come from the TIMED region
            ms[0], fl[0], by[0], n[0] = timed
        names = ["conv3x3_gn_silu_igemm", "conv1x1_igemm", "attention_core", "gn_channel_moments", "other"]
        breakdown = {names[k]: {"ms": round(ms[k], 3), "launches": int(n[k]), "tflops": round(fl[k] / ms[k] / 1e9, 2) if ms[k] > 0 else None,
                                "algorithmic_GBps": round(by[k] / ms[k] / 1e6, 1) if ms[k] > 0 else None} for k in range(K) if n[k] > 0}
        if n[0] > 0:
            ach = fl[0] / (ms[0] * 1e-3) / 1e12
            split = args.precision in ("f16x3", "f16mx")
            mx = args.precision == "f16mx"
            plain = args.precision in ("f16", "bf16")  # reduced precision (BASELINE configs[2]); never the default
            peak = F16_MFMA_PEAK_TFLOPS if (split or plain) else FP32_MFMA_PEAK_TFLOPS
            kname = ("conv_igemm_split_kernel<9,...> (fused GroupNorm+SiLU+conv3x3, fp16 hi/lo x3 v_mfma_f32_32x32x16_f16, fp32 accumulate; "
                     "achieved counts ALGORITHMIC FLOPs, the matrix cores execute 3x that)") if split else \
                "conv_split2_kernel<9,...,TERMS=0> (fused GroupNorm+SiLU+conv3x3, fp32 operands, v_mfma_f32_32x32x2_f32)"
            kname = kname.replace("conv_igemm_split_kernel", "conv_split2_kernel")
            if plain:
                kname = "conv_split2_kernel<9,...,TERMS=1> (fused GroupNorm+SiLU+conv3x3, fp16 operands, one v_mfma_f32_32x32x16_f16 per product, fp32 accumulate)"
            roofline = {"bound": "mfma", "kernel": kname,
                        "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                        "traffic": None, "launches": int(n[0]), "avg_launch_ms": round(ms[0] / n[0], 4),
                        "flops_per_launch": round(fl[0] / n[0], 1), "algorithmic_bytes_per_launch": round(by[0] / n[0], 1),
                        "share_of_step_time": round(ms[0] * 1e-3 / dt, 3)}
            if mx:
                roofline["kernel"] = ("conv_split2_kernel<9,...,TERMS=2> (fused GroupNorm+SiLU+conv3x3: fp16 hi*hi on v_mfma_f32_32x32x16_f16 + both cross terms in one "
                                      "block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3), fp32 accumulate: 128 matrix-pipe cycles per 32x32x32 block instead of 192)")
                roofline["matrix_pipe_cycles_vs_plain_f16"] = 2.0
                roofline["executed_frac_of_f16_peak"] = round(2 * ach / peak, 4)
            elif split:  # the matrix cores execute three f16 MFMAs per algorithmic product
                roofline["executed_mfma_tflops"] = round(3 * ach, 2)
                roofline["executed_frac_of_f16_peak"] = round(3 * ach / peak, 4)
                roofline["fp32_mfma_peak_for_reference"] = FP32_MFMA_PEAK_TFLOPS
                # what the matrix pipes sustain under the socket power limit on operands that look like data
                # (tools/mfma_peak.hip, profiles/r01_mfma_power_limit.json): 1.43-1.67 PFLOP/s, not the 2.5 PFLOP/s of the data sheet
                roofline["power_limited_f16_mfma_tflops_measured"] = [1430.0, 1670.0]
                roofline["executed_frac_of_power_limited_peak"] = [round(3 * ach / 1670.0, 4), round(3 * ach / 1430.0, 4)]
            # HBM traffic is a PMC quantity: it cannot be read from inside this process.  It is IMPORTED from the committed rocprofv3
            # --pmc FETCH_SIZE / WRITE_SIZE passes of this same command (tools/prof_round.sh -> profiles/rNN_pmc_hbm_traffic.json), per
            # launch of the dominant variant -- and only when that profile was taken on the kernel source this build was made from.
            tr = imported_traffic(split and args.workload == "drmnet_step" and (args.batch, args.height, args.width) == (32, 128, 256), args.precision)
            # ... and it is set against the algorithmic bytes of THE SAME instantiation's launches (the library profiler's per-variant
            # totals of the timed region), not the family mean over all 3x3 variants
            dom = max(timed_variants.items(), key=lambda kv: kv[1]["ms"])[0] if timed_variants else None
            # [r4] ... and since round 4 it is MEASURED in this run where that is possible (single process, the headline shape): two short
            # rocprofv3 --pmc child runs of this command; the committed profile stays as the fallback and as a cross-check
            if (dom is not None and world == 1 and split and args.workload == "drmnet_step" and (args.batch, args.height, args.width) == (32, 128, 256)
                    and not args.no_live_traffic):
                live = live_traffic(args, dom)
                if live.get("traffic") is not None:
                    if tr.get("traffic") is not None:
                        live["traffic_committed_profile"] = tr["traffic"]
                    tr = live
            if dom is not None:
                dv = timed_variants[dom]
                dv_ach = dv["flops"] / (dv["ms"] * 1e-3) / 1e12
                roofline["dominant_variant"] = {
                    "kernel": dom, "launches": dv["launches"], "avg_launch_ms": round(dv["ms"] / dv["launches"], 4), "share_of_step_time": round(dv["ms"] * 1e-3 / dt, 3),
                    "achieved": round(dv_ach, 2), "frac": round(dv_ach / peak, 4), "flops_per_launch": round(dv["flops"] / dv["launches"], 1),
                    "algorithmic_bytes_per_launch": round(dv["bytes"] / dv["launches"], 1), "traffic": None, "traffic_ratio": None}
                if tr.get("traffic") is not None and tr.get("traffic_kernel") == dom:
                    roofline["dominant_variant"]["traffic"] = tr["traffic"]
                    roofline["dominant_variant"]["traffic_ratio"] = round(tr["traffic"] / (dv["bytes"] / dv["launches"]), 3)
            for k_ in ("mfma_busy", "clock_ghz", "mfma_busy_source"):
                if k_ in tr:
                    roofline[k_] = tr[k_]
            family_traffic = {k_: tr[k_] for k_ in ("family_traffic_conv1x1", "family_traffic_attention") if k_ in tr}
            roofline["traffic"] = tr.get("traffic")
            roofline["traffic_is_for"] = "dominant_variant (see that object for the matching algorithmic bytes and the ratio)" if tr.get("traffic") is not None else None
            for k_ in ("traffic_source", "traffic_note", "traffic_committed_profile"):
                if k_ in tr:
                    roofline[k_] = tr[k_]
        # the two other matrix families of the step (second, untimed pass: every family instrumented)
        if n[0] <= 0:
            family_traffic = {}
        for key_, k_idx, bound in (("roofline_conv1x1", 1, "hbm"), ("roofline_attention", 2, "mfma")):
            if n[k_idx] > 0 and ms[k_idx] > 0:
                tf_ = fl[k_idx] / (ms[k_idx] * 1e-3) / 1e12
                gbs = by[k_idx] / (ms[k_idx] * 1e-3) / 1e9
                mpeak = F16_MFMA_PEAK_TFLOPS if args.precision != "fp32" else FP32_MFMA_PEAK_TFLOPS
                extra[key_] = ({"bound": "hbm", "kernel": "conv_split2_kernel<1,...> (skip_connection / qkv / proj_out 1x1 convs; HBM-bound at the top levels, staging-bound below)",
                                "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4), "tflops": round(tf_, 2)}
                               if bound == "hbm" else
                               {"bound": "mfma", "kernel": "attention core (S = q k^T, softmax, P v; scores through the workspace)", "achieved": round(tf_, 2), "peak": mpeak,
                                "unit": "TFLOP/s", "frac": round(tf_ / mpeak, 4), "algorithmic_GBps": round(gbs, 1)})
                extra[key_].update({"launches": int(n[k_idx]), "ms_per_step": round(ms[k_idx] / args.steps, 3), "traffic": None,
                                    "measured_in": "second, untimed pass of the same steps with every family instrumented"})
                ft = family_traffic.get("family_traffic_conv1x1" if k_idx == 1 else "family_traffic_attention")
                if ft:  # PMC bytes of the family's kernels per step, from the same child runs as roofline.traffic (2 x FETCH_SIZE + WRITE_SIZE)
                    alg = by[k_idx] / args.steps
                    extra[key_].update({"traffic": ft["bytes_per_step"], "traffic_unit": "HBM bytes per step (all launches of the family)", "algorithmic_bytes_per_step": round(alg, 1),
                                        "traffic_ratio": round(ft["bytes_per_step"] / alg, 3) if alg > 0 else None, "traffic_launches_per_step": ft["launches_per_step"]})

    chain_all = None
    if (world > 1 or dist is not None) and args.workload == "drmnet_step" and args.precision in ("f16x3", "f16mx") and not args.no_secondary:
        chain_all = full_chain_all_ranks(args, model, dev, dist)  # (DRM_BENCH_DIST=1 exercises it with a single rank)
    if rank == 0:
        value = total_units / dt
        out = {
            "metric": "full-chain samples/sec" if args.workload == "estimate_chain" else "U-Net denoise steps/sec on 3x128x256 refmaps",
            "value": round(value, 3),
            "unit": "object images/sec" if args.workload == "estimate_chain" else "denoise steps/sec (samples x steps / s)",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "scaling_measured": bool(world > 1),  # (a single-GPU line carries no multi-GPU figure, measured or projected)
            "vs_baseline": None,
            "dtype": {"fp32": "f32", "f16x3": "emulated fp32: f16 hi/lo split, 3 MFMAs per product (22-bit operands), fp32 acc; 1e-4 contract",
                      "f16": "f16 operands, fp32 acc (reduced precision)",
                      "bf16": "bf16 operands, fp32 acc (reduced precision)",
                      "f16mx": "emulated fp32: f16 hi*hi + e4m3 cross terms (~15-bit products), fp32 acc; 1e-4 contract"}[args.precision],
            "dtype_note": {"fp32": "v_mfma_f32_32x32x2_f32, exact fp32 products",
                           "f16x3": "every fp32 operand split into fp16 hi + lo, 3 MFMAs per product, fp32 accumulate: ~2e-6 rel-L2 against the reference (the fp32 tolerances)",
                           "f16": "REDUCED PRECISION, ~1e-3 rel-L2: not the headline configuration",
                           "bf16": "REDUCED PRECISION (BASELINE configs[2] as written), ~1e-2 rel-L2: not the headline configuration",
                           "f16mx": "f16x3 with the GroupNorm-fed 3x3 convs on fp16 hi*hi + ONE block-scaled fp8 (e4m3) MFMA for both cross terms; emulated fp32 with ~15-bit "
                                    "products: 2.4e-5 .. 4.0e-5 rel-L2 per network against the reference at B = 1 .. 256, <= 5e-5 on the 150-step / 1000-step reference chains "
                                    "(1e-4 contract; tests/test_gpu_*.py run every BASELINE-shaped case in this mode); the f16x3 and exact-fp32 figures of the same run are "
                                    "in the f16x3 / strict_fp32 objects of this line"}[args.precision],
            "precision_requested": requested_precision,
            "precision_auto": auto_report,
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "batch_per_gpu": args.batch, "refmap": "3x128x128 (from 256x256 object images)" if args.workload == "estimate_chain" else f"3x{args.height}x{args.width}",
                       "weights": "seeded synthetic (no checkpoint offline)", "parallelism": f"batch-sharded x{world}, no collective",
                       "algorithmic_gflop_per_sample_step": gflop, "achieved_tflops_per_gpu": round(value / world * gflop / 1e3, 2)},
            "roofline": roofline,
            **extra,
            "kernel_breakdown": breakdown,
            "kernel_breakdown_note": "conv3x3 row and the roofline object: HIP events inside the timed region; other rows: a second, untimed pass of the same steps with every kernel family instrumented",
        }
        out["state_finite"] = state_finite
        if state0 is not None and world == 1 and args.workload == "drmnet_step" and args.precision == "f16mx" and not args.no_parity_check:
            try:
                out["timed_state_check"] = timed_state_check(args, model, dev, state0[0], state0[1], state_final)
            except Exception as e:  # noqa: BLE001
                out["timed_state_check"] = {"error": f"{type(e).__name__}: {e}"}
            restore_modes(model, chosen_modes)
            if out["timed_state_check"].get("timed_state_rel_l2", 0.0) >= 1e-4:
                print(f"bench.py: the timed state differs from its f16x3 re-run by {out['timed_state_check']['timed_state_rel_l2']:.2e} (contract 1e-4): "
                      "the headline arithmetic does not hold on this state", file=sys.stderr)
                print(json.dumps(out), flush=True)
                sys.exit(3)
        if chain_all is not None:
            out["full_chain_all_gpus"] = chain_all
        if args.workload == "drmnet_step" and not args.no_parity_check:
            out["parity_check"] = parity_check(model, dev, args.precision)
        if world == 1 and args.workload == "drmnet_step" and args.precision in ("f16x3", "f16mx") and not args.no_strict_fp32:
            if args.precision == "f16mx":  # the three-product split mode (every test at the fp32 tolerances) on the same workload
                try:
                    out["f16x3"] = strict_fp32_pass(args, model, dev, L, _lib, "f16x3")
                except Exception as e:  # noqa: BLE001
                    out["f16x3"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                out["strict_fp32"] = strict_fp32_pass(args, model, dev, L, _lib)
            except Exception as e:  # noqa: BLE001
                out["strict_fp32"] = {"error": f"{type(e).__name__}: {e}"}
            # the two other accurate arithmetics of the same run, at the top level of the line: `value` is the tolerance-mode figure
            # (narrower products than the reference's fp32, inside the 1e-4 contract); value_fp32_exact is BASELINE configs[1]'s "fp32" as written
            out["value_fp32_exact"] = out["strict_fp32"].get("value") if isinstance(out.get("strict_fp32"), dict) else None
            out["value_f16x3"] = out["f16x3"].get("value") if isinstance(out.get("f16x3"), dict) else (round(value, 3) if args.precision == "f16x3" else None)
            out["value_note"] = ("value: the mode named in dtype (emulated fp32 inside the 1e-4 rel-L2 contract, measured live in parity_check / timed_state_check); "
                                 "value_f16x3: 22-bit split operands (~2e-6); value_fp32_exact: v_mfma_f32_32x32x2_f32 -- all on this workload, same run")
            if requested_precision == "auto":
                model.set_precision("auto")  # (back to auto mode: the stored reports put every network on its own choice again, nothing is re-measured)
                model.calibrate_precision()
            restore_modes(model, chosen_modes)
        if world == 1 and args.workload == "drmnet_step" and args.precision in ("f16x3", "f16mx") and not args.no_secondary and (args.batch, args.height, args.width) == (32, 128, 256):
            try:  # (the headline line must survive a failure of an appended workload)
                restore_modes(model, chosen_modes)
                out["secondary"] = secondary_pass(args, model, dev)
                if "parity_check" in out and "obsnet_parity" in out["secondary"]:  # the third network, in the mode its chains ran in
                    out["parity_check"]["cases"].update(out["secondary"].pop("obsnet_parity"))
                    vals = [v for v in out["parity_check"]["cases"].values() if isinstance(v, float)]
                    out["parity_check"]["max_rel_l2"] = max(vals)
                    out["parity_check"]["within_tolerance"] = bool(max(vals) < out["parity_check"]["tolerance_rel_l2"])
            except Exception as e:  # noqa: BLE001
                out["secondary"] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline and args.workload == "drmnet_step":
            out["cpu_baseline"] = cpu_baseline(args)
        if json_fd is None:
            print(json.dumps(out), flush=True)
        else:
            os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
