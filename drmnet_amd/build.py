"""Builds drmnet_amd/csrc/libdrmnet_hip.so for gfx950 with hipcc (in-tree, so it travels with gpurun).

    python -m drmnet_amd.build [--force] [--jobs N]

hipcc cross-compiles without a GPU.  Objects are cached by source mtime under csrc/_obj/.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(CSRC, "libdrmnet_hip.so")
SOURCES = ["conv.hip", "conv_split.hip", "conv_split2.hip", "gn.hip", "attn.hip", "attn_flash.hip", "misc.hip", "refmap.hip", "transform.hip", "engine.hip", "samplers.hip", "abi.hip", "profiler.hip"]
HEADERS = ["common.h", "gn_fold.h", "engine.h", "samplers.h", "profiler.h", os.path.join("..", "..", "include", "drmnet_hip.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-ffp-contract=off"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(a: str, deps) -> bool:
    if not os.path.exists(a):
        return False
    t = os.path.getmtime(a)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, jobs: int = 4, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    cc = hipcc()
    todo = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(op)
        if force or not _newer(op, [sp] + hdrs):
            todo.append((sp, op))

    def compile_one(job):
        sp, op = job
        cmd = [cc] + FLAGS + ["-c", sp, "-o", op]
        if verbose:
            print("  hipcc", os.path.basename(sp), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {sp}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr.strip())

    if todo:
        with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
            list(ex.map(compile_one, todo))
    if todo or force or not _newer(LIB, objs):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print("  linked", LIB)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=4)
    a = ap.parse_args()
    try:
        build(a.force, a.jobs)
    except RuntimeError as e:
        print(e, file=sys.stderr)
        sys.exit(1)
