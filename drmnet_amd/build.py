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
SOURCES = ["conv.hip", "conv_split.hip", "conv_split2.hip", "gn.hip", "attn.hip", "attn_flash.hip", "misc.hip", "stemhead.hip", "refmap.hip", "transform.hip", "engine.hip", "samplers.hip", "abi.hip", "profiler.hip"]
# conv_split2.hip is also compiled once per (TAPS, TERMS) pair of its kernel template (-DDRM_S2_UNIT=10 * TAPS + TERMS): nine objects built in
# parallel instead of one 4.5-minute translation unit; the plain compile above holds the host-side rest
S2_UNITS = [92, 93, 13, 90, 10, 91, 11, 94, 14]  # (slowest first)


def units():
    """(source, object name, extra flags) of every object of the library"""
    out = [("conv_split2.hip", f"conv_split2_u{u}.o", [f"-DDRM_S2_UNIT={u}"]) for u in S2_UNITS]
    out += [(src, src.replace(".hip", ".o"), []) for src in SOURCES]
    return out
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
    for src, obj, extra in units():
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, obj)
        objs.append(op)
        if force or not _newer(op, [sp] + hdrs):
            todo.append((sp, op, extra))

    def compile_one(job):
        sp, op, extra = job
        cmd = [cc] + FLAGS + extra + ["-c", sp, "-o", op]
        if verbose:
            print("  hipcc", os.path.basename(sp), " ".join(extra), flush=True)
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


def build_variant(name: str, flags, srcs=None, s2_units=None, jobs: int = 8, out_dir: str | None = None) -> str:
    """An experimental library for a same-box A/B (tools/ab_bench.sh): the named sources (default: conv_split2.hip, every unit; `s2_units`
    restricts the recompiled kernel units, e.g. [92]) are recompiled with extra flags into /tmp, everything else comes from the product's
    objects (run build() first).  srcs = ["all"] recompiles every source (flags that change ConvArgs, e.g. -DDRM_S2_STAMP)."""
    build(False, jobs, verbose=False)
    srcs = srcs or ["conv_split2"]
    tmp = os.path.join("/tmp", f"drm_variant_{name}")
    os.makedirs(tmp, exist_ok=True)
    out_dir = out_dir or os.path.join(CSRC, "_ab")
    os.makedirs(out_dir, exist_ok=True)
    cc = hipcc()
    todo, objs = [], []
    for src, obj, extra in units():
        stem = src.replace(".hip", "")
        sel = "all" in srcs or stem in srcs
        if sel and stem == "conv_split2" and extra and s2_units is not None and int(extra[0].split("=")[1]) not in s2_units:
            sel = False
        if sel:
            op = os.path.join(tmp, obj)
            todo.append((os.path.join(CSRC, src), op, extra + list(flags)))
            objs.append(op)
        else:
            objs.append(os.path.join(OBJ, obj))

    def compile_one(job):
        sp, op, extra = job
        r = subprocess.run([cc] + FLAGS + extra + ["-c", sp, "-o", op], capture_output=True, text=True)
        with open(op + ".log", "w") as f:
            f.write(r.stderr)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {sp} {extra}:\n{r.stderr}")

    with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        list(ex.map(compile_one, todo))
    lib = os.path.join(out_dir, f"libdrmnet_hip_{name}.so")
    r = subprocess.run([cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    print("built", lib)
    return lib


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--variant", help="build csrc/_ab/libdrmnet_hip_<name>.so from --srcs recompiled with --flags (see build_variant)")
    ap.add_argument("--flags", default="")
    ap.add_argument("--srcs", default="conv_split2", help="comma-separated source stems, or 'all'")
    ap.add_argument("--units", default="", help="comma-separated conv_split2 kernel units to recompile (default: all of them)")
    a = ap.parse_args()
    try:
        if a.variant:
            build_variant(a.variant, a.flags.split(), a.srcs.split(","), [int(u) for u in a.units.split(",")] if a.units else None, max(a.jobs, 8))
            sys.exit(0)
        build(a.force, a.jobs)
    except RuntimeError as e:
        print(e, file=sys.stderr)
        sys.exit(1)
