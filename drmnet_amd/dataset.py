"""``BaseDataset`` -- the HDR <-> network-space maps either side of both samplers, on the HIP map kernels.

Operator surface of ``dataset.basedataset.BaseDataset`` (reference dataset/basedataset.py:10-112): ``size``, ``transform(x,
dynamic_normalize=False, mask=None)``, ``rescale(x)`` and the ``transform_func`` string -- ``_``-separated map names read like
function composition (``f_g`` = f(g(x))), so ``transform`` runs them right to left and ``rescale`` runs the inverses left to
right.  Here the string is compiled once into two op lists for ``drm_map_chain`` (csrc/transform.hip): a whole chain is ONE pass
over the tensor; the only data-dependent piece, the per-image masked (log10 min, log10 max) of ``normalizedLogarithmic``
(:63-69), is one reduction launch in front of the pass that needs it, and -- like the reference (:69) -- is remembered on the
dataset object for the matching ``rescale``.  GPU tensors only: there is no torch fallback.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch

from . import ops

# map name -> (forward step, inverse steps); None = identity.  "@" marks the clamp_before_exp argument, "b" the lowerbound value.
_TABLE = {
    "log": (("log_p1", 0.0), (("exp_m1", "@"),)),
    "log10": (("log10", 0.0), (("exp10", "@"),)),
    "0p1tom1p1": (("unit_to_signed", 0.0), (("signed_to_unit", 0.0),)),
    "normalizedLogarithmic": (("norm_log", 0.0), (("denorm_log", 0.0), ("exp10", "@"))),
}


def _require_gpu(x: torch.Tensor) -> torch.Tensor:
    if not x.is_cuda:
        raise RuntimeError("BaseDataset maps run on the GPU (drmnet_amd has no CPU path); got a CPU tensor")
    return x.float().contiguous()


class BaseDataset(torch.utils.data.Dataset):
    def __init__(self, size: int, transform_func: str = "log", clamp_before_exp: float = 0.0):
        self.size = size
        self.transform_func_str = transform_func
        # basedataset.py:25: a literal False selects 10; any other value is used as given, and a falsy one means "no clamp" (:88,:94)
        self.clamp_before_exp = 10 if clamp_before_exp is False else clamp_before_exp
        self.Logarithmic_params = None  # [log10min, log10max], each [B, 1, 1, 1] (or [1, 1, 1] for a 3-D input), set by transform(dynamic_normalize=True)
        names = transform_func.split("_")
        # forward program: elementwise segments (one drm_map_chain pass each) cut at every resize, which keeps its position in the
        # chain as in the reference (:29-35: log(resize(x)) is not resize(log(x)), and normalizedLogarithmic's statistics are taken at
        # the resolution its input has at that point)
        self._forward: List = [[]]
        for name in reversed(names):
            if name.startswith("resize"):
                self._forward.append("bilinear" if name == "resize" else name[len("resize"):].replace("-", "_").lower())
                self._forward.append([])
            else:
                self._forward[-1].extend(self._compile(name, inverse=False))
        self._inverse: List[Tuple[str, float]] = []
        for name in names:
            self._inverse.extend(self._compile(name, inverse=True))

    # ------------------------------------------------------------------ compilation of the name string
    def _compile(self, name: str, inverse: bool) -> List[Tuple[str, float]]:
        if not name or "_" in name:
            raise NotImplementedError(name)
        cap = float(self.clamp_before_exp) if self.clamp_before_exp else math.inf
        if name.startswith("resize"):
            return []  # shape change, not an elementwise map: handled in transform(); rescale leaves the size alone (:86-87)
        if name.startswith("lowerbound"):
            return [] if inverse else [("lowerbound", float(name[len("lowerbound"):]))]
        if name not in _TABLE:
            raise NotImplementedError(name)
        fwd, inv = _TABLE[name]
        return [(n, cap if a == "@" else a) for n, a in inv] if inverse else [fwd]

    # kept for callers that ask for a single named map (same (sic) spelling as the reference, :42 / :82)
    def get_tranfrom_func(self, func_name: str):
        steps = self._compile(func_name, inverse=False)
        return lambda x, **kw: self._run(x, steps, **kw)

    def get_rescale_func(self, func_name: str):
        steps = self._compile(func_name, inverse=True)
        return lambda x, **kw: self._run(x, steps)

    # ------------------------------------------------------------------ execution
    def _run(self, x: torch.Tensor, steps, dynamic_normalize: bool = False, mask: Optional[torch.Tensor] = None):
        if not x.is_cuda:
            raise RuntimeError("BaseDataset maps run on the GPU (drmnet_amd has no CPU path); got a CPU tensor")
        x = x.float().contiguous()
        names = [n for n, _ in steps]
        lo = hi = None
        if "norm_log" in names and dynamic_normalize:
            if mask is None:
                raise AssertionError("dynamic_normalize needs a mask")
            cut = names.index("norm_log")
            if cut:  # the statistics are taken on the output of the maps in front (e.g. lowerbound1e-6)
                x = ops.map_chain(x, steps[:cut])
                steps = steps[cut:]
            lo, hi = ops.masked_log_range(x, mask)
            keep = (-1,) + (1,) * 3 if x.ndim >= 4 else (1, 1, 1)
            self.Logarithmic_params = [lo.view(keep), hi.view(keep)]
        if any(n in ("norm_log", "denorm_log") for n, _ in steps):
            if self.Logarithmic_params is None:
                raise RuntimeError("normalizedLogarithmic: call transform(..., dynamic_normalize=True, mask=...) first")
            lo, hi = (p.to(x.device) for p in self.Logarithmic_params)
            if lo.ndim != x.ndim:
                raise AssertionError(f"{x.ndim}, {lo.ndim}, {hi.ndim}")
        return ops.map_chain(x, steps, lo=lo, hi=hi) if steps else x

    def transform(self, x: torch.Tensor, dynamic_normalize: bool = False, mask: torch.Tensor = None):
        assert x.size(-1) >= self.size
        y = x
        for seg in self._forward:
            if isinstance(seg, list):
                if seg:
                    y = self._run(y, seg, dynamic_normalize=dynamic_normalize, mask=mask)
            elif y.shape[-1] != self.size or y.shape[-2] != self.size:
                # (the shipped path never gets here: refmaps are produced at `size`) torchvision's resize(..., antialias=True), :44-50
                y = ops.resize(_require_gpu(y), (self.size, self.size), seg)
        if not y.is_cuda:
            raise RuntimeError("BaseDataset maps run on the GPU (drmnet_amd has no CPU path); got a CPU tensor")
        return y.float().contiguous()

    def rescale(self, x: torch.Tensor):
        return self._run(x, self._inverse)
