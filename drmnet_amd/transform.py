"""Mirror reflectance map -> lat-long environment map, and the LDR tone map, on the HIP kernels.

``mirmap2envmap`` / ``hdr2ldr`` keep the reference's names and arguments (utils/transform.py:106-144, utils/tonemap.py:4-9);
the work is done by csrc/transform.hip: one gather kernel that evaluates the warp geometry per envmap texel and fetches with
grid_sample's bilinear / border / align_corners=False arithmetic, and one reduce-then-map kernel for the tone map.
"""
from __future__ import annotations

from typing import List, Tuple, Union

import numpy as np
import torch

from . import ops


def mirmap2envmap(mirmap: torch.Tensor, output_shape: Tuple[int, int], view: Union[torch.Tensor, List[float]] = [0, 0, 1],
                  top: Union[torch.Tensor, List[float]] = [0, 1, 0], envmap_zenith: Union[torch.Tensor, List[float]] = [0, 1, 0],
                  envmap_left_edge: Union[torch.Tensor, List[float]] = [0, 0, -1], reverse_azimuth: bool = True,
                  log_scale_interpolation: bool = False) -> torch.Tensor:
    """mirmap [B, C, H, W] -> envmap [B, C, OH, OW].  Only the geometry the reference itself supports and uses is built into the
    kernel (``assert view == [0, 0, 1]`` at utils/transform.py:117 and the default frame everywhere it is called)."""
    as_list = lambda v: [float(t) for t in (v.tolist() if isinstance(v, torch.Tensor) else v)]
    assert as_list(view) == [0, 0, 1], "now support [0,0,1] view direction"
    if as_list(top) != [0, 1, 0] or as_list(envmap_zenith) != [0, 1, 0] or as_list(envmap_left_edge) != [0, 0, -1] or not reverse_azimuth:
        raise NotImplementedError("only the default frame (top/zenith +y, left edge -z, reversed azimuth) is built into the kernel")
    return ops.mirmap2envmap(mirmap, output_shape, log_scale_interpolation=log_scale_interpolation)


def hdr2ldr(x, mask=None, alpha: float = 0.18, gamma: float = 2.2):
    """utils/tonemap.py:4-9.  Accepts what the reference is given (a [H, W, 3] numpy array -> numpy array) as well as a device
    tensor (-> device tensor); numpy input is staged through the current GPU."""
    if isinstance(x, torch.Tensor):
        return ops.hdr2ldr(x, mask, alpha, gamma)
    dev = torch.device("cuda", torch.cuda.current_device())
    m = None if mask is None else torch.as_tensor(np.asarray(mask)).to(dev)
    return ops.hdr2ldr(torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).to(dev), m, alpha, gamma).cpu().numpy()
