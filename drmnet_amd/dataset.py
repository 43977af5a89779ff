"""HDR <-> network-space maps used either side of the samplers.

Mirror of ``dataset.basedataset.BaseDataset.transform / rescale`` (reference dataset/basedataset.py:29-112):
a ``transform_func`` string of ``_``-separated names applied right-to-left (f_g = f(g(x))) and inverted
left-to-right.  Elementwise torch ops on whatever device the tensor lives on -- host glue, not the hot path.
"""
from __future__ import annotations

import torch


class BaseDataset(torch.utils.data.Dataset):
    def __init__(self, size: int, transform_func: str = "log", clamp_before_exp: float = 0.0):
        self.size = size
        self.transform_func_str = transform_func
        self.clamp_before_exp = 10 if isinstance(clamp_before_exp, bool) and not clamp_before_exp else clamp_before_exp
        names = transform_func.split("_")
        self.transform_funcs = [self.get_tranfrom_func(n) for n in names[::-1]]
        self.rescale_funcs = [self.get_rescale_func(n) for n in names]

    def transform(self, x: torch.Tensor, dynamic_normalize: bool = False, mask: torch.Tensor = None):
        assert x.size(-1) >= self.size
        for func in self.transform_funcs:
            x = func(x, dynamic_normalize=dynamic_normalize, mask=mask)
        return x

    def rescale(self, x: torch.Tensor):
        for func in self.rescale_funcs:
            x = func(x)
        return x

    def get_tranfrom_func(self, func_name: str):  # (sic) reference spelling kept
        assert "_" not in func_name
        if func_name.startswith("resize"):
            def resize(x, **kwargs):
                if x.shape[-1] == self.size and x.shape[-2] == self.size:
                    return x
                mode = "bilinear" if len(func_name) == 6 else func_name[6:].replace("-", "_").lower()
                lead = x.shape[:-2]
                y = torch.nn.functional.interpolate(x.reshape(-1, 1, *x.shape[-2:]), size=(self.size, self.size), mode=mode,
                                                    antialias=mode in ("bilinear", "bicubic"),
                                                    align_corners=False if mode in ("bilinear", "bicubic") else None)
                return y.reshape(*lead, self.size, self.size)
            return resize
        elif func_name == "log":
            return lambda x, **kwargs: torch.log10(x + 1e-1) + 1
        elif func_name == "log10":
            return lambda x, **kwargs: torch.log10(x)
        elif func_name.startswith("lowerbound"):
            bottom = float(func_name[10:])
            return lambda x, **kwargs: torch.clip(x, bottom)
        elif func_name == "0p1tom1p1":
            return lambda x, **kwargs: x * 2 - 1
        elif func_name == "normalizedLogarithmic":
            def func(x: torch.Tensor, mask: torch.Tensor, dynamic_normalize: bool, **kwargs):
                if dynamic_normalize:
                    assert mask is not None
                    linearmax = (x * mask).amax(dim=(-1, -2, -3), keepdim=True)
                    log10max = torch.log10(linearmax)
                    log10min = torch.log10((x * mask + (1 - mask.float()) * linearmax).amin(dim=(-1, -2, -3), keepdim=True))
                    self.Logarithmic_params = [log10min, log10max]  # state kept on the dataset object (basedataset.py:69)
                log10min, log10max = self.Logarithmic_params
                assert x.ndim == log10min.ndim == log10max.ndim, f"{x.ndim}, {log10min.ndim}, {log10max.ndim}"
                log10min, log10max = log10min.to(x.device), log10max.to(x.device)
                return (torch.log10(x) - log10min) / (log10max - log10min)
            return func
        raise NotImplementedError(func_name)

    def get_rescale_func(self, func_name: str):
        do_nothing = lambda x, **kwargs: x
        assert "_" not in func_name
        if func_name.startswith("resize"):
            return do_nothing
        elif func_name == "log":
            if self.clamp_before_exp:
                return lambda x, **kwargs: torch.pow(10, torch.clamp(x - 1, max=self.clamp_before_exp)) - 1e-1
            return lambda x, **kwargs: torch.pow(10, x - 1) - 1e-1
        elif func_name == "log10":
            if self.clamp_before_exp:
                return lambda x, **kwargs: torch.pow(10, torch.clamp(x, max=self.clamp_before_exp))
            return lambda x, **kwargs: torch.pow(10, x)
        elif func_name.startswith("lowerbound"):
            return do_nothing
        elif func_name == "0p1tom1p1":
            return lambda x, **kwargs: (x + 1) / 2
        elif func_name == "normalizedLogarithmic":
            log10 = self.get_rescale_func("log10")

            def func(x: torch.Tensor, **kwargs):
                log10min, log10max = self.Logarithmic_params
                log10min, log10max = log10min.to(x.device), log10max.to(x.device)
                assert x.ndim == log10min.ndim == log10max.ndim
                return log10(x * (log10max - log10min) + log10min, **kwargs)
            return func
        raise NotImplementedError(func_name)
