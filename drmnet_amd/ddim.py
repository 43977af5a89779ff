"""``DDIMSampler`` -- same surface as ldm/models/diffusion/ddim.py:16-259, loop executed by drm_ddim_sample.

Schedule construction restates make_ddim_timesteps / make_ddim_sampling_parameters
(ldm/modules/diffusionmodules/util.py:46-74) including the reference's mixed fp32/fp64 rounding sequence
(``ndarray / Tensor`` dispatches to ``Tensor.__rtruediv__`` = reciprocal()*other), checked bit-for-bit against
tests/golden/ddim_schedule_eta*.npz.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    if ddim_discr_method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        ddim_timesteps = np.asarray(list(range(0, num_ddpm_timesteps, c)))
    elif ddim_discr_method == "quad":
        ddim_timesteps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * 0.8), num_ddim_timesteps)) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    steps_out = ddim_timesteps + 1
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps_out}")
    return steps_out


def make_ddim_sampling_parameters(alphacums: torch.Tensor, ddim_timesteps, eta, verbose=True):
    ac = alphacums.detach().cpu().float()
    idx = torch.from_numpy(np.asarray(ddim_timesteps))
    alphas = ac[idx]  # fp32
    alphas_prev = torch.tensor([float(ac[0])] + [float(v) for v in ac[idx[:-1]]], dtype=torch.float64)
    recip = (1 - alphas).reciprocal().double()  # the one fp32-rounded factor of util.py:69
    sigmas = eta * torch.sqrt(recip * (1 - alphas_prev) * (1 - alphas.double() / alphas_prev))
    return sigmas, alphas, alphas_prev


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self._ws = _lib.Workspace()

    def register_buffer(self, name, attr):
        if isinstance(attr, torch.Tensor):
            attr = attr.to(self.model.device)  # the reference hard-codes "cuda" here (ddim.py:23-27)
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0.0, verbose=True):
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps, verbose=verbose)
        alphas_cumprod = self.model.alphas_cumprod
        assert alphas_cumprod.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        sig, a, a_prev = make_ddim_sampling_parameters(alphas_cumprod, self.ddim_timesteps, ddim_eta, verbose=verbose)
        self.ddim_sigmas, self.ddim_alphas, self.ddim_alphas_prev = sig, a, a_prev
        self.ddim_sqrt_one_minus_alphas = torch.sqrt(1.0 - a)
        # the five fp32 scalars p_sample_ddim derives per index through torch.full + fp32 tensor ops (ddim.py:243-258)
        S = len(self.ddim_timesteps)
        coef = np.zeros((S, 5), dtype=np.float32)
        for i in range(S):
            a_t = torch.full((1,), float(a[i]))
            ap = torch.full((1,), float(a_prev[i]))
            sg = torch.full((1,), float(sig[i]))
            s1m = torch.full((1,), float(self.ddim_sqrt_one_minus_alphas[i]))
            coef[i] = [a_t.sqrt().item(), s1m.item(), ap.sqrt().item(), (1.0 - ap - sg**2).sqrt().item(), sg.item()]
        self.ddim_coef = coef

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None, quantize_x0=False,
               eta=0.0, mask=None, x0=None, temperature=1.0, noise_dropout=0.0, score_corrector=None, corrector_kwargs=None, verbose=True,
               x_T=None, log_every_t=100, unconditional_guidance_scale=1.0, unconditional_conditioning=None, noise=None, seed=None,
               num_steps=None, **kwargs):
        """ddim.py:64-126.  Extra keyword-only knobs: ``noise`` [steps,N,C,H,W] injects the per-step draws (parity mode),
        ``seed`` keys the Philox stream otherwise, ``num_steps`` truncates the chain (benchmarks)."""
        if conditioning is not None and not isinstance(conditioning, dict):
            if conditioning.shape[0] != batch_size:
                print(f"Warning: Got {conditioning.shape[0]} conditionings but batch-size is {batch_size}")
        if any(v is not None for v in (callback, img_callback, score_corrector)) or quantize_x0:
            raise NotImplementedError("callbacks / score corrector / quantize: host hooks of the reference with no device form here")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        size = (batch_size, *shape)
        return self.ddim_sampling(conditioning, size, x_T=x_T, log_every_t=log_every_t, verbose=verbose, noise=noise, seed=seed, num_steps=num_steps,
                                  mask=mask, x0=x0, temperature=temperature, mask_noise=kwargs.get("mask_noise"), noise_dropout=noise_dropout,
                                  dropout_keep=kwargs.get("dropout_keep"), unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning)

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, log_every_t=100, verbose=True, noise=None, seed=None, num_steps=None, mask=None, x0=None,
                      temperature=1.0, mask_noise=None, noise_dropout=0.0, callback=None, img_callback=None, quantize_denoised=False, score_corrector=None,
                      corrector_kwargs=None, unconditional_guidance_scale=1.0, unconditional_conditioning=None, ddim_use_original_steps=False,
                      timesteps=None, dropout_keep=None):
        """ddim.py:128-204 -> (final x, intermediates).  ``intermediates`` is the reference's record (ddim.py:171-204): "x_inter" and "pred_x0" start
        with x_T and receive (img, pred_x0) after the step of ``index`` whenever index % log_every_t == 0 or at the first step -- written by the
        update kernel of the steps the device table marks (csrc/samplers.hip), so the chain stays one device loop.
        ``mask`` / ``x0`` (ddim.py:175-178): before every step img = q_sample(x0, t) * mask + (1 - mask) * img, on the device (mask_blend_kernel;
        ``mask_noise`` [steps,N,C,H,W] injects q_sample's draws, else Philox).  ``temperature`` (ddim.py:255) scales the step noise: the sigma column
        of the coefficient table.  ``timesteps`` (ddim.py:156-158): the chain over the first ``subset_end`` entries of the DDIM schedule -- fewer rows of the host coefficient
        table (``ddim_use_original_steps`` raises here as it does, by AttributeError, in the reference).  ``unconditional_conditioning`` +
        ``unconditional_guidance_scale`` (ddim.py:225-232): a second forward per step on the unconditional conditioning, e = e_u + s (e_c - e_u).
        ``noise_dropout`` (ddim.py:256-257): F.dropout on the step noise (``dropout_keep`` [steps,N,C,H,W] injects the 0 / 1 masks, else Philox).
        Score correctors, callbacks and quantisation are host hooks with no device form: rejected."""
        from . import ops

        if any(v is not None for v in (callback, img_callback, score_corrector)) or quantize_denoised:
            raise NotImplementedError("callbacks / score corrector / quantize: host hooks of the reference with no device form here")
        if not 0.0 <= noise_dropout < 1.0:
            raise ValueError("noise_dropout: 0 <= p < 1")
        # classifier-free guidance (ddim.py:225-232): only with an unconditional conditioning AND a scale != 1, as the reference decides
        guided = unconditional_conditioning is not None and unconditional_guidance_scale != 1.0
        if (mask is None) != (x0 is None):
            raise ValueError("mask and x0 go together (ddim.py:176)")

        dev = self.model.betas.device
        if seed is None:
            seed = int(torch.randint(0, 2**62, (1,)).item())
        img = ops.randn(shape, seed, 0, dev) if x_T is None else _lib.require_gpu_tensor(x_T, "x_T").clone()
        x_start = img.clone()
        c = cond[0] if isinstance(cond, (list, tuple)) else cond
        c = _lib.require_gpu_tensor(c, "conditioning")
        noise = None if noise is None else _lib.require_gpu_tensor(noise, "noise")
        unet = self.model.model.diffusion_model
        h = unet.engine_handle()
        if getattr(self.model, "_auto_chain", None) is not None:  # auto mode: the model's chain probe may move the network to f16x3 for these weights
            self.model._auto_chain_probe(c)  # (on rows of the caller's conditioning, once per weight signature)
            h = unet.engine_handle()
        L = _lib.lib()
        n, _, hh, ww = shape
        ws = self._ws.get(int(L.drm_sampler_workspace_bytes(h, n, hh, ww)), dev)
        ts = np.ascontiguousarray(np.asarray(self.ddim_timesteps, dtype=np.int64))
        coef = np.ascontiguousarray(self.ddim_coef)
        if ddim_use_original_steps:
            # (the reference cannot run this either: p_sample_ddim reads self.model.ddim_sigmas_for_original_num_steps, ddim.py:242, which
            #  make_schedule registers on the SAMPLER, ddim.py:62 -> AttributeError on the first step)
            raise NotImplementedError("ddim_use_original_steps: unreachable in the reference too (ddim.py:242 reads a buffer the model does not have)")
        if timesteps is not None:  # ddim.py:156-158: the first subset_end entries of the DDIM schedule
            n_all = len(ts)
            subset_end = int(min(timesteps / n_all, 1) * n_all) - 1
            if subset_end < 1:
                raise ValueError(f"timesteps = {timesteps} leaves no DDIM step (subset_end = {subset_end})")
            ts, coef = np.ascontiguousarray(ts[:subset_end]), np.ascontiguousarray(coef[:subset_end])
        if temperature != 1.0:  # noise = sigma_t * randn * temperature (ddim.py:255); dir_xt keeps the un-scaled sigma (its own column)
            coef = coef.copy()
            coef[:, 4] = (torch.from_numpy(coef[:, 4]) * float(temperature)).numpy()
        S = len(ts)
        steps = S if not num_steps else min(int(num_steps), S)
        blend = keep = None
        if mask is not None:
            # q_sample(x0, ts) at the step's own t (ddpm.py:1052-1058: sqrt_alphas_cumprod[t] x0 + sqrt_one_minus_alphas_cumprod[t] noise)
            sa = self.model.sqrt_alphas_cumprod.detach().cpu().float().numpy()
            s1 = self.model.sqrt_one_minus_alphas_cumprod.detach().cpu().float().numpy()
            q = np.array([[sa[int(ts[S - 1 - j])], s1[int(ts[S - 1 - j])]] for j in range(steps)], dtype=np.float32)
            blend, keep = _lib.make_mask_blend(mask, x0, q, mask_noise, 0, tuple(img.shape))
        log_every_t = int(log_every_t) if log_every_t else 0
        slots = sum(1 for j in range(steps) if log_every_t > 0 and ((S - 1 - j) % log_every_t == 0 or j == 0))
        inter = {"x_inter": [x_start], "pred_x0": [x_start]}
        with torch.cuda.device(dev):
            if blend is not None or guided or noise_dropout > 0.0:
                uc = None
                if guided:
                    uc = unconditional_conditioning[0] if isinstance(unconditional_conditioning, (list, tuple)) else unconditional_conditioning
                opt, keep_o = _lib.make_sampler_options(tuple(img.shape), steps, blend=None if blend is None else (blend, keep), uncond=uc,
                                                        guidance_scale=unconditional_guidance_scale, noise_dropout=noise_dropout, dropout_keep=dropout_keep)
                log_x = torch.empty((max(slots, 1),) + tuple(img.shape), dtype=torch.float32, device=dev)
                log_p = torch.empty_like(log_x)
                n_logged = C.c_int32(0)
                _lib.check(L.drm_ddim_sample_ex(h, img.data_ptr(), c.data_ptr(), ts.ctypes.data_as(C.POINTER(C.c_int64)),
                                                coef.ctypes.data_as(C.POINTER(C.c_float)), S, int(num_steps or 0), _lib.ptr(noise), seed, C.byref(opt),
                                                log_every_t if slots else 0, log_x.data_ptr(), log_p.data_ptr(), slots, C.byref(n_logged), n, hh, ww,
                                                ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev)))
                torch.cuda.current_stream(dev).synchronize()  # (the options' tensors stay alive until the chain has run)
                del keep_o
                inter["x_inter"] += [log_x[k] for k in range(n_logged.value)]
                inter["pred_x0"] += [log_p[k] for k in range(n_logged.value)]
            elif slots == 0:
                _lib.check(L.drm_ddim_sample(h, img.data_ptr(), c.data_ptr(), ts.ctypes.data_as(C.POINTER(C.c_int64)),
                                             coef.ctypes.data_as(C.POINTER(C.c_float)), S, int(num_steps or 0), _lib.ptr(noise), seed, n, hh, ww,
                                             ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev)))
            else:
                log_x = torch.empty((slots,) + tuple(img.shape), dtype=torch.float32, device=dev)
                log_p = torch.empty_like(log_x)
                n_logged = C.c_int32(0)
                _lib.check(L.drm_ddim_sample_logged(h, img.data_ptr(), c.data_ptr(), ts.ctypes.data_as(C.POINTER(C.c_int64)),
                                                    coef.ctypes.data_as(C.POINTER(C.c_float)), S, int(num_steps or 0), _lib.ptr(noise), seed, log_every_t,
                                                    log_x.data_ptr(), log_p.data_ptr(), slots, C.byref(n_logged), n, hh, ww, ws.data_ptr(), ws.numel(),
                                                    _lib.stream_ptr(dev)))
                inter["x_inter"] += [log_x[k] for k in range(n_logged.value)]
                inter["pred_x0"] += [log_p[k] for k in range(n_logged.value)]
        return img, inter
