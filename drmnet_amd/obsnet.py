"""``ObsNetDiffusion`` -- host-side operator surface of the reference's conditional DDPM, sampling on the HIP engine.

Mirrors (inference half only; training, VQ/KL first stages, patch fold/unfold are out of scope):
  DDPM.__init__/register_schedule/ema_scope/init_from_ckpt        ldm/models/diffusion/ddpm.py:59-231
  DDPM.predict_start_from_noise / q_posterior / q_sample          ddpm.py:233-246, :306-312
  LatentDiffusion.__init__/apply_model/p_mean_variance/p_sample/sample/decode_first_stage
                                                                  ddpm.py:439-488, :916-1023, :1079-1167, :1315-1350, :731-789
  ObsNetDiffusion.__init__/p_sample_loop/sample_log/get_cond_for_predict   models/obsnet.py:35-137, :500-583, :656-704
``state_dict()`` keys equal the reference's (schedule buffers, ``model.diffusion_model.*``, ``model_ema.*``).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Union

import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops
from .config import instantiate_from_config
from .wrappers import DiffusionWrapper, IdentityFirstStage, LitEma, ema_weights, load_checkpoint


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2):
    """The beta tables of ldm/modules/diffusionmodules/util.py:21-43 that are spaced between two end points, in fp64: "linear" (every shipped
    config) is linear in sqrt(beta), "sqrt_linear" in beta, "sqrt" is the square root of a linear ramp.  (The cosine table is training-side.)"""
    def ramp(lo, hi):
        return torch.linspace(lo, hi, n_timestep, dtype=torch.float64)

    tables = {"linear": lambda: ramp(linear_start**0.5, linear_end**0.5) ** 2, "sqrt_linear": lambda: ramp(linear_start, linear_end),
              "sqrt": lambda: ramp(linear_start, linear_end) ** 0.5}
    if schedule not in tables:
        raise ValueError(f"schedule '{schedule}' unknown.")
    return tables[schedule]().numpy()


def extract_into_tensor(a, t, x_shape):
    """Per-sample table entries a[t] shaped to broadcast against x (util.py:96-99)."""
    return a[t].reshape((t.shape[0],) + (1,) * (len(x_shape) - 1))


# Constructor / YAML keys of the reference that steer only training, logging or the out-of-scope renderers (ldm/models/diffusion/ddpm.py:60-135,
# :442-512; models/obsnet.py:38-137).  They are ACCEPTED, so configs/**.yaml and reference-style constructor calls load unchanged, and never
# read: nothing on the sampling path depends on them.  Any other unknown key is an error.
_TRAINING_ONLY = frozenset({
    "loss_type", "monitor", "original_elbo_weight", "l_simple_weight", "scheduler_config", "use_positional_encodings", "logvar_init", "cosine_s",
    "first_stage_key", "cond_stage_trainable", "masked_loss", "obj_img_key", "cache_data", "refmap_cache_root", "objimg_cache_root", "envmap_dir",
    "img_renderer_config",
})


def _drop_training_only(kwargs: dict, who: str) -> None:
    unknown = sorted(set(kwargs) - _TRAINING_ONLY)
    if unknown:
        raise TypeError(f"{who}: unexpected parameter(s) {unknown}")


class DDPM(nn.Module):
    """The noise schedule + the wrapped eps-network (ddpm.py:59-231), inference half."""

    def __init__(self, unet_config, timesteps=1000, beta_schedule="linear", *, ckpt_path=None, ignore_keys=(), load_only_unet=False, use_ema=True,
                 image_size=256, channels=3, log_every_t=100, clip_denoised=True, linear_start=1e-4, linear_end=2e-2, given_betas=None,
                 v_posterior=0.0, conditioning_key=None, parameterization="eps", learn_logvar=False, **training_only):
        super().__init__()
        _drop_training_only(training_only, type(self).__name__)
        if parameterization != "eps":
            raise NotImplementedError('only the "eps" parameterization is on the shipped path')
        if learn_logvar:
            raise NotImplementedError("learn_logvar is training-only")
        self.parameterization = parameterization
        self.image_size, self.channels = image_size, channels
        self.clip_denoised, self.log_every_t, self.v_posterior = clip_denoised, log_every_t, v_posterior
        self.cond_stage_model = None
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self._weight_set = "live"
        self.use_ema = use_ema
        if use_ema:
            self.model_ema = LitEma(self.model)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=list(ignore_keys), only_model=load_only_unet)
        self.register_schedule(given_betas=given_betas, beta_schedule=beta_schedule, timesteps=timesteps, linear_start=linear_start, linear_end=linear_end)

    @property
    def device(self):
        return self.betas.device

    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4, linear_end=2e-2):
        """ldm/models/diffusion/ddpm.py:137-187: every table of the forward / posterior process is computed in fp64 numpy from the
        betas and registered as an fp32 buffer -- same names, same order, so ``state_dict()`` matches the reference's checkpoints
        (tests/golden/ddpm_schedule.npz pins the values bit for bit).  abar_t = prod(1 - beta); posterior q(x_{t-1} | x_t, x_0)
        has variance beta_t (1 - abar_{t-1}) / (1 - abar_t) (mixed with beta_t by v_posterior) and mean coefficients
        beta_t sqrt(abar_{t-1}) / (1 - abar_t) on x_0 and (1 - abar_{t-1}) sqrt(alpha_t) / (1 - abar_t) on x_t."""
        beta = given_betas if given_betas is not None else make_beta_schedule(beta_schedule, timesteps, linear_start, linear_end)
        (n_steps,) = beta.shape
        self.num_timesteps = int(n_steps)
        self.linear_start, self.linear_end = linear_start, linear_end
        alpha = 1.0 - beta
        abar = np.cumprod(alpha, axis=0)
        abar_prev = np.append(1.0, abar[:-1])
        post_var = (1 - self.v_posterior) * beta * (1.0 - abar_prev) / (1.0 - abar) + self.v_posterior * beta
        tables = (  # (buffer name, fp64 table) in the reference's registration order
            ("betas", beta),
            ("alphas_cumprod", abar),
            ("alphas_cumprod_prev", abar_prev),
            ("sqrt_alphas_cumprod", np.sqrt(abar)),
            ("sqrt_one_minus_alphas_cumprod", np.sqrt(1.0 - abar)),
            ("log_one_minus_alphas_cumprod", np.log(1.0 - abar)),
            ("sqrt_recip_alphas_cumprod", np.sqrt(1.0 / abar)),
            ("sqrt_recipm1_alphas_cumprod", np.sqrt(1.0 / abar - 1)),
            ("posterior_variance", post_var),
            ("posterior_log_variance_clipped", np.log(np.maximum(post_var, 1e-20))),  # the variance is 0 at t = 0
            ("posterior_mean_coef1", beta * np.sqrt(abar_prev) / (1.0 - abar)),
            ("posterior_mean_coef2", (1.0 - abar_prev) * np.sqrt(alpha) / (1.0 - abar)),
        )
        for name, table in tables:
            self.register_buffer(name, torch.tensor(table, dtype=torch.float32))

    def ema_scope(self, context=None):
        """ldm/models/diffusion/ddpm.py:189-202 -- ``with model.ema_scope(): ...`` samples with the EMA weights (wrappers.ema_weights)."""
        return ema_weights(self, [(self.model, self.model_ema)] if self.use_ema else [], context)

    def init_from_ckpt(self, path, ignore_keys=list(), only_model=False, verbose=True):
        """ddpm.py:204-231 / models/obsnet.py:139-160 (``only_model`` loads into the wrapped U-Net alone)."""
        load_checkpoint(self, path, ignore_keys, into=self.model if only_model else None, verbose=verbose)

    def predict_start_from_noise(self, x_t, t, noise):
        return (extract_into_tensor(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t
                - extract_into_tensor(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * noise)

    def q_posterior(self, x_start, x_t, t):
        mean = (extract_into_tensor(self.posterior_mean_coef1, t, x_t.shape) * x_start
                + extract_into_tensor(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        return mean, extract_into_tensor(self.posterior_variance, t, x_t.shape), extract_into_tensor(self.posterior_log_variance_clipped, t, x_t.shape)

    def q_sample(self, x_start, t, noise=None):
        noise = torch.randn_like(x_start) if noise is None else noise
        return (extract_into_tensor(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
                + extract_into_tensor(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)


class LatentDiffusion(DDPM):
    """DDPM + conditioning by concatenation behind an identity first stage (ddpm.py:439-512), inference half."""

    def __init__(self, first_stage_config, cond_stage_config, num_timesteps_cond=None, cond_stage_key="image", *, concat_mode=True, cond_stage_forward=None,
                 conditioning_key=None, scale_factor=1.0, scale_by_std=False, ckpt_path=None, ignore_keys=(), **ddpm_kwargs):
        if scale_by_std:
            raise NotImplementedError("scale_by_std is training-only")
        if (num_timesteps_cond or 1) != 1:
            raise NotImplementedError("num_timesteps_cond > 1 (shortened conditioning schedule) is not on the shipped path")
        if cond_stage_config not in ("__is_first_stage__", "__is_unconditional__"):
            raise NotImplementedError("a separate cond_stage_config is not on the shipped path")
        if conditioning_key is None:
            conditioning_key = "concat" if concat_mode else "crossattn"
        if cond_stage_config == "__is_unconditional__":
            conditioning_key = None
        super().__init__(conditioning_key=conditioning_key, **ddpm_kwargs)  # (the checkpoint is read below, once the whole module exists)
        self.num_timesteps_cond = 1
        self.concat_mode, self.cond_stage_key, self.cond_stage_forward, self.scale_factor = concat_mode, cond_stage_key, cond_stage_forward, scale_factor
        self.first_stage_model = instantiate_from_config(first_stage_config).eval()
        if not isinstance(self.first_stage_model, IdentityFirstStage):
            raise NotImplementedError("only IdentityFirstStage is used by the shipped configs")
        self.cond_stage_model = self.first_stage_model if cond_stage_config == "__is_first_stage__" else None
        self.clip_denoised = False  # (ddpm.py:492: LatentDiffusion overrides the DDPM default)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, list(ignore_keys))
        self._ws = _lib.Workspace()

    def set_precision(self, precision: str, probe=None):
        """Conv arithmetic of the U-Net: "fp32" (exact fp32 MFMA), "f16x3" (split fp16, fp32-accurate, ~2.5x faster), "f16mx" (f16x3 with
        fp8 cross terms on the 3x3 convs: ~4e-5 per forward, ~3x faster) or "f16" (reduced precision)."""
        self.model.diffusion_model.set_precision(precision)
        # "auto": the per-network probe (unet.py) + a chain probe before f16mx is kept (_auto_chain_probe)
        # ``probe`` [n,3,H,W]: conditioning refmaps of the CALLER for the chain probe; without it the first batch a sampler sees is handed to it
        self._auto_chain = {"tolerance": 5e-5, "steps": 8, "done": {}, "busy": False, "report": None,
                            "probe": None if probe is None else _lib.require_gpu_tensor(probe, "probe").detach()} if precision == "auto" else None
        return self

    @property
    def auto_chain_report(self):
        ac = getattr(self, "_auto_chain", None)
        return None if ac is None else ac["report"]

    def calibrate_precision(self, probe=None):
        """Auto mode: runs the network's probe and the chain probe now (weights on a GPU); returns the chain report.  ``probe``: conditioning refmaps
        of the caller to run the chain on (re-measured for these rows)."""
        ac = getattr(self, "_auto_chain", None)
        if ac is not None and probe is not None:
            ac["probe"] = _lib.require_gpu_tensor(probe, "probe").detach()
            ac["done"] = {k: v for k, v in ac["done"].items() if k[-1] != "data"}
        self.model.diffusion_model.calibrate_precision()
        self._auto_chain_probe()
        return self.auto_chain_report

    @torch.no_grad()
    def _auto_chain_probe(self, data=None) -> None:
        """Where the U-Net's own probe settled on f16mx: the first eight steps of the DDIM-50 chain (eta = 1, Philox noise from a fixed key: the steps with the
        largest 1 / sqrt(alpha_bar) amplification) from a seeded x_T, in f16mx and in f16x3; f16mx is kept only if every row of the state agrees to
        `tolerance` (5e-5, half the contract), otherwise the network runs in f16x3 for these weights.  The conditioning rows: the CALLER's (``data`` = the
        conditioning a sampler was called with, or the ``probe`` of set_precision / calibrate_precision; first and middle row at their own size), once
        per weight signature -- else two seeded synthetic refmaps at 128x128."""
        ac = getattr(self, "_auto_chain", None)
        unet = self.model.diffusion_model
        if ac is None or ac["busy"] or unet.auto_report is None:
            return
        if data is None:
            data = ac.get("probe")
        sigs = (unet._active_set, unet.__dict__["_auto"]["sig"])
        key = sigs + ("data" if data is not None else "synth",)
        if data is None and sigs + ("data",) in ac["done"]:
            key = sigs + ("data",)  # (measured on the caller's rows before: that record stands)
        if key in ac["done"]:
            ac["report"] = ac["done"][key]
            return
        if unet.precision != "f16mx":
            return
        ac["busy"] = True
        try:
            from . import synth
            from .ddim import DDIMSampler

            dev = next(unet.parameters()).device
            if data is not None:
                cond = data[[0, data.shape[0] // 2]] if data.shape[0] > 1 else data[:1]
                cond = cond.detach().to(dev, torch.float32).contiguous()
                B, H, W = cond.shape[0], cond.shape[2], cond.shape[3]
            else:
                B, H, W = 2, 128, 128
                cond = synth.synth_refmaps(B, H, W, 4321).to(dev)
            x_T = torch.randn((B, 3, H, W), generator=torch.Generator().manual_seed(20261004)).to(dev)
            smp = DDIMSampler(self)
            smp.make_schedule(50, ddim_eta=1.0, verbose=False)

            def chain():
                x, _ = smp.ddim_sampling(cond, (B, 3, H, W), x_T=x_T, seed=20261004, num_steps=ac["steps"], log_every_t=0, verbose=False)
                return x.double().flatten(1)

            a = chain()
            unet._set_mode("f16x3")
            b = chain()
            rows = ((a - b).norm(dim=1) / b.norm(dim=1).clamp_min(1e-300)).tolist()
            err = max(rows)
            kept = err <= ac["tolerance"] and bool(torch.isfinite(a).all())
            if kept:
                unet._set_mode("f16mx")
            else:
                unet.auto_override("f16x3", f"chain probe: {ac['steps']} DDIM steps differ from f16x3 by {err:.2e} > {ac['tolerance']:.0e}")
            ac["report"] = {"kept": kept, "rel_l2_chain_vs_f16x3": err, "rows": [float(f"{r:.3e}") for r in rows], "steps": ac["steps"], "tolerance": ac["tolerance"],
                            "probe_source": "caller" if data is not None else "synthetic",
                            "probe": f"{B}x3x{H}x{W} {'conditioning rows of the caller' if data is not None else 'seeded refmaps'}: first {ac['steps']} steps of the DDIM-50 chain (eta 1), worst row"}
            ac["done"][key] = ac["report"]
        finally:
            ac["busy"] = False

    def get_learned_conditioning(self, c):
        if self.cond_stage_forward is None:
            if hasattr(self.cond_stage_model, "encode") and callable(self.cond_stage_model.encode):
                return self.cond_stage_model.encode(c)
            return self.cond_stage_model(c)
        return getattr(self.cond_stage_model, self.cond_stage_forward)(c)

    def decode_first_stage(self, z, predict_cids=False, force_not_quantize=False):
        return self.first_stage_model.decode(1.0 / self.scale_factor * z)

    def apply_model(self, x_noisy, t, cond, return_ids=False):
        """ddpm.py:916-926,1017-1023: one U-Net forward on cat([x, cond], 1) (the cat is folded into the engine)."""
        if not isinstance(cond, dict):
            if not isinstance(cond, list):
                cond = [cond]
            key = "c_concat" if self.model.conditioning_key == "concat" else "c_crossattn"
            cond = {key: cond}
        return self.model(x_noisy, t, **cond)

    def p_mean_variance(self, x, c, t, clip_denoised: bool, return_x0=False, **unused):
        model_out = self.apply_model(x, t, c)
        x_recon = self.predict_start_from_noise(x, t=t, noise=model_out)
        if clip_denoised:
            x_recon.clamp_(-1.0, 1.0)
        model_mean, posterior_variance, posterior_log_variance = self.q_posterior(x_start=x_recon, x_t=x, t=t)
        if return_x0:
            return model_mean, posterior_variance, posterior_log_variance, x_recon
        return model_mean, posterior_variance, posterior_log_variance

    @torch.no_grad()
    def p_sample(self, x, c, t, clip_denoised=False, repeat_noise=False, return_x0=False, temperature=1.0, noise=None, **unused):
        """Single ancestral step (ddpm.py:1120-1167) -- per-step drop-in; the fused loop is ``p_sample_loop``."""
        b = x.shape[0]
        outs = self.p_mean_variance(x=x, c=c, t=t, clip_denoised=clip_denoised, return_x0=return_x0)
        model_mean, _, model_log_variance = outs[:3]
        noise = (torch.randn_like(x) if noise is None else noise) * temperature
        nonzero_mask = (1 - (t == 0).float()).reshape(b, *((1,) * (len(x.shape) - 1)))
        out = model_mean + nonzero_mask * (0.5 * model_log_variance).exp() * noise
        return (out, outs[3]) if return_x0 else out

    def ddpm_coef_table(self) -> np.ndarray:
        """[T,5] fp32: sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod, posterior_mean_coef1/2, exp(0.5*logvar)."""
        tab = torch.stack([self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod, self.posterior_mean_coef1,
                           self.posterior_mean_coef2, (0.5 * self.posterior_log_variance_clipped).exp()], dim=1)
        return np.ascontiguousarray(tab.detach().cpu().numpy().astype(np.float32))

    @torch.no_grad()
    def _ddpm_loop(self, cond, shape, x_T=None, timesteps=None, start_T=None, noise=None, seed=None, mask=None, x0=None, mask_noise=None, blend_when=1,
                   temperature=1.0, noise_dropout=0.0, dropout_keep=None):
        """The ancestral chain on the device (drm_ddpm_sample).  mask / x0: the known-region blending of the reference's two loops --
        ``blend_when`` 1 = LatentDiffusion.p_sample_loop (after p_sample, q_sample(x0, t): ddpm.py:1300-1302), 0 = ObsNetDiffusion.p_sample_loop (before
        p_sample, x0 itself at t == 0, else q_sample(x0, t - 1): models/obsnet.py:545-547); ``mask_noise`` [T,N,C,H,W] injects q_sample's draws.
        ``temperature`` scales the step noise (ddpm.py:1157: the exp(0.5 logvar) column); ``noise_dropout`` is p_sample's F.dropout on it (ddpm.py:1158-1159;
        ``dropout_keep`` [T,N,C,H,W] injects the 0 / 1 masks).  (The reference's p_sample_loop passes neither to p_sample: knobs of p_sample itself.)"""
        if (mask is None) != (x0 is None):
            raise ValueError("mask and x0 go together (ddpm.py:1286-1288)")
        dev = self.betas.device
        c = cond[0] if isinstance(cond, (list, tuple)) else cond
        c = _lib.require_gpu_tensor(c, "cond")
        if seed is None:
            seed = int(torch.randint(0, 2**62, (1,)).item())
        from . import ops

        img = ops.randn(shape, seed, 0, dev) if x_T is None else _lib.require_gpu_tensor(x_T, "x_T").clone()
        T = self.num_timesteps if timesteps is None else timesteps
        if start_T is not None:
            T = min(T, start_T)
        noise = None if noise is None else _lib.require_gpu_tensor(noise, "noise")
        unet = self.model.diffusion_model
        h = unet.engine_handle()
        if getattr(self, "_auto_chain", None) is not None:  # auto mode: the chain probe may move the network to f16x3 for these weights
            self._auto_chain_probe(c)  # (on rows of the caller's conditioning, once per weight signature)
            h = unet.engine_handle()
        L = _lib.lib()
        n, _, hh, ww = shape
        ws = self._ws.get(int(L.drm_sampler_workspace_bytes(h, n, hh, ww)), dev)
        pred_x0 = torch.empty_like(img)
        coef = self.ddpm_coef_table()
        if temperature != 1.0:
            coef = coef.copy()
            coef[:, 4] = (torch.from_numpy(coef[:, 4]) * float(temperature)).numpy()
        if not 0.0 <= noise_dropout < 1.0:
            raise ValueError("noise_dropout: 0 <= p < 1")
        if mask is not None or noise_dropout > 0.0:
            blend = None
            if mask is not None:
                sa = self.sqrt_alphas_cumprod.detach().cpu().float().numpy()
                s1 = self.sqrt_one_minus_alphas_cumprod.detach().cpu().float().numpy()
                q = np.zeros((T, 2), dtype=np.float32)
                for j in range(T):
                    t = T - 1 - j
                    if blend_when == 1:
                        q[j] = (sa[t], s1[t])
                    else:
                        q[j] = (1.0, 0.0) if t == 0 else (sa[t - 1], s1[t - 1])
                blend = _lib.make_mask_blend(mask, x0, q, mask_noise, blend_when, tuple(img.shape))
            opt, keep = _lib.make_sampler_options(tuple(img.shape), T, blend=blend, noise_dropout=noise_dropout, dropout_keep=dropout_keep)
            with torch.cuda.device(dev):
                _lib.check(L.drm_ddpm_sample_ex(h, img.data_ptr(), pred_x0.data_ptr(), c.data_ptr(), coef.ctypes.data_as(C.POINTER(C.c_float)), T,
                                                int(bool(self.clip_denoised)), _lib.ptr(noise), seed, C.byref(opt), n, hh, ww, ws.data_ptr(), ws.numel(),
                                                _lib.stream_ptr(dev)))
            torch.cuda.current_stream(dev).synchronize()  # (the options' tensors stay alive until the chain has run)
            del keep
            return img, pred_x0
        with torch.cuda.device(dev):
            _lib.check(L.drm_ddpm_sample(h, img.data_ptr(), pred_x0.data_ptr(), c.data_ptr(), coef.ctypes.data_as(C.POINTER(C.c_float)), T,
                                         int(bool(self.clip_denoised)), _lib.ptr(noise), seed, n, hh, ww, ws.data_ptr(), ws.numel(),
                                         _lib.stream_ptr(dev)))
        return img, pred_x0

    @torch.no_grad()
    def p_sample_loop(self, cond, shape, return_intermediates=False, x_T=None, verbose=True, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, start_T=None, log_every_t=None, noise=None, seed=None,
                      mask_noise=None, temperature=1.0, noise_dropout=0.0, dropout_keep=None):
        """ddpm.py:1253-1313 -> final img.  (intermediates: only the endpoints are kept; the loop runs on the device.)  mask / x0: ddpm.py:1300-1302."""
        if callback is not None or img_callback is not None or quantize_denoised:
            raise NotImplementedError("callbacks / quantize are not on the shipped path")
        img, _ = self._ddpm_loop(cond, shape, x_T, timesteps, start_T, noise, seed, mask, x0, mask_noise, 1, temperature, noise_dropout, dropout_keep)
        if return_intermediates:
            return img, [img]
        return img

    @torch.no_grad()
    def sample(self, cond, batch_size=16, return_intermediates=False, x_T=None, verbose=True, timesteps=None, quantize_denoised=False,
               mask=None, x0=None, shape=None, **kwargs):
        if shape is None:
            shape = (batch_size, self.channels, self.image_size, self.image_size)
        if cond is not None:
            cond = [c[:batch_size] for c in cond] if isinstance(cond, list) else cond[:batch_size]
        return self.p_sample_loop(cond, shape, return_intermediates=return_intermediates, x_T=x_T, verbose=verbose, timesteps=timesteps,
                                  quantize_denoised=quantize_denoised, mask=mask, x0=x0, **kwargs)


class ObsNetDiffusion(LatentDiffusion):
    """inpainting class (models/obsnet.py:35)."""

    def __init__(self, renderer_config=None, img_renderer_config=None, num_timesteps_cond=None, cond_stage_key="image", padding_mode="noise", *,
                 ddim_steps: Optional[int] = None, ddim_eta: float = 1.0, noisy_observe: float = 0.0, init_from_ckpt_verbose=True, first_stage_config=None,
                 cond_stage_config="__is_first_stage__", ckpt_path=None, ignore_keys=(), **kwargs):
        # (the leading five parameters keep the reference's positional order, models/obsnet.py:38-44; img_renderer_config renders training
        # images only and is never read)
        if first_stage_config is None:
            first_stage_config = {"target": "ldm.models.autoencoder.IdentityFirstStage"}
        super().__init__(first_stage_config, cond_stage_config, num_timesteps_cond, cond_stage_key, **kwargs)
        self.renderer = instantiate_from_config(renderer_config) if renderer_config is not None else None
        self.padding_mode, self.noisy_observe = padding_mode, noisy_observe
        self.ddim_steps, self.ddim_eta = ddim_steps, ddim_eta
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, list(ignore_keys), verbose=init_from_ckpt_verbose)
        self.eval()

    @torch.no_grad()
    def p_sample_loop(self, cond, shape, return_intermediates=False, x_T=None, verbose=True, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, start_T=None, log_every_t=None, noise=None, seed=None,
                      mask_noise=None, temperature=1.0, noise_dropout=0.0, dropout_keep=None):
        """models/obsnet.py:500-564: like LatentDiffusion.p_sample_loop but returns pred_x0 of the LAST step; mask / x0 blend BEFORE p_sample
        (x0 itself at t == 0, else q_sample(x0, t - 1): models/obsnet.py:545-547)."""
        if callback is not None or img_callback is not None or quantize_denoised:
            raise NotImplementedError("callbacks / quantize are not on the shipped path")
        img, pred_x0 = self._ddpm_loop(cond, shape, x_T, timesteps, start_T, noise, seed, mask, x0, mask_noise, 0, temperature, noise_dropout, dropout_keep)
        if return_intermediates:
            return pred_x0, {"x_inter": [img], "pred_x0": [pred_x0]}
        return pred_x0

    @torch.no_grad()
    def sample_log(self, cond, batch_size, ddim, ddim_steps, **kwargs):
        """models/obsnet.py:566-583."""
        if ddim:
            from .ddim import DDIMSampler

            ddim_sampler = DDIMSampler(self)
            shape = (self.channels, self.image_size, self.image_size)
            samples, intermediates = ddim_sampler.sample(
                ddim_steps, batch_size, shape, cond, verbose=False,
                log_every_t=kwargs.pop("log_every_t", None) or max(self.log_every_t * ddim_steps // self.num_timesteps, 1), **kwargs)
        else:
            samples, intermediates = self.sample(cond=cond, batch_size=batch_size, return_intermediates=True, **kwargs)
        return samples, intermediates

    @torch.no_grad()
    def get_cond_for_predict(self, batch: Dict[str, Union[torch.Tensor, str]], bs: Optional[int] = None, force_c_encode: bool = False,
                             noise: Optional[torch.Tensor] = None):
        """models/obsnet.py:656-704 (cond_stage_key == 'raw_refmap').  The reference's in-place ``cond += ...`` also mutates
        ``c`` because IdentityFirstStage.encode returns the same tensor (SURVEY.md 7 'bug-compat aliasing'); restated explicitly:
        c = raw_refmap*mask + (1-mask)*noise."""
        if self.model.conditioning_key is None:
            mask = batch.get("mask")
            return None, (mask[:bs, None].float() if mask is not None else None), batch["tag"][:bs]
        if self.cond_stage_key != "raw_refmap":
            raise NotImplementedError(self.cond_stage_key)
        mask = batch["raw_refmask"][:bs, None].float()
        raw_refmap = self.ds.transform(batch["raw_refmap"][:bs], dynamic_normalize=True, mask=mask)
        cond = raw_refmap * mask
        if self.noisy_observe > 0:
            cond = self.noisy_observe * torch.randn_like(cond) + cond
        c = self.get_learned_conditioning(cond.to(self.device))
        if tuple(mask.shape[-2:]) != (self.image_size, self.image_size):  # :691 (default mode: nearest); a no-op on the shipped configs
            mask = ops.resize(mask, (self.image_size, self.image_size), "nearest")
        if self.padding_mode == "noise":
            nz = torch.randn_like(c) if noise is None else noise
            c = c + (1 - mask) * nz
        elif self.padding_mode != "zeros":
            raise NotImplementedError()
        return c, mask, batch["tag"][:bs]
