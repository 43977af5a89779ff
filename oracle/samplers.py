"""Oracle: sampler loops and schedules, restated (TEST INFRASTRUCTURE).

Follows the reference:
  DDPM.register_schedule                  ldm/models/diffusion/ddpm.py:137-187
  make_beta_schedule("linear")            ldm/modules/diffusionmodules/util.py:21-25
  make_ddim_timesteps / _sampling_params  util.py:46-74
  DDIMSampler.ddim_sampling/p_sample_ddim ldm/models/diffusion/ddim.py:128-259
  LatentDiffusion.p_sample / p_mean_variance / q_posterior   ddpm.py:1079-1167, :233-246
  ObsNetDiffusion.p_sample_loop           models/obsnet.py:500-564 (returns pred_x0 of last step)
  DRMNet.forward / get_schedule / get_brdf_out / check_convergence / p_sample_loop
                                          models/drmnet.py:452-456, :458-501, :390-396, :747-750, :782-847

All random draws are INJECTED (the reference draws from the global torch
generator with data-dependent shapes, SURVEY.md 7 "RNG parity"):
  * DRMNet loop: ``noise0`` [B,3,H,W]; ``step_noise[i]`` [B,3,H,W] -- row b is used
    by sample b at step i iff b is still active and did not converge at step i.
  * DDIM / DDPM: ``x_T`` and ``step_noise[j]`` for the j-th executed step.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

# ----------------------------------------------------------------------------- schedules


def ddpm_schedule(timesteps: int = 1000, linear_start: float = 1e-4, linear_end: float = 2e-2, v_posterior: float = 0.0):
    """fp64 numpy tables cast to fp32, as ddpm.py:137-176 (beta_schedule='linear')."""
    betas = (torch.linspace(linear_start**0.5, linear_end**0.5, timesteps, dtype=torch.float64) ** 2).numpy()
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    post_var = (1 - v_posterior) * betas * (1.0 - ac_prev) / (1.0 - ac) + v_posterior * betas
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)
    return {
        "betas": f32(betas),
        "alphas_cumprod": f32(ac),
        "alphas_cumprod_prev": f32(ac_prev),
        "sqrt_alphas_cumprod": f32(np.sqrt(ac)),
        "sqrt_one_minus_alphas_cumprod": f32(np.sqrt(1.0 - ac)),
        "log_one_minus_alphas_cumprod": f32(np.log(1.0 - ac)),
        "sqrt_recip_alphas_cumprod": f32(np.sqrt(1.0 / ac)),
        "sqrt_recipm1_alphas_cumprod": f32(np.sqrt(1.0 / ac - 1)),
        "posterior_variance": f32(post_var),
        "posterior_log_variance_clipped": f32(np.log(np.maximum(post_var, 1e-20))),
        "posterior_mean_coef1": f32(betas * np.sqrt(ac_prev) / (1.0 - ac)),
        "posterior_mean_coef2": f32((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)),
    }


def ddim_schedule(alphas_cumprod: torch.Tensor, num_ddpm: int, S: int, eta: float):
    """util.py:46-74 ('uniform'), ddim.py:29-62.

    NB the reference feeds the *fp32* alphas_cumprod buffer (``.cpu()``) into a mixed
    numpy/torch expression; the dtypes below reproduce its rounding sequence exactly
    (checked bit-for-bit against tests/golden/ddim_schedule_eta*.npz).
    """
    c = num_ddpm // S
    tau = np.asarray(list(range(0, num_ddpm, c))) + 1
    ac = alphas_cumprod.detach().cpu().float()
    a = ac[torch.from_numpy(tau)]  # fp32
    a_prev = torch.tensor([float(ac[0])] + [float(v) for v in ac[torch.from_numpy(tau[:-1])]], dtype=torch.float64)
    # util.py:69 mixes an fp32 tensor with a float64 array: ``ndarray / Tensor`` dispatches to
    # Tensor.__rtruediv__ = reciprocal() * other, so 1/(1-a) is rounded in fp32, the rest is float64
    recip_one_minus_a = (1 - a).reciprocal().double()
    sig = eta * torch.sqrt(recip_one_minus_a * (1 - a_prev) * (1 - a.double() / a_prev))
    return {
        "ddim_timesteps": tau,
        "ddim_alphas": a,
        "ddim_alphas_prev": a_prev,
        "ddim_sigmas": sig,
        "ddim_sqrt_one_minus_alphas": torch.sqrt(1.0 - a),  # ddim.py:57, fp32
    }


def ddim_step_coeffs(sched: dict) -> np.ndarray:
    """Per-index fp32 scalars the update uses: [sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev-sigma^2), sigma].

    ddim.py:243-258 builds each with torch.full(..., fp32) and then applies fp32
    tensor ops (.sqrt(), **2); this reproduces that rounding sequence.
    """
    S = len(sched["ddim_timesteps"])
    out = np.zeros((S, 5), dtype=np.float32)
    for i in range(S):
        a_t = torch.full((1,), float(sched["ddim_alphas"][i]))
        a_prev = torch.full((1,), float(sched["ddim_alphas_prev"][i]))
        sigma = torch.full((1,), float(sched["ddim_sigmas"][i]))
        s1m = torch.full((1,), float(sched["ddim_sqrt_one_minus_alphas"][i]))
        out[i, 0] = a_t.sqrt().item()
        out[i, 1] = s1m.item()
        out[i, 2] = a_prev.sqrt().item()
        out[i, 3] = (1.0 - a_prev - sigma**2).sqrt().item()
        out[i, 4] = sigma.item()
    return out


# ----------------------------------------------------------------------------- DDIM (ObsNet default)


@torch.no_grad()
def ddim_sample(
    eps_model: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
    cond: torch.Tensor,
    x_T: torch.Tensor,
    step_noise: Sequence[torch.Tensor],
    sched: dict,
    num_steps: Optional[int] = None,
    mask: Optional[torch.Tensor] = None,
    x0: Optional[torch.Tensor] = None,
    q_noise: Optional[Sequence[torch.Tensor]] = None,
    q_coef: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
    temperature: float = 1.0,
    uncond: Optional[torch.Tensor] = None,
    guidance_scale: float = 1.0,
    noise_dropout: float = 0.0,
    dropout_keep: Optional[Sequence[torch.Tensor]] = None,
) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """ddim.py:171-204 + :206-259.  eps_model(xc, t) with xc = cat([x, cond], 1)
    (DiffusionWrapper concat branch, ddpm.py:1527-1529).  Returns final x (not pred_x0).
    mask / x0 (ddim.py:175-178): before every step x = q_sample(x0, t) * mask + (1 - mask) * x with q_sample = sqrt_alphas_cumprod[t] x0 +
    sqrt_one_minus_alphas_cumprod[t] q_noise[j] (ddpm.py:1052-1058; q_coef = those two fp32 tables); temperature scales the step noise (ddim.py:255);
    uncond + guidance_scale: classifier-free guidance (ddim.py:225-232); noise_dropout + dropout_keep: F.dropout on the step noise (ddim.py:256-257)."""
    tau = sched["ddim_timesteps"]
    coef = ddim_step_coeffs(sched)
    total = len(tau)
    x = x_T
    xs = []
    b = x.shape[0]
    for j, index in enumerate(range(total - 1, -1, -1)):
        if num_steps is not None and j >= num_steps:
            break
        t = torch.full((b,), int(tau[index]), dtype=torch.long)
        if mask is not None:
            ti = int(tau[index])
            x = (q_coef[0][ti] * x0 + q_coef[1][ti] * q_noise[j]) * mask + (1.0 - mask) * x
        e = eps_model(torch.cat([x, cond], dim=1), t)
        if uncond is not None and guidance_scale != 1.0:  # classifier-free guidance, ddim.py:225-232
            e_u = eps_model(torch.cat([x, uncond], dim=1), t)
            e = e_u + guidance_scale * (e - e_u)
        sa, s1m, sap, sdir, sig = (torch.tensor(float(v)) for v in coef[index])
        pred_x0 = (x - s1m * e) / sa
        nz = sig * step_noise[j] * temperature
        if noise_dropout > 0.0:  # F.dropout(noise, p), ddim.py:256-257: kept elements scaled by 1 / (1 - p)
            nz = nz * dropout_keep[j] / (1.0 - noise_dropout)
        x = sap * pred_x0 + sdir * e + nz
        xs.append(x)
    return x, xs


# ----------------------------------------------------------------------------- ancestral DDPM (ObsNet, ddim_steps=None)


@torch.no_grad()
def ddpm_sample(
    eps_model: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
    cond: torch.Tensor,
    x_T: torch.Tensor,
    step_noise: Sequence[torch.Tensor],
    S: dict,
    timesteps: Optional[int] = None,
    start_T: Optional[int] = None,
    clip_denoised: bool = False,
    mask: Optional[torch.Tensor] = None,
    x0: Optional[torch.Tensor] = None,
    q_noise: Optional[Sequence[torch.Tensor]] = None,
    blend: str = "obsnet",
    temperature: float = 1.0,
    t_list: Optional[Sequence[int]] = None,
    noise_dropout: float = 0.0,
    dropout_keep: Optional[Sequence[torch.Tensor]] = None,
):
    """obsnet.py:500-564 over ddpm.py:1079-1167.  Returns (pred_x0_last, img_last, [img per step]).
    mask / x0: blend = "obsnet" (models/obsnet.py:545-547: BEFORE p_sample, img_orig = x0 at t == 0 else q_sample(x0, t - 1)) or "ldm"
    (ddpm.py:1300-1302: AFTER p_sample, q_sample(x0, t)); temperature scales the step noise (ddpm.py:1157); t_list: explicit timesteps (p_sample calls)."""
    def q_sample(t, nz):  # ddpm.py:1052-1058
        return S["sqrt_alphas_cumprod"][t] * x0 + S["sqrt_one_minus_alphas_cumprod"][t] * nz

    T = S["betas"].shape[0] if timesteps is None else timesteps
    if start_T is not None:
        T = min(T, start_T)
    img = x_T
    b = img.shape[0]
    imgs = []
    pred_x0 = None
    for j, i in enumerate(range(T - 1, -1, -1) if t_list is None else t_list):
        t = torch.full((b,), i, dtype=torch.long)
        if mask is not None and blend == "obsnet":
            img_orig = x0 if i == 0 else q_sample(i - 1, q_noise[j])
            img = img_orig * mask + (1.0 - mask) * img
        e = eps_model(torch.cat([img, cond], dim=1), t)
        x_recon = S["sqrt_recip_alphas_cumprod"][i] * img - S["sqrt_recipm1_alphas_cumprod"][i] * e  # ddpm.py:233-237
        if clip_denoised:
            x_recon = x_recon.clamp(-1.0, 1.0)
        mean = S["posterior_mean_coef1"][i] * x_recon + S["posterior_mean_coef2"][i] * img  # :239-246
        logvar = S["posterior_log_variance_clipped"][i]
        nonzero = 0.0 if i == 0 else 1.0
        nz = step_noise[j] * temperature
        if noise_dropout > 0.0:  # ddpm.py:1158-1159
            nz = nz * dropout_keep[j] / (1.0 - noise_dropout)
        img = mean + nonzero * (0.5 * logvar).exp() * nz  # :1156-1167
        if mask is not None and blend == "ldm":
            img = q_sample(i, q_noise[j]) * mask + (1.0 - mask) * img
        pred_x0 = x_recon
        imgs.append(img)
    return pred_x0, img, imgs


# ----------------------------------------------------------------------------- DRMNet reverse process


def gamma_pow(gamma: float, i: int) -> torch.Tensor:
    """drmnet.py:494-495: exp(i * ln(gamma)) evaluated in fp64, then cast to fp32."""
    return torch.exp(torch.tensor([float(i)], dtype=torch.float64) * math.log(gamma)).float()


def brdf_schedule(z_out: torch.Tensor, z0: torch.Tensor, gamma: float, i: int):
    """get_brdf_out in eval mode (drmnet.py:390-396): zk = clamp(z0 + gamma^i (z_out - z0), 0, 1), zK = clamp(z_out, 0, 1)."""
    zk = gamma_pow(gamma, i).unsqueeze(-1) * (z_out - z0) + z0
    return zk.clamp(0, 1), z_out.clamp(0, 1)


def check_convergence(zk: torch.Tensor, z0: torch.Tensor, eps: float) -> torch.Tensor:
    """drmnet.py:747-750."""
    d = torch.linalg.norm((zk - z0).abs(), dim=-1)
    return torch.logical_or(d < eps, d == 0)


@torch.no_grad()
def drmnet_sample(
    refnet: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
    illnet: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
    LrK: torch.Tensor,
    noise0: torch.Tensor,
    step_noise: Sequence[torch.Tensor],
    z0: torch.Tensor,
    gamma: float,
    epsilon: float,
    delta: float,
    max_timesteps: int,
    trace: Optional[list] = None,
):
    """drmnet.py:782-847.  refnet(xc, timesteps)->[n,6]; illnet(xc, delta_z)->[n,3,H,W] (z-MLP inside).

    Conditioning is the un-noised LrK for both nets (get_input_for_predict, drmnet.py:1037-1043).
    Returns (Lr_0, zK, K) with zK = NaN / K = max_timesteps for rows that never converge.
    """
    B = LrK.shape[0]
    zdim = z0.shape[0]
    Lr_k = LrK + delta * noise0
    active = torch.ones(B, dtype=torch.bool)
    K = torch.full((B,), max_timesteps, dtype=torch.int32)
    zK = torch.full((B, zdim), float("nan"), dtype=torch.float32)
    for i in range(max_timesteps):
        idx = torch.where(active)[0]
        xc = torch.cat([Lr_k[idx], LrK[idx]], dim=1)
        z_out = refnet(xc, torch.full((idx.numel(),), i, dtype=torch.long))
        zk, zKc = brdf_schedule(z_out, z0, gamma, i)
        out = illnet(xc, zk - z0)
        mean = Lr_k[idx] + out
        conv = check_convergence(zk, z0, epsilon)
        nc = ~conv
        mean[nc] = mean[nc] + step_noise[i][idx[nc]] * delta
        Lr_k[idx] = mean
        done = idx[conv]
        K[done] = i + 1
        zK[done] = zKc[conv]
        active[done] = False
        if trace is not None:
            trace.append({"Lr_k": Lr_k.clone(), "zk": zk.clone(), "idx": idx.clone()})
        if not bool(active.any()):
            break
    return Lr_k, zK, K
