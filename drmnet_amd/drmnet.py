"""``DRMNet`` -- host-side operator surface of the reference LightningModule, sampling on the HIP engine.

Mirrors models/drmnet.py of the reference by name, signature and return values for everything the
inference path touches (scripts/estimate.py:84-100):
  __init__ params (configs/drmnet/eval_drmnet.yaml), ema_scope :242-258, init_from_ckpt :260-277,
  apply_model :376-388, get_brdf_out :390-396, forward :452-456, get_schedule :458-501,
  check_convergence :747-750, p_mean_variance :752-770, p_sample (stub) :772-780,
  p_sample_loop :782-847, get_input_for_predict :1011-1045, decode_first_stage, r0toenvmap :931-941.
Training (p_losses, get_input, caches, log_images, Mitsuba rendering) is out of scope.

The module is a plain ``nn.Module`` (pytorch_lightning is not needed for inference); ``state_dict()`` has the
reference's keys, so ``drmnet.ckpt`` loads with ``init_from_ckpt``.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import _lib
from .config import instantiate_from_config
from .wrappers import DiffusionWrapper, IdentityFirstStage, LitEma, ZEmbDiffusionWrapper, ema_weights, load_checkpoint


# Constructor / YAML keys of the reference (models/drmnet.py:79-240) that steer only training, logging or dataset caches: accepted so that
# configs/drmnet/*.yaml load unchanged, never read.  Any other unknown key is an error.
_TRAINING_ONLY = frozenset({
    "loss_type", "monitor", "scheduler_config", "cond_stage_trainable", "l_refmap_weight", "l_refcode_weight", "sigma",
    "train_with_zk_gt", "train_with_zk_gt_switch_epoch", "cache_refmap", "refmap_cache_root", "envmap_dir",
})


class DRMNet(nn.Module):
    def __init__(self, illnet_config, refnet_config, renderer_config=None, max_timesteps: int = 250, *, ckpt_path: Optional[str] = None,
                 init_from_ckpt_verbose: bool = True, ignore_keys=(), use_ema: bool = True, input_key: str = "LrK", sigma_for_cond_xK: float = 0.0,
                 image_size: int = 128, channels: int = 3, log_every_k: int = 5, parameterization: str = "residual", concat_mode: bool = False,
                 conditioning_key: Optional[str] = None, scale_factor: float = 1.0, scale_by_std: bool = False, delta: float = 0.0125,
                 gamma: float = 0.9, epsilon: float = 0.001, brdf_param_names=("specular",), z0=(1.0,), model_emb_z: bool = True,
                 emb_z_crossattn: bool = False, refmap_input_scaler: Optional[float] = None, first_stage_config=None,
                 cond_stage_config="__is_first_stage__", basis_r0: Optional[torch.Tensor] = None, cond_stage_forward: Optional[str] = None,
                 **training_only):
        # (the leading four parameters keep the reference's positional order, models/drmnet.py:79-85: DRMNet(ill, ref, renderer_cfg, 250))
        super().__init__()
        unknown = sorted(set(training_only) - _TRAINING_ONLY)
        if unknown:
            raise TypeError(f"DRMNet: unexpected parameter(s) {unknown}")
        if parameterization != "residual":
            raise NotImplementedError('only the "residual" parameterization exists (models/drmnet.py:121)')
        if not concat_mode:
            raise AssertionError("This model only supports concat mode")
        if scale_by_std:
            raise NotImplementedError("scale_by_std is training-only")
        if cond_stage_forward is not None:
            # the reference reads it in get_learned_conditioning on the sampling path (models/drmnet.py:366-373): a config that sets it must
            # not silently behave differently
            raise NotImplementedError("cond_stage_forward: only the default (cond_stage_model.encode of the identity first stage) is on the shipped path")
        if cond_stage_config not in ("__is_first_stage__", "__is_unconditional__"):
            raise NotImplementedError("a separate cond_stage_config is not on the shipped path")
        # sampler constants (models/drmnet.py:782-847 reads them per step) and estimate.py's attribute surface
        self.parameterization = parameterization
        self.max_timesteps, self.gamma, self.epsilon, self.delta = max_timesteps, gamma, epsilon, delta
        self.log_every_k, self.input_key, self.sigma_for_cond_xK = log_every_k, input_key, sigma_for_cond_xK
        self.image_size, self.channels, self.scale_factor = image_size, channels, scale_factor
        self.brdf_param_names = list(brdf_param_names)
        self.refmap_input_scaler = refmap_input_scaler
        self.concat_mode = concat_mode
        self._z0 = torch.tensor(list(z0), dtype=torch.float32)
        self.zdim = len(self._z0)
        self.register_buffer("z0", self._z0)
        self.instantiate_brdf_model(renderer_config, basis_r0)
        # the two networks behind the reference's wrappers (state_dict keys illnet_model.* / refnet_model.* [+ *_ema.*])
        if conditioning_key is None:
            conditioning_key = "concat"
        if cond_stage_config == "__is_unconditional__":
            conditioning_key = None
        self.illnet_model = ZEmbDiffusionWrapper(illnet_config, conditioning_key, self.zdim, model_emb_z, emb_z_crossattn)
        self.refnet_model = DiffusionWrapper(refnet_config, conditioning_key)
        self.use_ema = use_ema
        if use_ema:
            self.illnet_model_ema = LitEma(self.illnet_model)
            self.refnet_model_ema = LitEma(self.refnet_model)
        self.first_stage_model = instantiate_from_config(first_stage_config or {"target": "ldm.models.autoencoder.IdentityFirstStage"}).eval()
        if not isinstance(self.first_stage_model, IdentityFirstStage):
            raise NotImplementedError("only IdentityFirstStage is used by the shipped configs")
        self.cond_stage_model = self.first_stage_model if cond_stage_config == "__is_first_stage__" else None
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=list(ignore_keys), verbose=init_from_ckpt_verbose)
        self.eval()
        self._samplers = {}  # weight set ("live" / "ema") -> (drm_drmnet handle, signature it was built from)
        self._weight_set = "live"
        self._ws = _lib.Workspace()

    # ------------------------------------------------------------------ plumbing kept from the reference
    @property
    def device(self):
        return self.z0.device

    def instantiate_brdf_model(self, config, basis_r0=None):
        """Reference renders basis_r0 (white envmap, BRDF z0) through Mitsuba (drmnet.py:328-347). Out of scope here:
        basis_r0 is an optional input (ones by default) -- see DESIGN.md.  For the shipped z0 = [1, 1, 1, 1, 0, 1] (white, fully metallic,
        roughness 0: Schlick F0 = base colour = 1) under the white environment the rendered quantity is analytically 1 on the sphere, so ones is
        its noise-free value; any other z0 needs the rendered basis (first screen)."""
        self.renderer = instantiate_from_config(config) if config is not None else None
        if basis_r0 is None:
            basis_r0 = torch.ones(3, self.image_size, self.image_size)
        self.register_buffer("basis_r0", basis_r0.float(), persistent=False)

    def ema_scope(self, context=None):
        """models/drmnet.py:242-258 -- ``with model.ema_scope(): ...`` samples with the EMA weights of both networks (and of the
        z-embedding MLP, which lives in illnet_model); see wrappers.ema_weights."""
        pairs = [(self.illnet_model, self.illnet_model_ema), (self.refnet_model, self.refnet_model_ema)] if self.use_ema else []
        return ema_weights(self, pairs, context)

    def init_from_ckpt(self, path, ignore_keys=list(), only_model=False, verbose=True):
        """models/drmnet.py:260-277 (``only_model`` loads into illnet_model alone, as there)."""
        load_checkpoint(self, path, ignore_keys, into=self.illnet_model if only_model else None, verbose=verbose)

    @torch.no_grad()
    def encode_first_stage(self, x):
        return self.first_stage_model.encode(x)

    def get_first_stage_encoding(self, encoder_posterior):
        assert isinstance(encoder_posterior, torch.Tensor)
        return self.scale_factor * encoder_posterior

    def decode_first_stage(self, z, predict_cids=False, force_not_quantize=False):
        z = 1.0 / self.scale_factor * z  # LatentDiffusion.decode_first_stage, ddpm.py:731-789 (identity first stage)
        return self.first_stage_model.decode(z)

    # ------------------------------------------------------------------ per-step pieces (reference semantics)
    def apply_model(self, model: DiffusionWrapper, input_refmap: torch.Tensor, k: Union[torch.Tensor, int], cond, rows=None):
        if not isinstance(cond, dict):
            if not isinstance(cond, list):
                cond = [cond]
            key = "c_concat" if model.conditioning_key == "concat" else "c_crossattn"
            cond = {key: cond}
        if isinstance(k, int):
            n = input_refmap.size(0) if rows is None else rows.numel()
            k = torch.full((n,), k, device=input_refmap.device)
        return model(input_refmap, k, rows=rows, **cond)

    def get_schedule(self, zK, z0=None, reversed_k=None, normalized_k=None, return_zkm1=False, power_precision=torch.double):
        """models/drmnet.py:458-501.  The BRDF code decays geometrically from zK towards the mirror code z0: after ``r`` reverse
        steps the offset is gamma^r (zK - z0), with gamma^r evaluated as exp(r ln gamma) in ``power_precision`` (fp64) and cast to
        fp32 before the multiply.  K = int(log_gamma(epsilon / |zK - z0|)) + 2 is the step count at which the offset falls below
        epsilon.  Exactly one of ``reversed_k`` (steps done) / ``normalized_k`` (fraction of K) selects the point.
        Returns (K, k, zk[, zk one step later]).  Tiny [n, z_dim] host-side math; the sampler kernels restate the reversed_k branch."""
        if (normalized_k is None) == (reversed_k is None):
            raise AssertionError("normalized_k and reversed_k are exclusive")
        anchor = self.z0 if z0 is None else z0.to(zK.device)
        offset = zK - anchor
        ln_gamma = math.log(self.gamma)
        K = (torch.log(self.epsilon / torch.linalg.norm(offset, dim=-1)) / ln_gamma).int() + 2
        if normalized_k is None:
            k = K - reversed_k - 1
            steps = torch.tensor([reversed_k], device=zK.device) if isinstance(reversed_k, int) else reversed_k
        else:
            K = K.clip(min=1).int()
            k = (normalized_k * K).int()
            steps = K - k - 1
        steps = steps.to(power_precision)

        def code_after(r):
            return torch.exp(r.unsqueeze(-1) * ln_gamma).float() * offset + anchor

        if return_zkm1:
            return K, k, code_after(steps), code_after(steps + 1)
        return K, k, code_after(steps)

    def get_brdf_out(self, brdf_model_out, reversed_k=None):
        zK = brdf_model_out
        _, _, zk = self.get_schedule(zK, reversed_k=reversed_k)
        if not self.training:
            zk = zk.clamp(0, 1)
            zK = zK.clamp(0, 1)
        return zk, zK

    def check_convergence(self, zk):
        distance = torch.linalg.norm((zk - self.z0).abs(), dim=-1)
        return torch.logical_or(distance < self.epsilon, distance == 0)

    def forward(self, Lr_k, illnet_cond, refnet_cond, reversed_k):
        z_out = self.apply_model(self.refnet_model, Lr_k, reversed_k, refnet_cond)
        zk, _ = self.get_brdf_out(z_out, reversed_k=reversed_k)
        Delta = zk - self.z0
        return self.apply_model(self.illnet_model, Lr_k, Delta, illnet_cond), z_out

    def p_mean_variance(self, Lr_k, illnet_cond, refnet_cond, reversed_k, return_model_out=False):
        model_out, z_out = self(Lr_k, illnet_cond, refnet_cond, reversed_k)
        model_mean = Lr_k + model_out
        if return_model_out:
            return model_mean, self.delta, z_out, model_out
        return model_mean, self.delta, z_out

    def p_sample(self, Lr_k, illnet_cond, refnet_cond, reversed_k, return_model_out=False):
        raise NotImplementedError("")  # the reference leaves this unimplemented too (drmnet.py:772-780)

    def set_precision(self, precision: str, probe: Optional[torch.Tensor] = None) -> "DRMNet":
        """Conv arithmetic of both networks: "fp32" (exact fp32 MFMA), "f16x3" (split fp16, fp32-accurate, ~2.5x faster), "f16mx" (f16x3 with
        fp8 cross terms on the 3x3 convs: ~3e-5 per network, ~3x faster), "auto" (f16mx per network only where a probe forward on the loaded
        weights agrees with f16x3 to 5e-5, else f16x3: unet.set_precision_auto), "f16" / "bf16" (reduced precision)."""
        self.illnet_model.diffusion_model.set_precision(precision)
        self.refnet_model.diffusion_model.set_precision(precision)
        # "auto": besides the per-network probes (unet.py), the CHAIN is measured before f16mx is kept -- see _auto_chain_probe
        # ``probe`` [n,3,H,W]: refmaps of the CALLER to measure the chain on (the first and the middle row are used) instead of the seeded synthetic
        # pair; without it the first batch p_sample_loop sees is handed to the probe (once per weight signature)
        self._auto_chain = {"tolerance": self.AUTO_CHAIN_TOLERANCE, "steps": self.AUTO_CHAIN_STEPS, "done": {}, "busy": False, "report": None,
                            "probe": None if probe is None else _lib.require_gpu_tensor(probe, "probe").detach()} if precision == "auto" else None
        return self

    AUTO_CHAIN_TOLERANCE = 5e-5  # half the 1e-4 contract, like the per-network probe
    AUTO_CHAIN_STEPS = 8

    @property
    def auto_chain_report(self) -> Optional[dict]:
        """{"kept", "rel_l2_chain_vs_f16x3" (worst row), "steps", "tolerance", "modes"} of the last chain probe; None outside auto mode / before it ran"""
        ac = getattr(self, "_auto_chain", None)
        return None if ac is None else ac["report"]

    def calibrate_precision(self, probe: Optional[torch.Tensor] = None) -> Optional[dict]:
        """Auto mode: runs the per-network probes and the chain probe now (weights on a GPU) and returns the chain report.  ``probe``: refmaps of the
        caller to run the chain on (re-measured for these rows even if a probe of these weights is on record)."""
        ac = getattr(self, "_auto_chain", None)
        if ac is not None and probe is not None:
            ac["probe"] = _lib.require_gpu_tensor(probe, "probe").detach()
            ac["done"] = {k: v for k, v in ac["done"].items() if k[-1] != "data"}
        self._engine()
        return self.auto_chain_report

    @torch.no_grad()
    def _auto_chain_probe(self, which: str, data: Optional[torch.Tensor] = None) -> None:
        """A per-network probe compares ONE forward; the sampler applies ~100 of them to its own output.  So where the networks settled on f16mx, eight
        reverse steps (RefNet -> schedule -> z-MLP -> IllNet -> update, Philox noise from a fixed key, every row active: drm_drmnet_step) are run in
        the chosen modes and in f16x3; f16mx is kept only if every row of the final state agrees to `tolerance`, otherwise BOTH networks run in f16x3
        for these weights.  The rows: the CALLER's refmaps when there are any -- ``data`` (the batch p_sample_loop was called with, or the ``probe`` of
        set_precision / calibrate_precision; first and middle row, at their own size), once per weight signature -- else two seeded synthetic refmaps at
        128x128 (calibrate_precision() before any data exists)."""
        from . import synth

        ac = self._auto_chain
        ill, ref = self.illnet_model.diffusion_model, self.refnet_model.diffusion_model
        if ill.auto_report is None or ref.auto_report is None:
            return
        if data is None:
            data = ac.get("probe")
        sigs = (which, ill.__dict__["_auto"]["sig"], ref.__dict__["_auto"]["sig"])
        key = sigs + ("data" if data is not None else "synth",)
        if data is None and sigs + ("data",) in ac["done"]:
            key = sigs + ("data",)  # (measured on the caller's rows before: that record stands)
        if key in ac["done"]:
            ac["report"] = ac["done"][key]
            return
        if "f16mx" not in (ill.precision, ref.precision):
            return
        ac["busy"] = True
        try:
            dev = next(ill.parameters()).device
            if data is not None:
                x = data[[0, data.shape[0] // 2]] if data.shape[0] > 1 else data[:1]
                x = x.detach().to(dev, torch.float32).contiguous()
                B, H, W = x.shape[0], x.shape[2], x.shape[3]
            else:
                B, H, W = 2, 128, 128
                x = synth.synth_refmaps(B, H, W, 4321).to(dev)
            L = _lib.lib()
            chosen = (ill.precision, ref.precision)

            def chain():
                h = self._engine_raw()
                ws = self._ws.get(int(L.drm_drmnet_workspace_bytes(h, B, H, W)), dev)
                Lr_k = x.clone()
                with torch.cuda.device(dev):
                    for i in range(ac["steps"]):
                        _lib.check(L.drm_drmnet_step(h, Lr_k.data_ptr(), x.data_ptr(), None, B, i, None, 20261004, None, None, None, B, H, W, ws.data_ptr(),
                                                     ws.numel(), _lib.stream_ptr(dev)))
                return Lr_k.double().flatten(1)

            a = chain()
            ill._set_mode("f16x3")
            ref._set_mode("f16x3")
            b = chain()
            rows = ((a - b).norm(dim=1) / b.norm(dim=1).clamp_min(1e-300)).tolist()
            err = max(rows)
            kept = err <= ac["tolerance"] and bool(torch.isfinite(a).all())
            if kept:
                ill._set_mode(chosen[0])
                ref._set_mode(chosen[1])
            else:
                why = f"chain probe: {ac['steps']} DRMNet steps differ from f16x3 by {err:.2e} > {ac['tolerance']:.0e}"
                ill.auto_override("f16x3", why)
                ref.auto_override("f16x3", why)
            ac["report"] = {"kept": kept, "rel_l2_chain_vs_f16x3": err, "rows": [float(f"{r:.3e}") for r in rows], "steps": ac["steps"], "tolerance": ac["tolerance"],
                            "modes": {"illnet": chosen[0], "refnet": chosen[1]}, "probe_source": "caller" if data is not None else "synthetic",
                            "probe": f"{B}x3x{H}x{W} {'rows of the caller' if data is not None else 'seeded refmaps'}, {ac['steps']} reverse steps, worst row"}
            ac["done"][key] = ac["report"]
        finally:
            ac["busy"] = False

    # ------------------------------------------------------------------ the device sampler
    def _engine(self, data: Optional[torch.Tensor] = None):
        """_engine_raw() behind the auto mode's chain probe (which may move both networks to f16x3 for the current weights); ``data``: the refmaps
        the caller is about to sample from -- the chain probe runs on rows of them (once per weight signature)."""
        h = self._engine_raw()
        ac = getattr(self, "_auto_chain", None)
        if ac is not None and not ac["busy"]:
            self._auto_chain_probe(getattr(self, "_weight_set", "live"), data)
            h = self._engine_raw()
        return h

    def _engine_raw(self):
        """The device sampler handle for the weight set that is live right now ("live" parameters, or the EMA shadow inside
        ``ema_scope``): one handle per set, rebuilt only when something it was built from changes."""
        which = getattr(self, "_weight_set", "live")
        ill, ref = self.illnet_model.diffusion_model, self.refnet_model.diffusion_model
        hi, hr = ill.engine_handle(), ref.engine_handle()
        if which == "ema":
            zp = self.illnet_model_ema.shadow_for(self.illnet_model.z_emb_param_names())
        else:
            zp = [p.detach() for p in self.illnet_model.z_emb_params()]
        for p in zp:
            _lib.require_gpu_tensor(p, "z_emb_layer parameter")
        sig = (hi.value, hr.value, ill.precision, ref.precision, tuple((p.data_ptr(), p._version) for p in zp), float(self.gamma), float(self.epsilon),
               float(self.delta), int(self.max_timesteps), tuple(self._z0.tolist()))
        cached = self._samplers.get(which)
        if cached is not None and cached[1] == sig:
            return cached[0]
        self._free_sampler(which)
        cfg = _lib.DrmnetCfg()
        cfg.z_dim = self.zdim
        cfg.max_timesteps = int(self.max_timesteps)
        cfg.gamma = float(self.gamma)
        cfg.epsilon = float(self.epsilon)
        cfg.delta = float(self.delta)
        for i, v in enumerate(self._z0.tolist()):
            cfg.z0[i] = v
        h = C.c_void_p()
        torch.cuda.current_stream(zp[0].device).synchronize()
        with torch.cuda.device(zp[0].device):
            _lib.check(_lib.lib().drm_drmnet_create(hi, hr, _lib.ptr_array(zp), C.byref(cfg), C.byref(h)))
        if getattr(self, "_batch_parts", None) is not None:
            _lib.check(_lib.lib().drm_drmnet_set_batch_parts(h, int(self._batch_parts)))
        if getattr(self, "_batch_part_min", None) is not None:
            _lib.check(_lib.lib().drm_drmnet_set_batch_part_min(h, int(self._batch_part_min)))
        self._samplers[which] = (h, sig)
        return h

    def set_batch_parts(self, parts: int, min_rows: int = None):
        """Row ranges a reverse step is forked into on internal streams (drm_drmnet_set_batch_parts; library default 2 from 64 rows per part, 1 = off).
        min_rows: rows per part from which the fork engages (drm_drmnet_set_batch_part_min; tests pass 1)."""
        self._batch_parts = int(parts)
        if min_rows is not None:
            self._batch_part_min = int(min_rows)
        for h, _ in getattr(self, "_samplers", {}).values():
            _lib.check(_lib.lib().drm_drmnet_set_batch_parts(h, int(parts)))
            if min_rows is not None:
                _lib.check(_lib.lib().drm_drmnet_set_batch_part_min(h, int(min_rows)))
        return self

    def _free_sampler(self, which=None):
        for k in ([which] if which else list(getattr(self, "_samplers", {}))):
            entry = self._samplers.pop(k, None)
            if entry is not None:
                _lib.lib().drm_drmnet_destroy(entry[0])

    def __del__(self):
        try:
            self._free_sampler()
        except Exception:
            pass

    @staticmethod
    def _one_cond(cond, LrK):
        c = cond[0] if isinstance(cond, (list, tuple)) else cond
        if isinstance(cond, (list, tuple)) and len(cond) != 1:
            raise NotImplementedError("one concat conditioning tensor expected")
        return _lib.require_gpu_tensor(c, "cond")

    @torch.no_grad()
    def p_sample_loop(self, Lr_K, illnet_cond, refnet_cond, return_intermediates=False, verbose=True, log_every_k=None,
                      noise0=None, step_noise=None, seed=None, early_exit=True):
        """drmnet.py:782-847.  Extra keyword-only knobs (not in the reference): ``noise0`` [B,3,H,W] and ``step_noise``
        [max_timesteps,B,3,H,W] inject the random draws (parity mode; row b of step_noise[i] is used by sample b iff it is
        still active and not converged at step i); otherwise noise comes from the library's Philox stream keyed by ``seed``
        (drawn from torch's generator when None).  ``early_exit=False`` keeps every sample active for max_timesteps steps.
        Returns (Lr_0, zK, K[, intermediates]) exactly as the reference."""
        log_every_k = log_every_k or self.log_every_k
        LrK = _lib.require_gpu_tensor(Lr_K, "Lr_K")
        dev = LrK.device
        cond = self._one_cond(illnet_cond, LrK)
        cond_r = self._one_cond(refnet_cond, LrK)
        if cond_r.data_ptr() != cond.data_ptr() and not torch.equal(cond_r, cond):
            raise NotImplementedError("illnet_cond and refnet_cond are the same tensor on the shipped path (drmnet.py:1043)")
        B, _, H, W = LrK.shape
        if seed is None:
            seed = int(torch.randint(0, 2**62, (1,)).item())
        noise0 = None if noise0 is None else _lib.require_gpu_tensor(noise0, "noise0")
        step_noise = None if step_noise is None else _lib.require_gpu_tensor(step_noise, "step_noise")
        if step_noise is not None and tuple(step_noise.shape) != (self.max_timesteps, B, 3, H, W):
            raise RuntimeError("step_noise must be [max_timesteps, B, 3, H, W]")
        h = self._engine(LrK)
        L = _lib.lib()
        ws = self._ws.get(int(L.drm_drmnet_workspace_bytes(h, B, H, W)), dev)
        Lr0 = torch.empty_like(LrK)
        zK = torch.empty((B, self.zdim), dtype=torch.float32, device=dev)
        K = torch.empty((B,), dtype=torch.int32, device=dev)
        if not return_intermediates:
            steps = C.c_int32(0)
            with torch.cuda.device(dev):
                _lib.check(L.drm_drmnet_sample(h, LrK.data_ptr(), cond.data_ptr(), _lib.ptr(noise0), _lib.ptr(step_noise), seed, int(bool(early_exit)),
                                               Lr0.data_ptr(), zK.data_ptr(), K.data_ptr(), C.byref(steps), B, H, W, ws.data_ptr(), ws.numel(),
                                               _lib.stream_ptr(dev)))
            self.last_steps = int(steps.value)
            return Lr0, zK, K
        # intermediates requested: drive the loop from the host, one drm_drmnet_step per iteration (same kernels)
        from . import ops

        n0 = noise0 if noise0 is not None else ops.randn(LrK.shape, seed, 0, dev)
        Lr_k = LrK + self.delta * n0
        intermediates = {"Lrk_inter": [Lr_k.clone()], "zk_inter": []}
        zK.fill_(float("nan"))
        K.fill_(self.max_timesteps)
        active = torch.arange(B, dtype=torch.int32, device=dev)
        for i in range(self.max_timesteps):
            n = active.numel()
            zk = torch.empty((n, self.zdim), device=dev)
            zKc = torch.empty((n, self.zdim), device=dev)
            conv = torch.empty((n,), dtype=torch.int32, device=dev)
            nz = None if step_noise is None else step_noise[i]
            with torch.cuda.device(dev):
                _lib.check(L.drm_drmnet_step(h, Lr_k.data_ptr(), cond.data_ptr(), active.data_ptr(), n, i, _lib.ptr(nz), seed, zk.data_ptr(),
                                             zKc.data_ptr(), conv.data_ptr(), B, H, W, ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev)))
            if i % log_every_k == 0:
                z_log = torch.full((B, self.zdim), float("nan"), device=dev)
                z_log[active.long()] = zk
                intermediates["zk_inter"].append(z_log)
                Lr_log = torch.zeros_like(Lr_k)
                Lr_log[active.long()] = Lr_k[active.long()]
                intermediates["Lrk_inter"].append(Lr_log)
            if early_exit:
                cb = conv.bool()
                done = active[cb].long()
                K[done] = i + 1
                zK[done] = zKc[cb]
                active = active[~cb].contiguous()
                if active.numel() == 0:
                    break
        return Lr_k, zK, K, intermediates

    # ------------------------------------------------------------------ estimate.py glue
    @torch.no_grad()
    def get_input_for_predict(self, batch, bs: Optional[int] = None):
        """models/drmnet.py:1011-1045: the first ``bs`` refmaps are exposure-normalised (each scaled so the geometric mean of its
        luminance over lit pixels equals ``refmap_input_scaler``; the factors are kept in ``self.normalizing_scale`` for the way
        back, scripts/estimate.py:99-100), mapped to network space by ``ds.transform`` and returned with the conditioning lists
        of the two networks (the same tensor, optionally jittered by ``sigma_for_cond_xK``).  The luminance reduction and the
        scale + log map run as two HIP launches (csrc/transform.hip)."""
        from . import ops

        src = batch[self.input_key]
        n = len(src) if bs is None else min(len(src), bs)
        scaled = self.refmap_input_scaler is not None

        def to_network_space(x):
            x = _lib.require_gpu_tensor(x[:n], self.input_key)
            if scaled:
                x = ops.map_chain(x, [("img_mul", 0.0)], scale=self.normalizing_scale)
            return self.ds.transform(x)

        if scaled:
            self.normalizing_scale = ops.luminance_scale(_lib.require_gpu_tensor(src[:n], self.input_key), self.refmap_input_scaler)
        LrK = self.get_first_stage_encoding(self.encode_first_stage(to_network_space(src)))
        Lr0 = to_network_space(batch["Lr0"]) if batch.get("Lr0") is not None else None
        cond = LrK if self.sigma_for_cond_xK <= 0 else self.sigma_for_cond_xK * torch.randn_like(LrK) + LrK
        illnet_c = [cond]
        return LrK, Lr0, illnet_c, illnet_c, batch["tag"][:n]

    def r0toenvmap(self, r0: torch.Tensor, envshape: Optional[Tuple[int]] = None) -> torch.Tensor:
        """models/drmnet.py:931-941: r0 / basis_r0, warped from the mirror-ball parametrisation to a lat-long map, channels last
        ([B, H, W, 3]) -- one HIP gather kernel (division fused into the fetch)."""
        from . import ops

        if envshape is None:
            envshape = (self.image_size, self.image_size * 2)
        return ops.mirmap2envmap(r0, envshape, basis=self.basis_r0.to(r0.device), channels_last=True)
