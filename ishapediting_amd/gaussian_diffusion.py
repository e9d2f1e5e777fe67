"""Host-side mirror of guided_diffusion/{gaussian_diffusion,respace}.py for the sampling path.

Schedule tables are float64 numpy like the reference's (gaussian_diffusion.py:118-169, respace.py:71-85);
each per-step lookup is cast to fp32 exactly where `_extract_into_tensor` does (:1035-1048).  The step
arithmetic itself (p_mean_variance :232-331, sample variants :443/:498-510) is one HIP kernel
(`ishap_ddpm_step`); the model call goes to the HIP UNet.  DDIM sampling (:654-847; optional faster sampler,
SURVEY 8f rank 4) is mode 3 of the same kernel.  Training losses and the VLB are not on the path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib

_STEP_RNG = os.environ.get("ISHAP_STEP_RNG", "1") != "0"      # the step kernel draws its own noise (GaussianDiffusion._draw)


def get_named_beta_schedule(schedule_name: str, num_diffusion_timesteps: int) -> np.ndarray:
    if schedule_name != "linear":
        raise NotImplementedError(f"unknown beta schedule: {schedule_name}")
    scale = 1000 / num_diffusion_timesteps
    return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)


def space_timesteps(num_timesteps: int, section_counts) -> set:
    """respace.py:6-59."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired = int(section_counts[len("ddim"):])
            for i in range(1, num_timesteps):
                if len(range(0, num_timesteps, i)) == desired:
                    return set(range(0, num_timesteps, i))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = divmod(num_timesteps, len(section_counts))
    start_idx, all_steps = 0, []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        frac = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            all_steps.append(start_idx + round(cur))
            cur += frac
        start_idx += size
    return set(all_steps)


class SpacedDiffusion:
    """LEARNED_RANGE variance + EPSILON mean (learn_sigma=True, predict_xstart=False: drag_utils.py:52-56)."""

    def __init__(self, use_timesteps, betas: np.ndarray, rescale_timesteps: bool = False):
        if rescale_timesteps:
            raise NotImplementedError("rescale_timesteps=False on the path (drag_utils.py:56)")
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(betas)
        base_ac = np.cumprod(1.0 - np.array(betas, dtype=np.float64), axis=0)
        last, new_betas, self.timestep_map = 1.0, [], []
        for i, ac in enumerate(base_ac):
            if i in self.use_timesteps:
                new_betas.append(1 - ac / last)
                last = ac
                self.timestep_map.append(i)
        betas = np.array(new_betas, dtype=np.float64)
        assert (betas > 0).all() and (betas <= 1).all()
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        self._log_betas = np.log(betas)

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _t_index(t) -> int:
        if torch.is_tensor(t):
            v = t.reshape(-1).tolist()
            assert all(x == v[0] for x in v), "one timestep per call (batch shares t), as every caller on the path does"
            return int(v[0])
        if isinstance(t, (list, tuple)):
            return int(t[0])
        return int(t)

    def _coefs(self, t: int, clip_denoised: bool, mode: int, eta: float = 0.0) -> _lib.StepCoefs:
        f32 = lambda a: float(np.float32(a[t]))                    # noqa: E731  (float64 table -> fp32, :1045)
        da = db = ds = 0.0
        if mode == 3:
            # ddim_sample (:688-699): the reference forms these from fp32 extracts with fp32 tensor ops, in this order
            ab, abp = np.float32(self.alphas_cumprod[t]), np.float32(self.alphas_cumprod_prev[t])
            one = np.float32(1)
            sigma = np.float32(eta) * np.sqrt((one - abp) / (one - ab)) * np.sqrt(one - ab / abp)
            da, db, ds = float(np.sqrt(abp)), float(np.sqrt(one - abp - sigma * sigma)), float(sigma)
        return _lib.StepCoefs(f32(self.posterior_log_variance_clipped), f32(self._log_betas),
                              f32(self.sqrt_recip_alphas_cumprod), f32(self.sqrt_recipm1_alphas_cumprod),
                              f32(self.posterior_mean_coef1), f32(self.posterior_mean_coef2),
                              0.0 if t == 0 else 1.0, int(bool(clip_denoised)), mode, da, db, ds)

    def prepare(self, model, indices, limit: int = 1024):
        """Hand the loop's timesteps to the model ahead of time (UNetModel.prepare_timesteps): the timestep-embedding
        products do not depend on x, so they are computed once per loop instead of once per step.  Purely an
        optimisation -- models without the method, or loops longer than `limit`, run as before.  The default covers the
        1000-step generate loop (164 KB per timestep for the full model: 164 MB).  """
        prep = getattr(model, "prepare_timesteps", None)
        idx = [int(t) for t in indices]
        if prep is not None and 0 < len(idx) <= limit:
            prep([self.timestep_map[t] for t in idx])

    def _model(self, model, x, t: int, feat_layer: int, **model_kwargs):
        ts = torch.tensor([self.timestep_map[t]] * x.shape[0])      # _WrappedModel, respace.py:122-127
        if feat_layer < 0:
            return model(x, ts, **model_kwargs), None
        return model(x, ts, feat_layer=feat_layer, **model_kwargs)

    def _draw(self, x, want_tensor):
        """The noise of one step when the caller injected none (the reference's th.randn_like(x), gaussian_diffusion.py:443 /
        :493 / :702).  Default: the step kernel draws it itself (include/ishap.h, ishap_step_coefs::rng -- Philox4x32-10 keyed by
        the device's default torch generator seed, counter offset taken from and advanced on that generator, so torch.manual_seed
        makes runs repeatable as with randn_like; the VALUES are not those torch.randn would give, as they are not the reference's
        on another device either) -- one launch and two 6.3 MB tensor passes fewer per step.  Returns (noise tensor or None,
        (seed, offset, noise_out tensor or None) or None).  ISHAP_STEP_RNG=0: torch.randn_like."""
        if _STEP_RNG and x.is_cuda:
            g = torch.cuda.default_generators[x.device.index]
            if hasattr(g, "get_offset"):
                seed, off = int(g.initial_seed()) & ((1 << 64) - 1), int(g.get_offset())
                g.set_offset(off + 4)
                return None, (seed, off, torch.empty_like(x) if want_tensor else None)
        return torch.randn_like(x), None

    def _step(self, x, model_output, t, noise, variance_in, clip_denoised, mode, want=("sample",), eta=0.0, guide=None, rng=None):
        N, Cc = x.shape[:2]
        HW = int(np.prod(x.shape[2:]))
        assert model_output.shape[1] == 2 * Cc
        outs = {k: torch.empty_like(x) for k in want}
        k = self._coefs(t, clip_denoised, mode, eta)
        if rng is not None:                   # (seed, offset, noise_out): the kernel draws the noise (noise must be None)
            assert noise is None
            k.rng, k.rng_seed, k.rng_offset = 1, rng[0], rng[1]
            k.noise_out = None if rng[2] is None else rng[2].data_ptr()
        if guide is not None:                 # (d loss / d x, scale): the guided update in the same pass
            grad, scale = guide
            assert grad.shape == x.shape and grad.dtype == torch.float32 and grad.is_contiguous()
            outs["guided"] = torch.empty_like(x)
            with torch.cuda.device(x.device):
                _lib.check(_lib.lib().ishap_ddpm_step_guided(
                    x.data_ptr(), model_output.data_ptr(), _lib.ptr(noise), _lib.ptr(variance_in), C.byref(k), N, Cc, HW,
                    grad.data_ptr(), scale, None, outs["guided"].data_ptr(), _lib.ptr(outs.get("sample")),
                    _lib.ptr(outs.get("variance")), _lib.stream_ptr(x.device)))
            return outs
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().ishap_ddpm_step(
                x.data_ptr(), model_output.data_ptr(), _lib.ptr(noise), _lib.ptr(variance_in), C.byref(k), N, Cc, HW,
                _lib.ptr(outs.get("sample")), _lib.ptr(outs.get("pred_xstart")), _lib.ptr(outs.get("variance")),
                _lib.ptr(outs.get("mean")), _lib.stream_ptr(x.device)))
        return outs

    @staticmethod
    def _prep(x):
        return x.detach().to(dtype=torch.float32).contiguous()

    def _guidance_step_autograd(self, model, x, ti, noise, variance, variance_noise, clip_denoised, feat_layer):
        """p_sample_guidance with the graph kept (gaussian_diffusion.py:232-331 for LEARNED_RANGE + EPSILON, :489-510): what
        the reference's autograd records when `x.requires_grad` (synthesize_latent(calc_grad=True), drag_utils.py:86-87).
        The model call is differentiable.UNetCall (re-forward + the library's input-gradient passes); the step arithmetic
        is a handful of elementwise torch ops on the device -- the same fp32 formulas ishap_ddpm_step evaluates."""
        from .differentiable import unet_call
        f32 = lambda a: float(np.float32(a[ti]))                   # noqa: E731
        Cc = x.shape[1]
        mo, inter = unet_call(model, x, [self.timestep_map[ti]] * x.shape[0], feat_layer)
        eps, v = torch.split(mo, Cc, dim=1)
        frac = (v + 1) / 2
        var = torch.exp(frac * f32(self._log_betas) + (1 - frac) * f32(self.posterior_log_variance_clipped))
        x0 = f32(self.sqrt_recip_alphas_cumprod) * x - f32(self.sqrt_recipm1_alphas_cumprod) * eps
        if clip_denoised:
            x0 = x0.clamp(-1, 1)
        mean = f32(self.posterior_mean_coef1) * x0 + f32(self.posterior_mean_coef2) * x
        if variance_noise is not None:
            return {"sample": mean + variance_noise, "inter_feat": inter, "variance": var}
        noise = noise if noise is not None else torch.randn_like(x)
        used = var if variance is None else variance
        sample = mean + (0.0 if ti == 0 else 1.0) * torch.sqrt(used) * noise
        return {"sample": sample, "pred_xstart": x0, "inter_feat": inter, "model_output": eps, "noise": noise,
                "variance": used, "mean": mean}

    # ------------------------------------------------------------------ reference surface
    def p_sample_guidance(self, model, x, t, noise=None, variance=None, variance_noise=None, clip_denoised=True,
                          denoised_fn=None, cond_fn=None, model_kwargs=None, feat_layer=-1, keep_for_backward=False,
                          want_inter_feat=True, between=None, overlap=True, guided_scale=None, want_noise=True):
        """gaussian_diffusion.py:446-510.  Returns the same dict keys.
        `between`: a callable run after the model call and before the step arithmetic -- the drag loop passes its loss +
        backward here; with `overlap` the model then runs the part of the network those do not need (everything after the tap)
        beside them, and the step arithmetic waits for it.  Results are identical with and without it.
        `guided_scale` (with a `between` that returns d loss / d x): the drag loop's update `sample + variance * scale * grad`
        (drag_utils.py:384-392) is formed by the step kernel itself and returned as "guided" -- one launch instead of two, the
        intermediate sample / variance tensors are not written.
        `want_noise=False`: the dict's "noise" may be None when the step drew the noise itself (`_draw`); the drag loop does
        not read it."""
        assert denoised_fn is None and cond_fn is None, "not used on the path"
        ti = self._t_index(t)
        if torch.is_tensor(x) and x.requires_grad and torch.is_grad_enabled() and hasattr(model, "backward_from_output"):
            assert not model_kwargs and between is None
            return self._guidance_step_autograd(model, x.to(torch.float32), ti, noise, variance, variance_noise, clip_denoised,
                                                feat_layer)
        x = self._prep(x)
        kw = dict(model_kwargs or {})
        if hasattr(model, "tap_ptr"):
            kw.update(keep_for_backward=keep_for_backward, want_inter_feat=want_inter_feat)
            if between is not None and overlap and feat_layer >= 0 and hasattr(model, "join_tail"):
                kw.update(overlap_tail=True)
        mo, inter = self._model(model, x, ti, feat_layer, **kw)
        grad = None
        if between is not None:
            try:
                grad = between()
            finally:
                # also when `between` raises: the planned tail writes `model_output` through the pointer the forward was given --
                # it must be enqueued and joined while that tensor is still alive (include/ishap.h, ishap_unet_forward)
                if kw.get("overlap_tail"):
                    if hasattr(model, "run_tail"):
                        model.run_tail()      # the tail the forward only planned
                    model.join_tail()
        if guided_scale is not None and grad is not None and variance_noise is None:
            rng = None
            if noise is None:
                noise, rng = self._draw(x, want_noise)
            noise = None if noise is None else self._prep(noise)
            vin = None if variance is None else self._prep(variance)
            o = self._step(x, mo, ti, noise, vin, clip_denoised, 0, (), guide=(grad, float(guided_scale)), rng=rng)
            return {"guided": o["guided"], "inter_feat": inter, "noise": noise if rng is None else rng[2]}
        if variance_noise is not None:
            o = self._step(x, mo, ti, self._prep(variance_noise), None, clip_denoised, 2, ("sample", "variance"))
            return {"sample": o["sample"], "inter_feat": inter, "variance": o["variance"]}
        rng = None
        if noise is None:
            noise, rng = self._draw(x, want_noise)
        noise = None if noise is None else self._prep(noise)
        vin = None if variance is None else self._prep(variance)
        o = self._step(x, mo, ti, noise, vin, clip_denoised, 0, ("sample", "pred_xstart", "variance", "mean"), rng=rng)
        return {"sample": o["sample"], "pred_xstart": o["pred_xstart"], "inter_feat": inter,
                "model_output": mo[:, :x.shape[1]], "noise": noise if rng is None else rng[2],
                "variance": o["variance"] if variance is None else variance, "mean": o["mean"]}

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, noise=None):
        """gaussian_diffusion.py:400-444 (`noise` injectable for parity runs; default randn_like as there)."""
        assert denoised_fn is None and cond_fn is None
        ti = self._t_index(t)
        x = self._prep(x)
        mo, _ = self._model(model, x, ti, -1, **(model_kwargs or {}))
        rng = None
        if noise is None:
            noise, rng = self._draw(x, False)
        noise = None if noise is None else self._prep(noise)
        o = self._step(x, mo, ti, noise, None, clip_denoised, 1, ("sample", "pred_xstart"), rng=rng)
        return {"sample": o["sample"], "pred_xstart": o["pred_xstart"]}

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=False, step_noise=None):
        """gaussian_diffusion.py:604-652."""
        if device is None:
            device = next(model.parameters()).device
        img = noise if noise is not None else torch.randn(*shape, device=device)
        indices = list(range(self.num_timesteps))[::-1]
        self.prepare(model, indices)
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        for i in indices:
            out = self.p_sample(model, img, i, clip_denoised=clip_denoised, model_kwargs=model_kwargs,
                                noise=None if step_noise is None else step_noise(i))
            yield out
            img = out["sample"]

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, progress=False, save_intermediate=False,
                      save_timestep_interval=20, step_noise=None):
        """gaussian_diffusion.py:534-602 (save_intermediate is a debugging aid there; not provided)."""
        assert not save_intermediate
        final = None
        for sample in self.p_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised,
                                                     model_kwargs=model_kwargs, device=device, progress=progress,
                                                     step_noise=step_noise):
            final = sample
        return final["sample"]

    # ------------------------------------------------------------------ DDIM (gaussian_diffusion.py:654-847)
    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0,
                    noise=None, feat_layer=-1, **kwargs):
        """gaussian_diffusion.py:654-705.  `noise` injectable for parity runs (default randn_like as there)."""
        assert denoised_fn is None and cond_fn is None
        ti = self._t_index(t)
        x = self._prep(x)
        mo, inter = self._model(model, x, ti, feat_layer, **(model_kwargs or {}))
        rng = None
        if noise is None:
            noise, rng = self._draw(x, False)
        noise = None if noise is None else self._prep(noise)
        o = self._step(x, mo, ti, noise, None, clip_denoised, 3, ("sample", "pred_xstart"), eta=eta, rng=rng)
        return {"sample": o["sample"], "pred_xstart": o["pred_xstart"], "inter_feat": inter,
                "model_output": mo[:, :x.shape[1]]}

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                     model_kwargs=None, device=None, progress=False, eta=0.0, step_noise=None):
        """gaussian_diffusion.py:797-846."""
        if device is None:
            device = next(model.parameters()).device
        img = noise if noise is not None else torch.randn(*shape, device=device)
        self.prepare(model, range(self.num_timesteps))
        for i in list(range(self.num_timesteps))[::-1]:
            out = self.ddim_sample(model, img, i, clip_denoised=clip_denoised, model_kwargs=model_kwargs, eta=eta,
                                   noise=None if step_noise is None else step_noise(i))
            yield out
            img = out["sample"]

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                         model_kwargs=None, device=None, progress=False, eta=0.0, step_noise=None, **_ignored):
        """gaussian_diffusion.py:762-795."""
        final = None
        for sample in self.ddim_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised,
                                                        model_kwargs=model_kwargs, device=device, progress=progress,
                                                        eta=eta, step_noise=step_noise):
            final = sample
        return final["sample"]

    def ddpm_inversion(self, model, x_0, steps, fwd_noise: Optional[Sequence[torch.Tensor]] = None, tap_sink=None,
                       **kwargs):
        """gaussian_diffusion.py:512-532.  `fwd_noise[i]` replaces randn_like at :522 for parity runs; `tap_sink()`
        is called after every reverse step so the caller can keep the resident tap (guidance cache)."""
        feat, variance_noise, variance = [], [], []
        x = self._prep(x_0)
        L = _lib.lib()
        self.prepare(model, range(steps))
        img_inter = [x]
        for i in range(steps):
            cof = np.float32(self.alphas_cumprod[i]) / np.float32(self.alphas_cumprod_prev[i])      # fp32 / fp32, :520
            a, b = float(np.sqrt(np.float32(cof))), float(np.sqrt(np.float32(1) - cof))
            eps = self._prep(fwd_noise[i]) if fwd_noise is not None else torch.randn_like(x)
            nxt = torch.empty_like(x)
            with torch.cuda.device(x.device):
                _lib.check(L.ishap_axpby(x.data_ptr(), eps.data_ptr(), a, b, x.numel(), nxt.data_ptr(),
                                         _lib.stream_ptr(x.device)))
            x = nxt
            img_inter.append(x)
        img = img_inter[-1]
        zero = torch.zeros_like(img)
        for i in range(steps - 1, -1, -1):
            outs = self.p_sample_guidance(model, img, i, noise=zero, **kwargs)
            if tap_sink is not None:
                tap_sink()
            variance.append(outs["variance"])
            feat.append(outs["inter_feat"])
            # z_i = x_i - mean_i and img = mean_i + z_i (:527-529) through the library's elementwise kernel (a, b = +-1 are
            # exact, so these are the reference's fp32 subtraction / addition): no torch arithmetic on the path
            z, nxt = torch.empty_like(img), torch.empty_like(img)
            with torch.cuda.device(img.device):
                s_ = _lib.stream_ptr(img.device)
                _lib.check(L.ishap_axpby(img_inter[i].data_ptr(), outs["mean"].data_ptr(), 1.0, -1.0, img.numel(), z.data_ptr(), s_))
                _lib.check(L.ishap_axpby(outs["mean"].data_ptr(), z.data_ptr(), 1.0, 1.0, img.numel(), nxt.data_ptr(), s_))
            variance_noise.append(z)
            img = nxt
        return {"inter_feat": feat, "latent": img_inter[-1], "variance_noise": variance_noise, "variance": variance,
                "sample": img}


def create_gaussian_diffusion(*, steps=1000, learn_sigma=True, sigma_small=False, noise_schedule="linear", use_kl=False,
                              predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False,
                              timestep_respacing=""):
    """script_util.py:389-427."""
    if not learn_sigma or predict_xstart or use_kl:
        raise NotImplementedError("the path uses learn_sigma=True, predict_xstart=False, MSE (drag_utils.py:52-56)")
    betas = get_named_beta_schedule(noise_schedule, steps)
    if not timestep_respacing:
        timestep_respacing = [steps]
    return SpacedDiffusion(space_timesteps(steps, timestep_respacing), betas, rescale_timesteps)
