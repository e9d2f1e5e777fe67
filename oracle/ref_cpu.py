"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A restatement, in plain PyTorch-CPU / numpy, of the reference's denoise-and-drag
hot path.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this file; the product (`ishapediting_amd/`) never does.

Pinned: every function here is checked against golden vectors produced by
running the reference's own code (tools/make_golden.py imports /root/reference
in the build container; fixtures under tests/golden/, test in
tests/test_oracle_golden.py).

Each function cites the reference file:line it follows (paths relative to the
reference root; gd = neural_field_diffusion/guided_diffusion).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# schedule tables (float64)  -- gd/gaussian_diffusion.py:18-42,118-169 ; gd/respace.py:6-59,71-85
# ----------------------------------------------------------------------------


def linear_betas(n: int) -> np.ndarray:
    scale = 1000 / n
    return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)


def space_timesteps(num_timesteps: int, section_counts) -> List[int]:
    """gd/respace.py:6-59."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):                      # :27-37: a fixed integer stride that gives exactly N steps
            want = int(section_counts[4:])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return sorted(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start = 0
    steps = []
    for i, cnt in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError(f"cannot divide section of {size} steps into {cnt}")
        stride = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            steps.append(start + round(cur))
            cur += stride
        start += size
    return sorted(set(steps))


class Tables:
    """Respaced diffusion constants. gd/respace.py:71-85 then gd/gaussian_diffusion.py:133-169."""

    def __init__(self, respacing: str, diffusion_steps: int = 1000):
        base_betas = linear_betas(diffusion_steps)
        base_ac = np.cumprod(1.0 - base_betas, axis=0)
        use = set(space_timesteps(diffusion_steps, respacing))
        last = 1.0
        betas, tmap = [], []
        for i, ac in enumerate(base_ac):
            if i in use:
                betas.append(1 - ac / last)
                last = ac
                tmap.append(i)
        betas = np.array(betas, dtype=np.float64)
        self.timestep_map = tmap
        self.betas = betas
        self.num_timesteps = len(betas)
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(
            np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        self.log_betas = np.log(betas)

    def f32(self, arr: np.ndarray, i: int) -> torch.Tensor:
        """gd/gaussian_diffusion.py:1035-1048: float64 table -> float32 scalar."""
        return torch.from_numpy(arr)[i].float()


# ----------------------------------------------------------------------------
# UNet  -- gd/unet.py, gd/nn.py
# ----------------------------------------------------------------------------


def timestep_embedding(t: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """gd/nn.py:102-120."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _gn(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """GroupNorm32: computed on x.float(), cast back. gd/nn.py:16-18,92-99."""
    return F.group_norm(x.float(), 32, w.float(), b.float(), eps=1e-5).type(x.dtype)


class UNetOracle:
    """Functional forward of UNetModel over a reference-keyed state_dict.

    `fp16=True` applies the reference's precision contract (gd/unet.py:618-624,
    gd/fp16_util.py:14-21): conv weights/biases of the torso in half, everything
    else fp32, activations of the torso in half.
    """

    def __init__(self, spec, sd: Dict[str, torch.Tensor], fp16: bool = False):
        from ishapediting_amd.unet_spec import is_torso_conv  # data description only (names), no arithmetic
        self.spec = spec
        self.cfg = spec.cfg
        self.fp16 = fp16
        self.dtype = torch.float16 if fp16 else torch.float32
        self.sd = {}
        for k, v in sd.items():
            v = v.detach().clone().float()
            if fp16 and is_torso_conv(k):
                v = v.half()
            self.sd[k] = v

    def p(self, name):
        return self.sd[name]

    # gd/unet.py:236-256
    def resblock(self, r, x, emb):
        P = r.path
        h = F.silu(_gn(x, self.p(f"{P}.in_layers.0.weight"), self.p(f"{P}.in_layers.0.bias")))
        if r.up:      # gd/unet.py:100-110 (nearest x2) applied to both branches, :237-242
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        elif r.down:  # gd/unet.py:129-140 (AvgPool2d 2,2)
            h = F.avg_pool2d(h, 2, 2)
            x = F.avg_pool2d(x, 2, 2)
        h = F.conv2d(h, self.p(f"{P}.in_layers.2.weight"), self.p(f"{P}.in_layers.2.bias"), padding=1)
        emb_out = F.linear(F.silu(emb), self.p(f"{P}.emb_layers.1.weight"),
                           self.p(f"{P}.emb_layers.1.bias")).type(h.dtype)[..., None, None]
        scale, shift = torch.chunk(emb_out, 2, dim=1)
        h = _gn(h, self.p(f"{P}.out_layers.0.weight"), self.p(f"{P}.out_layers.0.bias")) * (1 + scale) + shift
        h = F.silu(h)   # dropout inactive in eval (drag_utils.py:187,233)
        h = F.conv2d(h, self.p(f"{P}.out_layers.3.weight"), self.p(f"{P}.out_layers.3.bias"), padding=1)
        if r.cin != r.cout:
            x = F.conv2d(x, self.p(f"{P}.skip_connection.weight"), self.p(f"{P}.skip_connection.bias"))
        return x + h

    # gd/unet.py:299-305, 337-354 (legacy order: split heads, then q/k/v)
    def attention(self, a, x):
        P = a.path
        b, c, hh, ww = x.shape
        x = x.reshape(b, c, -1)
        qkv = F.conv1d(_gn(x, self.p(f"{P}.norm.weight"), self.p(f"{P}.norm.bias")),
                       self.p(f"{P}.qkv.weight"), self.p(f"{P}.qkv.bias"))
        bs, width, length = qkv.shape
        ch = width // (3 * a.heads)
        q, k, v = qkv.reshape(bs * a.heads, ch * 3, length).split(ch, dim=1)
        s = 1 / math.sqrt(math.sqrt(ch))
        w = torch.einsum("bct,bcs->bts", q * s, k * s)
        w = torch.softmax(w.float(), dim=-1).type(w.dtype)
        o = torch.einsum("bts,bcs->bct", w, v).reshape(bs, -1, length)
        o = F.conv1d(o, self.p(f"{P}.proj_out.weight"), self.p(f"{P}.proj_out.bias"))
        return (x + o).reshape(b, c, hh, ww)

    def block(self, blk, h, emb):
        for l in blk.layers:
            if l.kind == "conv":
                h = F.conv2d(h, self.p(f"{l.path}.weight"), self.p(f"{l.path}.bias"), padding=1)
            elif l.kind == "res":
                h = self.resblock(l, h, emb)
            else:
                h = self.attention(l, h)
        return h

    # gd/unet.py:634-671
    def forward(self, x: torch.Tensor, timesteps: torch.Tensor, feat_layer: int = -1, all_taps: bool = False):
        emb = timestep_embedding(timesteps, self.cfg.model_channels)
        emb = F.linear(emb, self.p("time_embed.0.weight"), self.p("time_embed.0.bias"))
        emb = F.linear(F.silu(emb), self.p("time_embed.2.weight"), self.p("time_embed.2.bias"))
        hs = []
        h = x.type(self.dtype)
        for blk in self.spec.input_blocks:
            h = self.block(blk, h, emb)
            hs.append(h)
        h = self.block(self.spec.middle_block, h, emb)
        inter = None
        taps = []
        for i, blk in enumerate(self.spec.output_blocks):
            h = torch.cat([h, hs.pop()], dim=1)
            h = self.block(blk, h, emb)
            if i == feat_layer:
                inter = h.clone()
            if all_taps:
                taps.append(h)
        h = h.type(x.dtype)
        h = F.silu(_gn(h, self.p("out.0.weight"), self.p("out.0.bias")))
        out = F.conv2d(h, self.p("out.2.weight"), self.p("out.2.bias"), padding=1)
        if all_taps:
            return out, taps
        if feat_layer < 0:
            return out
        return out, inter


# ----------------------------------------------------------------------------
# diffusion step  -- gd/gaussian_diffusion.py:232-338, 400-532 ; gd/respace.py:122-127
# ----------------------------------------------------------------------------


class DiffusionOracle:
    def __init__(self, tables: Tables):
        self.tb = tables

    def model_call(self, unet: UNetOracle, x, t: int, feat_layer: int):
        ts = torch.tensor([self.tb.timestep_map[t]] * x.shape[0])      # respace.py:122-127
        return unet.forward(x, ts, feat_layer=feat_layer)

    def mean_variance_from_output(self, model_output: torch.Tensor, x: torch.Tensor, t: int, clip_denoised=True):
        """gd/gaussian_diffusion.py:265-331 for LEARNED_RANGE + EPSILON."""
        tb = self.tb
        C = x.shape[1]
        eps, v = torch.split(model_output, C, dim=1)
        min_log = tb.f32(tb.posterior_log_variance_clipped, t)
        max_log = tb.f32(tb.log_betas, t)
        frac = (v + 1) / 2
        log_var = frac * max_log + (1 - frac) * min_log
        var = torch.exp(log_var)
        x0 = tb.f32(tb.sqrt_recip_alphas_cumprod, t) * x - tb.f32(tb.sqrt_recipm1_alphas_cumprod, t) * eps
        if clip_denoised:
            x0 = x0.clamp(-1, 1)
        mean = tb.f32(tb.posterior_mean_coef1, t) * x0 + tb.f32(tb.posterior_mean_coef2, t) * x
        return {"mean": mean, "variance": var, "log_variance": log_var, "pred_xstart": x0, "model_output": eps}

    def p_sample_guidance(self, unet, x, t: int, noise=None, variance=None, variance_noise=None,
                          clip_denoised=True, feat_layer=-1):
        """gd/gaussian_diffusion.py:446-510."""
        if feat_layer < 0:
            mo, inter = self.model_call(unet, x, t, -1), None
        else:
            mo, inter = self.model_call(unet, x, t, feat_layer)
        out = self.mean_variance_from_output(mo, x, t, clip_denoised)
        out["inter_feat"] = inter
        nonzero = 0.0 if t == 0 else 1.0
        if variance_noise is not None:
            return {"sample": out["mean"] + variance_noise, "inter_feat": inter, "variance": out["variance"]}
        noise = noise if noise is not None else torch.randn_like(x)
        var = out["variance"] if variance is None else variance
        sample = out["mean"] + nonzero * torch.sqrt(var) * noise
        return {"sample": sample, "pred_xstart": out["pred_xstart"], "inter_feat": inter,
                "model_output": out["model_output"], "noise": noise, "variance": var, "mean": out["mean"]}

    def p_sample(self, unet, x, t: int, noise, clip_denoised=True):
        """gd/gaussian_diffusion.py:400-444 (exp(0.5*logvar), the generate path)."""
        out = self.mean_variance_from_output(self.model_call(unet, x, t, -1), x, t, clip_denoised)
        nonzero = 0.0 if t == 0 else 1.0
        return {"sample": out["mean"] + nonzero * torch.exp(0.5 * out["log_variance"]) * noise,
                "pred_xstart": out["pred_xstart"]}

    def ddim_sample(self, unet, x, t: int, noise, eta: float = 0.0, clip_denoised=True):
        """gd/gaussian_diffusion.py:654-705 (eps re-derived from the clipped x0, :350-354; Equation 12)."""
        tb = self.tb
        out = self.mean_variance_from_output(self.model_call(unet, x, t, -1), x, t, clip_denoised)
        x0 = out["pred_xstart"]
        eps = (tb.f32(tb.sqrt_recip_alphas_cumprod, t) * x - x0) / tb.f32(tb.sqrt_recipm1_alphas_cumprod, t)
        ab, abp = tb.f32(tb.alphas_cumprod, t), tb.f32(tb.alphas_cumprod_prev, t)
        sigma = eta * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
        mean_pred = x0 * torch.sqrt(abp) + torch.sqrt(1 - abp - sigma ** 2) * eps
        nonzero = 0.0 if t == 0 else 1.0
        return {"sample": mean_pred + nonzero * sigma * noise, "pred_xstart": x0}

    def ddpm_inversion(self, unet, x0, steps: int, noises: List[torch.Tensor], clip_denoised=True, feat_layer=-1):
        """gd/gaussian_diffusion.py:512-532; `noises[i]` stands in for randn_like at :522."""
        tb = self.tb
        inter = [x0]
        x = x0
        for i in range(steps):
            cof = tb.f32(tb.alphas_cumprod, i) / tb.f32(tb.alphas_cumprod_prev, i)
            x = torch.sqrt(cof) * x + torch.sqrt(1 - cof) * noises[i]
            inter.append(x)
        img = inter[-1]
        feat, vn, var = [], [], []
        for i in range(steps - 1, -1, -1):
            o = self.p_sample_guidance(unet, img, i, noise=torch.zeros_like(img),
                                       clip_denoised=clip_denoised, feat_layer=feat_layer)
            var.append(o["variance"])
            feat.append(o["inter_feat"])
            vn.append(inter[i] - o["mean"])
            img = o["mean"] + vn[-1]
        return {"inter_feat": feat, "latent": inter[-1], "variance_noise": vn, "variance": var, "sample": img}


# ----------------------------------------------------------------------------
# drag: feature re-layout, lattice, loss  -- drag_utils.py:134-159, 302-398
# ----------------------------------------------------------------------------


def make_offsets(r: int) -> torch.Tensor:
    """drag_utils.py:134-138."""
    p = torch.arange(-r, r + 1)
    px, py, pz = torch.meshgrid(p, p, p, indexing="ij")
    return torch.stack([px.reshape(-1), py.reshape(-1), pz.reshape(-1)], dim=-1)


def resize_feat_align(feature: torch.Tensor) -> torch.Tensor:
    """drag_utils.py:141-159 with cat_var=True.  The nearest 'interpolate' over the channel axis
    (:146-151) picks source index floor(dst * src/dst_n)."""
    b, c2 = feature.shape[:2]
    assert c2 % 2 == 0 and b == 1
    c = c2 // 2
    mean, var = feature[:, :c], feature[:, c:]
    if c % 3:
        e = c - c % 3
        idx = torch.floor(torch.arange(e, dtype=torch.float32) * (c / e)).long()
        mean, var = mean[:, idx], var[:, idx]
    H, W = feature.shape[2:]
    return torch.cat((mean.reshape(3, -1, H, W), var.reshape(3, -1, H, W)), dim=1).float()


class DragSetup:
    """drag_utils.py:314-334: point lattices, three planar grids, and the per-plane complement
    texel lists used by the mask term."""

    def __init__(self, sources, targets, r1: int, voxel_size: float, width: int):
        src = torch.as_tensor(sources, dtype=torch.float32)
        tgt = torch.as_tensor(targets, dtype=torch.float32)
        off = make_offsets(r1)
        patch = src.unsqueeze(1) + voxel_size * off.unsqueeze(0)
        shift = tgt.unsqueeze(1) + voxel_size * off.unsqueeze(0)
        self.patch_grid = torch.stack((patch[..., :2], patch[..., 1:], patch[..., :3:2]), dim=0)   # 3,B,N1,2
        self.shift_grid = torch.stack((shift[..., :2], shift[..., 1:], shift[..., :3:2]), dim=0)
        pi = torch.round((patch + 1) * (width - 1) / 2).to(torch.int16).reshape(-1, 3)
        si = torch.round((shift + 1) * (width - 1) / 2).to(torch.int16).reshape(-1, 3)
        content = torch.cat((pi, si), dim=0).long()
        self.masks = []     # bool [W,W] per plane, True = texel NOT touched (indexed [row, col])
        for cols in ([1, 0], [2, 1], [2, 0]):
            touched = torch.zeros(width, width, dtype=torch.bool)
            rc = content[:, cols]
            ok = (rc[:, 0] >= 0) & (rc[:, 0] < width) & (rc[:, 1] >= 0) & (rc[:, 1] < width)
            touched[rc[ok, 0], rc[ok, 1]] = True
            self.masks.append(~touched)


def drag_loss(edit: torch.Tensor, orig: torch.Tensor, setup: DragSetup, cof: float, loss_type: str = "l2"):
    """drag_utils.py:355-382. `edit` may require grad."""
    patch = F.grid_sample(orig, setup.patch_grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    shift = F.grid_sample(edit, setup.shift_grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    C = orig.shape[1]
    nmask = sum(int(m.sum()) for m in setup.masks)
    if cof <= 0:
        mask_loss = 0.0
    else:
        tot = 0.0
        for p in range(3):
            d = (edit[p] - orig[p])[:, setup.masks[p]]
            tot = tot + (d.abs().sum() if loss_type == "l1" else (d ** 2).sum())
        mask_loss = tot / (C * nmask)
    if loss_type == "l1":
        return -F.l1_loss(shift, patch.detach()) - cof * mask_loss
    return -((shift.reshape(-1) - patch.detach().reshape(-1)) ** 2).mean() - cof * mask_loss


# ----------------------------------------------------------------------------
# triplane decoder  -- triplane_decoder/axisnetworks.py:78-90, 517-562 ; visualize.py:76-105
# ----------------------------------------------------------------------------


def decoder_forward(net: Dict[str, torch.Tensor], planes: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """planes [3,32,H,W] fp32; coords [N,3] -> logits [N]. Plane order xy, yz, xz (axisnetworks.py:549-551)."""
    c = coords.reshape(1, 1, -1, 3)
    f = 0
    for p, sl in enumerate((slice(0, 2), slice(1, 3), slice(0, 3, 2))):
        s = F.grid_sample(planes[p:p + 1], c[..., sl], mode="bilinear", padding_mode="zeros", align_corners=True)
        f = f + s.reshape(planes.shape[1], -1).t()
    y = 2 * np.pi * (f @ net["0._B"])
    y = torch.cat([torch.sin(y), torch.cos(y)], dim=-1)
    y = F.relu(F.linear(y, net["1.weight"], net["1.bias"]))
    y = F.relu(F.linear(y, net["3.weight"], net["3.bias"]))
    return F.linear(y, net["5.weight"], net["5.bias"]).reshape(-1)


def grid_coords(res: int) -> torch.Tensor:
    """visualize.py:79-86: linspace(-1,1,res)^3, 'ij' order, x slowest."""
    xx = torch.linspace(-1, 1, res)
    g = torch.meshgrid([xx, xx, xx], indexing="ij")
    return torch.stack(g, dim=-1).reshape(-1, 3)


def decode_volume(net, latent: torch.Tensor, rng, mid, res: int, chunk: int = 50000) -> torch.Tensor:
    """drag_utils.py:295-298 (un-normalise, reshape to 3 planes) + visualize.py:87-97 (chunked decode)."""
    S = latent.shape[-1]
    planes = (latent * rng + mid).reshape(3, 32, S, S)
    coords = grid_coords(res)
    out = torch.zeros(coords.shape[0])
    for h in range(0, coords.shape[0], chunk):
        out[h:h + chunk] = decoder_forward(net, planes, coords[h:h + chunk])
    return out.reshape(res, res, res)


# ----------------------------------------------------------------------------
# loops  -- drag_utils.py:252-280 (update_latent_params), 302-399 (training)
# ----------------------------------------------------------------------------


def sample_with_guidance_cache(diff: DiffusionOracle, unet: UNetOracle, img, num_steps, w_time, feat_layer, noises, progress=None):
    """drag_utils.py:266-277. noises[i] is the injected randn for loop index i.  progress(i): called once per step (long
    full-size runs print a line so that the job is seen to be alive)."""
    w = None
    cache = []
    with torch.no_grad():
        for i in range(num_steps - 1, -1, -1):
            o = diff.p_sample_guidance(unet, img, i, noise=noises[i], feat_layer=feat_layer)
            img = o["sample"]
            if i == w_time:
                w = img.clone()
            if i < w_time:
                cache.append(resize_feat_align(o["inter_feat"]))
            if progress:
                progress(i)
    return img, w, cache


def drag_loop(diff: DiffusionOracle, unet: UNetOracle, w, cache, setup: DragSetup, w_time, feat_layer, scale, cof,
              noises, loss_type="l2", progress=None):
    """drag_utils.py:336-398 (case 2: variance not fixed). Returns final latent and per-step losses."""
    img = w.clone().detach()
    losses = []
    for i in range(w_time - 1, -1, -1):
        img.requires_grad_(True)
        o = diff.p_sample_guidance(unet, img, i, noise=noises[i], feat_layer=feat_layer)
        edit = resize_feat_align(o["inter_feat"])
        loss = drag_loss(edit, cache[w_time - 1 - i], setup, cof, loss_type)
        g, = torch.autograd.grad(loss, img)
        losses.append(float(loss.detach()))
        img = (o["sample"] + o["variance"] * (scale * g)).detach()
        if progress:
            progress(i)
    return img, losses


def reconstruct_loop(diff: DiffusionOracle, unet: UNetOracle, net, img, rng, mid, coords, gts, noises, scale=600.0,
                     steps=None):
    """drag_utils.py:445-463 (train_triplane's guided loop) for given point batches: per step decode pred_xstart on the
    batch, loss = -BCEWithLogits, img <- sample + variance * scale * d loss / d img.  `steps` (loop indices, default the
    whole schedule T-1..0) lets shortened runs start mid-schedule; coords/gts/noises are indexed by position k."""
    T = diff.tb.num_timesteps
    imgs, losses, grads = [], [], []
    for k, i in enumerate(steps if steps is not None else range(T - 1, -1, -1)):
        img = img.detach().requires_grad_(True)
        o = diff.p_sample_guidance(unet, img, i, noise=noises[k])
        S = img.shape[-1]
        planes = (o["pred_xstart"] * rng + mid).reshape(3, 32, S, S)
        pred = decoder_forward(net, planes, coords[k])
        loss = -F.binary_cross_entropy_with_logits(pred, gts[k].reshape(-1))
        g, = torch.autograd.grad(loss, img)
        img = (o["sample"] + o["variance"] * (scale * g)).detach()
        imgs.append(img); losses.append(loss.detach()); grads.append(g)
    return imgs, losses, grads
