"""Autograd bridge for the one place the reference keeps a graph through the sampler: synthesize_latent(calc_grad=True)
(drag_utils.py:61-131, no caller in the reference; `img.requires_grad_(True)` at :86-87 makes every later
p_sample_guidance call differentiable w.r.t. the latent).

The engine has no tape: a context keeps the intermediates of ONE forward.  `UNetCall` therefore saves only (x, t) and, in
backward, re-runs that forward with keep_for_backward=1 and calls the library's two input-gradient entry points
(ishap_unet_backward_from_output for the cotangent of the model output, ishap_unet_backward_input for the cotangent of the
tap; the gradient is linear in them, so the two results add).  Cotangents travel as fp16 like the reference's autograd
tensors under its fp16 torso, carried with a power-of-two loss scale that the library removes
again.  Weight gradients are not produced (the reference never reads them on this path: the model is in eval mode and only
`img` is a leaf of interest)."""
from __future__ import annotations

import math

import torch


def _pow2_scale(t: torch.Tensor, target: float = 1024.0) -> float:
    """Largest power of two that keeps max|t| * scale <= target (fp16 cotangents: well inside 65504, far above subnormals)."""
    m = float(t.detach().abs().max())
    if not math.isfinite(m) or m == 0.0:
        return 1.0
    return float(2.0 ** math.floor(math.log2(target / m)))


class UNetCall(torch.autograd.Function):
    """model(x, timesteps, feat_layer=...) -> (out fp32 [N,2C,S,S], tap fp16 [N,Ct,St,St]) with d/dx."""

    @staticmethod
    def forward(ctx, x, model, timesteps, feat_layer):
        ctx.model, ctx.ts, ctx.feat_layer = model, [float(t) for t in timesteps], int(feat_layer)
        xd = x.detach().to(dtype=torch.float32).contiguous()
        ctx.save_for_backward(xd)
        if feat_layer >= 0:
            out, tap = model.forward(xd, ctx.ts, feat_layer=feat_layer, keep_for_backward=False)
            return out, tap
        out = model.forward(xd, ctx.ts, feat_layer=-1, keep_for_backward=False)
        empty = out.new_empty(0, dtype=torch.float16)
        ctx.mark_non_differentiable(empty)
        return out, empty

    @staticmethod
    def backward(ctx, d_out, d_tap):
        (x,) = ctx.saved_tensors
        m, fl = ctx.model, ctx.feat_layer
        dev = x.device
        use_out = d_out is not None and bool((d_out != 0).any())
        use_tap = fl >= 0 and d_tap is not None and d_tap.numel() > 0 and bool((d_tap != 0).any())
        dx = torch.zeros_like(x)
        if not (use_out or use_tap):
            return dx, None, None, None
        m.forward(x, ctx.ts, feat_layer=fl, keep_for_backward=True, want_inter_feat=False)     # the kept forward to differentiate
        if use_out:
            s = _pow2_scale(d_out)
            scale2 = torch.tensor([s, 1.0 / s], dtype=torch.float32, device=dev)
            dx += m.backward_from_output(d_out.float() * s, scale2)
        if use_tap:
            s = _pow2_scale(d_tap)
            scale2 = torch.tensor([s, 1.0 / s], dtype=torch.float32, device=dev)
            N, Ct, St, _ = d_tap.shape
            cot = (d_tap.float() * s).permute(0, 2, 3, 1).reshape(N, St * St, Ct).to(torch.float16).contiguous()
            dx += m.backward_input(cot, scale2)
        return dx, None, None, None


def unet_call(model, x, timesteps, feat_layer: int = -1):
    """Differentiable model call; returns (out, tap) -- tap is None without a feat_layer."""
    out, tap = UNetCall.apply(x, model, timesteps, feat_layer)
    return out, (tap if feat_layer >= 0 else None)
