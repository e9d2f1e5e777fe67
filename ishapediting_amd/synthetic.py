"""Deterministic synthetic weights and inputs (no checkpoints exist offline).

SURVEY.md §8(d): every conv/linear weight ~ N(0, 0.02) *including* the tensors
the reference zero-initialises (nn.py:67-73 -> unet.py:210-212,294,615), so no
block is dead; biases ~ N(0, 0.01); GroupNorm weight ~ 1 + N(0, 0.05),
bias ~ N(0, 0.05) (not the default 1/0, so the affine is exercised).
Values are drawn in key order from a CPU `torch.Generator`, so the same seed
gives the same state_dict here, in the golden generator and on the GPU box.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

from .unet_spec import DECODER_SHAPES, UNetConfig, is_torso_conv, param_shapes


def unet_state_dict(cfg: UNetConfig, seed: int = 1234, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """fp32 state_dict with the reference's key names.  `gain` scales the
    conv/linear std (fan-in aware: std = gain / sqrt(fan_in)), which keeps
    activations O(1) through the full-depth net so fp16 stays in range."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        is_norm = (".in_layers.0." in name or ".out_layers.0." in name or ".norm." in name
                   or name.startswith("out.0."))
        if is_norm:
            base = 1.0 if name.endswith("weight") else 0.0
            t = base + 0.05 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            t = 0.01 * torch.randn(shape, generator=g)
        else:
            fan_in = int(np.prod(shape[1:]))
            t = (gain / math.sqrt(fan_in)) * torch.randn(shape, generator=g)
        sd[name] = t.float()
    return sd


def unet_state_dict_offset(cfg: UNetConfig, seed: int, offset: float = 6.0, gain: float = 0.25) -> Dict[str, torch.Tensor]:
    """Like unet_state_dict, but every torso conv bias is (a per-GroupNorm-group constant of magnitude ~`offset`, random
    sign) + N(0, 0.01), and the conv weights are small (`gain`): the activations a GroupNorm reads then sit at a group
    mean tens of times their spread -- the regime where a one-pass fp32 variance cancels (nn.py:16-18 works in fp32 on
    x.float()).  Groups follow GroupNorm32's split of the conv's OUTPUT channels into 32 groups."""
    sd = unet_state_dict(cfg, seed, gain)
    g = torch.Generator().manual_seed(seed + 7)
    for name, t in sd.items():
        if not (name.endswith(".bias") and is_torso_conv(name)) or ".qkv." in name:
            continue            # qkv feeds the softmax, not a GroupNorm: its bias stays small
        C = t.shape[0]
        if C % 32:
            continue
        per_group = (torch.rand(32, generator=g) * 0.5 + 0.75) * offset * torch.sign(torch.randn(32, generator=g))
        sd[name] = (per_group.repeat_interleave(C // 32) + 0.01 * torch.randn(C, generator=g)).float()
    return sd


def round_torso_to_fp16(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Values the fp16 torso would hold (fp16_util.py:14-21), kept as fp32
    storage so an fp32 oracle and the fp16 device path share exact weights."""
    out = {}
    for k, v in sd.items():
        out[k] = v.half().float() if is_torso_conv(k) else v
    return out


def decoder_state_dict(seed: int = 4321) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in DECODER_SHAPES.items():
        if name == "0._B":
            t = torch.randn(shape, generator=g)            # scale=1, axisnetworks.py:527
        elif name.endswith("bias"):
            t = 0.1 * torch.randn(shape, generator=g)
        else:
            t = torch.randn(shape, generator=g) / math.sqrt(shape[1])
        sd[name] = t.float()
    return sd


def latent(k: int, channels: int = 96, size: int = 128) -> np.ndarray:
    """main.py:282-285: np.random.seed(latent id); randn(1,96,S,S)."""
    rs = np.random.RandomState(k)
    return rs.randn(1, channels, size, size).astype(np.float32)


def step_noise(seed: int, shape) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g)


def handles(b: int = 3, seed: int = 7, step: float = 0.15):
    rs = np.random.RandomState(seed)
    src = rs.uniform(-0.5, 0.5, size=(b, 3)).astype(np.float32)
    d = rs.randn(b, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tgt = (src + step * d).astype(np.float32)
    return src, tgt


# BASELINE configs[3] at FULL length (drag_utils.py:401-471, :552-566): 200 reconstruction steps x 40 000 occupancy samples,
# DDPM inversion over w_time = 170, 170 guided drag iterations, 256^3 decode
C4_T, C4_W, C4_RES, C4_POINTS = 200, 170, 256, 40000


def c4_inputs(T: int = C4_T, W: int = C4_W, points: int = C4_POINTS, seed: int = 2024):
    """Seeded inputs of the full-length C4 chain, identical for the device run, the oracle run and the committed fixture
    (tools/make_c4_fixture.py).  The 'real shape' is the union of four ellipsoids bench.py's C4 leg uses as its synthetic
    airplane, sampled analytically (occupancy = inside any ellipsoid).  Returns (img0, batch(k), noise(tag, k))."""
    gen = torch.Generator().manual_seed(seed)
    img0 = torch.randn(1, 96, 128, 128, generator=gen)
    centres = torch.tensor([[0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [-0.55, 0.0, 0.12], [0.1, 0.0, 0.0]])
    radii = torch.tensor([[0.75, 0.10, 0.10], [0.12, 0.62, 0.03], [0.10, 0.22, 0.03], [0.10, 0.03, 0.20]])

    def batch(k):                       # a fresh batch per step from its own seed (DataLoader(shuffle=True), drag_utils.py:453)
        g = torch.Generator().manual_seed(seed * 1000 + k)
        c = torch.rand(points, 3, generator=g) * 2 - 1
        inside = (((c[:, None, :] - centres[None]) / radii[None]).pow(2).sum(-1) < 1).any(dim=1)
        return c, inside.float()

    def noise(tag, k):                  # one 6.3 MB tensor at a time (three chains x 170-200 steps would be 3.4 GB resident)
        return torch.randn(1, 96, 128, 128, generator=torch.Generator().manual_seed(seed * 7919 + tag * 1000 + k))
    return img0, batch, noise
