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
