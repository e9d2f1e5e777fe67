"""Helpers shared by the CPU and GPU test files (no arithmetic of the path lives here)."""
from argparse import Namespace

import numpy as np
import torch

from ishapediting_amd.unet_spec import UNetConfig


def small96_config() -> UNetConfig:
    """The 96-channel-latent small model of golden G11 / G12 (tools/make_golden.py:small96_config)."""
    return UNetConfig(image_size=16, in_channels=96, model_channels=32, out_channels=192, num_res_blocks=1,
                      attention_resolutions="8", channel_mult=(1, 2), num_head_channels=32)


def small96_args(Tn: int, batch: int = 1, **over) -> Namespace:
    """The Namespace drag_utils.get_args / generate.py build, shrunk to small96_config."""
    ns = Namespace(clip_denoised=True, num_samples=batch, batch_size=batch, use_ddim=False, model_path=None,
                   stats_dir=None, explicit_normalization=True, save_dir=None, num_steps=Tn, image_size=16,
                   num_channels=32, num_res_blocks=1, num_heads=4, num_heads_upsample=-1, num_head_channels=32,
                   attention_resolutions="8", channel_mult="1,2", dropout=0.1, class_cond=False, shape_resolution=32,
                   use_checkpoint=False, use_scale_shift_norm=True, resblock_updown=True, use_fp16=True,
                   use_new_attention_order=False, in_out_channels=96, learn_sigma=True, diffusion_steps=1000,
                   noise_schedule="linear", timestep_respacing=str(Tn), w_time=2, feat_layer=1, loss_type="l2",
                   use_kl=False, predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False,
                   points_size=20000, points_uniform_ratio=0.5, decoder_ckpt=None)
    for k, v in over.items():
        setattr(ns, k, v)
    return ns


def redraw_generate_noise(g, B: int):
    """Golden G12 stores seeds, not noise: redraw the reference's RNG stream (th.randn(*shape), then one randn per
    step, gaussian_diffusion.py:629,437) on the CPU generator and check it against the stored float64 checksums."""
    Tn = int(g["T"])
    shape = (B, 96, 16, 16)
    torch.manual_seed(int(g[f"b{B}_seed"]))
    init = torch.randn(*shape)
    steps = torch.stack([torch.randn(*shape) for _ in range(Tn)])
    chk = g[f"b{B}_noise_check"]
    got = np.array([float(init.double().sum()), float(steps.double().pow(2).sum()), float(steps[-1, -1, -1, -1, -1])])
    np.testing.assert_allclose(got, chk, rtol=1e-12, atol=0, err_msg="torch's CPU RNG stream differs from the one the fixture was made with")
    return init, steps


def redraw_ddim_noise(g):
    """Golden G13's RNG stream: th.randn(*shape) then one randn per step under torch.manual_seed(seed)."""
    Tn = int(g["T"])
    shape = (2, 96, 16, 16)
    torch.manual_seed(int(g["seed"]))
    init = torch.randn(*shape)
    steps = torch.stack([torch.randn(*shape) for _ in range(Tn)])
    got = np.array([float(init.double().sum()), float(steps.double().pow(2).sum()), float(steps[-1, -1, -1, -1, -1])])
    np.testing.assert_allclose(got, g["noise_check"], rtol=1e-12, atol=0, err_msg="torch's CPU RNG stream differs from the fixture's")
    return init, steps


def philox4x32_10(ctr, key):
    """Philox4x32-10 on numpy uint32 arrays (Salmon et al., SC'11): ctr = 4 arrays, key = 2 scalars.  Test-side restatement."""
    c = [np.asarray(v, dtype=np.uint64) for v in ctr]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & MASK, p1 & MASK, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & MASK, p0 & MASK]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & MASK, (k1 + np.uint64(0xBB67AE85)) & MASK
    return [v.astype(np.uint32) for v in c]
