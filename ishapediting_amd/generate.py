"""Mirror of the reference's generate.py CLI (:14-98): DDPM-sample triplanes, save `<save_dir>/triplanes/{i}.npy`
(CHW), decode each on the 256^3 grid and write `<save_dir>/objects/{i}.obj`.

    python -m ishapediting_amd.generate --ddpm_ckpt ... --decoder_ckpt ... --stats_dir ... [--num_steps 256]
    python -m ishapediting_amd.generate --synthetic        # seeded random weights (no checkpoints offline)
"""
from __future__ import annotations

import argparse
import os
import time
from argparse import Namespace

import numpy as np
import torch as th

from . import image_sample, mesh as mesh_backend
from .triplane_decoder import MultiTriplane, decode_volume


def build_parser():
    p = argparse.ArgumentParser(description="Generate a set of triplanes and their corresponding meshes")
    p.add_argument("--resolution", type=str, default=128)
    p.add_argument("--ddpm_ckpt", type=str, default="models/chairs/ddpm_chairs_ckpts/ema_0.9999_200000.pt")
    p.add_argument("--decoder_ckpt", type=str, default="models/chairs/chair_decoder.pt")
    p.add_argument("--stats_dir", type=str, default="models/chairs/statistics/chairs_triplanes_stats")
    p.add_argument("--save_dir", type=str, default="samples/chairs_samples")
    p.add_argument("--num_samples", type=int, default=8)
    p.add_argument("--batch_size", type=int, default=8)
    p.add_argument("--num_steps", type=int, default=256)
    p.add_argument("--shape_resolution", type=int, default=256)
    p.add_argument("--synthetic", action="store_true", help="seeded random weights instead of checkpoints")
    return p


def ddpm_namespace(args) -> Namespace:
    """generate.py:63-70."""
    return Namespace(
        clip_denoised=True, num_samples=args.num_samples, batch_size=args.batch_size, use_ddim=False,
        model_path=args.ddpm_ckpt, stats_dir=args.stats_dir, explicit_normalization=True, save_dir=args.save_dir,
        save_intermediate=False, save_timestep_interval=20, image_size=int(args.resolution), num_channels=256,
        num_res_blocks=2, num_heads=4, num_heads_upsample=-1, num_head_channels=64, attention_resolutions="32,16,8",
        channel_mult="", dropout=0.1, class_cond=False, use_checkpoint=False, use_scale_shift_norm=True,
        resblock_updown=True, use_fp16=True, use_new_attention_order=False, in_out_channels=96, learn_sigma=True,
        diffusion_steps=1000, noise_schedule="linear", timestep_respacing=str(args.num_steps), use_kl=False,
        predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False)


def decode_to_obj(triplane_chw: np.ndarray, decoder: MultiTriplane, res: int, out_path: str):
    """visualize.main + create_obj (:36-73): decode on the dense grid, level-0 surface, vertices / 255 * 2 - 1."""
    lat = th.as_tensor(triplane_chw, dtype=th.float32, device=decoder.device).reshape(1, 96, *triplane_chw.shape[-2:])
    vol = decode_volume(decoder, lat, 1.0, 0.0, res)      # the saved triplane is already un-normalised
    mesh_backend.export_obj(vol, out_path, scale_div=255.0)
    return vol


def main(argv=None, overrides=None):
    """generate.py:14-98.  `overrides`: attribute values replacing the hard-wired model hyper-parameters of the
    namespace (:63-70) -- used by tests that run the CLI on a small model."""
    args = build_parser().parse_args(argv)
    os.makedirs(args.save_dir, exist_ok=True)
    ddpm_args = ddpm_namespace(args)
    for k, v in (overrides or {}).items():
        setattr(ddpm_args, k, v)
    sd = bounds = dec_sd = None
    if args.synthetic:
        from . import synthetic
        from .unet_spec import full_config
        sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict(full_config(), 1234))
        bounds = (-np.ones(96, np.float32), np.ones(96, np.float32))
        dec_sd = synthetic.decoder_state_dict()
    t1 = time.time()
    samples = image_sample.noise2shape(args=ddpm_args, state_dict=sd, bounds=bounds)
    t2 = time.time()
    print("ddpm time:", t2 - t1)
    if int(os.environ.get("RANK", "0")) != 0:
        return
    os.makedirs(f"{args.save_dir}/triplanes", exist_ok=True)
    samples = np.transpose(samples, [0, 3, 1, 2])
    for idx, triplane in enumerate(samples):
        np.save(f"{args.save_dir}/triplanes/{idx}.npy", triplane)
    os.makedirs(f"{args.save_dir}/objects", exist_ok=True)
    from . import visualize
    for idx, triplane in enumerate(samples):
        print(f"Decoding triplane {idx}...")
        decoder_args = Namespace(input=f"{args.save_dir}/triplanes/{idx}.npy", output=f"{args.save_dir}/objects/{idx}.obj",
                                 model_path=args.decoder_ckpt, res=args.shape_resolution)
        visualize.main(args=decoder_args, state_dict=dec_sd)        # generate.py:88-95
    print("Done!")
    print("decode time:", time.time() - t2)


if __name__ == "__main__":
    main()
