"""Factory mirror of guided_diffusion/script_util.py:42-129,389-455 for the config the path uses."""
from __future__ import annotations

from .gaussian_diffusion import create_gaussian_diffusion
from .unet import UNetModel
from .unet_spec import UNetConfig


def diffusion_defaults():
    return dict(learn_sigma=False, diffusion_steps=1000, noise_schedule="linear", timestep_respacing="", use_kl=False,
                predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False)


def model_and_diffusion_defaults():
    """script_util.py:42-65 (same keys; `args_to_dict(args, defaults.keys())` is how callers pick kwargs)."""
    res = dict(image_size=128, num_channels=256, num_res_blocks=2, num_heads=4, num_heads_upsample=-1,
               num_head_channels=-1, attention_resolutions="16,8", channel_mult="", dropout=0.0, class_cond=False,
               use_checkpoint=False, use_scale_shift_norm=True, resblock_updown=False, use_fp16=False,
               use_new_attention_order=False, in_out_channels=3)
    res.update(diffusion_defaults())
    return res


def args_to_dict(args, keys):
    return {k: getattr(args, k) for k in keys}


def create_model(image_size, num_channels, num_res_blocks, channel_mult="", learn_sigma=False, class_cond=False,
                 use_checkpoint=False, attention_resolutions="16", num_heads=1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, dropout=0, resblock_updown=False, use_fp16=False,
                 use_new_attention_order=False, in_out_channels=3, device=None, max_batch=1):
    """script_util.py:132-187."""
    if class_cond or use_new_attention_order or not use_scale_shift_norm or not resblock_updown:
        raise NotImplementedError("only the configuration on the path is built (drag_utils.py:44-57)")
    if num_head_channels == -1:
        raise NotImplementedError("num_head_channels must be given (64 on the path)")
    mult = () if channel_mult == "" else tuple(int(m) for m in channel_mult.split(","))
    cfg = UNetConfig(image_size=int(image_size), in_channels=in_out_channels, model_channels=num_channels,
                     out_channels=in_out_channels * (2 if learn_sigma else 1), num_res_blocks=num_res_blocks,
                     attention_resolutions=attention_resolutions, channel_mult=mult, num_head_channels=num_head_channels,
                     num_heads=num_heads, use_scale_shift_norm=use_scale_shift_norm, resblock_updown=resblock_updown,
                     use_fp16=use_fp16)
    return UNetModel(cfg, device=device, max_batch=max_batch)


def create_model_and_diffusion(image_size, class_cond, learn_sigma, num_channels, num_res_blocks, channel_mult, num_heads,
                               num_head_channels, num_heads_upsample, attention_resolutions, dropout, diffusion_steps,
                               noise_schedule, timestep_respacing, use_kl, predict_xstart, rescale_timesteps,
                               rescale_learned_sigmas, use_checkpoint, use_scale_shift_norm, resblock_updown, use_fp16,
                               use_new_attention_order, in_out_channels, device=None, max_batch=1):
    """script_util.py:74-129."""
    model = create_model(image_size, num_channels, num_res_blocks, channel_mult=channel_mult, learn_sigma=learn_sigma,
                         class_cond=class_cond, use_checkpoint=use_checkpoint, attention_resolutions=attention_resolutions,
                         num_heads=num_heads, num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample,
                         use_scale_shift_norm=use_scale_shift_norm, dropout=dropout, resblock_updown=resblock_updown,
                         use_fp16=use_fp16, use_new_attention_order=use_new_attention_order,
                         in_out_channels=in_out_channels, device=device, max_batch=max_batch)
    diffusion = create_gaussian_diffusion(steps=diffusion_steps, learn_sigma=learn_sigma, noise_schedule=noise_schedule,
                                          use_kl=use_kl, predict_xstart=predict_xstart,
                                          rescale_timesteps=rescale_timesteps,
                                          rescale_learned_sigmas=rescale_learned_sigmas,
                                          timestep_respacing=timestep_respacing)
    return model, diffusion
