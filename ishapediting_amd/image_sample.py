"""Mirror of guided_diffusion/image_sample.py:noise2shape (:138-201), the driver behind generate.py.

Each rank samples its own batch on its own GPU; the un-normalised triplanes are exchanged with one all_gather
(RCCL when launched under torchrun, exactly the reference's single collective, :191-192).  The MPI bootstrap of
dist_util.py is replaced by torch.distributed's env:// rendezvous (RANK / WORLD_SIZE / MASTER_ADDR).
"""
from __future__ import annotations

import os

import numpy as np
import torch as th
import torch.distributed as dist

from .script_util import args_to_dict, create_model_and_diffusion, model_and_diffusion_defaults


def setup_dist():
    """dist_util.py:21-43 without mpi4py: join the process group described by the environment, if any."""
    if dist.is_available() and not dist.is_initialized() and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if th.cuda.is_available() else "gloo")


def dev():
    """dist_util.py:46-53."""
    return th.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))


def unnormalize(sample, stats_dir=None, lower_bound=None, upper_bound=None):
    """normalization.py:6-15: x * (max-min)/2 + (max+min)/2 per channel."""
    if stats_dir is not None:
        lower_bound = np.load(f"{stats_dir}/lower_bound.npy")
        upper_bound = np.load(f"{stats_dir}/upper_bound.npy")
    mn = th.as_tensor(np.asarray(lower_bound, dtype=np.float32).reshape(1, -1, 1, 1), device=sample.device)
    mx = th.as_tensor(np.asarray(upper_bound, dtype=np.float32).reshape(1, -1, 1, 1), device=sample.device)
    return sample * ((mx - mn) / 2) + (mn + mx) / 2


def gather_samples(sample: th.Tensor, num_samples: int) -> np.ndarray:
    """The tail of noise2shape (image_sample.py:188-197): NCHW -> NHWC, one all_gather over the ranks (rank order), the
    batches concatenated along axis 0 and cut to `num_samples`.  Single process: the local batch alone."""
    sample = sample.permute(0, 2, 3, 1).contiguous()
    if dist.is_initialized() and dist.get_world_size() > 1:
        gathered = [th.zeros_like(sample) for _ in range(dist.get_world_size())]
        dist.all_gather(gathered, sample)
        dist.barrier()
    else:
        gathered = [sample]
    arr = np.concatenate([s.cpu().numpy() for s in gathered], axis=0)
    return arr[:num_samples]


def noise2shape(args=None, state_dict=None, bounds=None, noise=None, step_noise=None):
    """image_sample.py:138-201.  `state_dict` / `bounds` let callers without checkpoint files (tests, synthetic
    benchmarks) pass weights and (lower, upper) normalisation bounds directly."""
    setup_dist()
    device = dev()
    model, diffusion = create_model_and_diffusion(**args_to_dict(args, model_and_diffusion_defaults().keys()),
                                                  device=device, max_batch=args.batch_size)
    sd = state_dict if state_dict is not None else th.load(args.model_path, map_location="cpu")
    model.load_state_dict(sd)
    if args.use_fp16:
        model.convert_to_fp16()
    model.eval()
    shape = (args.batch_size, 96, args.image_size, args.image_size)
    sample_fn = diffusion.p_sample_loop if not getattr(args, "use_ddim", False) else diffusion.ddim_sample_loop   # :166-168
    sample = sample_fn(model, shape, noise=noise, clip_denoised=args.clip_denoised, device=device, step_noise=step_noise)
    if args.explicit_normalization:
        if bounds is not None:
            sample = unnormalize(sample, lower_bound=bounds[0], upper_bound=bounds[1])
        else:
            sample = unnormalize(sample, stats_dir=args.stats_dir)
    return gather_samples(sample, args.num_samples)
