"""One edit per GPU: independent shapes shard across ranks with no data-path collective; the only exchange is the
final gather of occupancy volumes to rank 0 (reference analogue: dist.all_gather of triplanes,
guided_diffusion/image_sample.py:191-192).  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_edits(n_edits: int, world: int, rank: int) -> List[int]:
    """Indices of the edits this rank owns (round-robin, so mixed categories spread evenly)."""
    return list(range(rank, n_edits, world))


def gather_volumes(volumes: Sequence[torch.Tensor], n_edits: int, dst: int = 0, full_shape=None,
                   recv: Optional[List[torch.Tensor]] = None) -> Optional[List[torch.Tensor]]:
    """Gather every rank's decoded volumes on `dst`, returned in edit order.  Ranks may own different counts
    (n_edits need not divide the world size): missing slots are padded with an empty-marker volume.
    `full_shape` (every rank passes the same value) skips the shape agreement round; `recv` = `world` receive buffers
    on `dst` to reuse across calls (they are returned, so only when each rank owns one edit)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        return list(volumes)
    per = (n_edits + world - 1) // world
    like = volumes[0] if volumes else None
    if full_shape is not None:
        full = tuple(int(v) for v in full_shape)
        # a rank that owns no edit still takes part in the gather: its padding lives on its own device (the current
        # CUDA device under RCCL, the CPU under gloo)
        if like is not None:
            dev = like.device
        elif dist.get_backend() == "nccl":
            dev = torch.device("cuda", torch.cuda.current_device())
        else:
            dev = torch.device("cpu")
    else:
        shape = torch.tensor(list(like.shape) if like is not None else [0, 0, 0], device=like.device if like is not None else "cpu")
        # every rank owns at least one edit when n_edits >= world; otherwise agree on the shape first
        shapes = [torch.zeros_like(shape) for _ in range(world)]
        dist.all_gather(shapes, shape)
        full = max((tuple(int(v) for v in s.tolist()) for s in shapes), key=lambda t: t[0] * t[1] * t[2])
        dev = like.device if like is not None else shapes[0].device
    out_all: List[torch.Tensor] = [None] * n_edits   # type: ignore
    for k in range(per):
        mine = volumes[k] if k < len(volumes) else torch.zeros(full, dtype=torch.float32, device=dev)
        bufs = None
        if rank == dst:
            bufs = recv if (recv is not None and per == 1) else [torch.empty(full, dtype=torch.float32, device=dev) for _ in range(world)]
        dist.gather(mine.contiguous(), bufs, dst=dst)
        if rank == dst:
            for r in range(world):
                idx = r + k * world
                if idx < n_edits:
                    out_all[idx] = bufs[r]
    return out_all if rank == dst else None
