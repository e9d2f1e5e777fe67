"""Mirror of triplane_decoder/visualize.py's call surface over the decode and surface kernels.

Same names, arguments and return types as the reference functions third parties call:

  create_obj(model, obj_idx, res=128, max_batch_size=50000, output_path='output.obj')      visualize.py:36-73
  create_obj_o3d(model, obj_idx, res=128, max_batch_size=50000) -> mesh                     visualize.py:76-105
  main(args=None, feature=None)                                                             visualize.py:108-128

`model` is ishapediting_amd.triplane_decoder.MultiTriplane with its `embeddings` set (drag_utils.py:295-298,
generate.py:95).  The reference decodes the res^3 grid in `max_batch_size`-point chunks with a host round trip each
(:89-95) and runs PyMCubes on the CPU; here the whole grid is one kernel launch and the level-0 surface is extracted on
the device, so `max_batch_size` is accepted and ignored.  Each function keeps its own vertex convention: create_obj
divides by 255 (:72), create_obj_o3d by res (:101).
"""
from __future__ import annotations

import argparse

import numpy as np
import torch

from . import mesh as mesh_backend
from .triplane_decoder import MultiTriplane, decode_planes_grid


def decode_grid(model: MultiTriplane, obj_idx: int, res: int) -> torch.Tensor:
    """The dense-grid half both functions share (:41-67, :79-97): logits on linspace(-1,1,res)^3 ('ij') -> [res,res,res]
    on the device."""
    model.eval()
    return decode_planes_grid(model, model._planes(obj_idx), res)


def create_obj(model: MultiTriplane, obj_idx: int, res: int = 128, max_batch_size: int = 50000,
               output_path: str = "output.obj"):
    """visualize.py:36-73: level-0 surface of the decoded grid, vertices / 255 * 2 - 1, written as Wavefront OBJ."""
    del max_batch_size
    vol = decode_grid(model, obj_idx, res)
    mesh_backend.export_obj(vol, output_path, scale_div=255.0)
    return vol


def create_obj_o3d(model: MultiTriplane, obj_idx: int, res: int = 128, max_batch_size: int = 50000):
    """visualize.py:76-105: the mesh of the decoded grid, vertices / res * 2 - 1, NOT smoothed (get_mesh applies
    filter_smooth_simple itself, drag_utils.py:300).  Returns what mesh.BACKEND selects: an OccupancyMesh ("device"), an
    open3d.geometry.TriangleMesh built from the device surface ("open3d") or by PyMCubes ("third_party")."""
    del max_batch_size
    vol = decode_grid(model, obj_idx, res)
    return mesh_backend.volume_to_mesh(vol, res, smooth_iterations=0)


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--input", type=str)
    p.add_argument("--output", type=str, required=True)
    p.add_argument("--model_path", type=str, default="models/epoch_24_decoder_loss=25.37570571899414.pt", required=False)
    p.add_argument("--res", type=int, default=128, required=False)
    return p


def main(args=None, feature=None, state_dict=None):
    """visualize.py:108-128: load the decoder (`args.model_path`), take the triplanes from `args.input` (a .npy that
    reshapes to 3x32x128x128) or from `feature`, decode at `args.res` and write `args.output`.  `state_dict` replaces the
    checkpoint file (tests, synthetic runs: no checkpoints exist offline)."""
    if args is None:
        args = build_parser().parse_args()
    device = torch.device("cuda", torch.cuda.current_device())
    model = MultiTriplane(1, input_dim=3, output_dim=1, device=device)
    model.net.load_state_dict(state_dict if state_dict is not None else torch.load(args.model_path, map_location="cpu"))
    model.eval()
    triplanes = np.load(args.input) if feature is None else feature
    triplanes = np.asarray(triplanes.detach().cpu() if torch.is_tensor(triplanes) else triplanes, dtype=np.float32)
    side = int(round((triplanes.size // 96) ** 0.5))
    triplanes = triplanes.reshape(3, 32, side, side)      # the reference hard-codes 128 (:121); any square side works here
    for i in range(3):
        model.embeddings[i] = torch.as_tensor(triplanes[[i]], device=device)
    create_obj(model, 0, res=args.res, output_path=args.output)


if __name__ == "__main__":
    main()
