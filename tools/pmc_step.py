"""Short driver for rocprofv3 --pmc passes: same kernels and shapes as bench.py's C3 edit, but only 4 sampling
steps + 2 guided iterations + one 256^3 decode (a full edit is ~11 000 dispatches; the counter pass only needs a few
launches of each kernel)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from ishapediting_amd import synthetic
    from ishapediting_amd.drag_utils import DragStuff, get_args
    from ishapediting_amd.unet_spec import full_config
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    args = get_args(["--w_time", "2", "--num_steps", "4", "--shape_resolution", "256"])
    ds = DragStuff(dev, args=args)
    sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict(full_config(), 1234))
    ds.load_weights(sd, synthetic.decoder_state_dict(4321), -np.ones(96, np.float32), np.ones(96, np.float32))
    del sd
    ds.update_latent_params(img=synthetic.latent(0))
    src, tgt = synthetic.handles(3, seed=7)
    for _ in ds.training(src, tgt, scale=1200, cof=0.4):
        pass
    torch.cuda.synchronize()
    print("ok", float(ds.volume.abs().mean()))


if __name__ == "__main__":
    main()
