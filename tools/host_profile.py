"""cProfile of one C3 edit on the host (ctypes calls are charged to their Python caller's tottime)."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from ishapediting_amd import synthetic
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ds = bench.make_dragstuff(dev, 1234)
    src, tgt = synthetic.handles(bench.HANDLES, seed=7)
    ds.update_latent_params(img=synthetic.latent(0))
    bench.one_edit(ds, src, tgt)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    bench.one_edit(ds, src, tgt)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
    main()
