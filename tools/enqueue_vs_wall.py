"""How far ahead of the GPU does the host run?  Times one C3 edit twice: until the Python call returns (everything
enqueued) and until the device is idle.  enqueue ~ wall => launch-bound; enqueue << wall => GPU-bound."""
import os
import sys
import time

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
    for _ in range(3):
        t0 = time.time()
        for _ in ds.training(src, tgt, scale=1200, cof=0.4):
            pass
        t1 = time.time()
        torch.cuda.synchronize()
        t2 = time.time()
        print(f"enqueue {1e3 * (t1 - t0):.1f} ms   wall {1e3 * (t2 - t0):.1f} ms", flush=True)


if __name__ == "__main__":
    main()
