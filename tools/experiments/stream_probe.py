"""Does the stream the path is enqueued on matter?  One C3 edit on torch's default (null) stream vs on a created stream."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    side = torch.cuda.Stream()
    for name, ctx in (("default stream", None), ("created stream", side), ("default stream", None), ("created stream", side)):
        times = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.time()
            if ctx is None:
                bench.one_edit(ds, src, tgt)
            else:
                with torch.cuda.stream(ctx):
                    bench.one_edit(ds, src, tgt)
            torch.cuda.synchronize()
            times.append(time.time() - t0)
        print(f"{name}: {min(times) * 1e3:.1f} ms per edit (best of 3)", flush=True)


if __name__ == "__main__":
    main()
