"""How far ahead of the GPU does the host run?  One C3 edit: the host time at which each guided iteration has been
enqueued (the generator yields) against the device time at which it has finished (an event recorded at the yield).
host << device => GPU-bound with the host running ahead; host ~ device => the launches themselves are the limit."""
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
        start = torch.cuda.Event(enable_timing=True)
        start.record()
        t0 = time.time()
        host, evs = [], []
        for _ in ds.training(src, tgt, scale=1200, cof=0.4):
            host.append(time.time() - t0)
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)
        torch.cuda.synchronize()
        wall = time.time() - t0
        devt = [start.elapsed_time(e) for e in evs]
        k = len(host) - 1
        print(f"iteration {k + 1}: enqueued at {1e3 * host[k]:.1f} ms, finished on the device at {devt[k]:.1f} ms; "
              f"iteration 10: {1e3 * host[9]:.1f} / {devt[9]:.1f} ms; whole edit {1e3 * wall:.1f} ms", flush=True)


if __name__ == "__main__":
    main()
