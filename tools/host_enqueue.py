"""Host time to ENQUEUE one guided step (the generator iteration returns when everything is in the queues; the GPU was idle at its
start, so nothing blocks) against the GPU time of the step, plain sequence and overlapped forward tail, same process.
If enqueue time >= GPU time the loop is host-bound in that mode.  Usage: python tools/host_enqueue.py"""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from ishapediting_amd import synthetic
    from ishapediting_amd import drag_utils as du
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ds = bench.make_dragstuff(dev, 1234)
    src, tgt = synthetic.handles(bench.HANDLES, seed=7)
    ds.update_latent_params(img=synthetic.latent(0))
    for overlap in (True, False, True, False):
        du._OVERLAP_TAIL = overlap
        host, total = [], []
        for rep in range(2):
            it = ds.training(src, tgt, scale=1200, cof=0.4)
            k = 0
            while True:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                try:
                    next(it)
                except StopIteration:
                    break
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                if rep > 0 and 2 <= k < 39:
                    host.append((t1 - t0) * 1e3)
                    total.append((t2 - t0) * 1e3)
                k += 1
        # free-running rate for comparison
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in ds.training(src, tgt, scale=1200, cof=0.4):
            pass
        torch.cuda.synchronize()
        free = (time.perf_counter() - t0) * 1e3 / 40
        print(f"{'overlapped tail' if overlap else 'plain sequence '}: host enqueue of one step {statistics.median(host):.3f} ms "
              f"(p90 {sorted(host)[int(0.9 * len(host))]:.3f}); step from an idle GPU (enqueue + drain) {statistics.median(total):.3f} ms; "
              f"free-running {free:.3f} ms per step incl. the final decode")


if __name__ == "__main__":
    main()
