#!/usr/bin/env python3
"""How many independent C3 edits should share one MI355X?  k model contexts on k streams driven by k host threads, k = 1 .. 4:
throughput (s/shape) and latency per edit.  Development probe; bench.py reports k = 2 next to the headline."""
import os, sys, time, threading
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ishapediting_amd import synthetic

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ctxs = []
for k in range(4):
    d = bench.make_dragstuff(dev, 1234 + k)
    d.update_latent_params(img=synthetic.latent(k))
    ctxs.append((d, *synthetic.handles(3, seed=7 + k), torch.cuda.Stream(dev)))
torch.cuda.synchronize()
reps = 2
for n in (1, 2, 3, 4, 2, 1):
    def run(c):
        d, s, t, st = c
        with torch.cuda.stream(st):
            for _ in range(reps):
                bench.one_edit(d, s, t)
    for warm in (True, False):
        ths = [threading.Thread(target=run, args=(c,)) for c in ctxs[:n]]
        torch.cuda.synchronize(); t0 = time.time()
        [t.start() for t in ths]; [t.join() for t in ths]
        torch.cuda.synchronize(); dt = time.time() - t0
    print(f"{n} concurrent edits: {dt / (n * reps):.4f} s/shape throughput, {dt / reps:.4f} s latency per edit", flush=True)
