"""Repeatability of the default edit path under load: N full C3 edits (overlapped forward tail, XCD-local GroupNorm rendezvous,
fused step) must give bitwise the same final latent and 256^3 volume every time, with no device status error -- alone on the GPU
and beside a SECOND PROCESS that keeps the chip busy with matrix products (other processes are invisible to the library's
tenancy guard: this is the uneven-load case MI355X_MICROARCH.md asks every in-launch hand-off to be tested under).
Usage: python tools/stress_repeat.py [--edits 40] [--load]"""
import argparse
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

LOAD = r"""
import torch, time
a = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
b = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
t0 = time.time()
while time.time() - t0 < %f:
    for _ in range(7):
        c = a @ b
    torch.cuda.synchronize()
    time.sleep(0.0007)          # bursts: the edit sees the chip alternately free and taken
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edits", type=int, default=40)
    ap.add_argument("--load", action="store_true")
    a = ap.parse_args()
    from ishapediting_amd import _lib, synthetic
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ds = bench.make_dragstuff(dev, 1234)
    src, tgt = synthetic.handles(bench.HANDLES, seed=7)
    ds.update_latent_params(img=synthetic.latent(0))
    noise = [synthetic.step_noise(500 + i, (1, 96, 128, 128)).to(dev) for i in range(bench.GUIDED_STEPS)]
    ds.step_noise = lambda i: noise[i]                      # the same injected noise every edit
    bench.one_edit(ds, src, tgt)
    torch.cuda.synchronize()
    ref_lat, ref_vol = ds.tri_feat.clone(), ds.volume.clone()
    child = None
    if a.load:
        child = subprocess.Popen([sys.executable, "-c", LOAD % (a.edits * 0.45 + 20.0)])
        time.sleep(8.0)                                     # the child's first import of torch
    bad = 0
    t0 = time.time()
    for k in range(a.edits):
        bench.one_edit(ds, src, tgt)
        torch.cuda.synchronize()
        same = torch.equal(ds.tri_feat, ref_lat) and torch.equal(ds.volume, ref_vol)
        st = int(_lib.lib().ishap_device_status())
        if not same or st != 0:
            bad += 1
            print(f"edit {k}: bitwise equal {same}, device status {st} ({_lib.lib().ishap_last_error().decode()})", flush=True)
    dt = (time.time() - t0) / a.edits
    if child is not None:
        child.terminate()
        child.wait()
    print(f"{a.edits} edits {'beside a loading process' if a.load else 'alone'}: {bad} differing or failing, {dt:.4f} s per edit", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
