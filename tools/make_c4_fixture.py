#!/usr/bin/env python3
"""Compact oracle fixture of BASELINE configs[3] at FULL length, end to end.

Runs the pinned fp32 CPU oracle (oracle/ref_cpu.py) ONCE, here in the build container, from the seeds of
`ishapediting_amd.synthetic.c4_inputs`: 200 reconstruction steps x 40 000 points with injected batches and noise
(drag_utils.py:442-463) -> DDPM inversion over 170 steps (gaussian_diffusion.py:512-532, drag_utils.py:552-566) -> 170
guided drag iterations (drag_utils.py:336-398) -> 256^3 decode.  Unlike `tools/parity_report.py --c4 oracle` no stage starts
from a device result: every stage continues from the oracle's own previous stage, so the fixture is a pure function of the
seeds and the device chain can be held against it end to end by the driver-run suite
(tests/test_gpu_fullsize.py::test_c4_full_length_end_to_end_vs_the_committed_oracle_fixture).

Kept (<= 10 MB): the final latent as fp32; the reconstruction and the inverted latent w on every 4th channel as fp32;
the 200 + 170 losses; the norms of the 170 variance-noise tensors; for both decodes the logits of the 64^3 sub-grid
[::4, ::4, ::4] of the 256^3 volume as fp16 with their sign bits (taken from the fp32 logits), and the number of inside voxels.

    python tools/make_c4_fixture.py [--threads 6] [--out tests/golden/g16_c4_full_length.npz]

~25 min on 6 host threads.  Test infrastructure: imports the oracle; nothing under ishapediting_amd/ does."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LO, HI = -0.05, 0.05            # the un-normalisation bounds of the C4 runs (tools/parity_report.py, bench.py's C4 leg)
CH_STEP = 4                     # channel subsample of the two intermediate latents


class Lazy:                     # the oracle's loops index coords[k] / gts[k] / noises[k]: one tensor alive at a time
    def __init__(self, f): self.f = f
    def __getitem__(self, k): return self.f(k)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "g16_c4_full_length.npz"))
    ap.add_argument("--steps", type=int, nargs=2, default=None, metavar=("T", "W"), help="shortened chain (smoke run of this script)")
    a = ap.parse_args()
    from ishapediting_amd import synthetic
    from ishapediting_amd.synthetic import C4_POINTS, C4_RES, C4_T, C4_W, c4_inputs
    from ishapediting_amd.unet_spec import build_spec, full_config
    from oracle import ref_cpu as O
    torch.set_num_threads(a.threads)
    T, W = (a.steps if a.steps else (C4_T, C4_W))
    res = C4_RES
    img0, batch, noise = c4_inputs()
    cfg = full_config()
    sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 1234))
    dec_sd = synthetic.decoder_state_dict(4321)
    rng = torch.full((1, 96, 1, 1), (HI - LO) / 2)
    mid = torch.full((1, 96, 1, 1), (HI + LO) / 2)
    src, tgt = synthetic.handles(3)
    net = O.UNetOracle(build_spec(cfg), sd, fp16=False)
    diff = O.DiffusionOracle(O.Tables(str(C4_T)))
    t0 = time.time()
    say = lambda m: print(f"[{time.time() - t0:6.0f} s] {m}", file=sys.stderr, flush=True)

    # ---- stage 1: reconstruction (train_triplane's guided loop), positions k = 0 .. T-1 <-> loop index i = C4_T-1-k ----
    img, loss_rec = img0, []
    for k in range(T):
        i = C4_T - 1 - k
        imgs, losses, _ = O.reconstruct_loop(diff, net, dec_sd, img, rng, mid, Lazy(lambda _k, k=k: batch(k)[0]), Lazy(lambda _k, k=k: batch(k)[1]),
                                             Lazy(lambda _k, k=k: noise(0, k)), scale=600.0, steps=[i])
        img = imgs[-1]
        loss_rec.append(float(losses[-1]))
        if k % 10 == 0: say(f"reconstruction step {k} / {T}: loss {loss_rec[-1]:.5f}")
    rec = img.detach()
    with torch.no_grad():
        vol_rec = O.decode_volume(dec_sd, rec, rng, mid, res)
    say("reconstruction decoded")
    # ---- stage 2: inversion over W from the oracle's own reconstruction ----
    with torch.no_grad():
        inv = diff.ddpm_inversion(net, rec, W, Lazy(lambda k: noise(1, k)), feat_layer=8)
    say("inversion done")
    w = inv["latent"]
    vn_norm = torch.stack([v.flatten().norm() for v in inv["variance_noise"]]).numpy()
    cache = [O.resize_feat_align(f) for f in inv["inter_feat"]]
    del inv
    # ---- stage 3: W guided drag iterations from the oracle's own w and guidance cache ----
    setup = O.DragSetup(src, tgt, 12, 2.0 / res, cache[0].shape[-1])
    final, loss_drag = O.drag_loop(diff, net, w, cache, setup, W, 8, 1200.0, 0.4, Lazy(lambda i: noise(2, W - 1 - i)),
                                   progress=lambda i: say(f"drag step {i}") if i % 10 == 0 else None)
    final = final.detach()
    with torch.no_grad():
        vol = O.decode_volume(dec_sd, final, rng, mid, res)
    say("final decode done")
    sub = lambda v: v[::4, ::4, ::4].contiguous()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    np.savez_compressed(
        a.out,
        meta=np.array([T, W, res, C4_POINTS, CH_STEP, 2024, 1234, 4321]),      # steps, decode res, points, channel step, seeds (inputs, UNet, decoder)
        bounds=np.array([LO, HI], np.float32),
        rec_sub=rec[:, ::CH_STEP].numpy().astype(np.float32), w_sub=w[:, ::CH_STEP].numpy().astype(np.float32), final=final.numpy().astype(np.float32),
        rec_norm=np.array(float(rec.norm())), w_norm=np.array(float(w.norm())),
        loss_rec=np.array(loss_rec, np.float64), loss_drag=np.array([float(x) for x in loss_drag], np.float64), vn_norm=vn_norm.astype(np.float32),
        vol_rec_sub=sub(vol_rec).numpy().astype(np.float16), vol_rec_sub_bits=np.packbits((sub(vol_rec) > 0).numpy().reshape(-1)), vol_rec_inside=np.array(int((vol_rec > 0).sum())),
        vol_sub=sub(vol).numpy().astype(np.float16), vol_sub_bits=np.packbits((sub(vol) > 0).numpy().reshape(-1)), vol_inside=np.array(int((vol > 0).sum())),
        vol_rms=np.array(float(vol.pow(2).mean().sqrt())), vol_rec_rms=np.array(float(vol_rec.pow(2).mean().sqrt())),
        seconds=np.array(time.time() - t0), threads=np.array(a.threads))
    say(f"wrote {a.out} ({os.path.getsize(a.out) / 1e6:.1f} MB)")


if __name__ == "__main__":
    main()
