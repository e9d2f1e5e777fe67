#!/usr/bin/env python3
"""Full-size UNet forward on the GPU: parity vs the CPU oracle (optional) and timing."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ishapediting_amd import synthetic  # noqa: E402
from ishapediting_amd.unet import UNetModel  # noqa: E402
from ishapediting_amd.unet_spec import build_spec, full_config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--feat_layer", type=int, default=8)
    a = ap.parse_args()
    cfg = full_config()
    t0 = time.time()
    sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 1234))
    print(f"weights: {time.time() - t0:.1f}s", flush=True)
    dev = torch.device("cuda", 0)
    m = UNetModel(cfg, dev)
    t0 = time.time()
    m.load_state_dict(sd)
    print(f"load+pack: {time.time() - t0:.1f}s", flush=True)
    x = torch.from_numpy(synthetic.latent(0)).to(dev)
    ts = torch.tensor([500])
    out, feat = m(x, ts, feat_layer=a.feat_layer)
    torch.cuda.synchronize()
    print("out", tuple(out.shape), float(out.abs().mean()), "finite", bool(torch.isfinite(out).all()),
          "tap", tuple(feat.shape), float(feat.float().abs().mean()), flush=True)
    for keep in (False, True):
        for _ in range(3):
            m(x, ts, feat_layer=a.feat_layer, keep_for_backward=keep, want_inter_feat=False)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(a.iters):
            m(x, ts, feat_layer=a.feat_layer, keep_for_backward=keep, want_inter_feat=False)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / a.iters
        print(f"forward keep={keep}: {dt * 1e3:.3f} ms  -> {634.9 / dt / 1e3:.1f} TFLOP/s algorithmic", flush=True)
    if a.oracle:
        from oracle import ref_cpu as O
        torch.set_num_threads(os.cpu_count())
        net = O.UNetOracle(build_spec(cfg), sd, fp16=False)
        t0 = time.time()
        with torch.no_grad():
            o_ref, f_ref = net.forward(x.cpu(), ts, feat_layer=a.feat_layer)
        print(f"oracle fp32 forward: {time.time() - t0:.2f}s on {torch.get_num_threads()} threads", flush=True)
        r = float((out.cpu() - o_ref).norm() / o_ref.norm())
        rf = float((feat.float().cpu() - f_ref).norm() / f_ref.norm())
        print(f"rel L2 err: out {r:.3e}  tap {rf:.3e}  max|out err| {float((out.cpu() - o_ref).abs().max()):.3e}", flush=True)


if __name__ == "__main__":
    main()
