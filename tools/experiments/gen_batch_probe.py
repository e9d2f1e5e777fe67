"""Full-size UNet forward at batch B (the generate.py path): ms per forward and per sample."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from ishapediting_amd import synthetic
    from ishapediting_amd.unet import UNetModel
    from ishapediting_amd.unet_spec import full_config
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda", 0)
    cfg = full_config()
    m = UNetModel(cfg, dev, max_batch=B)
    m.load_state_dict(synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 1234)))
    x = torch.randn(B, 96, 128, 128, device=dev)
    ts = [500.0] * B
    m(x, ts)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        m(x, ts)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n * 1e3
    print(f"batch {B}: {dt:.2f} ms per forward, {dt / B:.2f} ms per sample, {634.9 * B / dt:.0f} TFLOP/s")


if __name__ == "__main__":
    main()
