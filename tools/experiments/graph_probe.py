"""Is the UNet forward/backward host-launch-bound?  Eager calls vs replay of the same launches captured in a HIP graph."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


def main():
    from ishapediting_amd import synthetic
    from ishapediting_amd.unet import UNetModel
    from ishapediting_amd.unet_spec import full_config
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    cfg = full_config()
    m = UNetModel(cfg, dev)
    m.load_state_dict(synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 1234)))
    x = torch.from_numpy(synthetic.latent(0)).to(dev)
    ts = [500.0]
    cot = (torch.randn((4096, 512)) * 0.05).half().to(dev)

    def fwd():
        return m(x, ts, feat_layer=8, keep_for_backward=True, want_inter_feat=False)

    def bwd():
        return m.backward_input(cot)

    def both():
        fwd()
        bwd()

    print(f"eager  forward {timeit(fwd):.3f} ms   forward+backward {timeit(both):.3f} ms", flush=True)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        both()
        torch.cuda.synchronize()
        gf = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gf, stream=s):
            fwd()
        gb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gb, stream=s):
            both()
        torch.cuda.synchronize()
        print(f"graph  forward {timeit(gf.replay):.3f} ms   forward+backward {timeit(gb.replay):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
