"""BASELINE.json configs[1] (C2) timed on one MI355X: the full 1000-step DDPM sample of the full-size model (batch 1 and 8),
the 256^3 occupancy decode and the device marching cubes + 10 smoothing sweeps.  Synthetic seeded weights (no checkpoints
offline), so the level set is a noise surface: the surface time is reported for an analytic sphere as well."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from ishapediting_amd import synthetic
    from ishapediting_amd.gaussian_diffusion import create_gaussian_diffusion
    from ishapediting_amd.mesh import extract_surface, smooth_mesh
    from ishapediting_amd.triplane_decoder import MultiTriplane, decode_volume
    from ishapediting_amd.unet import UNetModel
    from ishapediting_amd.unet_spec import full_config
    dev = torch.device("cuda", 0)
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    cfg = full_config()
    sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 1234))
    dec = MultiTriplane(1, device=dev)
    dec.net.load_state_dict(synthetic.decoder_state_dict(4321))
    for B in (1, 8):
        model = UNetModel(cfg, dev, max_batch=B)
        model.load_state_dict(sd)
        diff = create_gaussian_diffusion(timestep_respacing=str(steps))
        g = torch.Generator(device="cpu").manual_seed(5)
        noise = torch.randn((B, 96, 128, 128), generator=g).to(dev)
        diff.p_sample_loop(model, noise.shape, noise=noise, device=dev, progress=False) if steps <= 20 else None   # warm-up on short runs only
        torch.cuda.synchronize()
        t0 = time.time()
        sample = diff.p_sample_loop(model, noise.shape, noise=noise, device=dev)
        torch.cuda.synchronize()
        t1 = time.time()
        vol = decode_volume(dec, sample[:1], 1.0, 0.0, 256)
        torch.cuda.synchronize()
        t2 = time.time()
        print(f"C2 batch {B}: {steps}-step sample {t1 - t0:.3f} s = {1e3 * (t1 - t0) / steps:.2f} ms/step "
              f"({(t1 - t0) / B:.3f} s per shape), 256^3 decode {1e3 * (t2 - t1):.1f} ms", flush=True)
        del model
        torch.cuda.empty_cache()
    # surface of a shape-like volume (sphere) and of the random-weight volume
    lin = torch.linspace(-1, 1, 256, device=dev)
    zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
    sphere = 0.6 - torch.sqrt(xx * xx + yy * yy + zz * zz)
    for name, v in (("sphere", sphere), ("random-weight volume", vol)):
        extract_surface(v)
        torch.cuda.synchronize()
        t0 = time.time()
        verts, tris = extract_surface(v)
        sm = smooth_mesh(verts, tris, 10, box_max=255.0)
        torch.cuda.synchronize()
        print(f"marching cubes + 10 smoothing sweeps, {name}: {1e3 * (time.time() - t0):.1f} ms, {verts.shape[0]} vertices, "
              f"{tris.shape[0]} triangles", flush=True)


if __name__ == "__main__":
    main()
