"""full-size batch-N forward vs N single forwards, per image (debug aid for tests/test_gpu_batched_tiles.py)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ishapediting_amd import synthetic
from ishapediting_amd.unet import UNetModel
from ishapediting_amd.unet_spec import full_config
dev = torch.device("cuda", 0)
cfg = full_config()
m = UNetModel(cfg, dev, max_batch=8)
m.load_state_dict(synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 1234)))
g = torch.Generator().manual_seed(808)
x = torch.randn(8, 96, 128, 128, generator=g).to(dev)
ts_all = [999.0, 870.0, 641.0, 500.0, 333.0, 120.0, 37.0, 0.0]
rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))
singles = [m(x[b:b + 1], ts_all[b:b + 1], feat_layer=-1).clone() for b in range(8)]
singles_same_t = [m(x[b:b + 1], [500.0], feat_layer=-1).clone() for b in range(8)]
for n in (2, 3, 4, 6, 8):
    o = m(x[:n], ts_all[:n], feat_layer=-1).clone()
    o2 = m(x[:n], [500.0] * n, feat_layer=-1).clone()
    torch.cuda.synchronize()
    print(f"batch {n}: per-image rel vs single, own timesteps:", " ".join(f"{rel(o[b:b+1], singles[b]):.1e}" for b in range(n)),
          "| same timestep:", " ".join(f"{rel(o2[b:b+1], singles_same_t[b]):.1e}" for b in range(n)), flush=True)
