"""Stamps of the fused conv + split-K fix-up + GroupNorm launch (round 6 experiment, tools/experiments/fixup8_conv_splitk_groupnorm.patch
applied and the library built with -DFX_STAMPS): one full-size forward, then the s_memtime stamps (shader clock) of the LAST fused launch's
finishing workgroup of slice 0 of every n-tile: kernel entry, K loop done, fix-up entered, partial tile acknowledged + flag raised,
all 16 flags seen, 16 slices landed, group sums done, outputs stored."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ishapediting_amd import synthetic, _lib
from ishapediting_amd.unet import UNetModel
from ishapediting_amd.unet_spec import full_config

cfg = full_config()
dev = torch.device("cuda", 0)
m = UNetModel(cfg, dev)
m.load_state_dict(synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 1234)))
x = torch.from_numpy(synthetic.latent(2)).to(dev)
for _ in range(3):
    m(x, [617.0], feat_layer=8, keep_for_backward=True)
torch.cuda.synchronize()
L = C.CDLL(os.path.join(ROOT, "ishapediting_amd", "libishap_hip.so"))
buf = (C.c_ulonglong * 128)()
assert L.ishap_debug_fx_stamps(buf) == 0
st = np.array(buf[:], dtype=np.float64).reshape(16, 8)
order = [6, 7, 0, 1, 2, 3, 4, 5]
names = ["kernel entry", "K loop done", "fix-up entered (slice stores issued)", "stores acknowledged, barrier, flag raised", "all 16 flags seen",
         "16 slices landed (128 KB)", "sums added, y stored, group sums reduced", "normalised + stored"]
GHZ = 2.1                                          # s_memtime counts shader cycles; 100 cycles ~ 0.042-0.05 us under this load
d = np.diff(st[:, order], axis=1) / (GHZ * 1e3)
print(f"us at an assumed {GHZ} GHz, median over the 16 n-tiles (p10 .. p90)")
for k in range(1, 8):
    v = np.sort(d[:, k - 1])
    print(f"  {names[k]:46s} +{np.median(v):6.2f}   ({v[1]:.2f} .. {v[-2]:.2f})")
print(f"  entry -> end: {np.median((st[:, 5] - st[:, 6]) / (GHZ * 1e3)):.2f} us")
