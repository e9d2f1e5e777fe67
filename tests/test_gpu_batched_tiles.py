"""The 128x128-tile conv kernels on maps narrower than 128 pixels (several image rows per tile:
igemm4_kernel<128,128,64|32|16,...>, csrc/igemm4.hip).  At batch 1 the tile policy never picks them (too few tiles to fill
256 CUs); the batched generate path (generate.py:52, batch 8 -> image_sample.py:173-184) does.  Two checks:
  * against the ORACLE: a mid-size configuration whose 64^2 / 32^2 / 16^2 levels all have >= 128 channels, run in a process
    of its own with ISHAP_BIG_MIN=1 (the policy threshold is read once per process) -- forward output, every output-block
    tap, input gradients from a tap and from the output; the launch profile proves that the three kernels ran;
  * at FULL SIZE: batch 8 (where the default policy selects them on the 64^2 and 32^2 maps) against the same eight images
    pushed through one at a time -- same arithmetic, another tile shape and split-K policy, i.e. summation order only.
Needs an MI355X: -m gpu.  Nothing here reads /root/reference."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from ishapediting_amd import synthetic
from ishapediting_amd.unet_spec import UNetConfig, build_spec

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a = torch.as_tensor(a).detach().float().cpu()
    b = torch.as_tensor(b).detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _cfg_wide():
    """128 / 128 / 256 / 256 channels on 64^2 / 32^2 / 16^2 / 8^2 maps, attention at 16^2 and 8^2: with ISHAP_BIG_MIN=1 every
    3x3 layer of the first three levels (forward and input gradient, the output blocks' folded 1x1 skip included) has
    M % 128 == 0 and N >= 128, i.e. takes a 128x128 tile of 2 / 4 / 8 image rows."""
    return UNetConfig(image_size=64, in_channels=6, model_channels=128, out_channels=12, num_res_blocks=1,
                      attention_resolutions="16,8", channel_mult=(1, 1, 2, 2), num_head_channels=64)


_WORKER = r"""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, {root!r})
from ishapediting_amd import synthetic, _lib
from ishapediting_amd.unet import UNetModel
from ishapediting_amd.unet_spec import UNetConfig, build_spec
cfg = UNetConfig(image_size=64, in_channels=6, model_channels=128, out_channels=12, num_res_blocks=1,
                 attention_resolutions="16,8", channel_mult=(1, 1, 2, 2), num_head_channels=64)
dev = torch.device("cuda", 0)
m = UNetModel(cfg, dev)
m.load_state_dict(synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 191)))
inp = np.load({inp!r})
x = torch.from_numpy(inp["x"]).to(dev)
ts = [float(inp["ts"][0])]
nblk = len(build_spec(cfg).output_blocks)
L = _lib.lib()
res = {{}}
L.ishap_profile_begin()
for k in range(nblk):
    out, tap = m(x, ts, feat_layer=k, keep_for_backward=True)
    res[f"tap{{k}}"] = tap.float().cpu().numpy()
res["out"] = out.cpu().numpy()
kg = int(inp["kg"])
m(x, ts, feat_layer=kg, keep_for_backward=True, want_inter_feat=False)
res["gx_tap"] = m.backward_input(torch.from_numpy(inp["cot_tap"]).to(dev)).cpu().numpy()
m(x, ts, feat_layer=-1, keep_for_backward=True)
res["gx_out"] = m.backward_from_output(torch.from_numpy(inp["cot_out"]).to(dev)).cpu().numpy()
torch.cuda.synchronize()
buf = C.create_string_buffer(1 << 20)
n = L.ishap_profile_shapes(buf, len(buf))
assert n > 0
res["shapes"] = np.frombuffer(buf.value, dtype=np.uint8)
tot = (C.c_double * 48)()
L.ishap_profile_end(tot, 16)
res["variants"] = np.array(list(tot))
np.savez({out!r}, **res)
"""


def test_wide_tiles_on_narrow_maps_vs_oracle(tmp_path):
    """ISHAP_BIG_MIN=1: 128x128 tiles on the 64-, 32- and 16-pixel-wide maps of a mid-size model, against the fp32 oracle on
    the same fp16-rounded weights.  Tolerances of the other mid-size oracle tests: 1e-2 forward (relative L2), 2e-2 input
    gradients."""
    from oracle import ref_cpu as O
    cfg = _cfg_wide()
    spec = build_spec(cfg)
    sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 191))
    net = O.UNetOracle(spec, sd, fp16=False)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(1, 6, 64, 64, generator=g)
    ts = torch.tensor([433.0])
    nblk = len(spec.output_blocks)
    xr = x.clone().requires_grad_(True)
    ref_out, ref_taps = net.forward(xr, ts, all_taps=True)
    kg = nblk - 3
    ct_tap = torch.randn(ref_taps[kg].shape, generator=g) * 0.1
    ct_out = torch.randn(ref_out.shape, generator=g)
    (ref_g_tap,) = torch.autograd.grad((ref_taps[kg] * ct_tap).sum(), xr, retain_graph=True)
    (ref_g_out,) = torch.autograd.grad((ref_out * ct_out).sum(), xr)
    inp, outp = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    cot_tap = ct_tap[0].permute(1, 2, 0).reshape(1, -1, ct_tap.shape[1]).contiguous().half().numpy()
    np.savez(inp, x=x.numpy(), ts=ts.numpy(), kg=kg, cot_tap=cot_tap, cot_out=ct_out.numpy())
    env = dict(os.environ)
    env["ISHAP_BIG_MIN"] = "1"
    r = subprocess.run([sys.executable, "-c", _WORKER.format(root=ROOT, inp=inp, out=outp)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(outp)
    # the launches: CSV lines M,N,K,conv3,tile,ksplit,launches,... -- 3x3 launches on 128-row tiles at M = 4096 / 1024 / 256
    lines = [l.split(",") for l in bytes(got["shapes"]).decode().strip().split("\n")]
    big3 = {int(l[0]) for l in lines if int(l[3]) == 1 and int(l[4]) == 128}
    assert {4096, 1024, 256} <= big3, big3
    assert got["variants"][8 * 3] >= 20, got["variants"][8 * 3]          # variant 8 = igemm4's 128x128 tiles
    worst = 0.0
    for k in range(nblk):
        r_t = rel(got[f"tap{k}"], ref_taps[k])
        worst = max(worst, r_t)
        assert r_t < 1e-2, (k, r_t)
    r_o, r_gt, r_go = rel(got["out"], ref_out), rel(got["gx_tap"], ref_g_tap), rel(got["gx_out"], ref_g_out)
    print(f"wide tiles on narrow maps: worst tap {worst:.2e}, out {r_o:.2e}, grad from tap {r_gt:.2e}, full-depth grad {r_go:.2e}")
    assert r_o < 1e-2 and r_gt < 2e-2 and r_go < 2e-2


def test_full_size_batch_8_equals_eight_single_images():
    """generate.py's default batch (generate.py:52) through the full 421 M-parameter model: one batch-8 forward against the same
    eight (image, timestep) pairs one at a time through the same context.  The batch-8 launch profile must contain 128x128
    tiles on maps below 128 pixels (M = 8 * 4096 and 8 * 1024: the 64- and 32-wide maps), which no batch-1 launch selects.  Same products,
    another tile shape / split-K policy: relative L2 <= 2e-3 per image (the igemm2-vs-igemm4 bound of
    test_gpu_fullsize.py); the batch-1 results themselves are pinned to the oracle by C1."""
    import ctypes as C
    from ishapediting_amd import _lib
    from ishapediting_amd.unet import UNetModel
    from ishapediting_amd.unet_spec import full_config
    dev = torch.device("cuda", 0)
    cfg = full_config()
    m = UNetModel(cfg, dev, max_batch=8)
    m.load_state_dict(synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 1234)))
    g = torch.Generator().manual_seed(808)
    x = torch.randn(8, 96, 128, 128, generator=g).to(dev)
    ts = [999.0, 870.0, 641.0, 500.0, 333.0, 120.0, 37.0, 0.0]
    L = _lib.lib()
    L.ishap_profile_begin()
    out8 = m(x, ts, feat_layer=-1).clone()
    torch.cuda.synchronize()
    buf = C.create_string_buffer(1 << 20)
    assert L.ishap_profile_shapes(buf, len(buf)) > 0
    tot = (C.c_double * 48)()
    L.ishap_profile_end(tot, 16)
    lines = [l.split(",") for l in buf.value.decode().strip().split("\n")]
    big3 = {int(l[0]) for l in lines if int(l[3]) == 1 and int(l[4]) == 128}
    assert {8 * 4096, 8 * 1024} <= big3, big3          # (the 16-wide maps take them from batch 16 on: 16 x 6 tiles at batch 8 < 192; oracle test above)
    assert bool(torch.isfinite(out8).all())
    worst = 0.0
    for b in range(8):
        o1 = m(x[b:b + 1], ts[b:b + 1], feat_layer=-1)
        r = rel(out8[b:b + 1], o1)
        worst = max(worst, r)
        assert r < 2e-3, (b, r)
    print(f"full-size batch 8 vs eight single images: worst rel {worst:.2e}")
