"""Hand-written UNet backward + the whole drag loop vs the reference's own autograd results (golden)."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from ishapediting_amd import synthetic
from ishapediting_amd.unet_spec import build_spec, tiny_config

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def rel(a, b):
    a = a.detach().float().cpu()
    b = torch.as_tensor(b).float()
    return float((a - b).norm() / (b.norm() + 1e-30))


def nchw_to_tap(ct: torch.Tensor) -> torch.Tensor:
    """[1,C,S,S] -> resident tap layout [S*S, C] fp16."""
    return ct[0].permute(1, 2, 0).reshape(-1, ct.shape[1]).contiguous().half()


@pytest.mark.parametrize("nrb", [1, 2])
def test_input_gradient_from_every_tap(gold, nrb):
    """d sum(tap*ct)/dx vs autograd of the reference model (fp32).  Tolerance: relative L2 <= 2e-2
    (fp16 activations and fp16 gradient maps against an fp32 reference)."""
    from ishapediting_amd.unet import UNetModel
    g = gold("g4_tiny_unet")
    cfg = tiny_config(nrb)
    m = UNetModel(cfg, dev())
    m.load_state_dict(synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 100 + nrb)))
    x = T(g[f"nrb{nrb}_x"]).to(dev())
    ts = T(g[f"nrb{nrb}_ts"])
    for k in range(len(build_spec(cfg).output_blocks)):
        m(x, ts, feat_layer=k, keep_for_backward=True, want_inter_feat=False)
        cot = nchw_to_tap(T(g[f"nrb{nrb}_tap{k}_ct"])).to(dev())
        gx = m.backward_input(cot)
        torch.cuda.synchronize()
        r = rel(gx, g[f"nrb{nrb}_tap{k}_gx"])
        print(f"nrb={nrb} tap {k}: grad rel err {r:.3e}")
        assert r < 2e-2, (k, r)
    # loss-scaled path: tiny cotangent * 2^k in fp16, scale removed on exit
    k = 1
    m(x, ts, feat_layer=k, keep_for_backward=True, want_inter_feat=False)
    ct = T(g[f"nrb{nrb}_tap{k}_ct"])
    scale2 = torch.tensor([2.0 ** 20, 2.0 ** -20], device=dev())
    cot = nchw_to_tap(ct * 1e-6 * 2.0 ** 20).to(dev())
    gx = m.backward_input(cot, scale2)
    assert rel(gx, g[f"nrb{nrb}_tap{k}_gx"] * 1e-6) < 2e-2
    with pytest.raises(RuntimeError):
        m(x, ts, feat_layer=k)            # no keep_for_backward
        m.backward_input(cot)


@pytest.mark.parametrize("nrb", [1, 2])
def test_input_gradient_from_output(gold, nrb):
    from ishapediting_amd.unet import UNetModel
    g = gold("g4_tiny_unet")
    cfg = tiny_config(nrb)
    m = UNetModel(cfg, dev())
    m.load_state_dict(synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 100 + nrb)))
    x = T(g[f"nrb{nrb}_x"]).to(dev())
    m(x, T(g[f"nrb{nrb}_ts"]), feat_layer=-1, keep_for_backward=True)
    gx = m.backward_from_output(T(g[f"nrb{nrb}_out_ct"]).to(dev()))
    r = rel(gx, g[f"nrb{nrb}_out_gx"])
    print(f"nrb={nrb} full-depth grad rel err {r:.3e}")
    assert r < 2e-2


def _tiny_dragstuff(gold):
    from ishapediting_amd.drag_utils import DragStuff
    g = gold("g8_g9_tiny_loops")
    Tn, w_time, feat_layer, r1, B = g["meta"].tolist()
    args = Namespace(clip_denoised=True, num_samples=1, batch_size=1, use_ddim=False, num_steps=Tn, image_size=16,
                     num_channels=32, num_res_blocks=1, num_heads=4, num_heads_upsample=-1, num_head_channels=32,
                     attention_resolutions="8", channel_mult="1,2", dropout=0.1, class_cond=False, shape_resolution=32,
                     use_checkpoint=False, use_scale_shift_norm=True, resblock_updown=True, use_fp16=True,
                     use_new_attention_order=False, in_out_channels=6, learn_sigma=True, diffusion_steps=1000,
                     noise_schedule="linear", timestep_respacing=str(Tn), w_time=w_time, feat_layer=feat_layer,
                     loss_type="l2", use_kl=False, predict_xstart=False, rescale_timesteps=False,
                     rescale_learned_sigmas=False, explicit_normalization=False)
    ds = DragStuff(dev(), args=args)
    cfg = tiny_config(1)
    ds.model.load_state_dict(synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 101)))
    captured = []
    ds.get_mesh = lambda tri_feat=None, img=None, t=0: captured.append((tri_feat, img, t))
    return ds, g, captured, (Tn, w_time, feat_layer, r1, B)


def test_sampling_cache_and_drag_loop_vs_reference_run(gold):
    """The reference's own update_latent_params + training generator (run on CPU fp32, fixed noise) vs DragStuff
    on the device.  Tolerances (fp16 torso, 6 + 3 chained steps): latents 2e-2 relative L2."""
    from ishapediting_amd.drag_utils import resize_feat_align
    ds, g, captured, (Tn, w_time, feat_layer, r1, B) = _tiny_dragstuff(gold)
    ns = T(g["loop_noise_sampling"]).to(dev())
    ds.step_noise = lambda i: ns[Tn - 1 - i]
    final = ds.update_latent_params(img=g["loop_latent0"])
    torch.cuda.synchronize()
    assert rel(ds.w, g["loop_w"]) < 1e-2
    assert rel(final, g["loop_final_unguided"]) < 2e-2
    assert len(ds.feature_guidance) == w_time
    ch, sz = ds.model.tap_shape(feat_layer)
    for k, tap in enumerate(ds.feature_guidance):
        nchw = tap.reshape(sz, sz, ch).permute(2, 0, 1).unsqueeze(0).float()
        assert rel(resize_feat_align(nchw), g["loop_guidance"][k]) < 2e-2
    # drag: same handles, scale, cof and per-step noise as the reference run
    dn = T(g["drag_noise"]).to(dev())
    ds.step_noise = lambda i: dn[w_time - 1 - i]
    ds.set_offset1(r1)
    ds.voxel_size = 2.0 / 32
    # start from the reference's own w / cache so the comparison isolates the guided loop
    ds.w = T(g["loop_w"]).to(dev())
    prog = list(ds.training(g["drag_sources"], g["drag_targets"], scale=50.0, cof=0.4))
    torch.cuda.synchronize()
    np.testing.assert_allclose(prog, g["drag_progress"])
    tri, img, t = captured[-1]
    assert t == int(g["drag_stop_time"]) == 0
    r = rel(img, g["drag_final"])
    # what the guidance contributed, so a dead gradient cannot hide inside the tolerance
    ds.step_noise = lambda i: dn[w_time - 1 - i]
    list(ds.training(g["drag_sources"], g["drag_targets"], scale=0.0, cof=0.4))
    unguided = captured[-1][1]
    effect = rel(unguided, g["drag_final"])
    print(f"drag final latent rel err {r:.3e}; guidance effect {effect:.3e}")
    assert r < 2e-2
    assert effect > 5 * r, "guidance effect is not resolved by the tolerance"


def test_training_stop_flag(gold):
    """drag_utils.py:337-339,399: clearing train_flag stops after the current iteration and the remaining steps
    run unguided inside get_mesh(img, t=stop_time)."""
    ds, g, captured, (Tn, w_time, feat_layer, r1, B) = _tiny_dragstuff(gold)
    ds.update_latent_params(img=g["loop_latent0"])
    ds.set_offset1(r1)
    ds.voxel_size = 2.0 / 32
    got = []
    for v in ds.training(g["drag_sources"], g["drag_targets"], scale=50.0, cof=0.4):
        got.append(v)
        ds.train_flag = False
    assert got == [0.0]
    assert captured[-1][2] == w_time - 1
