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


def test_reconstruction_loop_vs_reference_run(gold):
    """DragStuff.reconstruct (decoder loss+backward kernel, x0 bridge, loss-scaled full-depth UNet backward) vs the
    same loop over the reference's objects (fp32 CPU).  Tolerance: loss 1e-3 rel, gradients / latents 2e-2 rel L2."""
    from ishapediting_amd.drag_utils import DragStuff
    g = gold("g11_reconstruct")
    Tn = int(g["T"])
    args = Namespace(clip_denoised=True, num_samples=1, batch_size=1, use_ddim=False, num_steps=Tn, image_size=16,
                     num_channels=32, num_res_blocks=1, num_heads=4, num_heads_upsample=-1, num_head_channels=32,
                     attention_resolutions="8", channel_mult="1,2", dropout=0.1, class_cond=False, shape_resolution=32,
                     use_checkpoint=False, use_scale_shift_norm=True, resblock_updown=True, use_fp16=True,
                     use_new_attention_order=False, in_out_channels=96, learn_sigma=True, diffusion_steps=1000,
                     noise_schedule="linear", timestep_respacing=str(Tn), w_time=2, feat_layer=1, loss_type="l2",
                     use_kl=False, predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False,
                     explicit_normalization=True)
    ds = DragStuff(dev(), args=args)
    from ishapediting_amd.unet_spec import UNetConfig
    cfg = UNetConfig(image_size=16, in_channels=96, model_channels=32, out_channels=192, num_res_blocks=1,
                     attention_resolutions="8", channel_mult=(1, 2), num_head_channels=32)
    ds.model.load_state_dict(synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 202)))
    ds.decoder.net.load_state_dict(synthetic.decoder_state_dict())
    ds.range, ds.middle = T(g["range"]).to(dev()), T(g["middle"]).to(dev())
    noise, coords, gts = T(g["noise"]).to(dev()), T(g["coords"]).to(dev()), T(g["gt"]).to(dev())
    ds.step_noise = lambda i: noise[Tn - 1 - i]
    # One step at a time from the reference's own latents: the first step (t = 999, pred_xstart = clamp(157*(x - eps)))
    # amplifies fp16 noise by two orders of magnitude, so a chained comparison would only measure that sensitivity.
    prev = T(g["img0"])
    for k in range(Tn):
        i = Tn - 1 - k
        img = ds.reconstruct(None, None, scale=600, img=prev, batch_fn=lambda ii: (coords[Tn - 1 - ii], gts[Tn - 1 - ii]),
                             steps=[i])
        torch.cuda.synchronize()
        loss = float(ds.last_losses[0])
        r = rel(img, g["imgs"][k])
        step_effect = rel(T(g["imgs"][k]), prev if k else g["img0"])
        print(f"step {k} (t={i}): loss {loss:.6f} vs {float(g['losses'][k]):.6f}; latent rel err {r:.3e}")
        assert abs(loss - float(g["losses"][k])) <= 1e-3 * abs(float(g["losses"][k]))
        if k >= 1:
            assert r < 2e-2, (k, r)
        else:
            # step 0 runs at t = 999: pred_xstart = clamp(157.1*x - 157.1*eps), so the fp16 torso's ~1e-3 relative error of
            # eps is amplified 157x before the clip (only the elements that stay inside [-1, 1] carry it on); measured 0.11
            assert r < 0.25, (k, r)
            assert r < 0.5 * step_effect, "step 0: the update itself must stay resolved"
        prev = T(g["imgs"][k])
    # the guidance term itself (scale 600 vs 0) must be resolved by that tolerance
    i = 1
    base = T(g["imgs"][Tn - 3])
    g_on = ds.reconstruct(None, None, scale=600, img=base, batch_fn=lambda ii: (coords[Tn - 1 - ii], gts[Tn - 1 - ii]), steps=[i])
    g_off = ds.reconstruct(None, None, scale=0, img=base, batch_fn=lambda ii: (coords[Tn - 1 - ii], gts[Tn - 1 - ii]), steps=[i])
    effect = rel(g_off, g["imgs"][Tn - 2])
    err = rel(g_on, g["imgs"][Tn - 2])
    print(f"guidance effect {effect:.3e} vs error {err:.3e}")
    assert effect > 5 * err


def test_x0_bridge_kernel_vs_autograd():
    """ishap_x0_grad_to_cotangent vs autograd of planes = clamp(sr*x - srm1*eps, -1, 1)*range + middle for a fixed eps."""
    import ctypes as C
    from ishapediting_amd import _lib
    gen = torch.Generator().manual_seed(13)
    S, sr, srm1 = 16, 1.7716, 1.4624
    x = torch.randn(1, 96, S, S, generator=gen)
    eps = torch.randn(1, 192, S, S, generator=gen)
    rng = torch.rand(96, generator=gen) + 0.5
    dplanes_chw = torch.randn(3, 32, S, S, generator=gen)            # d loss / d planes, reference layout
    xr = x.clone().requires_grad_(True)
    er = eps[:, :96].clone().requires_grad_(True)
    x0 = (sr * xr - srm1 * er).clamp(-1, 1)
    planes = (x0 * rng.reshape(1, 96, 1, 1)).reshape(3, 32, S, S)
    gx, ge = torch.autograd.grad((planes * dplanes_chw).sum(), (xr, er))
    d = dev()
    hwc = dplanes_chw.permute(0, 2, 3, 1).contiguous().to(d)
    g_direct = torch.empty(1, 96, S, S, device=d)
    cot = torch.empty(1, 192, S, S, device=d)
    rng_d, x_d, eps_d = rng.to(d), x.to(d), eps.to(d)      # keep the device tensors alive across the launch
    _lib.check(_lib.lib().ishap_x0_grad_to_cotangent(hwc.data_ptr(), rng_d.data_ptr(), x_d.data_ptr(), eps_d.data_ptr(),
                                                     sr, srm1, 1, S, g_direct.data_ptr(), cot.data_ptr(),
                                                     _lib.stream_ptr(d)))
    torch.cuda.synchronize()
    np.testing.assert_allclose(g_direct.cpu().numpy(), gx.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(cot[:, :96].cpu().numpy(), ge.numpy(), rtol=1e-5, atol=1e-6)
    assert float(cot[:, 96:].abs().max()) == 0.0
