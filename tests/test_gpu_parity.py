"""HIP path vs. the CPU oracle and the reference's golden vectors.  Needs an MI355X: -m gpu.
Every call goes through the C ABI of libishap_hip.so (ctypes); nothing here reads /root/reference."""
import ctypes as C

import numpy as np
import pytest
import torch

from ishapediting_amd import synthetic
from ishapediting_amd.unet_spec import build_spec, tiny_config

pytestmark = pytest.mark.gpu

T = torch.from_numpy


def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda", 0)


def rel_err(a, b):
    a = a.detach().float().cpu()
    b = torch.as_tensor(b).float()
    return float((a - b).norm() / (b.norm() + 1e-30)), float((a - b).abs().max())


# ---------------------------------------------------------------------------------------------- UNet forward
@pytest.mark.parametrize("nrb", [1, 2])
def test_tiny_unet_forward_vs_reference_golden(gold, nrb):
    """fp16-torso device path vs the reference's fp16 (CPU) run and its fp32 run of the same weights.
    Tolerance: relative L2 <= 5e-3 vs the fp16 reference run (both carry fp16 rounding noise),
    <= 1e-2 vs fp32."""
    from ishapediting_amd.unet import UNetModel
    g = gold("g4_tiny_unet")
    cfg = tiny_config(nrb)
    sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 100 + nrb))
    m = UNetModel(cfg, dev())
    m.load_state_dict(sd, strict=True)
    x = T(g[f"nrb{nrb}_x"]).to(dev())
    ts = T(g[f"nrb{nrb}_ts"])
    k = int(g[f"nrb{nrb}_tap_fp16_idx"])
    out, feat = m(x, ts, feat_layer=k)
    torch.cuda.synchronize()
    assert feat.dtype == torch.float16 and out.dtype == torch.float32
    r16, a16 = rel_err(out, g[f"nrb{nrb}_out_fp16"])
    r32, a32 = rel_err(out, g[f"nrb{nrb}_out"])
    f16, _ = rel_err(feat, g[f"nrb{nrb}_tap_fp16"])
    f32, _ = rel_err(feat, g[f"nrb{nrb}_tap{k}"])
    print(f"nrb={nrb}: out rel vs fp16-ref {r16:.2e} (max {a16:.2e}), vs fp32-ref {r32:.2e}; tap {f16:.2e}/{f32:.2e}")
    assert r16 < 5e-3 and f16 < 5e-3
    assert r32 < 1e-2 and f32 < 1e-2
    # every tap index, against the fp32 reference run
    for kk in range(len(build_spec(cfg).output_blocks)):
        _, ft = m(x, ts, feat_layer=kk)
        r, _ = rel_err(ft, g[f"nrb{nrb}_tap{kk}"])
        assert r < 1e-2, (kk, r)
    # feat_layer < 0 returns only the output (unet.py:668-669)
    o2 = m(x, ts)
    assert torch.is_tensor(o2) and torch.equal(o2, out)


def test_tiny_unet_param_table_matches_spec():
    from ishapediting_amd.unet import UNetModel
    from ishapediting_amd.unet_spec import param_shapes
    cfg = tiny_config(2)
    m = UNetModel(cfg, dev())
    assert m.param_table() == param_shapes(cfg)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 6, 16, 16, device=dev()), [0])      # weights not loaded -> loud failure
    sd = synthetic.unet_state_dict(cfg, 1)
    bad = dict(sd)
    bad.pop("out.2.bias")
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad, strict=True)


# ---------------------------------------------------------------------------------------------- diffusion step
def test_ddpm_step_vs_golden(gold):
    from ishapediting_amd.gaussian_diffusion import create_gaussian_diffusion
    g = gold("g2_steps")
    d = create_gaussian_diffusion(timestep_respacing="40")
    x, mo, noise, vn = (T(g[k]).to(dev()) for k in ("x", "model_output", "noise", "variance_noise"))

    class Fixed:
        def __call__(self, x, ts, feat_layer=-1, **kw):
            self.ts = ts
            return (mo, None) if feat_layer >= 0 else mo
    for t in (0, 1, 17, 39):
        f = Fixed()
        o = d.p_sample_guidance(f, x, torch.tensor([t]), noise=noise)
        assert int(f.ts[0]) == int(g[f"t{t}_ts"][0])
        for k in ("sample", "pred_xstart", "variance", "mean"):
            np.testing.assert_allclose(o[k].cpu().numpy(), g[f"t{t}_{k}"], rtol=2e-5, atol=2e-6, err_msg=f"{t} {k}")
        o2 = d.p_sample_guidance(f, x, torch.tensor([t]), noise=noise, clip_denoised=False)
        np.testing.assert_allclose(o2["sample"].cpu().numpy(), g[f"t{t}_sample_noclip"], rtol=2e-5, atol=1e-5)
        o3 = d.p_sample_guidance(f, x, torch.tensor([t]), variance_noise=vn)
        np.testing.assert_allclose(o3["sample"].cpu().numpy(), g[f"t{t}_sample_vn"], rtol=2e-5, atol=2e-6)
        o4 = d.p_sample(f, x, torch.tensor([t]), noise=T(g["psample_noise"]).to(dev()))
        np.testing.assert_allclose(o4["sample"].cpu().numpy(), g[f"t{t}_psample"], rtol=2e-5, atol=2e-6)


def test_step_draws_its_own_noise(gold):
    """ishap_step_coefs::rng (include/ishap.h): with no injected noise the step kernel draws th.randn_like(x) itself
    (gaussian_diffusion.py:493).  The noise it reports equals the documented construction -- Philox4x32-10, key = seed ^
    0x9E3779B97F4A7C15, counter = {vector index, offset}, two Box-Muller pairs -- to fp32 rounding of log / sin / cos (2e-5 abs);
    the step's outputs are BITWISE those of the same step with that noise injected; torch.manual_seed repeats it, consecutive
    steps differ; the moments are a standard normal's."""
    from ishapediting_amd import gaussian_diffusion as gdm
    assert gdm._STEP_RNG
    g = gold("g2_steps")
    d = gdm.create_gaussian_diffusion(timestep_respacing="40")
    x, mo = (T(g[k]).to(dev()) for k in ("x", "model_output"))

    class Fixed:
        def __call__(self, x, ts, feat_layer=-1, **kw):
            return (mo, None) if feat_layer >= 0 else mo
    torch.manual_seed(4242)
    gen = torch.cuda.default_generators[dev().index]
    seed, off = gen.initial_seed(), gen.get_offset()
    o = d.p_sample_guidance(Fixed(), x, torch.tensor([17]))
    assert gen.get_offset() == off + 4
    noise = o["noise"].cpu().numpy().reshape(-1)
    n4 = noise.size // 4
    vec = np.arange(n4, dtype=np.uint64)
    key = seed ^ 0x9E3779B97F4A7C15
    from tests.helpers import philox4x32_10
    w = philox4x32_10([vec & 0xFFFFFFFF, vec >> 32, np.full(n4, off & 0xFFFFFFFF), np.full(n4, off >> 32)], (key & 0xFFFFFFFF, key >> 32))
    u = [((v >> 8).astype(np.float64) + 0.5) * 2.0 ** -24 for v in w]
    want = np.empty((n4, 4))
    for h in range(2):
        rad = np.sqrt(-2.0 * np.log(u[2 * h]))
        want[:, 2 * h] = rad * np.cos(2 * np.pi * u[2 * h + 1])
        want[:, 2 * h + 1] = rad * np.sin(2 * np.pi * u[2 * h + 1])
    np.testing.assert_allclose(noise, want.reshape(-1), rtol=0, atol=2e-5)
    o_inj = d.p_sample_guidance(Fixed(), x, torch.tensor([17]), noise=o["noise"])
    for k in ("sample", "pred_xstart", "variance", "mean"):
        assert torch.equal(o[k], o_inj[k]), k
    o_next = d.p_sample_guidance(Fixed(), x, torch.tensor([17]))
    assert not torch.equal(o_next["noise"], o["noise"])
    torch.manual_seed(4242)
    o_again = d.p_sample_guidance(Fixed(), x, torch.tensor([17]))
    assert torch.equal(o_again["noise"], o["noise"]) and torch.equal(o_again["sample"], o["sample"])
    # p_sample / want_noise=False: no noise tensor, same values (same seed and offset)
    torch.manual_seed(4242)
    o_quiet = d.p_sample_guidance(Fixed(), x, torch.tensor([17]), want_noise=False)
    assert o_quiet["noise"] is None and torch.equal(o_quiet["sample"], o["sample"])
    # moments over 1.5 M draws of a full-size latent
    xb = torch.zeros(1, 96, 128, 128, device=dev())
    mb = torch.zeros(1, 192, 128, 128, device=dev())

    class Zero:
        def __call__(self, x, ts, feat_layer=-1, **kw):
            return (mb, None) if feat_layer >= 0 else mb
    z = d.p_sample_guidance(Zero(), xb, torch.tensor([20]))["noise"].double().reshape(-1)
    n = z.numel()
    assert abs(float(z.mean())) < 5.0 / n ** 0.5
    assert abs(float(z.var()) - 1.0) < 5.0 * (2.0 / n) ** 0.5
    assert abs(float((z ** 3).mean())) < 5.0 * (15.0 / n) ** 0.5
    assert abs(float((z ** 4).mean()) - 3.0) < 5.0 * (96.0 / n) ** 0.5
    assert float(z.abs().max()) < 6.5


# ---------------------------------------------------------------------------------------------- decoder
def test_decoder_points_vs_golden(gold):
    """fp32 MFMA decode vs the reference MultiTriplane (incl. border and out-of-range coords).
    Tolerance: |logit error| <= 1e-4 + 1e-4*|logit| (fp32 summation-order differences only)."""
    from ishapediting_amd.triplane_decoder import MultiTriplane
    g = gold("g6_decoder")
    dec = MultiTriplane(1, device=dev())
    dec.net.load_state_dict(synthetic.decoder_state_dict())
    planes = T(g["planes"]).to(dev())
    for i in range(3):
        dec.embeddings[i] = planes[[i]]
    logits = dec(0, T(g["coords"]).to(dev()).unsqueeze(0))
    assert logits.shape == (1, 2048, 1)
    np.testing.assert_allclose(logits.reshape(-1).cpu().numpy(), g["logits"], rtol=1e-4, atol=1e-4)


def test_decoder_grid_vs_oracle():
    from oracle import ref_cpu as O
    from ishapediting_amd.triplane_decoder import MultiTriplane, decode_volume
    net = synthetic.decoder_state_dict()
    gen = torch.Generator().manual_seed(3)
    latent = torch.randn(1, 96, 32, 32, generator=gen) * 0.5
    rng = torch.rand(1, 96, 1, 1, generator=gen) + 0.5
    mid = torch.randn(1, 96, 1, 1, generator=gen) * 0.1
    want = O.decode_volume(net, latent, rng, mid, 24)
    dec = MultiTriplane(1, device=dev())
    dec.net.load_state_dict(net)
    got = decode_volume(dec, latent.to(dev()), rng.to(dev()), mid.to(dev()), 24)
    assert got.shape == (24, 24, 24)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------------------------------------- drag loss
def _tap_from_planes(feat):
    """[3,Cc,W,W] fp32 (fp16-representable) -> NHWC fp16 tap [W*W][ld] + chmap, plane p channel c at p*Cc+c."""
    P, Cc, W, _ = feat.shape
    ld = ((P * Cc + 31) // 32) * 32
    tap = torch.zeros(W * W, ld, dtype=torch.float16)
    tap[:, :P * Cc] = feat.reshape(P * Cc, W * W).t().half()
    chmap = torch.arange(P * Cc, dtype=torch.int32).reshape(P, Cc)
    return tap, chmap, ld


@pytest.mark.parametrize("loss_type", ["l2", "l1"])
@pytest.mark.parametrize("cof", [0.0, 0.4])
def test_drag_loss_gradient_vs_golden(gold, loss_type, cof):
    """d loss / d feature from the HIP drag kernels vs the gradient the reference's own training()
    produced (autograd).  Tolerance: 1e-4 relative + 1e-9 absolute (fp32 atomics reorder sums)."""
    from oracle import ref_cpu as O
    from ishapediting_amd.drag_utils import DragKernels
    g = gold("g7_drag")
    edit, chmap, ld = _tap_from_planes(T(g["edit"]))
    orig, _, _ = _tap_from_planes(T(g["orig"]))
    dk = DragKernels(dev(), W=16, ld=ld, chmap=chmap, r=int(g["r1"]), voxel=float(g["voxel_size"]),
                     loss_type=loss_type)
    dk.setup(g["sources"], g["targets"], cof)
    e_d, o_d = edit.to(dev()), orig.to(dev())
    grad, loss = dk.loss_grad(e_d, o_d)
    torch.cuda.synchronize()
    got = grad.cpu()[:, :60].t().reshape(3, 20, 16, 16)
    want = g[f"{loss_type}_cof{cof}_grad"]
    np.testing.assert_allclose(got.numpy(), want, rtol=1e-4, atol=1e-9)
    setup = O.DragSetup(g["sources"], g["targets"], int(g["r1"]), float(g["voxel_size"]), 16)
    lw = O.drag_loss(T(g["edit"]), T(g["orig"]), setup, cof, loss_type)
    assert abs(float(loss.cpu()) - float(lw)) <= 1e-5 * abs(float(lw)) + 1e-8
    # the fused form the guided step uses (terms | gather + max | scale) equals the two separate calls bit for bit
    cot_a, sc_a = dk.scaled_cotangent()
    cot_a, sc_a, grad_a, loss_a = cot_a.clone(), sc_a.clone(), grad.clone(), loss.clone()
    slot = torch.zeros(1, dtype=torch.float32, device=dev())
    cot_b, sc_b = dk.loss_cotangent_ptr(e_d.data_ptr(), o_d.data_ptr(), loss_out=slot)
    torch.cuda.synchronize()
    assert torch.equal(cot_a, cot_b) and torch.equal(sc_a, sc_b) and torch.equal(grad_a, dk.grad) and torch.equal(loss_a, slot)
    assert int(dk.gfx.abs().max()) == 0 and int(dk.acc.abs().max()) == 0      # scratch left zero for the next call
    # mask bitmap equals the reference's complement sets
    touched = (dk.touched.cpu().reshape(3, 16, 16) & 1).bool()      # bit 0; bit 1 marks the motion scatter's footprint
    for p in range(3):
        assert torch.equal(~touched[p], setup.masks[p])


def test_decoder_points_loss_grad_vs_oracle_autograd():
    """decode_bwd.hip (loss = -BCEWithLogits(decoder(coords), gt), d loss / d planes) vs torch autograd on the oracle
    decoder.  Tolerance 1e-3 relative L2 (fp32 both sides; atomics reorder the scatter)."""
    from oracle import ref_cpu as O
    from ishapediting_amd.triplane_decoder import MultiTriplane
    net = synthetic.decoder_state_dict()
    gen = torch.Generator().manual_seed(9)
    planes = (torch.randn(3, 32, 16, 16, generator=gen) * 0.5).requires_grad_(True)
    coords = torch.rand(1024, 3, generator=gen) * 2.2 - 1.1
    gt = (torch.rand(1024, generator=gen) > 0.5).float()
    pred = O.decoder_forward(net, planes, coords)
    loss = -torch.nn.functional.binary_cross_entropy_with_logits(pred, gt)
    gp, = torch.autograd.grad(loss, planes)
    dec = MultiTriplane(1, device=dev())
    dec.net.load_state_dict(net)
    hwc = planes.detach().permute(0, 2, 3, 1).contiguous().to(dev())
    l, dplanes, logits = dec.points_loss_grad(hwc, coords.to(dev()), gt.to(dev()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(logits.cpu().numpy(), pred.detach().numpy(), rtol=1e-4, atol=1e-4)
    assert abs(float(l) - float(loss)) < 1e-5
    got = dplanes.cpu().permute(0, 3, 1, 2)
    r = float((got - gp).norm() / gp.norm())
    print(f"decoder backward rel err {r:.3e}")
    assert r < 1e-3


def test_noise2shape_batch_equals_single_samples():
    """generate.py's driver (image_sample.py:138-201) at batch 3 vs the same three samples one by one: the batched
    kernels (M = N*HW rows, per-image GroupNorm / attention) must not mix images."""
    from argparse import Namespace
    from ishapediting_amd import image_sample
    from ishapediting_amd.unet_spec import UNetConfig
    cfg = UNetConfig(image_size=16, in_channels=96, model_channels=32, out_channels=192, num_res_blocks=1,
                     attention_resolutions="8", channel_mult=(1, 2), num_head_channels=32)
    sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 202))
    Tn, B = 4, 3

    def args(bs):
        return Namespace(clip_denoised=True, num_samples=bs, batch_size=bs, use_ddim=False, model_path=None, stats_dir=None,
                         explicit_normalization=True, image_size=16, num_channels=32, num_res_blocks=1, num_heads=4,
                         num_heads_upsample=-1, num_head_channels=32, attention_resolutions="8", channel_mult="1,2",
                         dropout=0.1, class_cond=False, use_checkpoint=False, use_scale_shift_norm=True,
                         resblock_updown=True, use_fp16=True, use_new_attention_order=False, in_out_channels=96,
                         learn_sigma=True, diffusion_steps=1000, noise_schedule="linear", timestep_respacing=str(Tn),
                         use_kl=False, predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False)
    gen = torch.Generator().manual_seed(5)
    x0 = torch.randn(B, 96, 16, 16, generator=gen).to(dev())
    step = torch.randn(Tn, B, 96, 16, 16, generator=gen).to(dev())
    lo, hi = -np.linspace(0.5, 1.5, 96).astype(np.float32), np.linspace(1.0, 2.0, 96).astype(np.float32)
    full = image_sample.noise2shape(args(B), state_dict=sd, bounds=(lo, hi), noise=x0, step_noise=lambda i: step[i])
    assert full.shape == (B, 16, 16, 96) and np.isfinite(full).all()
    # images must not mix: permuting the batch permutes the result exactly (same kernels on both sides, so this is
    # bitwise; batch-vs-single would not be -- different row counts pick different kernels, and the 4-step sampler
    # amplifies last-bit differences by 1/sqrt(alpha_bar) ~ 157 at t = 999)
    perm = [2, 0, 1]
    full_p = image_sample.noise2shape(args(B), state_dict=sd, bounds=(lo, hi), noise=x0[perm],
                                      step_noise=lambda i: step[i][perm])
    np.testing.assert_array_equal(full_p, full[perm])
    # one UNet forward, batch 3 vs the three single images: relative L2 <= 5e-3 (fp16 torso, kernel choice may differ)
    from ishapediting_amd.unet import UNetModel
    m3, m1 = UNetModel(cfg, dev(), max_batch=B), UNetModel(cfg, dev())
    m3.load_state_dict(sd)
    m1.load_state_dict(sd)
    ts = [37.0, 501.0, 999.0]
    out3 = m3(x0, ts)
    for b in range(B):
        out1 = m1(x0[b:b + 1], ts[b:b + 1])
        r, _ = rel_err(out3[b:b + 1], out1.cpu().numpy())
        assert r < 5e-3, (b, r)


def test_small_map_gemm_path_vs_oracle():
    """A 64-channel configuration (K a multiple of 64 everywhere) so the 8x8 maps take the one-launch small-map GEMM
    kernel (csrc/igemm_skinny.hip) for their 1x1 layers and the LDS-DMA tiled kernel elsewhere -- the tiny golden
    configuration (32 channels) only reaches the register-staged kernel.  Forward and input gradient against the
    oracle (fp32, same fp16-rounded weights): relative L2 <= 1e-2 / 2e-2 as for the golden configuration."""
    from oracle import ref_cpu as O
    from ishapediting_amd.unet import UNetModel
    from ishapediting_amd.unet_spec import UNetConfig, build_spec
    cfg = UNetConfig(image_size=16, in_channels=6, model_channels=64, out_channels=12, num_res_blocks=1,
                     attention_resolutions="8", channel_mult=(1, 2), num_head_channels=64)
    sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 77))
    m = UNetModel(cfg, dev())
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 6, 16, 16, generator=g)
    ts = torch.tensor([321.0])
    net = O.UNetOracle(build_spec(cfg), sd, fp16=False)
    nblk = len(build_spec(cfg).output_blocks)
    k = nblk - 2
    xr = x.clone().requires_grad_(True)
    ref_out, ref_tap = net.forward(xr, ts, feat_layer=k)
    ct = torch.randn(ref_tap.shape, generator=g) * 0.1
    (ref_gx,) = torch.autograd.grad((ref_tap * ct).sum(), xr)
    out, tap = m(x.to(dev()), ts, feat_layer=k, keep_for_backward=True)
    r_out, _ = rel_err(out, ref_out.detach().numpy())
    r_tap, _ = rel_err(tap, ref_tap.detach().numpy())
    cot = ct[0].permute(1, 2, 0).reshape(-1, ct.shape[1]).contiguous().half().to(dev())
    gx = m.backward_input(cot)
    torch.cuda.synchronize()
    r_gx, _ = rel_err(gx, ref_gx.numpy())
    print(f"64-channel config: out {r_out:.2e} tap {r_tap:.2e} grad {r_gx:.2e}")
    assert r_out < 1e-2 and r_tap < 1e-2 and r_gx < 2e-2


def test_batched_forward_and_backward_on_the_64_channel_configuration():
    """Batch 2 through the LDS-DMA kernels, the fused skip concatenation, the epilogue statistics and the split gradient
    outputs (all indexed per image) against the two samples run one by one: relative L2 <= 5e-3 (kernel choice
    differs with the row count; no image may leak into the other -- a leak would be O(1))."""
    from ishapediting_amd.unet import UNetModel
    from ishapediting_amd.unet_spec import UNetConfig, build_spec
    cfg = UNetConfig(image_size=16, in_channels=6, model_channels=64, out_channels=12, num_res_blocks=1,
                     attention_resolutions="8", channel_mult=(1, 2), num_head_channels=64)
    sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 78))
    m2, m1 = UNetModel(cfg, dev(), max_batch=2), UNetModel(cfg, dev())
    m2.load_state_dict(sd)
    m1.load_state_dict(sd)
    g = torch.Generator().manual_seed(10)
    x = torch.randn(2, 6, 16, 16, generator=g).to(dev())
    ts = [123.0, 876.0]
    k = len(build_spec(cfg).output_blocks) - 2
    ch, sz = m1.tap_shape(k)
    cot = (torch.randn(2, sz * sz, ch, generator=g) * 0.1).half().to(dev())
    out2, _ = m2(x, ts, feat_layer=k, keep_for_backward=True)
    gx2 = m2.backward_input(cot)
    for b in range(2):
        o1, _ = m1(x[b:b + 1], ts[b:b + 1], feat_layer=k, keep_for_backward=True)
        g1 = m1.backward_input(cot[b:b + 1].contiguous())
        r_o, _ = rel_err(out2[b:b + 1], o1.cpu().numpy())
        r_g, _ = rel_err(gx2[b:b + 1], g1.cpu().numpy())
        print(f"batch element {b}: out {r_o:.2e} grad {r_g:.2e}")
        assert r_o < 5e-3 and r_g < 1e-2
