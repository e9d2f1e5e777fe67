#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own code on CPU.

Runs only in the build container (needs /root/reference); the fixtures it
writes are data (inputs + the reference's outputs) and are committed.  Nothing
under tests/ or the package reads /root/reference at run time.

Weights are not stored: they are re-derived from a seed by
ishapediting_amd.synthetic (same function here and in the tests), loaded into
the reference model with strict=True -- which also pins the state_dict key
table of ishapediting_amd.unet_spec against the reference module tree.
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "neural_field_diffusion"))

# inert stubs for packages that are not installed here and are not on the arithmetic path
for name in ("mcubes", "open3d", "blobfile", "mpi4py", "matplotlib", "matplotlib.pyplot"):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
_mpi = types.ModuleType("mpi4py.MPI")


class _Comm:
    rank = 0
    size = 1

    def Get_rank(self):
        return 0

    def Get_size(self):
        return 1

    def bcast(self, x, root=0):
        return x


_mpi.COMM_WORLD = _Comm()
sys.modules["mpi4py"].MPI = _mpi
sys.modules["mpi4py.MPI"] = _mpi
_WHICH = sys.argv[1:] or None      # fixture names given on the command line (the reference's argparse must not see them)
sys.argv = ["x"]

import torch  # noqa: E402

torch.set_num_threads(8)

from neural_field_diffusion.guided_diffusion import gaussian_diffusion as gd  # noqa: E402
from neural_field_diffusion.guided_diffusion import unet as ref_unet  # noqa: E402
from neural_field_diffusion.guided_diffusion.respace import SpacedDiffusion, space_timesteps  # noqa: E402
from neural_field_diffusion.guided_diffusion.script_util import create_model_and_diffusion  # noqa: E402
from neural_field_diffusion.guided_diffusion.nn import timestep_embedding  # noqa: E402
from triplane_decoder.axisnetworks import MultiTriplane  # noqa: E402
import drag_utils as ref_drag  # noqa: E402

from ishapediting_amd import synthetic  # noqa: E402
from ishapediting_amd.unet_spec import UNetConfig, full_config, param_shapes, tiny_config  # noqa: E402


def ref_model_and_diffusion(cfg: UNetConfig, respacing: str, use_fp16=False):
    return create_model_and_diffusion(
        image_size=cfg.image_size, class_cond=False, learn_sigma=True, num_channels=cfg.model_channels,
        num_res_blocks=cfg.num_res_blocks, channel_mult=",".join(str(m) for m in cfg.channel_mult),
        num_heads=4, num_head_channels=cfg.num_head_channels, num_heads_upsample=-1,
        attention_resolutions=cfg.attention_resolutions, dropout=0.1, diffusion_steps=1000,
        noise_schedule="linear", timestep_respacing=respacing, use_kl=False, predict_xstart=False,
        rescale_timesteps=False, rescale_learned_sigmas=False, use_checkpoint=False,
        use_scale_shift_norm=True, resblock_updown=True, use_fp16=use_fp16, use_new_attention_order=False,
        in_out_channels=cfg.in_channels)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def g1_schedules():
    out = {}
    for T in (10, 40, 200, 256, 1000):
        _, d = ref_model_and_diffusion(tiny_config(), str(T))
        out[f"T{T}_timestep_map"] = np.array(d.timestep_map)
        for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                  "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                  "posterior_mean_coef1", "posterior_mean_coef2"):
            out[f"T{T}_{k}"] = getattr(d, k)
    save("g1_schedules", **out)


class _FakeModel:
    """Returns a fixed model output so the step arithmetic is isolated from the UNet."""

    def __init__(self, out, feat):
        self.out, self.feat = out, feat

    def __call__(self, x, ts, feat_layer=-1, **kw):
        self.ts = ts
        return (self.out, self.feat) if feat_layer >= 0 else self.out


def g2_steps():
    g = torch.Generator().manual_seed(11)
    _, d = ref_model_and_diffusion(tiny_config(), "40")
    x = torch.randn(1, 6, 8, 8, generator=g)
    mo = torch.randn(1, 12, 8, 8, generator=g) * 1.5
    noise = torch.randn(1, 6, 8, 8, generator=g)
    vn = torch.randn(1, 6, 8, 8, generator=g)
    out = {"x": x, "model_output": mo, "noise": noise, "variance_noise": vn}
    for t in (0, 1, 17, 39):
        m = _FakeModel(mo, None)
        tt = torch.tensor([t])
        o = d.p_sample_guidance(m, x, tt, noise=noise)
        out[f"t{t}_ts"] = m.ts
        for k in ("sample", "pred_xstart", "variance", "mean", "model_output"):
            out[f"t{t}_{k}"] = o[k]
        o2 = d.p_sample_guidance(m, x, tt, noise=noise, clip_denoised=False)
        out[f"t{t}_sample_noclip"] = o2["sample"]
        o3 = d.p_sample_guidance(m, x, tt, variance_noise=vn)
        out[f"t{t}_sample_vn"] = o3["sample"]
        torch.manual_seed(5)
        o4 = d.p_sample(m, x, tt)
        out[f"t{t}_psample"] = o4["sample"]
    torch.manual_seed(5)
    out["psample_noise"] = torch.randn_like(x)
    save("g2_steps", **out)


def g3_primitives():
    g = torch.Generator().manual_seed(21)
    out = {}
    t = torch.tensor([0, 5, 999])
    out["temb_t"] = t
    out["temb_32"] = timestep_embedding(t, 32)
    out["temb_256"] = timestep_embedding(t, 256)
    # resize_feat_align (G5)
    f512 = torch.randn(1, 512, 4, 4, generator=g).half()
    out["rfa_in_512"] = f512.float()
    out["rfa_out_512"] = ref_drag.resize_feat_align(f512)
    f64 = torch.randn(1, 64, 4, 4, generator=g)
    out["rfa_in_64"] = f64
    out["rfa_out_64"] = ref_drag.resize_feat_align(f64)
    f96 = torch.randn(1, 96, 4, 4, generator=g)
    out["rfa_in_96"] = f96
    out["rfa_out_96"] = ref_drag.resize_feat_align(f96)
    out["offsets_r2"] = ref_drag.make_offsets(2, "cpu")
    save("g3_primitives", **out)


def load_ref_unet(cfg, seed, respacing, gain=1.0, fp16=False, sd=None):
    """fp16=True: the reference's own precision contract (create with use_fp16=True, then convert_to_fp16(), as
    generate.py:67 / drag_utils.py:51,232 do) -- the torso runs in half on the CPU (torch 2.10 supports it)."""
    model, diff = ref_model_and_diffusion(cfg, respacing, use_fp16=fp16)
    if sd is None:
        sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, seed, gain))
    assert set(sd) == set(model.state_dict()), "unet_spec key table differs from the reference module tree"
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    model.load_state_dict(sd, strict=True)
    if fp16:
        model.convert_to_fp16()
    model.eval()
    return model, diff


def g4_tiny_unet():
    out = {}
    for nrb in (1, 2):
        cfg = tiny_config(nrb)
        model, _ = load_ref_unet(cfg, 100 + nrb, "10")
        g = torch.Generator().manual_seed(31 + nrb)
        x = torch.randn(1, 6, 16, 16, generator=g)
        ts = torch.tensor([437])
        n_out = len(model.output_blocks)
        taps = []
        for k in range(n_out):
            xx = x.clone().requires_grad_(True)
            o, feat = model(xx, ts, feat_layer=k)
            # fixed functional of the tap: sum(feat * cotangent)
            ct = torch.randn(feat.shape, generator=torch.Generator().manual_seed(900 + k))
            gx, = torch.autograd.grad((feat * ct).sum(), xx)
            out[f"nrb{nrb}_tap{k}"] = feat
            out[f"nrb{nrb}_tap{k}_ct"] = ct
            out[f"nrb{nrb}_tap{k}_gx"] = gx
            taps.append(feat)
        xx = x.clone().requires_grad_(True)
        o = model(xx, ts)
        ct = torch.randn(o.shape, generator=torch.Generator().manual_seed(999))
        gx, = torch.autograd.grad((o * ct).sum(), xx)
        out[f"nrb{nrb}_x"] = x
        out[f"nrb{nrb}_ts"] = ts
        out[f"nrb{nrb}_out"] = o
        out[f"nrb{nrb}_out_ct"] = ct
        out[f"nrb{nrb}_out_gx"] = gx
        # fp16 torso (reference precision contract) on the same weights
        model.convert_to_fp16()
        model.dtype = torch.float16
        with torch.no_grad():
            o16, f16 = model(x, ts, feat_layer=n_out // 2)
        out[f"nrb{nrb}_out_fp16"] = o16
        out[f"nrb{nrb}_tap_fp16"] = f16.float()
        out[f"nrb{nrb}_tap_fp16_idx"] = n_out // 2
    save("g4_tiny_unet", **out)


def g6_decoder():
    dec = MultiTriplane(1, input_dim=3, output_dim=1, device="cpu")
    net_sd = synthetic.decoder_state_dict()
    assert set(net_sd) == set(dec.net.state_dict())
    dec.net.load_state_dict(net_sd)
    dec.eval()
    g = torch.Generator().manual_seed(41)
    planes = torch.randn(3, 32, 16, 16, generator=g) * 0.5
    coords = torch.rand(2048, 3, generator=g) * 2.4 - 1.2      # includes out-of-range
    coords[:8] = torch.tensor([[-1, -1, -1], [1, 1, 1], [-1, 1, 0], [0, 0, 0],
                               [1, -1, 1], [0.999, 0.5, -1], [1.0001, 0, 0], [-1.0001, 0.3, 0.3]])
    for i in range(3):
        dec.embeddings[i] = planes[[i]]
    with torch.no_grad():
        logits = dec(0, coords.unsqueeze(0)).reshape(-1)
    save("g6_decoder", planes=planes, coords=coords, logits=logits)


def g7_drag():
    g = torch.Generator().manual_seed(51)
    W, C, B, r1 = 16, 20, 2, 2
    orig = torch.randn(3, C, W, W, generator=g).half().float()          # fp16-representable: the device taps are fp16
    edit0 = (orig + 0.3 * torch.randn(3, C, W, W, generator=g)).half().float()
    src = np.array([[0.1, -0.2, 0.3], [-0.4, 0.5, 0.0]], dtype=np.float32)
    tgt = np.array([[0.3, -0.1, 0.2], [-0.3, 0.3, 0.93]], dtype=np.float32)   # second one hits the border
    voxel = 2.0 / 32
    out = {"orig": orig, "edit": edit0, "sources": src, "targets": tgt, "r1": r1, "voxel_size": voxel}

    # replay drag_utils.py:314-334 and :355-383 through the reference's own code path
    ds = object.__new__(ref_drag.DragStuff)
    ds.device = torch.device("cpu")
    ds.offset1 = ref_drag.make_offsets(r1, "cpu")
    ds.voxel_size = voxel

    class _Args:
        num_samples = 1
        w_time = 2
        feat_layer = 0
        loss_type = "l2"
    ds.args = _Args()
    for loss_type in ("l2", "l1"):
        for cof in (0.0, 0.4):
            ds.args.loss_type = loss_type
            cap = {}

            class _Diff:
                def p_sample_guidance(self, model, img, t, feat_layer=0):
                    e = img      # the "latent" IS the edit feature here, so grad wrt it is d loss / d edit
                    cap["e"] = e
                    return {"inter_feat": _Passthrough(e), "sample": torch.zeros_like(img),
                            "variance": torch.ones_like(img)}
            # resize_feat_align wants a [1,2c,H,W] tensor; bypass by monkeypatching on the module
            ds.diffusion = _Diff()
            ds.model = None
            ds.feature_guidance = [orig, orig]
            ds.w = edit0.clone()
            ds.get_mesh = lambda img=None, t=0, **kw: cap.setdefault("final", img)
            saved = ref_drag.resize_feat_align
            ref_drag.resize_feat_align = lambda f: f.t
            try:
                # one guided iteration, then the cooperative stop (drag_utils.py:337-339):
                # the loop breaks with stop_time=1 and hands img (= 0 + 1*scale*grad) to get_mesh
                prog = []
                for v in ds.training(src, tgt, scale=1.0, cof=cof):
                    prog.append(v)
                    ds.train_flag = False
            finally:
                ref_drag.resize_feat_align = saved
            # img_new = 0 + 1 * scale * grad  => grad wrt edit feature
            out[f"{loss_type}_cof{cof}_grad"] = cap["final"]
            out[f"{loss_type}_cof{cof}_progress"] = np.array(prog)
    save("g7_drag", **out)


class _Passthrough:
    def __init__(self, t):
        self.t = t


def g7b_drag_loss_values():
    """Loss values (not only gradients) via the reference formulas evaluated by torch on the same tensors
    are covered in g9; here we store the mask index sets the reference builds."""
    pass


def g8_g9_tiny_loops(fp16=False, name="g8_g9_tiny_loops"):
    cfg = tiny_config(1)
    T, w_time, feat_layer, r1, B = 6, 3, 1, 2, 2
    model, diff = load_ref_unet(cfg, 101, str(T), fp16=fp16)
    g = torch.Generator().manual_seed(61)
    out = {}
    # ---- G8: ddpm_inversion (noise via torch.manual_seed) ----
    x0 = torch.randn(1, 6, 16, 16, generator=g).clamp(-1, 1)
    torch.manual_seed(77)
    inv = diff.ddpm_inversion(model, x0, w_time, clip_denoised=True, feat_layer=feat_layer)
    torch.manual_seed(77)
    fwd_noise = [torch.randn_like(x0) for _ in range(w_time)]
    out["inv_x0"] = x0
    out["inv_fwd_noise"] = torch.stack(fwd_noise)
    out["inv_latent"] = inv["latent"]
    out["inv_sample"] = inv["sample"]
    out["inv_variance_noise"] = torch.stack(inv["variance_noise"])
    out["inv_variance"] = torch.stack(inv["variance"])
    out["inv_inter_feat"] = torch.stack(inv["inter_feat"])
    # ---- G9: update_latent_params + training through the reference's DragStuff methods ----
    ds = object.__new__(ref_drag.DragStuff)
    ds.device = torch.device("cpu")
    ds.model, ds.diffusion = model, diff

    class _Args:
        num_samples = 1
        num_steps = T
        image_size = 16
        clip_denoised = True
        loss_type = "l2"
    _Args.w_time = w_time
    _Args.feat_layer = feat_layer
    ds.args = _Args()
    ds.offset1 = ref_drag.make_offsets(r1, "cpu")
    ds.voxel_size = 2.0 / 32
    ds.feature_guidance = []
    ds.w = ds.w0 = None
    finals = []
    ds.get_mesh = lambda tri_feat=None, img=None, t=0: finals.append((tri_feat, img, t))
    latent0 = torch.randn(1, 6, 16, 16, generator=g)
    # per-step noise: the reference draws randn_like inside p_sample_guidance; pre-draw the same stream
    torch.manual_seed(123)
    noises = [torch.randn(1, 6, 16, 16) for _ in range(T)]          # consumed in loop order i=T-1..0
    torch.manual_seed(123)
    final_unguided = ds.update_latent_params(img=latent0)
    out["loop_latent0"] = latent0
    out["loop_noise_sampling"] = torch.stack(noises)                # [k] used at loop position k (i=T-1-k)
    out["loop_final_unguided"] = final_unguided
    out["loop_w"] = ds.w
    out["loop_guidance"] = torch.stack(ds.feature_guidance)
    src = np.array([[0.1, -0.2, 0.3], [-0.4, 0.5, 0.0]], dtype=np.float32)
    tgt = np.array([[0.3, -0.1, 0.2], [-0.3, 0.3, 0.2]], dtype=np.float32)
    torch.manual_seed(321)
    dnoise = [torch.randn(1, 6, 16, 16) for _ in range(w_time)]
    torch.manual_seed(321)
    prog = list(ds.training(src, tgt, scale=50.0, cof=0.4))
    out["drag_sources"], out["drag_targets"] = src, tgt
    out["drag_noise"] = torch.stack(dnoise)
    out["drag_progress"] = np.array(prog)
    out["drag_final"] = finals[-1][1]
    out["drag_stop_time"] = finals[-1][2]
    out["meta"] = np.array([T, w_time, feat_layer, r1, B])
    if fp16:        # the inputs are those of the fp32 fixture (same seeds): keep only what the fp16 torso changes
        out = {k: v for k, v in out.items() if k in ("inv_latent", "inv_sample", "inv_variance_noise", "inv_variance",
                                                     "inv_inter_feat", "loop_final_unguided", "loop_w", "loop_guidance",
                                                     "drag_final", "drag_stop_time", "drag_progress", "meta")}
    save(name, **out)


def small96_config():
    return UNetConfig(image_size=16, in_channels=96, model_channels=32, out_channels=192, num_res_blocks=1,
                      attention_resolutions="8", channel_mult=(1, 2), num_head_channels=32)


def g11_reconstruct(fp16=False, name="g11_reconstruct"):
    """Two steps of train_triplane's guided loop (drag_utils.py:445-463), restated line by line over the
    reference's own model / diffusion / MultiTriplane objects (the method itself needs Open3D for its sampling)."""
    import torch as th
    cfg = small96_config()
    T = 4
    model, diff = load_ref_unet(cfg, 202, str(T), fp16=fp16)
    dec = MultiTriplane(1, input_dim=3, output_dim=1, device="cpu")
    dec.net.load_state_dict(synthetic.decoder_state_dict())
    dec.eval()
    for p in dec.net.parameters():
        p.requires_grad = False
    g = th.Generator().manual_seed(71)
    # small normalisation range: keeps the random decoder's Fourier phases O(1) rad, so the fixture measures the
    # arithmetic and not the amplification of fp16 noise through sin/cos of 30-rad phases
    rng = (th.rand(1, 96, 1, 1, generator=g) + 0.5) * 0.04
    mid = th.randn(1, 96, 1, 1, generator=g) * 0.01
    img = th.randn(1, 96, 16, 16, generator=g)
    coords = th.rand(T, 2048, 3, generator=g) * 2 - 1
    gts = (th.rand(T, 2048, 1, generator=g) > 0.5).float()
    noises = th.randn(T, 1, 96, 16, 16, generator=g)
    out = {"img0": img, "range": rng, "middle": mid, "coords": coords, "gt": gts, "noise": noises, "T": T}
    scale = 600

    def one_step(m, d, img_in, k, i):
        img_in = img_in.detach().clone().requires_grad_(True)
        outs = d.p_sample_guidance(m, img_in, th.tensor([i]), noise=noises[k])
        predict_x0 = (outs["pred_xstart"] * rng + mid).reshape(3, 32, 16, 16)
        for j in range(3):
            dec.embeddings[j] = predict_x0[[j]]
        prediction = dec(0, coords[k].unsqueeze(0)).squeeze(0)
        loss = -th.nn.BCEWithLogitsLoss()(prediction, gts[k])
        loss.backward()
        grads1 = img_in.grad.clone().detach()
        with th.no_grad():
            new = (outs["sample"] + outs["variance"] * (scale * grads1)).clone().detach()
        return new, loss.detach(), grads1

    # the fp16 fixture also records, for every step, what the reference's fp32 model makes of the SAME input state: the
    # distance between the two is the reference's own one-step precision spread, the yardstick of the like-for-like test
    model32, diff32 = load_ref_unet(cfg, 202, str(T)) if fp16 else (None, None)
    imgs, losses, grads, imgs32 = [], [], [], []
    for k, i in enumerate(range(T - 1, -1, -1)):
        if fp16:
            imgs32.append(one_step(model32, diff32, img, k, i)[0])
        img, loss, grads1 = one_step(model, diff, img, k, i)
        imgs.append(img); losses.append(loss); grads.append(grads1)
    out["imgs"] = th.stack(imgs); out["losses"] = th.stack(losses); out["grads"] = th.stack(grads)
    if fp16:
        out["imgs_fp32_same_input"] = th.stack(imgs32)
        out = {k: out[k] for k in ("imgs", "losses", "grads", "T", "imgs_fp32_same_input")}
    save(name, **out)



def _seed_module(mod, seed):
    """Seeded non-zero values for every parameter of a reference module (zero_module tensors included)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in mod.state_dict().items():
        if v.dim() == 1 and ("in_layers.0" in k or "out_layers.0" in k or "norm" in k):
            t = (1.0 if k.endswith("weight") else 0.0) + 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith("bias"):
            t = 0.05 * torch.randn(v.shape, generator=g)
        else:
            t = torch.randn(v.shape, generator=g) / float(np.sqrt(np.prod(v.shape[1:])))
        sd[k] = t
    mod.load_state_dict(sd)
    mod.eval()
    return sd


def g3b_block_primitives():
    """the reference's own ResBlock (plain / channel change / up / down, FiLM) and AttentionBlock
    (legacy qkv order) classes, standalone, with seeded weights; GroupNorm32 + SiLU alone.  Inputs carry a per-group
    mean offset so a one-pass variance would show."""
    from neural_field_diffusion.guided_diffusion.nn import normalization
    g = torch.Generator().manual_seed(23)
    out = {}
    emb = torch.randn(2, 64, generator=g)
    out["emb"] = emb
    cases = {"plain": dict(channels=32, out_channels=32), "chan": dict(channels=64, out_channels=32),
             "up": dict(channels=32, out_channels=32, up=True), "down": dict(channels=32, out_channels=32, down=True)}
    for name, kw in cases.items():
        blk = ref_unet.ResBlock(kw["channels"], 64, 0.1, out_channels=kw["out_channels"], use_scale_shift_norm=True,
                                dims=2, use_checkpoint=False, up=kw.get("up", False), down=kw.get("down", False))
        sd = _seed_module(blk, 300 + len(out))
        x = torch.randn(2, kw["channels"], 8, 8, generator=g) + 3.0 * torch.randn(2, kw["channels"], 1, 1, generator=g)
        with torch.no_grad():
            y = blk(x, emb)
        out[f"res_{name}_x"] = x
        out[f"res_{name}_y"] = y
        for k, v in sd.items():
            out[f"res_{name}_sd.{k}"] = v
    att = ref_unet.AttentionBlock(64, num_heads=4, num_head_channels=32, use_checkpoint=False, use_new_attention_order=False)
    sd = _seed_module(att, 401)
    x = torch.randn(2, 64, 4, 4, generator=g)
    with torch.no_grad():
        y = att._forward(x)
    out["attn_x"], out["attn_y"] = x, y
    for k, v in sd.items():
        out[f"attn_sd.{k}"] = v
    gn = normalization(64)
    sd = _seed_module(gn, 402)
    x = (torch.randn(2, 64, 8, 8, generator=g) * 0.1 + 100.0 * torch.randn(2, 64, 1, 1, generator=g).sign()).half()
    with torch.no_grad():
        y = torch.nn.functional.silu(gn(x.float()))
    out["gn_x"], out["gn_y"], out["gn_w"], out["gn_b"] = x.float(), y, sd["weight"], sd["bias"]
    save("g3b_block_primitives", **out)


def g4b_block_outputs():
    """Every block's output (input_blocks[i], middle_block, output_blocks[i]) of the tiny UNet runs of G4 (fp32 reference),
    so a device mismatch is localised to one block."""
    out = {}
    for nrb in (1, 2):
        cfg = tiny_config(nrb)
        model, _ = load_ref_unet(cfg, 100 + nrb, "10")
        g = torch.Generator().manual_seed(31 + nrb)
        x = torch.randn(1, 6, 16, 16, generator=g)
        ts = torch.tensor([437])
        rec = {}
        hooks = []
        for i, b in enumerate(model.input_blocks):
            hooks.append(b.register_forward_hook(lambda m, a, o, i=i: rec.__setitem__(f"in{i}", o.detach().clone())))
        hooks.append(model.middle_block.register_forward_hook(lambda m, a, o: rec.__setitem__("mid", o.detach().clone())))
        for i, b in enumerate(model.output_blocks):
            hooks.append(b.register_forward_hook(lambda m, a, o, i=i: rec.__setitem__(f"out{i}", o.detach().clone())))
        with torch.no_grad():
            model(x, ts)
        for h in hooks:
            h.remove()
        for k, v in rec.items():
            out[f"nrb{nrb}_{k}"] = v.half()        # fp16 storage: the device tensors are fp16 and the tolerance is 1e-2
    save("g4b_block_outputs", **out)


def g12_generate(fp16=False, name="g12_generate"):
    """The generate path (image_sample.py:173-192): the reference's p_sample_loop with its own RNG draws (noise=None:
    th.randn(*shape) then randn_like per step, gaussian_diffusion.py:629,437), its unnormalize (normalization.py:6-15)
    and the NHWC permute, for batch 1 and 3 on small96_config."""
    import tempfile
    from neural_field_diffusion.guided_diffusion.normalization import unnormalize
    cfg = small96_config()
    T = 5
    model, diff = load_ref_unet(cfg, 303, str(T), fp16=fp16)
    g = torch.Generator().manual_seed(81)
    lower = -(torch.rand(96, generator=g) + 0.5).numpy().astype(np.float32)
    upper = (torch.rand(96, generator=g) + 0.5).numpy().astype(np.float32)
    out = {"T": T, "lower_bound": lower, "upper_bound": upper}
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "lower_bound.npy"), lower)
        np.save(os.path.join(d, "upper_bound.npy"), upper)
        for B in (1, 3):
            shape = (B, 96, 16, 16)
            torch.manual_seed(500 + B)
            sample = diff.p_sample_loop(model, shape, clip_denoised=True, model_kwargs={})
            torch.manual_seed(500 + B)
            init = torch.randn(*shape)
            steps = torch.stack([torch.randn(*shape) for _ in range(T)])     # consumed in loop order i = T-1 .. 0
            arr = unnormalize(sample, stats_dir=d).permute(0, 2, 3, 1).contiguous()
            # the noise itself is not stored (incompressible megabytes): tests redraw it from the same seed on the CPU
            # generator and check these float64 checksums before using it
            out[f"b{B}_seed"] = 500 + B
            out[f"b{B}_noise_check"] = np.array([float(init.double().sum()), float(steps.double().pow(2).sum()),
                                                  float(steps[-1, -1, -1, -1, -1])], dtype=np.float64)
            out[f"b{B}_sample"] = sample
            out[f"b{B}_arr"] = arr
    if fp16:
        out = {k: v for k, v in out.items() if k.endswith("_sample") or k.endswith("_arr") or k == "T"}
    save(name, **out)


def g13_ddim():
    """ddim_sample_loop (gaussian_diffusion.py:762-846) on small96_config with timestep_respacing='ddim8': eta = 0 (the
    deterministic sampler) and eta = 0.7 (per-step noise drawn under torch.manual_seed, redrawn by the tests)."""
    cfg = small96_config()
    model, diff = load_ref_unet(cfg, 404, "ddim8")
    out = {"T": diff.num_timesteps, "timestep_map": np.array(diff.timestep_map)}
    shape = (2, 96, 16, 16)
    for eta in (0.0, 0.7):
        torch.manual_seed(600)
        sample = diff.ddim_sample_loop(model, shape, clip_denoised=True, model_kwargs={}, eta=eta)
        out[f"eta{eta}_sample"] = sample
    torch.manual_seed(600)
    init = torch.randn(*shape)
    steps = torch.stack([torch.randn(*shape) for _ in range(diff.num_timesteps)])
    out["seed"] = 600
    out["noise_check"] = np.array([float(init.double().sum()), float(steps.double().pow(2).sum()),
                                   float(steps[-1, -1, -1, -1, -1])], dtype=np.float64)
    save("g13_ddim", **out)


def g14_fp16_loops():
    """the chained loops of G8/G9, G11 and G12 re-run by the reference with its own fp16 torso
    (model.convert_to_fp16(), unet.py:618-624 / fp16_util.py:14-21) on the SAME inputs and seeds, so the device path is
    compared like for like (fp16 torso vs fp16 torso) and not against an fp32 run."""
    g8_g9_tiny_loops(fp16=True, name="g14a_tiny_loops_fp16")
    g11_reconstruct(fp16=True, name="g14b_reconstruct_fp16")
    g12_generate(fp16=True, name="g14c_generate_fp16")


def offset64_config():
    """64 channels on 64^2 and 32^2 maps, 128 on 16^2 with attention: the 64^2 level runs the full-map GroupNorm route
    (statistics gathered as 64-bit fixed-point sums in the producing convolution's epilogue), the 32^2 / 16^2 levels the
    group-local route with several workgroups per (image, group)."""
    return UNetConfig(image_size=64, in_channels=6, model_channels=64, out_channels=12, num_res_blocks=1,
                      attention_resolutions="4", channel_mult=(1, 1, 2), num_head_channels=64)


def g15_large_group_means():
    """VERDICT r2 item 1b / ADVICE: GroupNorm inputs whose group mean is far from zero (nn.py:16-18 computes in fp32 on
    x.float(); a one-pass variance in fp32 cancels).  (i) offset64_config with conv biases that give every GroupNorm
    group a mean of 30-60 times its spread (synthetic.unet_state_dict_offset): output, two taps and the input gradient
    from one tap and from the output, from the reference in fp32 and with its fp16 torso.  (ii) GroupNorm32 (+ SiLU)
    alone on a 32x32 map with |mean| = 100, std 0.1 (1000x): output and the input gradient of sum(y * ct)."""
    from neural_field_diffusion.guided_diffusion.nn import normalization
    out = {}
    cfg = offset64_config()
    sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict_offset(cfg, 141, offset=6.0))
    g = torch.Generator().manual_seed(142)
    x = torch.randn(1, 6, 64, 64, generator=g)
    ts = torch.tensor([311])
    n_out = 3 * 2
    taps_k = (1, n_out - 2)
    for tag, fp16 in (("f32", False), ("f16", True)):
        model, _ = load_ref_unet(cfg, 0, "10", fp16=fp16, sd=sd)
        assert len(model.output_blocks) == n_out
        rec = {}
        hooks = []
        if not fp16:
            # ratio |group mean| / group std of every GroupNorm input, to document what the fixture exercises
            def probe(m, a, name):
                v = a[0].detach().float()
                N, C = v.shape[:2]
                gv = v.reshape(N, 32, -1)
                rec[name] = float((gv.mean(-1).abs() / (gv.std(-1) + 1e-12)).median())
            for name, mod in model.named_modules():
                if mod.__class__.__name__ == "GroupNorm32":
                    hooks.append(mod.register_forward_hook(lambda m, a, o, name=name: probe(m, a, name)))
        for k in taps_k:
            xx = x.clone().requires_grad_(True)
            o, feat = model(xx, ts, feat_layer=k)
            out[f"{tag}_tap{k}"] = feat.half()          # the device tap is fp16; tolerance 1e-2
            if k == taps_k[0]:
                ct = torch.randn(feat.shape, generator=torch.Generator().manual_seed(1400 + k)) * 0.1
                gx, = torch.autograd.grad((feat.float() * ct).sum(), xx)
                out[f"tap{k}_ct"] = ct
                out[f"{tag}_tap{k}_gx"] = gx
        for h in hooks:
            h.remove()
        xx = x.clone().requires_grad_(True)
        o = model(xx, ts)
        ct = torch.randn(o.shape, generator=torch.Generator().manual_seed(1499)) * 0.1
        gx, = torch.autograd.grad((o * ct).sum(), xx)
        out[f"{tag}_out"] = o
        out["out_ct"] = ct
        out[f"{tag}_out_gx"] = gx
        if not fp16:
            out["ratio_names"] = np.array(sorted(rec))
            out["ratio_median"] = np.array([rec[k] for k in sorted(rec)])
            print("median |mean|/std per GroupNorm input:", {k: round(v, 1) for k, v in rec.items()})
    out["x"], out["ts"], out["taps_k"] = x, ts, np.array(taps_k)
    # (ii) GroupNorm32 + SiLU alone, 1000x
    gn = normalization(64)
    sdg = _seed_module(gn, 1402)
    gg = torch.Generator().manual_seed(143)
    xg = (torch.randn(1, 64, 32, 32, generator=gg) * 0.1 + 100.0 * torch.randn(1, 64, 1, 1, generator=gg).sign()).half().float()
    ctg = torch.randn(1, 64, 32, 32, generator=gg).half().float()
    xr = xg.clone().requires_grad_(True)
    y = torch.nn.functional.silu(gn(xr.float()))
    gxg, = torch.autograd.grad((y * ctg).sum(), xr)
    yp = gn(xr.float())
    gxp, = torch.autograd.grad((yp * ctg).sum(), xr)
    out["gn_x"], out["gn_ct"], out["gn_w"], out["gn_b"] = xg.half(), ctg.half(), sdg["weight"], sdg["bias"]     # fp16-exact values
    out["gn_y_silu"], out["gn_gx_silu"], out["gn_y_plain"], out["gn_gx_plain"] = y, gxg, yp, gxp
    save("g15_large_group_means", **out)


def g10_full_keys():
    """Key table + parameter count of the full-size model (structure only, no tensors stored)."""
    cfg = full_config()
    model, _ = ref_model_and_diffusion(cfg, "200")
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    mine = param_shapes(cfg)
    assert shapes == mine, "full-size key/shape table mismatch"
    n = sum(int(np.prod(s)) for s in shapes.values())
    # which tensors convert_to_fp16 halves
    model.convert_to_fp16()
    halved = sorted(k for k, v in model.state_dict().items() if v.dtype == torch.float16)
    save("g10_full_keys", keys=np.array(sorted(shapes)), n_params=n, halved=np.array(halved))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = _WHICH
    todo = [g1_schedules, g2_steps, g3_primitives, g3b_block_primitives, g4_tiny_unet, g4b_block_outputs, g6_decoder, g7_drag,
            g8_g9_tiny_loops, g10_full_keys, g11_reconstruct, g12_generate, g13_ddim, g14_fp16_loops, g15_large_group_means]
    if which:
        todo = [f for f in todo if f.__name__ in which]
    with torch.no_grad():
        pass
    for fn in todo:
        print("==", fn.__name__)
        fn()
