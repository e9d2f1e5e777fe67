"""The CPU oracle (oracle/ref_cpu.py) against golden vectors produced by the reference's own code
(tools/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from ishapediting_amd import synthetic
from ishapediting_amd.unet_spec import build_spec, full_config, param_shapes, tiny_config, is_torso_conv

T = torch.from_numpy


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("steps", [10, 40, 200, 256, 1000])
def test_schedule_tables(gold, steps):
    g = gold("g1_schedules")
    tb = O.Tables(str(steps))
    assert tb.timestep_map == g[f"T{steps}_timestep_map"].tolist()
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
              "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
              "posterior_mean_coef1", "posterior_mean_coef2"):
        np.testing.assert_array_equal(getattr(tb, k), g[f"T{steps}_{k}"])   # float64, bit-exact


def test_step_arithmetic(gold):
    g = gold("g2_steps")
    d = O.DiffusionOracle(O.Tables("40"))
    x, mo, noise, vn = (T(g[k]) for k in ("x", "model_output", "noise", "variance_noise"))
    for t in (0, 1, 17, 39):
        assert int(g[f"t{t}_ts"][0]) == d.tb.timestep_map[t]
        o = d.mean_variance_from_output(mo, x, t, True)
        nz = 0.0 if t == 0 else 1.0
        close(o["mean"], g[f"t{t}_mean"])
        close(o["variance"], g[f"t{t}_variance"])
        close(o["pred_xstart"], g[f"t{t}_pred_xstart"])
        close(o["mean"] + nz * torch.sqrt(o["variance"]) * noise, g[f"t{t}_sample"])
        o2 = d.mean_variance_from_output(mo, x, t, False)
        close(o2["mean"] + nz * torch.sqrt(o2["variance"]) * noise, g[f"t{t}_sample_noclip"], atol=1e-5)
        close(o["mean"] + vn, g[f"t{t}_sample_vn"])
        close(o["mean"] + nz * torch.exp(0.5 * o["log_variance"]) * T(g["psample_noise"]), g[f"t{t}_psample"])


def test_primitives(gold):
    g = gold("g3_primitives")
    t = T(g["temb_t"])
    close(O.timestep_embedding(t, 32), g["temb_32"])
    close(O.timestep_embedding(t, 256), g["temb_256"])
    for c in (512, 64, 96):
        out = O.resize_feat_align(T(g[f"rfa_in_{c}"]))
        np.testing.assert_array_equal(out.numpy(), g[f"rfa_out_{c}"])
    np.testing.assert_array_equal(O.make_offsets(2).numpy(), g["offsets_r2"])


@pytest.mark.parametrize("nrb", [1, 2])
def test_tiny_unet_forward_and_input_grad(gold, nrb):
    g = gold("g4_tiny_unet")
    cfg = tiny_config(nrb)
    spec = build_spec(cfg)
    sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 100 + nrb))
    net = O.UNetOracle(spec, sd, fp16=False)
    x = T(g[f"nrb{nrb}_x"])
    ts = T(g[f"nrb{nrb}_ts"])
    xx = x.clone().requires_grad_(True)
    out, taps = net.forward(xx, ts, all_taps=True)
    close(out, g[f"nrb{nrb}_out"], rtol=1e-4, atol=1e-5)
    gx, = torch.autograd.grad((out * T(g[f"nrb{nrb}_out_ct"])).sum(), xx, retain_graph=True)
    close(gx, g[f"nrb{nrb}_out_gx"], rtol=1e-3, atol=1e-5)
    for k, tap in enumerate(taps):
        close(tap, g[f"nrb{nrb}_tap{k}"], rtol=1e-4, atol=1e-5)
        gx, = torch.autograd.grad((tap * T(g[f"nrb{nrb}_tap{k}_ct"])).sum(), xx, retain_graph=True)
        close(gx, g[f"nrb{nrb}_tap{k}_gx"], rtol=1e-3, atol=1e-5)
    # fp16 torso: same precision contract as convert_to_fp16 (unet.py:618-624)
    net16 = O.UNetOracle(spec, sd, fp16=True)
    k = int(g[f"nrb{nrb}_tap_fp16_idx"])
    with torch.no_grad():
        o16, f16 = net16.forward(x, ts, feat_layer=k)
    close(o16, g[f"nrb{nrb}_out_fp16"], rtol=1e-3, atol=1e-3)
    close(f16.float(), g[f"nrb{nrb}_tap_fp16"], rtol=1e-3, atol=1e-3)


def test_decoder(gold):
    g = gold("g6_decoder")
    net = synthetic.decoder_state_dict()
    logits = O.decoder_forward(net, T(g["planes"]), T(g["coords"]))
    close(logits, g["logits"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("loss_type", ["l2", "l1"])
@pytest.mark.parametrize("cof", [0.0, 0.4])
def test_drag_loss_gradient(gold, loss_type, cof):
    g = gold("g7_drag")
    setup = O.DragSetup(g["sources"], g["targets"], int(g["r1"]), float(g["voxel_size"]), 16)
    edit = T(g["edit"]).clone().requires_grad_(True)
    loss = O.drag_loss(edit, T(g["orig"]), setup, cof, loss_type)
    gr, = torch.autograd.grad(loss, edit)
    close(gr, g[f"{loss_type}_cof{cof}_grad"], rtol=1e-4, atol=1e-8)
    assert g[f"{loss_type}_cof{cof}_progress"].tolist() == [0.0]


def _tiny_loop_objects():
    cfg = tiny_config(1)
    spec = build_spec(cfg)
    sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 101))
    return spec, O.UNetOracle(spec, sd, fp16=False)


def test_ddpm_inversion(gold):
    g = gold("g8_g9_tiny_loops")
    Tn, w_time, feat_layer, r1, B = g["meta"].tolist()
    _, net = _tiny_loop_objects()
    d = O.DiffusionOracle(O.Tables(str(Tn)))
    with torch.no_grad():
        inv = d.ddpm_inversion(net, T(g["inv_x0"]), w_time, list(T(g["inv_fwd_noise"])), feat_layer=feat_layer)
    close(inv["latent"], g["inv_latent"], rtol=1e-5, atol=1e-6)
    close(inv["sample"], g["inv_sample"], rtol=1e-4, atol=1e-5)
    # round-trip identity by construction (gaussian_diffusion.py:530-531): sample == x_0 up to rounding
    close(inv["sample"], g["inv_x0"], rtol=0, atol=1e-5)
    close(torch.stack(inv["variance_noise"]), g["inv_variance_noise"], rtol=1e-3, atol=1e-5)
    close(torch.stack(inv["variance"]), g["inv_variance"], rtol=1e-4, atol=1e-7)
    close(torch.stack(inv["inter_feat"]), g["inv_inter_feat"], rtol=1e-3, atol=1e-5)


def test_sampling_and_drag_loops(gold):
    g = gold("g8_g9_tiny_loops")
    Tn, w_time, feat_layer, r1, B = g["meta"].tolist()
    _, net = _tiny_loop_objects()
    d = O.DiffusionOracle(O.Tables(str(Tn)))
    ns = T(g["loop_noise_sampling"])
    noises = {Tn - 1 - k: ns[k] for k in range(Tn)}
    img, w, cache = O.sample_with_guidance_cache(d, net, T(g["loop_latent0"]), Tn, w_time, feat_layer, noises)
    close(w, g["loop_w"], rtol=1e-4, atol=1e-5)
    close(img, g["loop_final_unguided"], rtol=1e-3, atol=1e-4)
    close(torch.stack(cache), g["loop_guidance"], rtol=1e-3, atol=1e-4)
    setup = O.DragSetup(g["drag_sources"], g["drag_targets"], r1, 2.0 / 32, cache[0].shape[-1])
    dn = T(g["drag_noise"])
    dnoise = {w_time - 1 - k: dn[k] for k in range(w_time)}
    final, losses = O.drag_loop(d, net, T(g["loop_w"]), list(T(g["loop_guidance"])), setup, w_time, feat_layer,
                                50.0, 0.4, dnoise)
    close(final, g["drag_final"], rtol=1e-3, atol=1e-4)
    assert int(g["drag_stop_time"]) == 0
    np.testing.assert_allclose(g["drag_progress"], [1 - i / (w_time - 1.0) for i in range(w_time - 1, -1, -1)])


def test_full_model_key_table(gold):
    g = gold("g10_full_keys")
    shapes = param_shapes(full_config())
    assert sorted(shapes) == g["keys"].tolist()
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(g["n_params"])
    assert sorted(k for k in shapes if is_torso_conv(k)) == g["halved"].tolist()


def test_reconstruction_loop(gold):
    """train_triplane's guided loop (drag_utils.py:445-463) vs the run over the reference's own objects."""
    from ishapediting_amd.unet_spec import UNetConfig
    g = gold("g11_reconstruct")
    cfg = UNetConfig(image_size=16, in_channels=96, model_channels=32, out_channels=192, num_res_blocks=1,
                     attention_resolutions="8", channel_mult=(1, 2), num_head_channels=32)
    net = O.UNetOracle(build_spec(cfg), synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 202)), fp16=False)
    d = O.DiffusionOracle(O.Tables(str(int(g["T"]))))
    imgs, losses, grads = O.reconstruct_loop(d, net, synthetic.decoder_state_dict(), T(g["img0"]), T(g["range"]),
                                             T(g["middle"]), T(g["coords"]), T(g["gt"]), T(g["noise"]))
    close(torch.stack(losses), g["losses"], rtol=1e-4, atol=1e-6)
    close(torch.stack(grads), g["grads"], rtol=2e-3, atol=1e-7)
    close(torch.stack(imgs), g["imgs"], rtol=1e-3, atol=1e-4)


def test_surface_extraction_on_analytic_sphere():
    """The surface checker (oracle/surface_cpu.py) itself, on a sphere SDF: vertices lie on the sphere,
    the mesh is closed (every edge shared by two faces) and Chamfer(sphere, same sphere) ~ 0."""
    from oracle.surface_cpu import chamfer_distance, marching_tetrahedra, mc_vertices, smooth_simple
    res, r = 32, 9.3
    ax = torch.arange(res, dtype=torch.float32) - (res - 1) / 2
    x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
    vol = r - torch.sqrt(x * x + y * y + z * z)              # > 0 inside
    pv = mc_vertices(vol)
    rad = torch.linalg.norm(pv - (res - 1) / 2, dim=1)
    assert pv.shape[0] > 1000 and float((rad - r).abs().max()) < 0.05
    verts, faces = marching_tetrahedra(vol)
    rad = torch.linalg.norm(verts - (res - 1) / 2, dim=1)
    assert float((rad - r).abs().max()) < 0.08
    e = torch.cat([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]]).sort(dim=1).values
    _, counts = torch.unique(e, dim=0, return_counts=True)
    assert int(counts.min()) == 2 and int(counts.max()) == 2        # watertight
    assert chamfer_distance(pv, pv.clone(), 10 ** 6) < 1e-6          # no subsampling: identical sets
    shifted = pv + torch.tensor([0.5, 0.0, 0.0])
    assert 0.05 < chamfer_distance(pv, shifted, 2000) < 0.6
    assert verts.shape[0] - 3 * faces.shape[0] // 2 + faces.shape[0] == 2      # Euler characteristic of a sphere
    sm = smooth_simple(verts, faces, 10)
    rs = torch.linalg.norm(sm - (res - 1) / 2, dim=1)
    assert 0.9 * r < float(rs.mean()) < r and float(rs.std()) < 0.1      # Laplacian smoothing shrinks the sphere slightly


def _sub_sd(g, prefix, path):
    return {f"{path}.{k[len(prefix):]}": T(g[k]) for k in g.files if k.startswith(prefix)}


@pytest.mark.parametrize("case", ["plain", "chan", "up", "down"])
def test_resblock_primitive_vs_reference_class(gold, case):
    """the oracle's ResBlock against the reference's own ResBlock class standalone (unet.py:160-256):
    plain, channel change (1x1 skip), up (nearest x2 on both branches), down (AvgPool2d), FiLM scale/shift."""
    from ishapediting_amd.unet_spec import ResSpec
    g = gold("g3b_block_primitives")
    sd = _sub_sd(g, f"res_{case}_sd.", "blk")
    cin = sd["blk.in_layers.2.weight"].shape[1]
    cout = sd["blk.in_layers.2.weight"].shape[0]
    spec = ResSpec("blk", cin, cout, up=case == "up", down=case == "down")
    net = O.UNetOracle.__new__(O.UNetOracle)
    net.sd, net.fp16, net.dtype = sd, False, torch.float32
    y = net.resblock(spec, T(g[f"res_{case}_x"]), T(g["emb"]))
    close(y, g[f"res_{case}_y"], rtol=1e-4, atol=1e-5)


def test_attention_primitive_vs_reference_class(gold):
    """AttentionBlock + QKVAttentionLegacy (unet.py:299-305,337-354): heads split BEFORE q/k/v."""
    from ishapediting_amd.unet_spec import AttnSpec
    g = gold("g3b_block_primitives")
    sd = _sub_sd(g, "attn_sd.", "att")
    net = O.UNetOracle.__new__(O.UNetOracle)
    net.sd, net.fp16, net.dtype = sd, False, torch.float32
    y = net.attention(AttnSpec("att", 64, 2), T(g["attn_x"]))
    close(y, g["attn_y"], rtol=1e-4, atol=1e-5)


def test_groupnorm32_silu_with_large_group_means(gold):
    """GroupNorm32 (nn.py:16-18) + SiLU on fp16 inputs whose groups sit at |mean| = 100, std 0.1: a one-pass
    E[x^2] - E[x]^2 in fp32 would lose the variance here."""
    g = gold("g3b_block_primitives")
    y = torch.nn.functional.silu(O._gn(T(g["gn_x"]), T(g["gn_w"]), T(g["gn_b"])))
    close(y, g["gn_y"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("nrb", [1, 2])
def test_tiny_unet_block_outputs(gold, nrb):
    """Every block's output of the tiny UNet vs the reference's forward hooks (localises a mismatch to one block)."""
    g4, gb = gold("g4_tiny_unet"), gold("g4b_block_outputs")
    cfg = tiny_config(nrb)
    spec = build_spec(cfg)
    net = O.UNetOracle(spec, synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 100 + nrb)), fp16=False)
    rec = {}
    orig_block = net.block

    def tap(blk, h, emb):
        o = orig_block(blk, h, emb)
        rec[blk.name] = o
        return o
    net.block = tap
    with torch.no_grad():
        net.forward(T(g4[f"nrb{nrb}_x"]), T(g4[f"nrb{nrb}_ts"]))
    names = [f"in{i}" for i in range(len(spec.input_blocks))] + ["mid"] + [f"out{i}" for i in range(len(spec.output_blocks))]
    for nm, blk in zip(names, spec.all_blocks()):
        ref = T(gb[f"nrb{nrb}_{nm}"]).float()
        e = float((rec[blk.name] - ref).norm() / ref.norm())
        assert e < 1e-3, (nm, e)            # the fixture is stored in fp16 (2^-11 relative)


@pytest.mark.parametrize("B", [1, 3])
def test_generate_path_p_sample_loop(gold, B):
    """image_sample.py:173-192: p_sample_loop with the reference's own RNG order (initial randn, then randn_like per
    step), unnormalize, NHWC -- the oracle's p_sample chain on the redrawn noise."""
    from tests.helpers import small96_config, redraw_generate_noise
    g = gold("g12_generate")
    Tn = int(g["T"])
    init, steps = redraw_generate_noise(g, B)
    cfg = small96_config()
    net = O.UNetOracle(build_spec(cfg), synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 303)), fp16=False)
    d = O.DiffusionOracle(O.Tables(str(Tn)))
    img = init
    with torch.no_grad():
        for k, i in enumerate(range(Tn - 1, -1, -1)):
            img = d.p_sample(net, img, i, steps[k])["sample"]
    close(img, g[f"b{B}_sample"], rtol=1e-3, atol=1e-4)
    lo, hi = T(g["lower_bound"]).reshape(1, -1, 1, 1), T(g["upper_bound"]).reshape(1, -1, 1, 1)
    arr = (img * ((hi - lo) / 2) + (lo + hi) / 2).permute(0, 2, 3, 1)
    close(arr, g[f"b{B}_arr"], rtol=1e-3, atol=1e-4)


def test_marching_cubes_statement_topology():
    """The CPU statement of marching cubes (oracle/surface_cpu.py:marching_cubes, table derived in tools/make_mc_table.py):
    sphere -> Euler characteristic 2, torus -> 0, both closed and consistently oriented with outward normals, vertex count =
    the number of sign-changing grid edges; a noise volume (every ambiguous face configuration occurs) has no edge shared
    by more than two triangles."""
    from oracle.surface_cpu import marching_cubes, mc_vertices
    res = 32
    ax = torch.arange(res, dtype=torch.float32) - (res - 1) / 2
    x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
    for vol, chi in ((9.3 - torch.sqrt(x * x + y * y + z * z), 2),
                     (3.4 - torch.sqrt((torch.sqrt(x * x + y * y) - 9.0) ** 2 + z * z), 0)):
        v, f = marching_cubes(vol)
        assert v.shape[0] == mc_vertices(vol).shape[0]
        e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        _, counts = torch.unique(e.sort(dim=1).values, dim=0, return_counts=True)
        assert int(counts.min()) == 2 and int(counts.max()) == 2
        assert torch.unique(e[:, 0] * v.shape[0] + e[:, 1]).numel() == e.shape[0]
        assert v.shape[0] - counts.numel() + f.shape[0] == chi
        A, B, C = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
        assert float((A * torch.cross(B, C, dim=1)).sum()) > 0
    noise = torch.randn(12, 12, 12, generator=torch.Generator().manual_seed(0))
    v, f = marching_cubes(noise)
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).sort(dim=1).values
    _, counts = torch.unique(e, dim=0, return_counts=True)
    assert int(counts.max()) == 2 and v.shape[0] == mc_vertices(noise).shape[0]
    # tie rule: voxels exactly at the level sit with the values below it in BOTH statements (`value > level` is the cut)
    ties = torch.randint(-1, 2, (12, 12, 12), generator=torch.Generator().manual_seed(3)).float()
    v, f = marching_cubes(ties)
    assert v.shape[0] == mc_vertices(ties).shape[0]
    at = torch.zeros(4, 4, 4); at[1, 1, 1] = 1.0; at[2, 1, 1] = 0.0          # one voxel above, its neighbours at the level
    assert mc_vertices(at).shape[0] == 6 and marching_cubes(at)[0].shape[0] == 6
    assert mc_vertices(-at).shape[0] == 0                                     # one voxel below, the rest AT the level: no surface


@pytest.mark.parametrize("eta", [0.0, 0.7])
def test_ddim_sample_loop(gold, eta):
    """ddim_sample_loop (gaussian_diffusion.py:762-846, timestep_respacing='ddim8') vs the reference's own run."""
    from tests.helpers import small96_config, redraw_ddim_noise
    g = gold("g13_ddim")
    Tn = int(g["T"])
    init, steps = redraw_ddim_noise(g)
    cfg = small96_config()
    net = O.UNetOracle(build_spec(cfg), synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 404)), fp16=False)
    d = O.DiffusionOracle(O.Tables("ddim8"))
    assert d.tb.timestep_map == g["timestep_map"].tolist()
    img = init
    with torch.no_grad():
        for k, i in enumerate(range(Tn - 1, -1, -1)):
            img = d.ddim_sample(net, img, i, steps[k], eta=eta)["sample"]
    close(img, g[f"eta{eta}_sample"], rtol=1e-3, atol=1e-4)


# ------------------------------------------------------------------------------------------ round 3 fixtures
def _offset64_config():
    from ishapediting_amd.unet_spec import UNetConfig
    return UNetConfig(image_size=64, in_channels=6, model_channels=64, out_channels=12, num_res_blocks=1,
                      attention_resolutions="4", channel_mult=(1, 1, 2), num_head_channels=64)


def test_model_with_large_group_means_oracle_vs_reference(gold):
    """Golden G15 (i): the offset64 model whose GroupNorm inputs sit at group means 40-80x their spread, fp32: the oracle's
    output, two taps and both input gradients against the reference's."""
    g = gold("g15_large_group_means")
    cfg = _offset64_config()
    sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict_offset(cfg, 141, offset=6.0))
    net = O.UNetOracle(build_spec(cfg), sd, fp16=False)
    x, ts = T(g["x"]), T(g["ts"])
    xx = x.clone().requires_grad_(True)
    out, taps = net.forward(xx, ts, all_taps=True)
    k1, k2 = [int(v) for v in g["taps_k"]]
    rel = lambda a, b: float((a.detach().float() - T(np.asarray(b, np.float32))).norm() / T(np.asarray(b, np.float32)).norm())
    assert rel(out, g["f32_out"]) < 1e-4
    assert rel(taps[k1], g[f"f32_tap{k1}"]) < 1e-3 and rel(taps[k2], g[f"f32_tap{k2}"]) < 1e-3      # fixture taps are fp16-stored
    gx, = torch.autograd.grad((taps[k1] * T(g[f"tap{k1}_ct"])).sum(), xx, retain_graph=True)
    assert rel(gx, g[f"f32_tap{k1}_gx"]) < 1e-3
    gx, = torch.autograd.grad((out * T(g["out_ct"])).sum(), xx)
    assert rel(gx, g["f32_out_gx"]) < 1e-3
    assert float(np.median(g["ratio_median"])) > 20 and float(g["ratio_median"].max()) > 60   # what the fixture claims to exercise


def test_groupnorm32_1000x_forward_and_input_gradient(gold):
    """Golden G15 (ii): GroupNorm32 (+ SiLU) on a 32x32 map at |mean| = 100 / std 0.1, outputs and input gradients."""
    g = gold("g15_large_group_means")
    x = T(g["gn_x"]).float().requires_grad_(True)
    w, b, ct = T(g["gn_w"]), T(g["gn_b"]), T(g["gn_ct"]).float()
    y = O._gn(x, w, b)
    close(y, g["gn_y_plain"], rtol=1e-4, atol=1e-4)
    gx, = torch.autograd.grad((y * ct).sum(), x, retain_graph=True)
    close(gx, g["gn_gx_plain"], rtol=1e-3, atol=2e-3)
    ys = torch.nn.functional.silu(y)
    close(ys, g["gn_y_silu"], rtol=1e-4, atol=1e-4)
    gx, = torch.autograd.grad((ys * ct).sum(), x)
    close(gx, g["gn_gx_silu"], rtol=1e-3, atol=2e-3)


def test_fp16_torso_loop_fixtures_are_consistent_with_the_fp32_ones(gold):
    """Golden G14a-c hold the reference's fp16-torso runs of the G8 / G11 / G12 loops on the same inputs and seeds: same
    shapes, and at the distance from the fp32 runs that an fp16 torso explains (so a fixture generated from other inputs,
    or with the torso left in fp32, fails here)."""
    a, a16 = gold("g8_g9_tiny_loops"), gold("g14a_tiny_loops_fp16")
    np.testing.assert_array_equal(a["meta"], a16["meta"])
    np.testing.assert_allclose(a16["inv_latent"], a["inv_latent"], rtol=0, atol=1e-6)       # pure forward noising: no model
    for k in ("inv_variance", "inv_variance_noise", "inv_inter_feat", "loop_w", "drag_final"):
        assert a[k].shape == a16[k].shape
        r = float(np.linalg.norm(a16[k] - a[k]) / np.linalg.norm(a[k]))
        assert 1e-6 < r < 3e-2, (k, r)
    b, b16 = gold("g11_reconstruct"), gold("g14b_reconstruct_fp16")
    assert b16["imgs"].shape == b["imgs"].shape == b16["imgs_fp32_same_input"].shape
    np.testing.assert_allclose(b16["imgs_fp32_same_input"][0], b["imgs"][0], rtol=1e-5, atol=1e-5)   # same input at step 0
    c, c16 = gold("g12_generate"), gold("g14c_generate_fp16")
    for B in (1, 3):
        r = float(np.linalg.norm(c16[f"b{B}_arr"] - c[f"b{B}_arr"]) / np.linalg.norm(c[f"b{B}_arr"]))
        assert 1e-6 < r < 1e-2, (B, r)
