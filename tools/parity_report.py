#!/usr/bin/env python3
"""Full-size parity report: a shortened drag edit (T DDPM steps, the last W of them recorded as guidance, then W
guided iterations, then the occupancy decode) on the device vs the fp32 CPU oracle with identical seeds, weights,
handles and injected noise.  Writes latent / logit errors, sign flips, marching-cubes vertex counts and the Chamfer
distance of meshProcess.py:18-35 to a JSON file (profiles/).  Test infrastructure (imports the oracle)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def surface_report(vol_gpu, vol_cpu, va, vb, res, dev, extract_surface, mesh_chamfer):
    """the device marching cubes on both volumes: vertex count (= the checker's sign-changing grid edges), closedness, and
    calc_chamfer as the reference defines it -- 20 000 points sampled uniformly by area on each surface"""
    out = {}
    meshes = []
    for name, vol in (("device", vol_gpu), ("oracle", vol_cpu)):
        v, t = extract_surface(vol.to(dev), 0.0, method="marching_cubes")
        meshes.append((v / res * 2 - 1, t))
        e = torch.cat([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]]).long()
        key = torch.minimum(e[:, 0], e[:, 1]) * (v.shape[0] + 1) + torch.maximum(e[:, 0], e[:, 1])
        _, cnt = torch.unique(key, return_counts=True)
        out[f"device_mc_vertices_on_{name}_volume"] = int(v.shape[0])
        out[f"device_mc_triangles_on_{name}_volume"] = int(t.shape[0])
        out[f"device_mc_open_edges_on_{name}_volume"] = int((cnt == 1).sum())      # edges on the volume's faces only
    out["device_mc_equals_checker_vertex_count"] = out["device_mc_vertices_on_device_volume"] == int(va.shape[0])
    out["device_mc_equals_checker_vertex_count_on_oracle_volume"] = out["device_mc_vertices_on_oracle_volume"] == int(vb.shape[0])
    # sampled by area at the reference's default 20 000 points: on a surface of this area the value IS its sampling floor
    # (reported next to it) -- a floor, not a parity measurement
    out["chamfer_area_uniform_20k_FLOOR_LIMITED"] = mesh_chamfer(meshes[0], meshes[1], 20000)
    out["chamfer_area_uniform_20k_floor"] = mesh_chamfer(meshes[1], (meshes[1][0].clone(), meshes[1][1].clone()), 20000, seed=1)
    return out


def surface_area(mesh):
    v, t = mesh
    t = t.long()
    return float(0.5 * torch.cross(v[t[:, 1]] - v[t[:, 0]], v[t[:, 2]] - v[t[:, 0]], dim=1).norm(dim=1).sum())


def lowpass(lat, k):
    """box filter k x k (average pool + bilinear upsample): a LINEAR map of the latent, applied to both runs alike"""
    import torch.nn.functional as F
    return F.interpolate(F.avg_pool2d(lat, k), scale_factor=k, mode="bilinear", align_corners=False)


def shape_like_report(dec_sd, dev, res, extract_surface, mesh_chamfer, chamfer_distance, mc_vertices, edit_dev=None, edit_cpu=None,
                      point_num=500000, smooth=0.005, amp=0.0005, lowpass_k=0):
    """The decode + surface comparison on a SHAPE-LIKE level set.  A smooth, low-amplitude triplane latent makes the
    random-weight decoder a smooth function of position; the EDIT's own results (the final latents of the device run and
    of the oracle run: sampling + guided iterations with the same seeds, weights and handles) ride on it as a small
    perturbation, so the two surfaces differ by exactly what the edit path's numerics differ by -- the random-weight UNet
    alone gives a volume-filling noise surface on which any sampled Chamfer sits on its sampling floor.
    Chamfer is calc_chamfer's (meshProcess.py:18-35): `point_num` points drawn uniformly by area on each mesh, the two
    mean squared nearest-neighbour distances added; its sampling floor for a surface of area A is ~ 2 A / (pi N) per
    direction pair, so on this surface (A ~ 47) N = 500 000 puts the floor below the 1e-4 target (20 000, the reference's default, sits at 1e-3)."""
    from ishapediting_amd.triplane_decoder import MultiTriplane, decode_volume
    from oracle import ref_cpu as O
    S = 128
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, S), torch.linspace(-1, 1, S), indexing="ij")
    g = torch.Generator().manual_seed(17)
    lat = torch.zeros(1, 96, S, S)
    for c in range(96):                                  # a few low-frequency modes per channel
        a, b, p, q = torch.randn(4, generator=g)
        lat[0, c] = smooth * (a * torch.cos(1.5 * xx + p) + b * torch.cos(1.5 * yy + q))
    # amp: the edit's latent (|x| <= 1) rides on the smooth field at a tenth of its amplitude
    if lowpass_k:
        # FULL-AMPLITUDE variant: the edit's final latent itself, low-passed so that the random-weight decoder sees a smooth
        # field (a shape-like level set); nothing is scaled down, low-frequency differences of the two runs pass 1:1
        lat_dev = amp * lowpass(edit_dev.float().cpu(), lowpass_k)
        lat_cpu = amp * lowpass(edit_cpu.float().cpu(), lowpass_k)
    else:
        lat_dev = lat + (amp * edit_dev.float().cpu() if edit_dev is not None else 0)
        lat_cpu = lat + (amp * edit_cpu.float().cpu() if edit_cpu is not None else 0)
    dec = MultiTriplane(1, device=dev)
    dec.net.load_state_dict(dec_sd)
    vg = decode_volume(dec, lat_dev.to(dev), 1.0, 0.0, res)
    torch.cuda.synchronize()
    vc = O.decode_volume(dec_sd, lat_cpu, 1.0, 0.0, res)
    level = float(vc.median())                           # cut the field where it splits the volume in two
    meshes, out = [], {}
    for name, vol in (("device", vg), ("oracle", vc.to(dev))):
        v, t = extract_surface(vol, level, method="marching_cubes")
        meshes.append((v / res * 2 - 1, t))
        out[f"vertices_{name}_volume"] = int(v.shape[0])
        out[f"triangles_{name}_volume"] = int(t.shape[0])
    va = mc_vertices(vg.cpu(), level) / res * 2 - 1
    vb = mc_vertices(vc, level) / res * 2 - 1
    area = surface_area(meshes[1])
    out.update({
        "what": (f"decode of {amp} x lowpass_{lowpass_k}(the edit's final latent): the edit at full relative amplitude on a smooth field" if lowpass_k else
                 f"decode of (smooth triplane of amplitude {smooth} + {amp} x the edit's final latent): the edit is ATTENUATED to {amp / smooth:.2f} of the carrier, so this "
                 "case checks decode + surface extraction, not the edit path" if edit_dev is not None else "decode of a smooth triplane"),
        "device_mc_equals_checker_vertex_count": out["vertices_device_volume"] == int(va.shape[0]),
        "device_mc_equals_checker_vertex_count_on_oracle_volume": out["vertices_oracle_volume"] == int(vb.shape[0]),
        "latent_rel_l2_after_filter": float((lat_dev - lat_cpu).norm() / lat_cpu.norm()),
        "res": res, "level": level, "logit_max_abs_err": float((vg.cpu() - vc).abs().max()), "logit_rms": float(vc.pow(2).mean().sqrt()),
        "sign_flips_about_level": int(((vg.cpu() > level) != (vc > level)).sum()),
        "checker_vertex_count_device_volume": int(va.shape[0]), "checker_vertex_count_oracle_volume": int(vb.shape[0]),
        "surface_area": area,
        "chamfer_all_vertices": chamfer_distance(va.to(dev), vb.to(dev), None),
        "chamfer_area_uniform_20k_FLOOR_LIMITED": mesh_chamfer(meshes[0], meshes[1], 20000),
        "chamfer_area_uniform_20k_floor": mesh_chamfer(meshes[1], (meshes[1][0].clone(), meshes[1][1].clone()), 20000, seed=1),
        f"chamfer_area_uniform_{point_num // 1000}k_FLOOR_LIMITED": mesh_chamfer(meshes[0], meshes[1], point_num),
        f"chamfer_area_uniform_{point_num // 1000}k_floor": mesh_chamfer(meshes[1], (meshes[1][0].clone(), meshes[1][1].clone()), point_num, seed=1),
        f"chamfer_floor_estimate_2A_over_piN_{point_num // 1000}k": 2.0 * area / (3.141592653589793 * point_num),
    })
    return out


# ------------------------------------------------------------------------------------------------------------------
# --c4: BASELINE configs[3] at FULL length (drag_utils.py:401-471, :552-566): 200 reconstruction steps x 40 000 occupancy
# samples with injected batches (`batch_fn`), DDPM inversion over w_time = 170, 170 guided drag iterations, 256^3 decode.
# The fp32 CPU oracle needs ~1 h of host time for this chain (every reconstruction / drag step is a full forward +
# autograd backward of the 421 M-parameter model), more than one GPU lease lasts: the two sides run as two STAGES from
# the same seeds -- `--c4 device` on the MI355X writes its stage results (a few MB: latents, losses, sign bits and a
# subsample of the volume) to an .npz, `--c4 oracle` (any host, no GPU) re-creates the inputs from the seeds, runs the
# oracle stage by stage FROM THE DEVICE'S STAGE INPUTS (so each tolerance measures one stage, as the shortened chain in
# tests/test_gpu_fullsize.py does) and writes the report.
# ------------------------------------------------------------------------------------------------------------------
from ishapediting_amd.synthetic import C4_POINTS, C4_RES, C4_T, C4_W, c4_inputs  # noqa: E402  (shared with tools/make_c4_fixture.py and the GPU test)


def c4_device(out_path):
    from ishapediting_amd import synthetic
    from ishapediting_amd.drag_utils import DragStuff, get_args
    from ishapediting_amd.unet_spec import full_config
    dev = torch.device("cuda", 0)
    T, W, res = C4_T, C4_W, C4_RES
    img0, batch, noise = c4_inputs()
    sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict(full_config(), 1234))
    dec_sd = synthetic.decoder_state_dict(4321)
    lo, hi = -0.05 * np.ones(96, np.float32), 0.05 * np.ones(96, np.float32)
    src, tgt = synthetic.handles(3)
    d = DragStuff(dev, args=get_args(["--w_time", str(W), "--num_steps", str(T), "--shape_resolution", str(res)]))
    d.load_weights(sd, dec_sd, lo, hi)
    t0 = time.time()
    d.step_noise = lambda i: noise(0, T - 1 - i).to(dev)                 # position k = T - 1 - i of the oracle's loop
    bf = lambda i: tuple(t.to(dev) for t in batch(T - 1 - i))
    rec = d.reconstruct(None, None, scale=600, img=img0, batch_fn=bf)
    loss_rec = [float(l) for l in d.last_losses]
    d.clear_params()
    d.get_mesh(tri_feat=rec)                                             # train_triplane decodes the reconstruction (drag_utils.py:464-465)
    vol_rec_bits = np.packbits((d.volume > 0).cpu().numpy().reshape(-1))
    d.latent_inversion(rec, fwd_noise=[noise(1, k).to(dev) for k in range(W)])
    w_dev = d.w.clone()
    vn = torch.stack(d.variance_noise)                                   # [W, 1, 96, 128, 128], loop order (i = W-1 .. 0)
    vn_norm = vn.flatten(1).norm(dim=1).cpu().numpy()
    vn_keep = vn[[0, W // 2, W - 1]].half().cpu().numpy()
    d.step_noise = lambda i: noise(2, W - 1 - i).to(dev)
    for _ in d.training(src, tgt, scale=1200.0, cof=0.4):
        pass
    torch.cuda.synchronize()
    secs = time.time() - t0
    vol = d.volume.cpu()
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    np.savez_compressed(out_path, rec=rec.cpu().numpy(), loss_rec=np.array(loss_rec), w=w_dev.cpu().numpy(), vn_norm=vn_norm, vn_keep=vn_keep,
                        final=d.tri_feat.cpu().numpy(), loss_drag=np.array([float(l) for l in d.last_losses]),
                        vol_bits=np.packbits((vol > 0).numpy().reshape(-1)), vol_rec_bits=vol_rec_bits,
                        vol_sub=vol[::4, ::4, ::4].numpy().astype(np.float32), seconds=np.array(secs))
    print(json.dumps({"c4_device": out_path, "seconds_device": round(secs, 2), "loss_rec_first_last": [loss_rec[0], loss_rec[-1]]}))


def c4_oracle(dev_path, out_path, threads):
    from ishapediting_amd import synthetic
    from ishapediting_amd.unet_spec import build_spec, full_config
    from oracle import ref_cpu as O
    torch.set_num_threads(threads)
    T, W, res = C4_T, C4_W, C4_RES
    g = np.load(dev_path)
    img0, batch, noise = c4_inputs()
    cfg = full_config()
    sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 1234))
    dec_sd = synthetic.decoder_state_dict(4321)
    lo, hi = -0.05 * np.ones(96, np.float32), 0.05 * np.ones(96, np.float32)
    rng = torch.from_numpy((hi - lo) / 2).reshape(1, 96, 1, 1)
    mid = torch.from_numpy((hi + lo) / 2).reshape(1, 96, 1, 1)
    src, tgt = synthetic.handles(3)
    net = O.UNetOracle(build_spec(cfg), sd, fp16=False)
    diff = O.DiffusionOracle(O.Tables(str(T)))
    rel = lambda x, y: float((torch.as_tensor(x).float() - y).norm() / y.norm())
    t0 = time.time()
    say = lambda m: print(f"[{time.time() - t0:6.0f} s] {m}", file=sys.stderr, flush=True)
    # ---- stage 1: reconstruction, 200 steps, from the common img0 (the one stage both sides start identically) ----
    class Lazy:                          # reconstruct_loop indexes coords[k] / gts[k] / noises[k]
        def __init__(self, f): self.f = f
        def __getitem__(self, k): return self.f(k)
    coords, gts, nz = Lazy(lambda k: batch(k)[0]), Lazy(lambda k: batch(k)[1]), Lazy(lambda k: noise(0, k))
    img, loss_rec_ref = img0, []
    for k, i in enumerate(range(T - 1, -1, -1)):
        imgs, losses, _ = O.reconstruct_loop(diff, net, dec_sd, img, rng, mid, Lazy(lambda _k, k=k: coords[k]), Lazy(lambda _k, k=k: gts[k]),
                                             Lazy(lambda _k, k=k: nz[k]), scale=600.0, steps=[i])
        img = imgs[-1]
        loss_rec_ref.append(float(losses[-1]))
        if k % 10 == 0: say(f"reconstruction step {k} / {T}: loss {loss_rec_ref[-1]:.5f} (device {float(g['loss_rec'][k]):.5f})")
    rec_ref = img
    rec_dev = torch.from_numpy(g["rec"])
    vol_rec_ref = O.decode_volume(dec_sd, rec_dev, rng, mid, res)        # decode of the DEVICE's reconstruction: the decode stage alone
    rec_flips = int((np.unpackbits(g["vol_rec_bits"])[:vol_rec_ref.numel()].astype(bool) != (vol_rec_ref > 0).numpy().reshape(-1)).sum())
    # ---- stage 2: inversion over W from the device's reconstruction ----
    with torch.no_grad():
        inv = diff.ddpm_inversion(net, rec_dev, W, Lazy(lambda k: noise(1, k)), feat_layer=8)
    say("inversion done")
    w_dev = torch.from_numpy(g["w"])
    vn_ref = torch.stack(inv["variance_noise"])
    vn_norm_ref = vn_ref.flatten(1).norm(dim=1).numpy()
    vn_keep_ref = vn_ref[[0, W // 2, W - 1]]
    cache = [O.resize_feat_align(f) for f in inv["inter_feat"]]
    del inv["inter_feat"]
    # ---- stage 3: 170 guided drag iterations from the device's w with the oracle's own guidance cache ----
    setup = O.DragSetup(src, tgt, 12, 2.0 / res, cache[0].shape[-1])
    final_ref, loss_drag_ref = O.drag_loop(diff, net, w_dev, cache, setup, W, 8, 1200.0, 0.4, Lazy(lambda i: noise(2, W - 1 - i)),
                                           progress=lambda i: say(f"drag step {i}") if i % 10 == 0 else None)
    with torch.no_grad():
        vol_ref = O.decode_volume(dec_sd, final_ref, rng, mid, res)
    final_dev = torch.from_numpy(g["final"])
    bits_ref = (vol_ref > 0).numpy().reshape(-1)
    flips = int((np.unpackbits(g["vol_bits"])[:bits_ref.size].astype(bool) != bits_ref).sum())
    sub_ref = vol_ref[::4, ::4, ::4]
    ld, lr = g["loss_rec"], np.array(loss_rec_ref)
    dd, dr = g["loss_drag"], np.array([float(x) for x in loss_drag_ref])
    rep = {
        "config": {"chain": "C4 full length", "reconstruction_steps": T, "points_per_step": C4_POINTS, "inversion_steps": W, "drag_steps": W,
                   "decode_res": res, "weights": "synthetic seed 1234 (421M params)", "stages": "each oracle stage starts from the DEVICE's stage input"},
        "reconstruction_latent_rel_l2": rel(rec_dev, rec_ref),
        "reconstruction_loss_max_rel_diff": float(np.max(np.abs(ld - lr) / np.maximum(np.abs(lr), 1e-30))),
        "reconstruction_loss_first_last_device": [float(ld[0]), float(ld[-1])], "reconstruction_loss_first_last_oracle": [float(lr[0]), float(lr[-1])],
        "reconstruction_moved_latent_rel_l2": rel(rec_ref, img0),
        "reconstruction_decode_sign_flips": rec_flips, "voxels": int(vol_ref.numel()),
        "inversion_latent_rel_l2": rel(w_dev, inv["latent"]),
        "inversion_variance_noise_norm_max_rel_diff": float(np.max(np.abs(g["vn_norm"] - vn_norm_ref) / vn_norm_ref)),
        "inversion_variance_noise_rel_l2_first_mid_last": [rel(torch.from_numpy(g["vn_keep"][j].astype(np.float32)), vn_keep_ref[j]) for j in range(3)],
        "drag_final_latent_rel_l2": rel(final_dev, final_ref.detach()),
        "drag_loss_max_rel_diff": float(np.max(np.abs(dd - dr) / np.maximum(np.abs(dr), 1e-30))),
        "drag_moved_latent_rel_l2": rel(final_ref.detach(), w_dev),
        "final_sign_flips": flips,
        "final_logit_rms_err_on_subsample": float((torch.from_numpy(g["vol_sub"]) - sub_ref).pow(2).mean().sqrt()), "final_logit_rms": float(vol_ref.pow(2).mean().sqrt()),
        "seconds_device": float(g["seconds"]), "seconds_oracle_cpu": round(time.time() - t0, 1), "oracle_threads": threads,
    }
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c4", choices=["device", "oracle"], default=None, help="full-length C4 chain, one stage (see the section comment)")
    ap.add_argument("--c4-file", default=os.path.join(ROOT, "gpurun_out", "c4_device.npz"))
    ap.add_argument("--T", type=int, default=12)
    ap.add_argument("--W", type=int, default=4)
    ap.add_argument("--res", type=int, default=96)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "round4_parity.json"))
    ap.add_argument("--threads", type=int, default=min(16, os.cpu_count()))
    a = ap.parse_args()
    if a.c4 == "device":
        return c4_device(a.c4_file)
    if a.c4 == "oracle":
        return c4_oracle(a.c4_file, a.out, a.threads)
    from ishapediting_amd import synthetic
    from ishapediting_amd.drag_utils import DragStuff, get_args
    from ishapediting_amd.mesh import chamfer_distance, extract_surface, mesh_chamfer
    from oracle.surface_cpu import mc_vertices          # checker-side vertex sets for both volumes
    from ishapediting_amd.unet_spec import build_spec, full_config
    from oracle import ref_cpu as O
    dev = torch.device("cuda", 0)
    args = get_args(["--w_time", str(a.W), "--num_steps", str(a.T), "--shape_resolution", str(a.res)])
    cfg = full_config()
    sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 1234))
    dec_sd = synthetic.decoder_state_dict(4321)
    lo, hi = -0.05 * np.ones(96, np.float32), 0.05 * np.ones(96, np.float32)    # realistic small feature range
    src, tgt = synthetic.handles(3)
    gen = torch.Generator().manual_seed(99)
    lat = torch.from_numpy(synthetic.latent(0))
    n1 = [torch.randn(1, 96, 128, 128, generator=gen) for _ in range(a.T)]
    n2 = [torch.randn(1, 96, 128, 128, generator=gen) for _ in range(a.W)]
    scale, cof = 1200.0, 0.4
    # ---------------- device ----------------
    ds = DragStuff(dev, args=args)
    ds.load_weights(sd, dec_sd, lo, hi)
    ds.step_noise = lambda i: n1[a.T - 1 - i]
    t0 = time.time()
    ds.update_latent_params(img=lat)
    ds.step_noise = lambda i: n2[a.W - 1 - i]
    for _ in ds.training(src, tgt, scale=scale, cof=cof):
        pass
    torch.cuda.synchronize()
    t_gpu = time.time() - t0
    vol_gpu = ds.volume.cpu()
    lat_gpu = ds.tri_feat.cpu()
    # ---------------- oracle (fp32, CPU) ----------------
    torch.set_num_threads(a.threads)
    net = O.UNetOracle(build_spec(cfg), sd, fp16=False)
    diff = O.DiffusionOracle(O.Tables(str(a.T)))
    t0 = time.time()
    tick = lambda what: (lambda i: print(f"oracle {what} step {i} ({time.time() - t0:.0f} s)", file=sys.stderr, flush=True) if i % 10 == 0 else None)
    img, w, cache = O.sample_with_guidance_cache(diff, net, lat, a.T, a.W, 8, {a.T - 1 - k: n1[k] for k in range(a.T)}, progress=tick("sampling"))
    setup = O.DragSetup(src, tgt, 12, 2.0 / a.res, cache[0].shape[-1])
    final, losses = O.drag_loop(diff, net, w, cache, setup, a.W, 8, scale, cof, {a.W - 1 - k: n2[k] for k in range(a.W)}, progress=tick("guided"))
    rng = torch.from_numpy((hi - lo) / 2).reshape(1, 96, 1, 1)
    mid = torch.from_numpy((hi + lo) / 2).reshape(1, 96, 1, 1)
    vol_cpu = O.decode_volume(dec_sd, final, rng, mid, a.res)
    t_cpu = time.time() - t0
    vol_cpu_unedited = O.decode_volume(dec_sd, img, rng, mid, a.res)      # the oracle's mesh0 volume: the NULL for every metric below
    # ---------------- report ----------------
    rel = lambda x, y: float((x - y).norm() / y.norm())
    va, vb = mc_vertices(vol_gpu.cpu()) / a.res * 2 - 1, mc_vertices(vol_cpu.cpu()) / a.res * 2 - 1     # visualize.py:101 convention
    v0 = mc_vertices(vol_cpu_unedited) / a.res * 2 - 1
    big = max(va.shape[0], vb.shape[0]) > 1_500_000
    # every vertex of one surface against EVERY vertex of the other (no target thinning, hence no sampling floor); on
    # surfaces of millions of vertices the query side is a 200 000-point random subset (unbiased, mesh.chamfer_distance)
    allv = lambda x, y: chamfer_distance(x.to(dev), y.to(dev), None, query_num=200000 if big else None) if min(x.shape[0], y.shape[0]) > 0 else None
    dl, ol = np.array([float(l) for l in ds.last_losses]), np.array(losses)

    def guarded(name, fn):            # a secondary section must not cost the (minutes-long) oracle run its primary numbers
        try:
            return fn()
        except Exception as e:        # noqa: BLE001
            print(f"section {name} failed: {e!r}", file=sys.stderr, flush=True)
            return {f"{name}_error": repr(e)}
    rep = {
        "config": {"T": a.T, "guided_steps": a.W, "decode_res": a.res, "handles": 3, "scale": scale, "cof": cof,
                   "weights": "synthetic seed 1234 (421M params)", "feature_range": "+-0.05"},
        "latent_rel_l2": rel(lat_gpu, final),
        "w_rel_l2": rel(ds.w0.cpu(), w),
        "logit_max_abs_err": float((vol_gpu - vol_cpu).abs().max()),
        "logit_rms_err": float((vol_gpu - vol_cpu).pow(2).mean().sqrt()),
        "logit_rms": float(vol_cpu.pow(2).mean().sqrt()),
        "sign_flips": int(((vol_gpu > 0) != (vol_cpu > 0)).sum()), "voxels": int(vol_cpu.numel()),
        "mc_vertices_device": int(va.shape[0]), "mc_vertices_oracle": int(vb.shape[0]),
        "PARITY_chamfer_all_vertices": allv(va, vb),
        "chamfer_all_vertices_queries": "200000-point random query subset per direction, complete target set" if big else "all",
        # the same metrics against the oracle's UNEDITED volume: what a wrong edit would score (the metric's resolving power)
        "NULL_chamfer_all_vertices_device_edit_vs_oracle_unedited": allv(va, v0),
        "NULL_sign_flips_oracle_edit_vs_oracle_unedited": int(((vol_cpu > 0) != (vol_cpu_unedited > 0)).sum()),
        "NULL_logit_rms_oracle_edit_vs_oracle_unedited": float((vol_cpu - vol_cpu_unedited).pow(2).mean().sqrt()),
        "NULL_latent_rel_l2_oracle_edit_vs_oracle_unedited": rel(img, final),
        "chamfer_20k_samples_FLOOR_LIMITED": chamfer_distance(va.to(dev), vb.to(dev), 20000) if min(va.shape[0], vb.shape[0]) > 0 else None,
        "chamfer_20k_sampling_floor": chamfer_distance(vb.to(dev), vb.clone().to(dev), 20000, seed=1),
        **guarded("surface_report", lambda: surface_report(vol_gpu, vol_cpu, va, vb, a.res, dev, extract_surface, mesh_chamfer)),
        "shape_like_edit_attenuated": guarded("shape_like_edit_attenuated", lambda: shape_like_report(
            dec_sd, dev, 128, extract_surface, mesh_chamfer, chamfer_distance, mc_vertices, edit_dev=lat_gpu, edit_cpu=final)),
        "shape_like_edit_full_amplitude": guarded("shape_like_edit_full_amplitude", lambda: shape_like_report(
            dec_sd, dev, 128, extract_surface, mesh_chamfer, chamfer_distance, mc_vertices, edit_dev=lat_gpu, edit_cpu=final, amp=0.05, lowpass_k=16)),
        "drag_loss_max_rel_diff": float(np.max(np.abs(dl - ol) / np.maximum(np.abs(ol), 1e-30))) if len(dl) == len(ol) and len(ol) else None,
        "oracle_drag_losses": losses, "device_drag_losses": [float(l) for l in ds.last_losses],
        "seconds_device": round(t_gpu, 2), "seconds_oracle_cpu": round(t_cpu, 1),
    }
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
