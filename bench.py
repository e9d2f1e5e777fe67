#!/usr/bin/env python3
"""Benchmark of the denoise-and-drag hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one complete drag edit of one shape (BASELINE.json config 3 / SURVEY.md 8(d) "C3"):
40 classifier-guided iterations (UNet forward + drag loss gradient + UNet input-gradient backward +
guided DDPM update, drag_utils.py:336-398) followed by the 256^3 occupancy decode of get_mesh
(drag_utils.py:399,282-298).  Synthetic seeded weights/latents/handles (no checkpoints offline).
The untimed set-up per rank (update_latent_params: 200 DDPM steps that also record the 40 guidance
features) is reported as `unet_steps_per_s`.  Ranks edit independent shapes (weak scaling); the only
collective is the final gather of the occupancy volumes to rank 0 (RCCL), inside the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GUIDED_STEPS = 40          # w_time of config 3
NUM_STEPS = 200
HANDLES = 3
RES = 256
UNET_FWD_GFLOP = 634.9
GUIDED_GFLOP = 977.8
PEAK_MFMA_F16_TFLOPS = 2500.0   # dense, MI355X_MICROARCH.md


def make_dragstuff(device, seed):
    from ishapediting_amd import synthetic
    from ishapediting_amd.drag_utils import DragStuff, get_args
    args = get_args(["--w_time", str(GUIDED_STEPS), "--num_steps", str(NUM_STEPS), "--shape_resolution", str(RES)])
    ds = DragStuff(device, args=args)
    from ishapediting_amd.unet_spec import full_config
    sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict(full_config(), seed))
    ds.load_weights(sd, synthetic.decoder_state_dict(4321), -np.ones(96, np.float32), np.ones(96, np.float32))
    del sd
    return ds


def one_edit(ds, src, tgt):
    for _ in ds.training(src, tgt, scale=1200, cof=0.4):      # GUI defaults main.py:102,105
        pass
    return ds.volume


def cpu_baseline(seed):
    """The oracle (torch-CPU restatement of the reference path, fp32) on this host's cores, bounded sample:
    one guided step (forward + drag loss + autograd backward to the latent) and one 64^3 decode, extrapolated
    linearly to 40 guided steps + a 256^3 decode (x64 points)."""
    from oracle import ref_cpu as O
    from ishapediting_amd import synthetic
    from ishapediting_amd.unet_spec import build_spec, full_config
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    cfg = full_config()
    sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, seed))
    net = O.UNetOracle(build_spec(cfg), sd, fp16=False)
    diff = O.DiffusionOracle(O.Tables(str(NUM_STEPS)))
    img = torch.from_numpy(synthetic.latent(0))
    src, tgt = synthetic.handles(HANDLES)
    t0 = time.time()
    with torch.no_grad():
        o = diff.p_sample_guidance(net, img, GUIDED_STEPS - 1, noise=torch.zeros_like(img), feat_layer=8)
    t_fwd = time.time() - t0
    orig = O.resize_feat_align(o["inter_feat"])
    setup = O.DragSetup(src, tgt, 12, 2.0 / RES, orig.shape[-1])
    t0 = time.time()
    x = img.clone().requires_grad_(True)
    o = diff.p_sample_guidance(net, x, GUIDED_STEPS - 1, noise=torch.zeros_like(img), feat_layer=8)
    loss = O.drag_loss(O.resize_feat_align(o["inter_feat"]), orig + 0.01, setup, 0.4)
    torch.autograd.grad(loss, x)
    t_guided = time.time() - t0
    dec = synthetic.decoder_state_dict(4321)
    t0 = time.time()
    with torch.no_grad():
        O.decode_volume(dec, img, 1.0, 0.0, 64)
    t_dec64 = time.time() - t0
    est = GUIDED_STEPS * t_guided + 64 * t_dec64
    return {"value": round(est, 2), "unit": "s/shape", "cores": cores, "kind": "port",
            "sample": f"1 guided step ({t_guided:.2f}s; fwd-only {t_fwd:.2f}s) + 64^3 decode ({t_dec64:.2f}s) on {cores} "
                      f"threads, fp32 torch-CPU oracle; extrapolated x{GUIDED_STEPS} steps + x64 decode points"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shape-profile", default=None, help="write the per-shape conv/GEMM timing CSV of one edit here")
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    import torch.distributed as dist
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)

    from ishapediting_amd import synthetic, _lib
    ds = make_dragstuff(device, 1234 + rank % 3)           # three weight sets stand for chair / car / plane
    src, tgt = synthetic.handles(HANDLES, seed=7 + rank)
    # ---- set-up (untimed for the headline metric): 200 DDPM steps, the last 40 record guidance features ----
    torch.cuda.synchronize()
    t0 = time.time()
    ds.update_latent_params(img=synthetic.latent(rank))
    torch.cuda.synchronize()
    t_setup = time.time() - t0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    gather_buf = [torch.empty((RES, RES, RES), dtype=torch.float32, device=device) for _ in range(world)] \
        if (world > 1 and rank == 0) else None

    def step():
        vol = one_edit(ds, src, tgt)
        if world > 1:
            dist.gather(vol, gather_buf, dst=0)
        return vol

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.time()
    for _ in range(a.steps):
        vol = step()
    barrier()
    dt = time.time() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    shapes = a.steps * world
    sec_per_shape = dt / shapes

    # ---- roofline leg: the same edit once more with HIP events around every implicit-GEMM launch ----
    roofline = None
    if rank == 0:
        L = _lib.lib()
        L.ishap_profile_begin()
        one_edit(ds, src, tgt)
        torch.cuda.synchronize()
        NV = 7
        out = (C.c_double * (NV * 3))()
        L.ishap_profile_end(out, NV)
        if a.shape_profile:
            buf = C.create_string_buffer(1 << 16)
            L.ishap_profile_shapes(buf, len(buf))
            with open(a.shape_profile, "w") as f:
                f.write("M,N,K,conv3,tile,ksplit,launches,main_ms,reduce_ms,gflop\n" + buf.value.decode())
        # one entry per kernel symbol (the name rocprofv3 reports)
        names = ["void igemm2_kernel<128, 128, 4, true, 1>", "void igemm2_kernel<64, 64, 4, true, 1>",
                 "void igemm2_kernel<128, 128, 4, false, 1>", "void igemm2_kernel<64, 64, 4, false, 1>",
                 "void igemm2_kernel<64, 64, 4, true, 2>", "igemm_skinny_kernel<*, false>", "void igemm_kernel<128, 128, 32, 2, 2, true>"]
        v = max(range(NV), key=lambda i: out[i * 3 + 1])
        launches, ms, flops = out[v * 3], out[v * 3 + 1], out[v * 3 + 2]
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # HBM-side bytes per launch of that kernel from the committed PMC passes (profiles/pmc_traffic.json, produced by
        # tools/pmc_only.sh + tools/pmc_summary.py: counters cannot be collected from inside this process)
        traffic = None
        try:
            k = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"][names[v]]
            traffic = k["FETCH_SIZE"]["bytes_per_launch"] + k["WRITE_SIZE"]["bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        roofline = {"bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_MFMA_F16_TFLOPS, 4), "traffic": traffic, "kernel": names[v],
                    "launches_per_edit": int(launches), "avg_launch_us": round(ms * 1e3 / max(launches, 1), 2),
                    "flops_per_launch_avg": flops / max(launches, 1),
                    "share_of_edit_time": round(ms * 1e-3 / sec_per_shape / max(world, 1), 3) if world == 1 else None,
                    "all_variants": {names[i]: {"launches": int(out[i * 3]), "ms": round(out[i * 3 + 1], 3),
                                                "tflops": round(out[i * 3 + 2] / max(out[i * 3 + 1], 1e-9) / 1e9, 1)}
                                     for i in range(NV)}}
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(1234)

    if rank == 0:
        # surface of the last decoded volume on the device (reported separately from the headline, as BASELINE's metric
        # does for marching cubes): marching tetrahedra + 10 smoothing sweeps, csrc/surface.hip
        from ishapediting_amd.mesh import extract_surface, smooth_mesh
        extract_surface(vol)
        torch.cuda.synchronize()
        t0 = time.time()
        sv, st = extract_surface(vol)
        smooth_mesh(sv, st, 10)
        torch.cuda.synchronize()
        surface_ms = (time.time() - t0) * 1e3
        # the random-weight volume is noise (surface through almost every cell); a shape-like volume for scale:
        ax = torch.arange(RES, dtype=torch.float32, device=device) - 120.3
        sph = 90.4 - torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2)
        extract_surface(sph)
        torch.cuda.synchronize()
        t0 = time.time()
        sv2, st2 = extract_surface(sph)
        smooth_mesh(sv2, st2, 10)
        torch.cuda.synchronize()
        surface_ms_sphere = (time.time() - t0) * 1e3
        line = {
            "metric": "end-to-end drag-edit wall-clock (s) per shape", "value": round(sec_per_shape, 4), "unit": "s/shape",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
            "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "C3 chair drag-edit: 3 handle->target pairs, 40 guided iterations (UNet fwd + drag "
                                   "gradient + UNet input-grad bwd + DDPM update) + 256^3 occupancy decode",
                       "unet": "421M params, 1x96x128x128 latent, fp16 torso", "r1": 12, "scale": 1200, "cof": 0.4,
                       "shapes_per_gpu_per_step": 1},
            "shapes_per_s": round(shapes / dt, 4),
            "algorithmic_tflops_per_gpu": round((GUIDED_STEPS * GUIDED_GFLOP + 1185.0) / 1e3 / (dt / a.steps), 1),
            "unet_steps_per_s": round(NUM_STEPS / t_setup, 2),
            "setup_s": round(t_setup, 3),
            "surface_vertices": int(sv.shape[0]), "surface_triangles": int(st.shape[0]),
            "surface_extract_ms": round(surface_ms, 2),
            "surface_extract_ms_sphere256": round(surface_ms_sphere, 2), "sphere256_vertices": int(sv2.shape[0]),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if cpu:
            line["speedup_vs_cpu_baseline"] = round(cpu["value"] / sec_per_shape, 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
