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


def _cpu_info():
    model, logical = "unknown", os.cpu_count() or 1
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = logical
    # a container's CPU share (cgroup quota) is what the threads really get: more threads than that only thrash
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            usable = min(usable, max(1, int(float(q[0]) / float(q[1]))))
    except (OSError, ValueError, IndexError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                usable = min(usable, max(1, quota // period))
        except (OSError, ValueError):
            pass
    cap = int(os.environ.get("ISHAP_BENCH_CPU_THREADS", "0"))
    if cap > 0:
        usable = min(usable, cap)
    return model, logical, usable


def cpu_baseline(seed, reps=5):
    """The oracle (torch-CPU restatement of the reference path, fp32) on every core this process may use, bounded
    sample: `reps` guided steps (forward + drag loss + autograd backward to the latent), `reps` unguided
    steps and `reps` 64^3 decodes -- each kind after one discarded run, the MEDIAN taken -- then extrapolated linearly to
    C3 = 40 guided steps + a 256^3 decode (x64 points)."""
    from oracle import ref_cpu as O
    from ishapediting_amd import synthetic
    from ishapediting_amd.unet_spec import build_spec, full_config
    model, logical, cores = _cpu_info()
    torch.set_num_threads(cores)
    print(f"[bench] cpu baseline: {cores} threads on {model} ({logical} logical CPUs)", file=sys.stderr, flush=True)
    cfg = full_config()
    sd = synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, seed))
    net = O.UNetOracle(build_spec(cfg), sd, fp16=False)
    diff = O.DiffusionOracle(O.Tables(str(NUM_STEPS)))
    img = torch.from_numpy(synthetic.latent(0))
    src, tgt = synthetic.handles(HANDLES)
    with torch.no_grad():                                          # untimed warm-up (thread pool, allocator) + the guidance feature
        o = diff.p_sample_guidance(net, img, GUIDED_STEPS - 1, noise=torch.zeros_like(img), feat_layer=8)
    orig = O.resize_feat_align(o["inter_feat"])
    setup = O.DragSetup(src, tgt, 12, 2.0 / RES, orig.shape[-1])
    print("[bench] cpu baseline: warm-up forward done", file=sys.stderr, flush=True)
    # every kind: one UNTIMED first run (the first autograd pass builds its graph buffers and the allocator / thread pool are
    # still warming: round 5's samples read 1.27 / 0.95 / 0.62 s), then `reps` timed runs whose MEDIAN is reported
    t_fwd, t_guided, t_dec = [], [], []
    for r in range(reps + 1):
        t0 = time.time()
        with torch.no_grad():
            diff.p_sample_guidance(net, img, GUIDED_STEPS - 1 - r, noise=torch.zeros_like(img), feat_layer=8)
        if r > 0: t_fwd.append(time.time() - t0)
    for r in range(reps + 1):
        t0 = time.time()
        x = img.clone().requires_grad_(True)
        o = diff.p_sample_guidance(net, x, GUIDED_STEPS - 1 - r, noise=torch.zeros_like(img), feat_layer=8)
        loss = O.drag_loss(O.resize_feat_align(o["inter_feat"]), orig + 0.01, setup, 0.4)
        torch.autograd.grad(loss, x)
        dt_ = time.time() - t0
        if r > 0: t_guided.append(dt_)
        print(f"[bench] cpu baseline: guided step {r}{' (discarded)' if r == 0 else ''}: {dt_:.2f}s", file=sys.stderr, flush=True)
    dec = synthetic.decoder_state_dict(4321)
    for r in range(reps + 1):
        t0 = time.time()
        with torch.no_grad():
            O.decode_volume(dec, img, 1.0, 0.0, 64)
        if r > 0: t_dec.append(time.time() - t0)
    med = lambda v: sorted(v)[len(v) // 2]
    mg, mf, md = med(t_guided), med(t_fwd), med(t_dec)
    est = GUIDED_STEPS * mg + 64 * md
    return {"value": round(est, 2), "unit": "s/shape", "cores": cores, "kind": "port",
            "cpu_model": model, "logical_cpus": logical,
            "sample": f"after one discarded run of each kind: {reps} guided steps (median {mg:.2f}s), {reps} unguided steps (median {mf:.2f}s), "
                      f"{reps} 64^3 decodes (median {md:.2f}s) on {cores} threads of {model} ({logical} logical CPUs), fp32 torch-CPU oracle; "
                      f"extrapolated to {GUIDED_STEPS} guided steps + 64x the decode points (256^3)",
            "unguided_step_s": round(mf, 3), "guided_step_s": round(mg, 3), "decode64_s": round(md, 3),
            "guided_step_samples_s": [round(v, 3) for v in t_guided]}


def visible_gpu_count():
    """GPUs this process may use, WITHOUT touching HIP/HSA (torch.cuda.device_count() falls back to hipGetDeviceCount
    when amdsmi is missing, which would initialise the runtime in a parent that then starts children): the KFD topology
    in sysfs lists one node per agent, GPUs are the nodes with SIMDs; HIP_/ROCR_/CUDA_VISIBLE_DEVICES narrow that.
    None = cannot tell (let the ranks fail in set_device, which already ends the run non-zero)."""
    n = None
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        n = 0
        for d in os.listdir(base):
            props = dict(line.split(None, 1) for line in open(os.path.join(base, d, "properties")) if " " in line)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except (OSError, ValueError):
        n = None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            k = len([t for t in v.split(",") if t.strip() != ""])
            n = k if n is None else min(n, k)
    return n


def c2_generate_leg(device, steps=1000, batch=1):
    """BASELINE.json configs[1] (C2) once at batch `batch` (1, and 8 = generate.py's default batch_size,
    image_sample.py:173-184): the full `steps`-step DDPM sample of the 421M-parameter model (generate.py --num_steps 1000),
    the 256^3 decode and marching cubes + 10 smoothing sweeps of every sample.  Reported next to the headline (extra keys),
    never part of `value`."""
    from ishapediting_amd import synthetic
    from ishapediting_amd.gaussian_diffusion import create_gaussian_diffusion
    from ishapediting_amd.mesh import extract_surface, smooth_mesh
    from ishapediting_amd.triplane_decoder import MultiTriplane, decode_volume
    from ishapediting_amd.unet import UNetModel
    from ishapediting_amd.unet_spec import full_config
    cfg = full_config()
    model = UNetModel(cfg, device, max_batch=batch)
    model.load_state_dict(synthetic.round_torso_to_fp16(synthetic.unet_state_dict(cfg, 1234)))
    dec = MultiTriplane(1, device=device)
    dec.net.load_state_dict(synthetic.decoder_state_dict(4321))
    diff = create_gaussian_diffusion(timestep_respacing=str(steps))
    g = torch.Generator(device="cpu").manual_seed(5)
    noise = torch.randn((batch, 96, 128, 128), generator=g).to(device)
    warm = create_gaussian_diffusion(timestep_respacing="4")
    warm.p_sample_loop(model, noise.shape, noise=noise, device=device)        # kernels loaded, allocator warm
    torch.cuda.synchronize()
    t0 = time.time()
    sample = diff.p_sample_loop(model, noise.shape, noise=noise, device=device)
    torch.cuda.synchronize()
    t1 = time.time()
    vols = [decode_volume(dec, sample[b:b + 1], 1.0, 0.0, RES) for b in range(batch)]
    torch.cuda.synchronize()
    t2 = time.time()
    for vol in vols:
        verts, tris = extract_surface(vol)
        smooth_mesh(verts, tris, 10, box_max=float(RES - 1))
    torch.cuda.synchronize()
    t3 = time.time()
    del model
    if batch > 1:
        return {f"c2_batch{batch}_s_per_shape": round((t3 - t0) / batch, 4),
                f"c2_batch{batch}_unet_sample_steps_per_s": round(steps * batch / (t1 - t0), 1),
                f"c2_batch{batch}_sample_s": round(t1 - t0, 4), f"c2_batch{batch}_decode_ms_per_shape": round((t2 - t1) * 1e3 / batch, 2),
                f"c2_batch{batch}_marching_cubes_ms_per_shape": round((t3 - t2) * 1e3 / batch, 2)}
    return {"c2_s_per_shape": round(t3 - t0, 4), "c2_unet_steps_per_s": round(steps / (t1 - t0), 1),
            "c2_sample_s": round(t1 - t0, 4), "c2_decode_ms": round((t2 - t1) * 1e3, 2),
            "c2_marching_cubes_ms": round((t3 - t2) * 1e3, 2), "c2_steps": steps,
            "c2_note": "random-init weights decode to a noise volume: the marching-cubes time is that of ~16M vertices "
                       "(a shape-like 256^3 volume takes surface_extract_ms_sphere256)"}


def concurrent_edits_leg(device, ds, src, tgt, seed, reps=2):
    """Two and three INDEPENDENT C3 edits at once on one GPU: more model contexts (each its own weights, latent, guidance cache) on
    their own streams, each driven from its own host thread -- what a server with more queued edits than GPUs would do (BASELINE
    configs[4] shards 8 edits over 8 GPUs; with 24 queued it would run three per GPU).  Reported NEXT to the headline, never as
    it: the headline is one edit per GPU, as the reference's DragStuff handles one shape at a time (drag_utils.py:303-304).
    The latency-bound chains fill each other's idle compute units (a guided step is 408 dependent launches, most of them
    far from filling 256 CUs); the library's rendezvous tenancy (include/ishap.h) lets one context at a time use the in-launch
    GroupNorm rendezvous, the others run the same kernels with one workgroup per group.  tools/experiments/concurrent_probe.py: 1 / 2 / 3 / 4
    concurrent edits -> 0.1835 / 0.137 / 0.120 / 0.139 s per shape."""
    import threading
    from ishapediting_amd import synthetic
    ctxs = [(ds, src, tgt, torch.cuda.Stream(device))]
    for k in (1, 2):
        d = make_dragstuff(device, seed + k)
        d.update_latent_params(img=synthetic.latent(4 + k))
        ctxs.append((d, *synthetic.handles(HANDLES, seed=23 + k), torch.cuda.Stream(device)))
    torch.cuda.synchronize()
    out = {}
    for n in (2, 3):
        errs = []

        def run(c):
            d, s_, t_, stream = c
            try:
                with torch.cuda.stream(stream):
                    for _ in range(reps):
                        one_edit(d, s_, t_)
            except Exception as e:      # noqa: BLE001
                errs.append(e)
        for _warm in (True, False):
            ths = [threading.Thread(target=run, args=(c,)) for c in ctxs[:n]]
            torch.cuda.synchronize()
            t0 = time.time()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            torch.cuda.synchronize()
            dt = time.time() - t0
            if errs:
                raise errs[0]
        out[f"concurrent{n}_edits_s_per_shape"] = round(dt / (n * reps), 4)
        out[f"concurrent{n}_edits_latency_s"] = round(dt / reps, 4)
    out["concurrent_edits_note"] = ("throughput of 2 / 3 independent C3 edits sharing one GPU (one context, stream and host thread each); "
                                    "each edit then takes concurrent<n>_edits_latency_s; not the headline, which is one edit per GPU")
    del ctxs[1:]
    return out


def airplane_like_mesh(device, res=128):
    """SURVEY 8(d) C4 input: a synthetic watertight "airplane-like" mesh -- the union of four ellipsoids (fuselage, wing,
    tail plane, fin) -- extracted on the device from its implicit function (csrc/surface.hip), vertices in [-1, 1]^3."""
    from ishapediting_amd.mesh import extract_surface
    ax = torch.linspace(-1, 1, res, device=device)
    x, y, z = ax[:, None, None], ax[None, :, None], ax[None, None, :]
    parts = [((0.0, 0.0, 0.0), (0.80, 0.11, 0.11)), ((0.05, 0.0, 0.0), (0.16, 0.75, 0.035)),
             ((-0.68, 0.0, 0.02), (0.09, 0.27, 0.025)), ((-0.68, 0.0, 0.12), (0.10, 0.025, 0.16))]
    f = None
    for (cx, cy, cz), (rx, ry, rz) in parts:
        e = 1.0 - (((x - cx) / rx) ** 2 + ((y - cy) / ry) ** 2 + ((z - cz) / rz) ** 2)
        f = e if f is None else torch.maximum(f, e)
    v, t = extract_surface(f.contiguous())
    return v / (res - 1) * 2 - 1, t


def c4_real_shape_leg(device, shape_profile=None):
    """BASELINE.json configs[3] (C4) once at FULL length with the reference's defaults (num_steps 200, w_time 170,
    40 000-point batches, drag_utils.py:44-57): mesh -> 200 000 occupancy samples (on-device ray parity) ->
    train_triplane's guided reconstruction, 200 steps of {UNet forward, decoder BCE on 40 000 points + decoder backward,
    FULL-DEPTH UNet input-gradient backward, guided update} (drag_utils.py:442-463) -> 256^3 decode of the reconstruction
    -> ddpm_inversion over 170 steps (gaussian_diffusion.py:512-532) -> 170 guided drag iterations + final 256^3 decode
    (drag_utils.py:336-399).  Then one more reconstruction pass of 8 steps with HIP events on every implicit-GEMM launch:
    the per-shape table and the roofline record of the reconstruction step (1.278 TFLOP algorithmic)."""
    import tempfile
    from ishapediting_amd import synthetic, _lib
    from ishapediting_amd.drag_utils import DragStuff, get_args
    from ishapediting_amd import mesh as mesh_backend
    from ishapediting_amd.unet_spec import full_config
    args = get_args(["--shape_resolution", str(RES)])            # reference defaults: num_steps 200, w_time 170
    ds = DragStuff(device, args=args)
    ds.load_weights(synthetic.round_torso_to_fp16(synthetic.unet_state_dict(full_config(), 1236)), synthetic.decoder_state_dict(4321),
                    -np.ones(96, np.float32), np.ones(96, np.float32))
    v, t = airplane_like_mesh(device)
    src, tgt = synthetic.handles(HANDLES, seed=11)
    sync = torch.cuda.synchronize
    ds.reconstruct(*[x.to(device) for x in (torch.rand(50000, 3) * 2 - 1, torch.rand(50000).round())], steps=[199, 198])   # warm-up: kernels loaded
    sync()
    t0 = time.time()
    pts, occ = mesh_backend.sample_occupancy((v, t), None, True, args.points_size, args.points_uniform_ratio, device=device)
    pts = torch.as_tensor(np.asarray(pts) if not torch.is_tensor(pts) else pts, dtype=torch.float32).to(device)
    occ = torch.as_tensor(np.asarray(occ) if not torch.is_tensor(occ) else occ, dtype=torch.float32).reshape(-1).to(device)
    sync()
    t1 = time.time()
    img = ds.reconstruct(pts, occ)                                # drag_utils.py:442-463
    sync()
    t2 = time.time()
    ds.clear_params()
    ds.mesh = ds.get_mesh(tri_feat=img)                           # :467-469 (256^3 decode + surface of the reconstruction)
    sync()
    t3 = time.time()
    ds.latent_inversion(tri_feat=img)                             # :471 -> gaussian_diffusion.py:512-532 (+ decode of the inverted sample)
    sync()
    t4 = time.time()
    for _ in ds.training(src, tgt, scale=1200, cof=0.4):
        pass
    sync()
    t5 = time.time()
    n_rec, n_w = args.num_steps, args.w_time
    out = {"c4_s_per_shape": round(t5 - t0, 4), "c4_occupancy_sampling_s": round(t1 - t0, 4),
           "c4_reconstruct_s": round(t2 - t1, 4), "c4_reconstruct_ms_per_step": round((t2 - t1) * 1e3 / n_rec, 3),
           "c4_reconstruct_tflops": round(1.278 * n_rec / (t2 - t1), 1),
           "c4_reconstruction_decode_and_surface_s": round(t3 - t2, 4),
           "c4_inversion_s": round(t4 - t3, 4), "c4_inversion_unet_steps_per_s": round(n_w / (t4 - t3), 1),
           "c4_drag_s": round(t5 - t4, 4), "c4_drag_ms_per_guided_step": round((t5 - t4) * 1e3 / n_w, 3),
           "c4_mesh_vertices": int(v.shape[0]), "c4_occupancy_samples": int(pts.shape[0]),
           "c4_config": f"synthetic airplane-like mesh (union of 4 ellipsoids), {n_rec} reconstruction steps x 40000 points, inversion {n_w}, drag {n_w} guided iterations, 256^3 decodes"}
    # ---- the reconstruction step under the per-launch event profile ----
    L = _lib.lib()
    L.ishap_profile_begin()
    steps = list(range(107, 99, -1))
    sync()
    tp0 = time.time()
    ds.reconstruct(pts, occ, steps=steps)
    sync()
    tp = time.time() - tp0
    NV = 13
    buf3 = (C.c_double * (NV * 3))()
    L.ishap_profile_end(buf3, NV)
    buf = C.create_string_buffer(1 << 17)
    L.ishap_profile_shapes(buf, len(buf))
    csv = buf.value.decode()
    if shape_profile:
        with open(shape_profile, "w") as f:
            f.write(f"# {len(steps)} reconstruction steps (UNet forward + full-depth input-gradient backward)\nM,N,K,conv3,tile,ksplit,launches,main_ms,reduce_ms,gflop\n" + csv)
    conv_ms = conv_gf = 0.0
    for row in csv.strip().splitlines():
        M_, N_, K_, c3, _tile, _ks, n_, main_ms, red_ms, gf = (float(x) for x in row.split(","))
        conv_ms += main_ms + red_ms
        conv_gf += gf
    out["c4_reconstruct_roofline"] = {
        "bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_MFMA_F16_TFLOPS,
        "achieved": round(1.278 * n_rec / (t2 - t1), 1), "frac": round(1.278 * n_rec / (t2 - t1) / PEAK_MFMA_F16_TFLOPS, 4),
        "what": "whole reconstruction step: 1.278 TFLOP algorithmic (SURVEY 8d) / measured step time of the 200-step run",
        "conv_gemm_ms_per_step": round(conv_ms / len(steps), 3), "conv_gemm_gflop_per_step": round(conv_gf / len(steps), 1),
        "conv_gemm_tflops": round(conv_gf / max(conv_ms, 1e-9), 1),
        "profiled_step_ms": round(tp * 1e3 / len(steps), 3)}
    del ds
    return out


def spawn_workers(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (each its own
    interpreter; this parent never touches the GPU), relay rank 0's JSON line, exit with the worst status."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    have = visible_gpu_count()                                 # sysfs / environment only: this parent never loads HIP
    if have is not None and a.gpus > have and not a.rehearse:
        sys.exit(f"bench.py: --gpus {a.gpus} but only {have} GPU(s) are visible")
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if a.rehearse else str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if a.rehearse:
            env["ISHAP_GN_PARTS"] = "1"      # several PROCESSES share the GPU: no in-launch rendezvous (include/ishap.h, Tenancy)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:                                                # a rank that dies must not leave the others waiting in a collective
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = max(rc, abs(code))
                for q in live:
                    q.terminate()
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c2", action="store_true", help="skip the generate-path leg (BASELINE configs[1]) after the headline")
    ap.add_argument("--c2-steps", type=int, default=1000)
    ap.add_argument("--no-c2-batch8", action="store_true", help="skip the batch-8 generate leg (generate.py's default batch)")
    ap.add_argument("--no-c4", action="store_true", help="skip the real-shape leg (BASELINE configs[3]) after the headline")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the two-concurrent-edits leg")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on ONE GPU: every rank uses cuda:0 and the collectives run over gloo on host copies -- a rehearsal of "
                         "the multi-rank control flow (spawn, rendezvous, barriers, gather, per-rank report), NOT a measurement")
    ap.add_argument("--c4-shape-profile", default=None, help="write the per-shape conv/GEMM CSV of the reconstruction step here")
    ap.add_argument("--shape-profile", default=None, help="write the per-shape conv/GEMM timing CSV of one edit here")
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and "RANK" not in os.environ:
        spawn_workers(a)                                       # does not return
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    import torch.distributed as dist
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    comm_dev = device                                          # where collective operands live: the GPU under RCCL
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.rehearse:
            comm_dev = torch.device("cpu")
            dist.init_process_group("gloo")
        else:
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

    from ishapediting_amd.parallel import gather_volumes
    recv = [torch.empty((RES, RES, RES), dtype=torch.float32, device=comm_dev) for _ in range(world)] \
        if (world > 1 and rank == 0) else None

    # per-rank diagnostics of the N > 1 run (no host synchronisation inside the timed region): events on this rank's
    # compute stream around the edit and around the gather call
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.steps + a.warmup)]
    it = [0]

    def step():
        e0, e1, e2 = ev[it[0]]
        it[0] += 1
        e0.record()
        vol = one_edit(ds, src, tgt)
        e1.record()
        if world > 1:     # the path's only collective: every rank's occupancy volume to rank 0 (RCCL gather, 67 MB per rank)
            gather_volumes([vol.to(comm_dev)], world, dst=0, full_shape=(RES, RES, RES), recv=recv)
        e2.record()
        return vol

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.time()
    for _ in range(a.steps):
        vol = step()
    barrier()
    dt = time.time() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    shapes = a.steps * world
    sec_per_shape = dt / shapes
    torch.cuda.synchronize()
    mine = torch.tensor([sum(e[0].elapsed_time(e[1]) for e in ev[a.warmup:]) / a.steps,
                         sum(e[1].elapsed_time(e[2]) for e in ev[a.warmup:]) / a.steps, t_setup * 1e3], dtype=torch.float64, device=comm_dev)
    per_rank = [mine.clone() for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, mine)
    per_rank = [{"rank": r, "edit_ms": round(float(v[0]), 2), "gather_ms": round(float(v[1]), 2), "setup_s": round(float(v[2]) / 1e3, 2)}
                for r, v in enumerate(per_rank)]
    print(f"[bench] rank {rank}: edit {per_rank[rank]['edit_ms']} ms, gather call {per_rank[rank]['gather_ms']} ms per step "
          f"(stream time; rank 0's gather includes waiting for the slowest rank)", file=sys.stderr, flush=True)

    # ---- roofline leg: the same edit once more with HIP events around every implicit-GEMM launch ----
    roofline = small_maps = step_breakdown = None
    if rank == 0:
        L = _lib.lib()
        # per-launch durations are measured in the PLAIN launch sequence: in the timed (default) configuration the forward tail
        # shares the chip with loss + backward, and a launch's duration inside that window is not the kernel's own
        ds.overlap_tail = False                  # this DragStuff only (not the module default); restored whatever happens
        L.ishap_profile_begin()
        try:
            one_edit(ds, src, tgt)
            torch.cuda.synchronize()
        finally:
            ds.overlap_tail = None               # back to the module default
        NV = 13
        out = (C.c_double * (NV * 3))()
        L.ishap_profile_end(out, NV)
        buf = C.create_string_buffer(1 << 17)
        L.ishap_profile_shapes(buf, len(buf))
        shape_csv = buf.value.decode()
        if a.shape_profile:
            with open(a.shape_profile, "w") as f:
                f.write("M,N,K,conv3,tile,ksplit,launches,main_ms,reduce_ms,gflop\n" + shape_csv)
        # second record: the HBM-bound class -- every conv / GEMM on the 8x8 and 16x16 maps (M <= 256 rows at
        # batch 1) streams its weights once; algorithmic bytes = fp16 weights + input map + output map per launch
        sm_bytes = sm_ms = sm_launch = 0.0
        for row in shape_csv.strip().splitlines():
            M_, N_, K_, c3, _tile, _ks, n_, main_ms, red_ms, _gf = (float(v) for v in row.split(","))
            if M_ <= 256:
                cin = K_ / 9 if c3 else K_
                sm_bytes += n_ * (N_ * K_ * 2 + M_ * cin * 2 + M_ * N_ * 2)
                sm_ms += main_ms + red_ms
                sm_launch += n_
        small_maps = None
        if sm_ms > 0:
            gbs = sm_bytes / (sm_ms * 1e-3) / 1e9
            small_maps = {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                          "traffic": None, "kernel": "all conv/GEMM launches on the 8x8 and 16x16 maps (M <= 256)",
                          "launches_per_edit": int(sm_launch), "ms_per_edit": round(sm_ms, 2),
                          "algorithmic_bytes_per_edit": int(sm_bytes), "algorithmic_bytes_per_launch": int(sm_bytes / max(sm_launch, 1))}
        # the same launches' conv3x3 / 1x1 split per guided step (live, from the event records above)
        c3_ms = c3_n = g1_ms = g1_n = 0.0
        for row in shape_csv.strip().splitlines():
            _M, _N, _K, c3, _tile, _ks, n_, main_ms, red_ms, _gf = (float(v) for v in row.split(","))
            if c3: c3_ms += main_ms + red_ms; c3_n += n_
            else: g1_ms += main_ms + red_ms; g1_n += n_
        # one entry per kernel symbol (the name rocprofv3 reports)
        names = ["igemm2_kernel<128, 128, 4, true, 1>", "igemm2_kernel<64, 64, 4, true, 1>",
                 "igemm2_kernel<128, 128, 4, false, 1>", "igemm2_kernel<64, 64, 4, false, 1>",
                 "igemm2_kernel<64, 64, 4, true, 2>", "igemm_skinny_kernel<*, false>", "igemm_kernel<64, 64, 32, 2, 2, *>",
                 "(unused)", "igemm4_kernel<128, 128, 128, 5, 3, 1>", "igemm4_kernel<64, 64, *, *, 3, 1>",
                 "igemm4_kernel<64, 64, *, 6, 3, 2>", "igemm4_kernel<64, 64, 8, 4, 3, 1>", "igemm4_kernel<128, 64, 64, 6, 3, 1>"]
        # (the '*'s stand for the map width 16 / 32 / 64 and the weight-ring depth 6 / 4 (4 for slices of 9-12 K-steps); the width-8
        # symbol -- the sliced, weight-streaming launches of the 8x8 maps -- has an entry of its own)
        v = max(range(NV), key=lambda i: out[i * 3 + 1])
        launches, ms, flops = out[v * 3], out[v * 3 + 1], out[v * 3 + 2]
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # HBM-side bytes per launch of that kernel from the committed PMC passes (profiles/pmc_traffic.json, produced by
        # tools/pmc_only.sh + tools/pmc_summary.py: counters cannot be collected from inside this process)
        pj = None
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        except (OSError, ValueError):
            pass

        def counters(name):
            """HBM-side bytes per launch and MFMA-busy share of one entry of `names` from the committed counter passes
            (profiles/pmc_traffic.json: tools/final_profile.sh writes it from the same run as the round's pmc_*.txt files)"""
            if pj is None:
                return None, None
            import re as _re
            try:
                pat = _re.compile("^" + _re.escape(name).replace("\\*", "[0-9]+") + "$")
                ks = [k for key, k in pj["kernels"].items() if pat.match(key) and (key == name or key not in names)]   # a '*' stands for a template argument
                nd = sum(k["FETCH_SIZE"]["dispatches"] for k in ks)
                byt = int(sum((k["FETCH_SIZE"]["bytes_per_launch"] + k["WRITE_SIZE"]["bytes_per_launch"]) * k["FETCH_SIZE"]["dispatches"] for k in ks) / nd)
                mf = [k["mfma_util_pct"] * k["FETCH_SIZE"]["dispatches"] for k in ks if "mfma_util_pct" in k]
                return byt, (round(sum(mf) / nd / 100.0, 4) if len(mf) == len(ks) else None)
            except (KeyError, ValueError, ZeroDivisionError):
                return None, None
        traffic, mfma_busy = counters(names[v])
        if small_maps is not None and pj is not None and "small_maps" in pj:
            small_maps["traffic"] = pj["small_maps"].get("bytes_per_launch")
            small_maps["traffic_source"] = "committed profiles/pmc_traffic.json, group `small_maps`: " + str(pj["small_maps"].get("what"))
        # per-class breakdown of one guided step in the plain launch sequence: conv / GEMM classes live (event records of this run,
        # 40 guided steps + the decode's none), the other classes from the committed kernel trace of the same build
        step_breakdown = {"conv3x3": {"ms": round(c3_ms / GUIDED_STEPS, 4), "launches": round(c3_n / GUIDED_STEPS, 1)},
                          "gemm1x1": {"ms": round(g1_ms / GUIDED_STEPS, 4), "launches": round(g1_n / GUIDED_STEPS, 1)},
                          "source": "conv3x3 / gemm1x1: HIP events on every implicit-GEMM dispatch of this run, per guided step"}
        try:
            sb = json.load(open(os.path.join(ROOT, "profiles", "step_breakdown.json")))
            step_breakdown["trace"] = sb
            step_breakdown["source"] += "; `trace`: committed profiles/step_breakdown.json (tools/step_timeline.py --json over the rocprofv3 " \
                                        "kernel trace of tools/final_profile.sh, median guided step, plain sequence: ms and launches per kernel class)"
        except (OSError, ValueError):
            pass
        traffic_source = None if traffic is None else (
            "committed profiles/pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / MFMA passes of tools/pmc_step.py ("
            + str(pj.get("source", "tools/final_profile.sh")) + "); not collected live (counters cannot be read from inside this process)")
        # the single kernel SYMBOL with the most time (the dominant entry above may be a group of symbols that differ only in the
        # map width / ring depth template arguments)
        single = [i for i in range(NV) if "*" not in names[i]]
        t1 = max(single, key=lambda i: out[i * 3 + 1])
        t1_tf = out[t1 * 3 + 2] / max(out[t1 * 3 + 1], 1e-9) / 1e9
        t1_traffic, t1_busy = counters(names[t1])
        top_single = {"kernel": names[t1], "achieved": round(t1_tf, 1), "frac": round(t1_tf / PEAK_MFMA_F16_TFLOPS, 4),
                      "launches_per_edit": int(out[t1 * 3]), "avg_launch_us": round(out[t1 * 3 + 1] * 1e3 / max(out[t1 * 3], 1), 2),
                      "share_of_edit_time": round(out[t1 * 3 + 1] * 1e-3 / sec_per_shape, 3) if world == 1 else None,
                      "traffic": t1_traffic, "mfma_busy": t1_busy}
        roofline = {"bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_MFMA_F16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "mfma_busy": mfma_busy,
                    "kernel": names[v],
                    "measured_in": "plain launch sequence (ISHAP_OVERLAP_TAIL=0) with HIP events attached to every implicit-GEMM dispatch; "
                                   "the timed region runs the forward tail beside loss + backward",
                    "launches_per_edit": int(launches), "avg_launch_us": round(ms * 1e3 / max(launches, 1), 2),
                    "flops_per_launch_avg": flops / max(launches, 1),
                    "share_of_edit_time": round(ms * 1e-3 / sec_per_shape / max(world, 1), 3) if world == 1 else None,
                    "top_single_symbol": top_single,
                    "all_variants": {names[i]: {"launches": int(out[i * 3]), "ms": round(out[i * 3 + 1], 3),
                                                "tflops": round(out[i * 3 + 2] / max(out[i * 3 + 1], 1e-9) / 1e9, 1)}
                                     for i in range(NV) if out[i * 3] > 0}}
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(1234)

    if rank == 0:
        # surface of the last decoded volume on the device (reported separately from the headline, as BASELINE's metric
        # does for marching cubes): marching cubes (256-case table) + 10 smoothing sweeps, csrc/surface.hip
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
            "roofline": roofline, "roofline_small_maps": small_maps, "step_breakdown": step_breakdown, "cpu_baseline": cpu,
            "per_rank": per_rank,
        }
        if a.rehearse:
            line["data"] = "synthetic; REHEARSAL (all ranks on one GPU, gloo on host copies): not a measurement"
        if cpu:
            line["speedup_vs_cpu_baseline"] = round(cpu["value"] / sec_per_shape, 1)
        if world == 1 and not a.no_concurrent:
            line.update(concurrent_edits_leg(device, ds, src, tgt, 1235))
        if world == 1 and not (a.no_c2 and a.no_c4):
            del ds                                                   # the edit context's arena + guidance cache
            torch.cuda.empty_cache()
        if world == 1 and not a.no_c2:
            line.update(c2_generate_leg(device, a.c2_steps))
            torch.cuda.empty_cache()
            if not a.no_c2_batch8:
                line.update(c2_generate_leg(device, a.c2_steps, batch=8))
                torch.cuda.empty_cache()
        if world == 1 and not a.no_c4:
            line.update(c4_real_shape_leg(device, a.c4_shape_profile))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
