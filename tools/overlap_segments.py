"""Where the backward pass loses time while the forward tail runs beside it -- UN-TRACED (rocprofv3's queue interception distorts
two-queue runs, profiles/round5_overlap_tail_ab.txt): the library records a timing event after every block of the backward pass and
around the deferred tail on its side stream (ISHAP_BWD_MARKS=1, ishap_unet_marks, include/ishap.h).  One process, the same context:
guided steps in the plain sequence, then with the overlapped tail (drag_utils._OVERLAP_TAIL); medians over the steps of two edits.
Prints per backward segment: plain us | overlapped us | ratio, and the tail's span relative to the backward's start.
Usage: python tools/overlap_segments.py"""
import ctypes as C
import os
import statistics
import sys

os.environ["ISHAP_BWD_MARKS"] = "1"
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def label(tag, res):
    if tag == 0:
        return "start"
    if tag == 999:
        return "layout conversion (end)"
    kind, i = ("out", tag - 100) if tag < 200 else (("mid", 0) if tag == 200 else ("in", tag - 300))
    r = res.get((kind, i))
    return f"{kind}{i if kind != 'mid' else ''}" + (f" ({r}x{r})" if r else "")


def main():
    from ishapediting_amd import _lib, synthetic
    from ishapediting_amd import drag_utils as du
    from ishapediting_amd.unet_spec import build_spec, full_config
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ds = bench.make_dragstuff(dev, 1234)
    src, tgt = synthetic.handles(bench.HANDLES, seed=7)
    ds.update_latent_params(img=synthetic.latent(0))
    L = _lib.lib()
    h = ds.model._h
    # map sizes per block, for the labels
    spec = build_spec(full_config())
    res = {}
    for name, blocks in (("in", spec.input_blocks), ("out", spec.output_blocks)):
        for i, b in enumerate(blocks):
            r = getattr(b, "res_in", None) or getattr(b, "res", None)
            if r:
                res[(name, i)] = r
    tags = (C.c_int * 64)()
    ms = (C.c_float * 64)()
    tb, te = C.c_float(), C.c_float()

    # torch events on the caller's stream around the phases of a step: model call (forward, to the tap when overlapping) | loss |
    # backward | tail join + DDPM step
    phase = []
    cur = {}
    diff, model = ds.diffusion, ds.model
    inner_model, inner_bwd, inner_psg = diff._model, model.backward_input, diff.p_sample_guidance

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    import time
    host_model = []

    def model_timed(*a, **kw):
        cur["a"] = ev()
        t0 = time.perf_counter()
        out = inner_model(*a, **kw)
        host_model.append((time.perf_counter() - t0) * 1e3)
        cur["b"] = ev()
        return out

    def bwd_timed(*a, **kw):
        cur["c"] = ev()
        out = inner_bwd(*a, **kw)
        cur["d"] = ev()
        return out

    def psg_timed(*a, **kw):
        out = inner_psg(*a, **kw)
        cur["e"] = ev()
        phase.append(dict(cur))
        return out
    diff._model, model.backward_input, diff.p_sample_guidance = model_timed, bwd_timed, psg_timed

    def collect(overlap):
        du._OVERLAP_TAIL = overlap
        rows, tails = [], []
        phase.clear()
        host_model.clear()
        for _ in range(2):
            for _ in ds.training(src, tgt, scale=1200, cof=0.4):
                n = L.ishap_unet_marks(h, tags, ms, 64, C.byref(tb), C.byref(te))
                rows.append([(tags[k], ms[k]) for k in range(n)])
                tails.append((tb.value, te.value))
        ph = {k: statistics.median(p[k[0]].elapsed_time(p[k[1]]) for p in phase) for k in ("ab", "bc", "cd", "de")}
        ph["host_model"] = statistics.median(host_model)
        return rows, tails, ph
    collect(True)                                  # warm both modes
    plain, _, ph_p = collect(False)
    over, tails, ph_o = collect(True)
    print("phases of a guided step on the caller's stream (ms, medians):   model call | drag loss | backward | tail join + DDPM step + update | sum")
    for name, ph in (("plain sequence", ph_p), ("overlapped tail", ph_o)):
        print(f"  {name:16s} {ph['ab']:.3f} | {ph['bc']:.3f} | {ph['cd']:.3f} | {ph['de']:.3f} | {ph['ab'] + ph['bc'] + ph['cd'] + ph['de']:.3f}"
              f"     (host time inside the model call: {ph['host_model']:.3f} ms)")
    n = len(plain[0])
    print(f"{'segment (ends at the mark after ...)':38s} {'plain us':>10s} {'overlapped us':>14s} {'ratio':>7s} {'ends at, overlapped (ms)':>26s}")
    tot_p = tot_o = 0.0
    for k in range(1, n):
        dp = statistics.median(r[k][1] - r[k - 1][1] for r in plain) * 1e3
        do = statistics.median(r[k][1] - r[k - 1][1] for r in over) * 1e3
        at = statistics.median(r[k][1] for r in over)
        tot_p += dp
        tot_o += do
        print(f"{label(plain[0][k][0], res):38s} {dp:10.1f} {do:14.1f} {do / dp if dp > 0 else 0:7.2f} {at:26.3f}")
    print(f"{'backward pass':38s} {tot_p:10.1f} {tot_o:14.1f} {tot_o / tot_p:7.2f}")
    b = statistics.median(t[0] for t in tails)
    e = statistics.median(t[1] for t in tails)
    alone = ph_p["ab"] - ph_o["ab"]
    print(f"forward tail on the side stream: begins {b:.3f} ms and ends {e:.3f} ms after the backward's start (span {e - b:.3f} ms); alone, on "
          f"the whole chip, it takes {alone:.3f} ms (model call plain - overlapped); the backward pays {(tot_o - tot_p) / 1e3:.3f} ms for hosting it: "
          f"{100 * (1 - (tot_o - tot_p) / 1e3 / alone):.0f} % of the tail is hidden")


if __name__ == "__main__":
    main()
