"""Un-traced phase times of a guided step with the overlapped forward tail (rocprofv3's queue interception distorts two-queue
runs: under the tracer the caller's queue stood still for 0.3-1 ms after the fork, tools/fork_probe.hip shows no such stall).
torch events on the caller's stream: step start | loss + backward enqueued (between()) | tail joined | guided update done.
Plain mode:      t1 - t0 = full forward + loss + backward                     t2 - t1 = DDPM step + guided update
Overlapped mode: t1 - t0 = forward to the tap + loss + backward (contended)    t2 - t1 = what the tail still needs after the backward + step
Usage: [ISHAP_OVERLAP_TAIL=1 ISHAP_TAIL_DEFER_WGS=128] python tools/overlap_phases.py"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from ishapediting_amd import synthetic
    from ishapediting_amd import gaussian_diffusion as gd
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ds = bench.make_dragstuff(dev, 1234)
    src, tgt = synthetic.handles(bench.HANDLES, seed=7)
    ds.update_latent_params(img=synthetic.latent(0))
    bench.one_edit(ds, src, tgt)
    torch.cuda.synchronize()
    marks = []
    orig = gd.SpacedDiffusion.p_sample_guidance if hasattr(gd, "SpacedDiffusion") else None
    cls = type(ds.diffusion)
    inner = cls.p_sample_guidance

    def timed(self, model, x, t, **kw):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        between = kw.get("between")
        if between is not None:
            def b2():
                g = between()
                ev[1].record()
                return g
            kw["between"] = b2
        out = inner(self, model, x, t, **kw)
        ev[2].record()
        marks.append(ev)
        return out
    cls.p_sample_guidance = timed
    ov = os.environ.get("ISHAP_OVERLAP_TAIL", "1") == "1"
    for _ in range(2):
        marks.clear()
        for _ in ds.training(src, tgt, scale=1200, cof=0.4):
            pass
        torch.cuda.synchronize()
    a = [m[0].elapsed_time(m[1]) for m in marks]
    b = [m[1].elapsed_time(m[2]) for m in marks]
    what = (f"overlapped (ISHAP_TAIL_DEFER_WGS={os.environ.get('ISHAP_TAIL_DEFER_WGS', '128')})"
            if ov else "plain sequence")
    print(f"{what}: step start -> loss + backward done {statistics.median(a):.3f} ms; -> tail joined + DDPM step / update "
          f"{statistics.median(b):.3f} ms; sum {statistics.median(a) + statistics.median(b):.3f} ms")


if __name__ == "__main__":
    main()
