"""From a rocprofv3 kernel_trace.csv: one guided step = the dispatches between two consecutive drag-loss launches (drag_terms_kernel; drag_motion_kernel in older traces).
Prints, for the median step: wall span, sum of kernel durations, idle (gaps), launches, and the per-kernel breakdown with
the gap that precedes each kernel.  --json FILE: the same step as ms / launches per kernel CLASS (conv3x3, gemm1x1, groupnorm,
attention, other) -- profiles/step_breakdown.json, which bench.py quotes as `step_breakdown.trace`.
Usage: step_timeline.py kernel_trace.csv [--list] [--json out.json]"""
import csv
import re
import statistics
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith(("drag_terms_kernel", "drag_motion_kernel"))]
steps = []
for a, b in zip(marks, marks[1:]):
    seg = rows[a:b]
    wall = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    steps.append((wall, busy, len(seg), a, b))
steps = [s for s in steps if s[2] < 2000]
steps.sort()
wall, busy, n, a, b = steps[len(steps) // 2]
print(f"median guided step: wall {wall/1e3:.1f} us, kernels {busy/1e3:.1f} us, idle {(wall-busy)/1e3:.1f} us ({100*(wall-busy)/wall:.1f} %), {n} launches, "
      f"mean gap {(wall-busy)/n/1e3:.2f} us")
def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^()]*>)?)", name)
    return (m.group(1) if m else name)[:56]
agg = {}
prev_end = rows[a][0]
for s, e, name in rows[a:b]:
    k = short(name)
    v = agg.setdefault(k, [0, 0.0, 0.0])
    v[0] += 1; v[1] += (e - s) / 1e3; v[2] += max(0, s - prev_end) / 1e3
    prev_end = e
for k, v in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"{v[1]+v[2]:8.1f} us  kernel {v[1]:8.1f}  gap-before {v[2]:7.1f}  n={v[0]:4d}  avg {v[1]/v[0]:6.2f} + {v[2]/v[0]:5.2f}  {k}")
if "--list" in sys.argv:
    prev_end = rows[a][0]
    for s, e, name in rows[a:b]:
        print(f"  +{(s-prev_end)/1e3:6.2f} {(e-s)/1e3:8.2f} us  {short(name)}")
        prev_end = e

if "--json" in sys.argv:
    import json

    def klass(k):
        if k.startswith("igemm4_kernel") or re.match(r"igemm2?_kernel<.*true", k) or k.startswith("conv3_"):
            return "conv3x3"
        if re.match(r"igemm2?_kernel<.*false", k) or k.startswith("igemm_skinny") or k.startswith("igemm_splitk_reduce"):
            return "gemm1x1"
        if k.startswith("gn_"):
            return "groupnorm"
        if k.startswith("attn"):
            return "attention"
        return "other"
    cls = {}
    for k, v in agg.items():
        c = cls.setdefault(klass(k), {"kernel_ms": 0.0, "gap_before_ms": 0.0, "launches": 0})
        c["kernel_ms"] += v[1] / 1e3
        c["gap_before_ms"] += v[2] / 1e3
        c["launches"] += v[0]
    for c in cls.values():
        c["kernel_ms"] = round(c["kernel_ms"], 4)
        c["gap_before_ms"] = round(c["gap_before_ms"], 4)
    out = {"what": "median guided step of the plain launch sequence (ISHAP_OVERLAP_TAIL=0) under rocprofv3 --kernel-trace, per kernel class: kernel "
                   "durations (a dependent launch's duration includes its boundary) and, separately, the idle gaps in front of the kernels (inflated by "
                   "the tracer: the un-traced step is shorter than step_wall_ms)",
           "step_wall_ms": round(wall / 1e6, 4), "kernel_ms": round(busy / 1e6, 4), "launches": n, "classes": cls}
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
