"""From a rocprofv3 kernel_trace.csv of a run with the overlapped forward tail (ISHAP_OVERLAP_TAIL=1): per guided step, the
window in which the side queue (the tail) runs, what the caller's queue (loss + backward) does inside it, and -- per
kernel symbol -- duration and gap-before INSIDE the window against the same symbol OUTSIDE any window (same run), i.e.
which kernels of the backward chain waited and for how long.  Usage: overlap_timeline.py kernel_trace.csv [baseline.csv] [--list]
With a second trace (the plain sequence on the same box) the per-symbol baseline comes from that file instead.
--list: the launches of both queues around the median window, start times relative to the window's start."""
import collections
import csv
import re
import statistics
import sys


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    return rows


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^()]*>)?)", name)
    return (m.group(1) if m else name)[:60]


LIST = "--list" in sys.argv
sys.argv = [a for a in sys.argv if a != "--list"]
rows = load(sys.argv[1])
qcount = collections.Counter(r[3] for r in rows)
main_q = qcount.most_common(1)[0][0]
side = [r for r in rows if r[3] != main_q]
if not side:
    print("one queue only: nothing overlapped")
    sys.exit(0)
# side-queue bursts: consecutive side kernels less than 300 us apart
bursts, cur = [], [side[0]]
for r in side[1:]:
    if r[0] - cur[-1][1] > 300_000:
        bursts.append(cur)
        cur = [r]
    else:
        cur.append(r)
bursts.append(cur)
bursts = [b for b in bursts if len(b) >= 5]
main = [r for r in rows if r[3] == main_q]
inside, outside = collections.defaultdict(list), collections.defaultdict(list)
windows = [(b[0][0], b[-1][1]) for b in bursts]
wi = 0
prev_end = None
for s, e, name, _ in main:
    while wi < len(windows) and windows[wi][1] < s:
        wi += 1
    gap = max(0, s - prev_end) if prev_end is not None else 0
    prev_end = e
    k = short(name)
    if wi < len(windows) and windows[wi][0] <= s <= windows[wi][1]:
        inside[k].append((e - s, gap))
    else:
        outside[k].append((e - s, gap))
if len(sys.argv) > 2:
    outside = collections.defaultdict(list)
    prev_end = None
    for s, e, name, _ in load(sys.argv[2]):
        gap = max(0, s - prev_end) if prev_end is not None else 0
        prev_end = e
        outside[short(name)].append((e - s, gap))
tail_len = [(w[1] - w[0]) / 1e3 for w in windows]
tail_busy = [sum(e - s for s, e, _, _ in b) / 1e3 for b in bursts]
print(f"queues {dict(qcount)}; {len(bursts)} tail bursts: window median {statistics.median(tail_len):.0f} us, kernel time in it "
      f"{statistics.median(tail_busy):.0f} us ({statistics.median([len(b) for b in bursts]):.0f} launches)")
# main-queue work inside the window
per_burst_main = []
for (w0, w1) in windows:
    seg = [(s, e) for s, e, _, _ in main if w0 <= s <= w1]
    if seg:
        per_burst_main.append((sum(e - s for s, e in seg) / 1e3, len(seg), (seg[-1][1] - seg[0][0]) / 1e3))
if per_burst_main:
    print(f"caller's queue inside a window (median): {statistics.median([p[1] for p in per_burst_main]):.0f} launches, kernel time "
          f"{statistics.median([p[0] for p in per_burst_main]):.0f} us over a span of {statistics.median([p[2] for p in per_burst_main]):.0f} us")
print(f"{'symbol':62s} {'n/window':>8s} {'in: dur':>9s} {'gap':>7s} | {'out: dur':>9s} {'gap':>7s} | {'extra us/window':>15s}")
tot = 0.0
lines = []
for k, v in inside.items():
    n = len(v) / len(bursts)
    di, gi = statistics.mean(x[0] for x in v) / 1e3, statistics.mean(x[1] for x in v) / 1e3
    if k in outside and outside[k]:
        do, go = statistics.mean(x[0] for x in outside[k]) / 1e3, statistics.mean(x[1] for x in outside[k]) / 1e3
    else:
        do = go = float("nan")
    extra = n * ((di + gi) - (do + go)) if do == do else float("nan")
    lines.append((extra if extra == extra else 0.0, f"{k:62s} {n:8.1f} {di:9.2f} {gi:7.2f} | {do:9.2f} {go:7.2f} | {extra:15.1f}"))
    if extra == extra:
        tot += extra
for _, l in sorted(lines, key=lambda x: -x[0]):
    print(l)
print(f"caller's queue: {tot:.0f} us per window slower than the same kernels outside a window")
# and the tail's own kernels: duration against the same symbol on the main queue outside windows (the plain forward of warm-up steps has none:
# use the second trace when given)
agg = collections.defaultdict(list)
for b in bursts:
    for s, e, name, _ in b:
        agg[short(name)].append(e - s)
print("tail kernels (side queue):")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    base = outside.get(k)
    bs = f"{statistics.mean(x[0] for x in base) / 1e3:8.2f}" if base else "     n/a"
    print(f"  {k:60s} n/window {len(v) / len(bursts):5.1f}  avg {statistics.mean(v) / 1e3:8.2f} us   (plain sequence: {bs})")

if LIST:
    order = sorted(range(len(windows)), key=lambda i: windows[i][1] - windows[i][0])
    w0, w1 = windows[order[len(order) // 2]]
    print(f"launches from 150 us before the median window to 300 us after it (window = [0, {(w1 - w0) / 1e3:.0f}] us):")
    for s_, e_, name, q in rows:
        if w0 - 150_000 <= s_ <= w1 + 300_000:
            print(f"  {'main' if q == main_q else 'SIDE'} {(s_ - w0) / 1e3:9.1f} +{(e_ - s_) / 1e3:7.1f}  {short(name)}")
