"""Rank the (tile, ksplit) candidates measured by tools/experiments/sweep_igemm.sh; a split adds one reduce launch (~5.5 us)."""
import collections
import re
import sys
best = collections.defaultdict(list)
for l in open(sys.argv[1]):
    m = re.match(r"ksize=(\d) gen2 H=(\d+) Cin=(\d+) Cout=(\d+) big=(\d) ksplit=(\d+) : ([\d.]+) us", l)
    if not m:
        continue
    ks, H, ci, co, big, sp, us = m.groups()
    us, sp = float(us), int(sp)
    best[(int(ks), int(H), int(ci), int(co))].append((us + (5.5 if sp > 1 else 0), us, int(big), sp))
for k, v in best.items():
    v.sort()
    print(k, " | ".join(f"big={b} ks={s}: {u:.1f} (+red {t:.1f})" for t, u, b, s in v[:5]))
