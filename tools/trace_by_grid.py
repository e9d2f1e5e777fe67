"""Aggregate a rocprofv3 kernel_trace.csv by (kernel name with template arguments, grid, workgroup, LDS bytes):
launches, total ms, average us.  Usage: trace_by_grid.py kernel_trace.csv [name substring]"""
import collections
import csv
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^()]*>)?)", name)
    return (m.group(1) if m else name)[:64]


agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    key = (short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")),
           r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?"), r.get("Scratch_Size", "?"))
    a = agg[key]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
tot = sum(v[1] for v in agg.values())
pat = sys.argv[2] if len(sys.argv) > 2 else ""
print(f"total {tot:.1f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if pat in k[0]:
        print(f"{v[1]:9.2f} ms {100 * v[1] / tot:5.2f}% n={v[0]:6d} avg={v[1] / v[0] * 1e3:8.2f}us grid={k[1]:>8} wg={k[2]:>4} lds={k[3]:>6} vgpr={k[4]:>3} scr={k[5]:>4} {k[0]}")
