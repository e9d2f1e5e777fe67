"""Per kernel symbol, from a rocprofv3 --pmc counter_collection.csv holding SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE,
SQ_INSTS_VALU_MFMA_MOPS_F16 and SQ_BUSY_CU_CYCLES: dispatches, and
  mfma_util  = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)   (rocprofv3's MfmaUtil formula: GRBM_GUI_ACTIVE is
               summed over the 8 XCDs, MI355X_MICROARCH.md 'DVFS give-back'; 256 CUs x 4 SIMDs)
  mfma_flops = SQ_INSTS_VALU_MFMA_MOPS_F16 * 512 per dispatch (the executed fp16 MFMA FLOPs).
Usage: pmc_mfma_summary.py counter_collection.csv"""
import collections
import csv
import re
import sys


def short(name):
    """kernel symbol with its template arguments, without the argument list / anonymous-namespace prefix"""
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^()]*>)?)", name)
    return (m.group(1) if m else name)[:70]

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    name = short(r["Kernel_Name"])
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (name, r.get("Dispatch_Id"))
    if key not in seen:
        seen.add(key)
        cnt[name] += 1
print(f"{'kernel':70s} {'dispatches':>10s} {'mfma_util_%':>11s} {'GFLOP/disp':>11s} {'busy_cu_%':>9s}")
for name, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if gui <= 0:
        continue
    util = 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024.0)
    gflop = c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) * 512 / 1e9 / max(cnt[name], 1)
    busy = 100.0 * c.get("SQ_BUSY_CU_CYCLES", 0.0) / (gui * 256.0)
    print(f"{name:70s} {cnt[name]:10d} {util:11.2f} {gflop:11.3f} {busy:9.1f}")
