"""Per kernel symbol, from a rocprofv3 --pmc counter_collection.csv holding SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT,
SQ_ACTIVE_INST_LDS, SQ_INSTS_LDS and GRBM_GUI_ACTIVE:
  lds_util      = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE/8 * 256 CUs)     (rocprofv3's LdsUtil formula; GRBM_GUI_ACTIVE is summed over
                  the 8 XCDs): the share of the kernel's cycles in which a CU's LDS was processing an index (read / write) pass
  bank_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE               (share of those cycles lost to bank conflicts)
  lds_insts     = SQ_INSTS_LDS per dispatch
Usage: pmc_lds_summary.py counter_collection.csv"""
import collections
import csv
import re
import sys


def short(name):
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
print(f"{'kernel':70s} {'dispatches':>10s} {'lds_util_%':>10s} {'conflict_%':>10s} {'lds_insts/disp':>14s}")
for name, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0.0)):
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if gui <= 0:
        continue
    idx = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    util = 100.0 * idx / (gui * 256.0)
    conf = 100.0 * c.get("SQ_LDS_BANK_CONFLICT", 0.0) / idx if idx > 0 else 0.0
    print(f"{name:70s} {cnt[name]:10d} {util:10.1f} {conf:10.1f} {c.get('SQ_INSTS_LDS', 0.0) / max(cnt[name], 1):14.0f}")
