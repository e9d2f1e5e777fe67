#!/bin/bash
# sweep the grid caps of the drag-loss launches (bench wall + per-kernel averages from a short trace)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for cfg in "1024 512" "1024 1024"; do
  set -- $cfg
  export ISHAP_DRAG_BLOCKS=$1 ISHAP_DRAG_OUT_BLOCKS=$2
  rm -rf /tmp/dp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/dp.json 2>/dev/null || exit 1
  python3 - "$1 $2" $(find /tmp/dp -name "*kernel_stats.csv") <<'PY'
import csv, json, sys
print("cap", sys.argv[1], "bench", json.load(open("/tmp/dp.json"))["value"], end="  ")
for r in csv.DictReader(open(sys.argv[2])):
    if r["Name"].startswith(("drag_terms", "drag_gather", "_Z17drag_scale")):
        print(r["Name"][:14], round(float(r["AverageNs"]) / 1e3, 1), "us", end="  ")
print()
PY
done
