#!/bin/bash
# grid cap of the big-map GroupNorm apply kernel: per-kernel averages from a short trace
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for cap in 1024 512 256 2048; do
  export ISHAP_GN_APPLY_BLOCKS=$cap
  rm -rf /tmp/gp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/gp.json 2>/dev/null || exit 1
  python3 - "$cap" $(find /tmp/gp -name "*kernel_stats.csv") <<'PY'
import csv, sys
tot = 0
print("cap", sys.argv[1], end="  ")
for r in csv.DictReader(open(sys.argv[2])):
    if "gn_apply_kernel" in r["Name"]:
        tot += float(r["TotalDurationNs"])
        print(r["Name"].split("gn_apply_kernel")[1][:28], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), end=" | ")
print(" total ms", round(tot / 1e6, 2))
PY
done
