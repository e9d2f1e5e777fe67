#!/bin/bash
# L2 hit-rate counter pass (its own run: --pmc with --kernel-trace only), per kernel symbol: TCC_HIT_sum / TCC_MISS_sum.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_l2 -- python3 $R/tools/pmc_step.py > $O/pmc_l2.json 2> $O/pmc_l2.err || exit 1
f=$(find $O/pmc_l2 -name "*counter_collection.csv"); python3 $R/tools/pmc_summary.py $f > $O/pmc_l2_raw.txt; rm -rf $O/pmc_l2
python3 - <<PY
import re, collections
d = collections.defaultdict(dict)
for line in open("$O/pmc_l2_raw.txt"):
    m = re.match(r"(.+?)\s+(TCC_HIT_sum|TCC_MISS_sum)\s+dispatches=\s*(\d+)\s+mean=\s*([\d.]+) KiB", line)
    if m: d[m.group(1).strip()][m.group(2)] = (int(m.group(3)), float(m.group(4)))
with open("$O/pmc_l2.txt", "w") as f:
    for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("TCC_MISS_sum", (0, 0))[1] * kv[1].get("TCC_MISS_sum", (0, 0))[0]):
        if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
            h, m_ = v["TCC_HIT_sum"][1], v["TCC_MISS_sum"][1]
            f.write(f"{k:70s} dispatches={v['TCC_HIT_sum'][0]:6d} hits/launch={h:12.0f} misses/launch={m_:12.0f} hit rate={h / max(h + m_, 1):.3f}\n")
PY
echo pmc l2 done
