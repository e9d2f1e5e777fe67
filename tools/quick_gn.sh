#!/bin/bash
# A/B aid: un-profiled bench value, then per-symbol kernel averages of the GroupNorm kernels from a kernel trace.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-ab}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-c2 > $O/bench.json 2> $O/bench.err || exit 1
python -c "import json;d=json.load(open('$O/bench.json'));print('s/shape',d['value'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-c2 > $O/trace_bench.json 2> $O/trace.err || exit 1
f=$(find $O/trace -name "*kernel_trace.csv")
python3 $R/tools/step_timeline.py $f > $O/step_timeline.txt
python3 $R/tools/trace_by_grid.py $f > $O/by_grid.txt
rm -rf $O/trace
head -1 $O/step_timeline.txt
grep -E "gn_|${2:-gn_}" $O/step_timeline.txt | head -${3:-30}
