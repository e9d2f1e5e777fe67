#!/bin/bash
# rocprofv3 kernel trace of a short bench run -> gpurun_out/<tag>_stats.csv, <tag>_bygrid.txt, <tag>_step.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-prof}
O=$R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O.json 2> $O.err || exit 1
cp $(find $O -name "*kernel_stats.csv") ${O}_stats.csv
python3 $R/tools/trace_by_grid.py $(find $O -name "*kernel_trace.csv") > ${O}_bygrid.txt
python3 $R/tools/step_timeline.py $(find $O -name "*kernel_trace.csv") --list > ${O}_step.txt; python3 $R/tools/overlap_check.py $(find $O -name "*kernel_trace.csv") > ${O}_overlap.txt
rm -rf $O
echo "$TAG done"
