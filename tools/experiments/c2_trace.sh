#!/bin/bash
# kernel trace of the generate path (C2, forward-only steps): per-kernel table of a 60-step sample at batch 1 and 8 -> gpurun_out/c2trace/
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/c2trace
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/experiments/c2_probe.py 60 > $O/c2.log 2> $O/c2.err || exit 1
f=$(find $O/trace -name "*kernel_trace.csv")
python3 $R/tools/trace_by_grid.py $f > $O/kernel_by_grid.txt
rm -rf $O/trace
cat $O/c2.log
head -40 $O/kernel_by_grid.txt
