#!/bin/bash
# step timeline + per-grid kernel table + per-shape conv table of the current build (no full bench line): gpurun_out/quick/
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/quick
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrent --no-c2 --no-c4 --shape-profile $O/shapes.csv > $O/shapes_bench.json 2> $O/shapes.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrent --no-c2 --no-c4 > $O/trace_bench.json 2> $O/trace.err || exit 1
f=$(find $O/trace -name "*kernel_trace.csv")
python3 $R/tools/trace_by_grid.py $f > $O/kernel_by_grid.txt
python3 $R/tools/step_timeline.py $f > $O/step_timeline.txt
python3 $R/tools/shape_table.py $O/shapes.csv > $O/shapes.txt
rm -rf $O/trace
head -50 $O/step_timeline.txt
