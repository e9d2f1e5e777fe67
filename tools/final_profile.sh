#!/bin/bash
# One GPU call producing the round's measurement artefacts under gpurun_out/final/ (copy into profiles/ afterwards).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout -k 10 700 python bench.py --c4-shape-profile $O/shapes_c4.csv > $O/bench.json 2> $O/bench.err || exit 1
tail -c 400 $O/bench.json
cd /tmp && export TMPDIR=/tmp
# per-shape table from live events around each conv launch (its own run: the event records open gaps in the timeline)
timeout -k 10 300 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrent --no-c2 --no-c4 --shape-profile $O/shapes.csv > $O/shapes_bench.json 2> $O/shapes.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrent --no-c2 --no-c4 > $O/trace_bench.json 2> $O/trace.err || exit 1
f=$(find $O/trace -name "*kernel_trace.csv")
python3 $R/tools/trace_by_grid.py $f > $O/kernel_by_grid.txt
python3 $R/tools/step_timeline.py $f > $O/step_timeline.txt
rm -f $f
cp $(find $O/trace -name "*kernel_stats.csv") $O/kernel_stats.csv
python3 $R/tools/shape_table.py $O/shapes.csv > $O/shapes.txt
grep -v '^#' $O/shapes_c4.csv > $O/shapes_c4_plain.csv; python3 $R/tools/shape_table.py $O/shapes_c4_plain.csv > $O/shapes_c4.txt
echo trace done
rm -rf $O/trace
