#!/bin/bash
# One GPU call producing the round's measurement artefacts under gpurun_out/final/ (copy into profiles/ afterwards, pmc_traffic.json
# included: bench.py quotes it, so it must come from the same run as the pmc_*.txt files next to it).
# The headline bench runs in the default configuration (overlapped forward tail); every per-kernel leg -- shape profile, kernel
# trace, counter passes -- runs the PLAIN launch sequence (ISHAP_OVERLAP_TAIL=0): per-kernel durations of two queues sharing the
# chip are not per-kernel figures, and rocprofv3's queue interception distorts two-queue runs (profiles/round5_overlap_tail_ab.txt).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python bench.py --c4-shape-profile $O/shapes_c4.csv > $O/bench.json 2> $O/bench.err || exit 1
tail -c 400 $O/bench.json
cd /tmp && export TMPDIR=/tmp
export ISHAP_OVERLAP_TAIL=0
# per-shape table from live events around each conv launch (its own run: the event records open gaps in the timeline)
timeout -k 10 300 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrent --no-c2 --no-c4 --shape-profile $O/shapes.csv > $O/shapes_bench.json 2> $O/shapes.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrent --no-c2 --no-c4 > $O/trace_bench.json 2> $O/trace.err || exit 1
f=$(find $O/trace -name "*kernel_trace.csv")
python3 $R/tools/trace_by_grid.py $f > $O/kernel_by_grid.txt
python3 $R/tools/step_timeline.py $f --json $O/step_breakdown.json > $O/step_timeline.txt
rm -f $f
cp $(find $O/trace -name "*kernel_stats.csv") $O/kernel_stats.csv
python3 $R/tools/shape_table.py $O/shapes.csv > $O/shapes.txt
grep -v '^#' $O/shapes_c4.csv > $O/shapes_c4_plain.csv; python3 $R/tools/shape_table.py $O/shapes_c4_plain.csv > $O/shapes_c4.txt
echo trace done
rm -rf $O/trace
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/tools/pmc_step.py > $O/pmc_$c.json 2> $O/pmc_$c.err || exit 1
  f=$(find $O/pmc_$c -name "*counter_collection.csv"); python3 $R/tools/pmc_summary.py $f > $O/pmc_$c.txt; rm -rf $O/pmc_$c
  echo pmc $c done
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 $R/tools/pmc_step.py > $O/pmc_mfma.json 2> $O/pmc_mfma.err || exit 1
f=$(find $O/pmc_mfma -name "*counter_collection.csv"); python3 $R/tools/pmc_mfma_summary.py $f > $O/pmc_mfma.txt; rm -rf $O/pmc_mfma
python3 $R/tools/pmc_traffic_json.py $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt $O/pmc_traffic.json $O/pmc_mfma.txt
echo pmc done
