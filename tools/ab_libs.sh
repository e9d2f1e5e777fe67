#!/bin/bash
# Same-box A/B of library builds: tools/ab_libs.sh build/lib_A.so build/lib_B.so ...  (two alternating rounds of the
# un-profiled bench, then one kernel trace per build: kernel time per guided step and the per-symbol lines matching $AB_GREP).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab
mkdir -p $O
cd $R
cp ishapediting_amd/libishap_hip.so $O/lib_installed.so            # restored at the end: a variant must not stay installed
trap 'cp $O/lib_installed.so $R/ishapediting_amd/libishap_hip.so' EXIT
for round in 1 2; do
  for lib in "$@"; do
    cp $lib ishapediting_amd/libishap_hip.so
    v=$(timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-c2 --no-c4 --no-concurrent 2> $O/bench.err | python -c "import json,sys;print(json.loads(sys.stdin.read())['value'])") || exit 1
    echo "round $round $(basename $lib) s/shape $v"
  done
done
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  n=$(basename $lib .so)
  cp $R/$lib $R/ishapediting_amd/libishap_hip.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$n -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-c2 --no-c4 --no-concurrent > $O/trace_$n.json 2> $O/trace_$n.err || exit 1
  f=$(find $O/trace_$n -name "*kernel_trace.csv")
  python3 $R/tools/step_timeline.py $f > $O/step_$n.txt
  rm -rf $O/trace_$n
  echo "== $n: $(head -1 $O/step_$n.txt)"
  grep -E "${AB_GREP:-gn_}" $O/step_$n.txt | head -${AB_LINES:-12}
done
