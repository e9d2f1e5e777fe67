#!/bin/bash
# Kernel traces of the plain sequence and of overlapped-tail settings on ONE box, then tools/overlap_timeline.py on each:
#   tools/overlap_trace.sh "ISHAP_TAIL_DEFER_WGS=128" "ISHAP_TAIL_DEFER_WGS=192" ...     (ISHAP_OVERLAP_TAIL=1 is added to every setting)
# Output: gpurun_out/overlap/{plain,<setting>}.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/overlap
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
trace() {   # name
  rocprofv3 --kernel-trace --output-format csv -d $O/t_$1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrent --no-c2 --no-c4 > $O/$1.json 2> $O/$1.err || return 1
  find $O/t_$1 -name "*kernel_trace.csv" | head -1
}
fp=$(trace plain) || exit 1
python3 $R/tools/step_timeline.py $fp > $O/plain.txt
for cfg in "$@"; do
  n=$(echo "$cfg" | tr ' =' '__')
  export ISHAP_OVERLAP_TAIL=1
  for kv in $cfg; do export "$kv"; done
  f=$(trace $n) || exit 1
  python3 $R/tools/overlap_check.py $f > $O/$n.txt
  python3 $R/tools/overlap_timeline.py $f $fp --list >> $O/$n.txt
  python3 -c "import json;print('s/shape under the tracer:', json.load(open('$O/$n.json'))['value'])" >> $O/$n.txt
  for kv in $cfg; do unset "${kv%%=*}"; done
  unset ISHAP_OVERLAP_TAIL
  rm -rf $O/t_$n
  head -4 $O/$n.txt
done
rm -rf $O/t_plain
