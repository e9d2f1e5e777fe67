#!/bin/bash
# per-symbol kernel time of one guided step for a list of environment settings, plain launch sequence, one box:
#   AB_GREP="gn_" tools/env_trace_ab.sh "ISHAP_GN_XCD=0" "ISHAP_GN_XCD=1"    -> gpurun_out/env_trace/<setting>.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/env_trace
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ISHAP_OVERLAP_TAIL=0
for cfg in "$@"; do
  n=$(echo "$cfg" | tr ' =' '__')
  for kv in $cfg; do export "$kv"; done
  rocprofv3 --kernel-trace --output-format csv -d $O/t_$n -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrent --no-c2 --no-c4 > $O/$n.json 2> $O/$n.err || exit 1
  f=$(find $O/t_$n -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/step_timeline.py $f > $O/$n.txt
  rm -rf $O/t_$n
  for kv in $cfg; do unset "${kv%%=*}"; done
  echo "== $cfg: $(head -1 $O/$n.txt)"
  grep -E "${AB_GREP:-gn_}" $O/$n.txt | awk '{k+=$4} END {printf "   sum of matching kernel time %.1f us\n", k}'
  grep -E "${AB_GREP:-gn_}" $O/$n.txt | head -${AB_LINES:-14}
done
