#!/bin/bash
# FETCH_SIZE per kernel symbol for a list of environment settings (one rocprofv3 --pmc pass each, plain launch sequence):
#   tools/pmc_fetch_ab.sh "ISHAP_IG4_NOUTER=0" "ISHAP_IG4_NOUTER=2"   -> gpurun_out/pmc_ab/<setting>.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_ab
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ISHAP_OVERLAP_TAIL=0
for cfg in "$@"; do
  n=$(echo "$cfg" | tr ' =' '__')
  for kv in $cfg; do export "$kv"; done
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p_$n -- python3 $R/tools/pmc_step.py > $O/$n.json 2> $O/$n.err || exit 1
  f=$(find $O/p_$n -name "*counter_collection.csv"); python3 $R/tools/pmc_summary.py $f > $O/$n.txt; rm -rf $O/p_$n
  for kv in $cfg; do unset "${kv%%=*}"; done
  echo "== $cfg"; grep -E "igemm4_kernel" $O/$n.txt | head -12
done
