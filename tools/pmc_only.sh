#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/tools/pmc_step.py > $O/pmc_$c.json 2> $O/pmc_$c.err || exit 1
  f=$(find $O/pmc_$c -name "*counter_collection.csv"); python3 $R/tools/pmc_summary.py $f > $O/pmc_$c.txt; rm -rf $O/pmc_$c
  echo pmc $c done
done
