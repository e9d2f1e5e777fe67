#!/bin/bash
# LDS-utilisation counter pass (its own run: --pmc with --kernel-trace only), summarised per kernel symbol.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_lds -- python3 $R/tools/pmc_step.py > $O/pmc_lds.json 2> $O/pmc_lds.err || exit 1
f=$(find $O/pmc_lds -name "*counter_collection.csv"); python3 $R/tools/experiments/pmc_lds_summary.py $f > $O/pmc_lds.txt; rm -rf $O/pmc_lds
echo pmc lds done
