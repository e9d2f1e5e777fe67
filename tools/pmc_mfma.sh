#!/bin/bash
# MFMA-utilisation counter pass (its own run: --pmc with --kernel-trace only), summarised per kernel symbol.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 $R/tools/pmc_step.py > $O/pmc_mfma.json 2> $O/pmc_mfma.err || exit 1
f=$(find $O/pmc_mfma -name "*counter_collection.csv"); python3 $R/tools/pmc_mfma_summary.py $f > $O/pmc_mfma.txt; rm -rf $O/pmc_mfma
echo pmc mfma done
