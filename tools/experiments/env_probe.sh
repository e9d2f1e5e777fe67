#!/bin/bash
# HIP-runtime environment knobs against the bench (3 edits each): does any of them lower the per-launch cost?
# (ROC_SYSTEM_SCOPE_SIGNAL=0 hangs the process on this stack: not in the list)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "$1: $(timeout -k 5 100 env $1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;print(json.load(sys.stdin)['value'])" 2>/dev/null || echo failed)"; }
for kv in AMD_OPT_FLUSH=0 GPU_FLUSH_ON_EXECUTION=1 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1 ROC_USE_FGS_KERNARG=0 DEBUG_HIP_KERNARG_COPY_OPT=0 ROC_ACTIVE_WAIT_TIMEOUT=0 AMD_DIRECT_DISPATCH=0 ROC_SKIP_KERNEL_ARG_COPY=1; do
  run "$kv"
done
