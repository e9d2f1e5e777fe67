#!/bin/bash
# in-situ A/B of launch-policy switches on one box: bench wall per edit (3 edits) per setting
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "$1: $(timeout -k 5 120 env $1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;print(json.load(sys.stdin)['value'])" 2>/dev/null || echo failed)"; }
run "X_DEFAULT=1"
for kv in ISHAP_S3_WGS=256 ISHAP_S3_WGS=512 ISHAP_S3_WGS=640 ISHAP_TEAM_STEPS=8 ISHAP_TEAM_STEPS=24 ISHAP_TEAM_TILES=512 ISHAP_SKINNY=2 ISHAP_SKINNY=0 ISHAP_SMALL3=2 ISHAP_LOCAL_GN=2; do
  run "$kv"
done
run "X_DEFAULT=2"
