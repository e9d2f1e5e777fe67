#!/bin/bash
# un-profiled s/shape for a list of environment settings, two alternating rounds on one box:  tools/env_ab.sh "A=1" "A=2 B=3" ...
# (AB_ROUNDS sets the number of rounds; a failing run prints the tail of its stderr instead of a number)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for round in $(seq 1 ${AB_ROUNDS:-2}); do
  for cfg in "$@"; do
    v=$(env $cfg timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-c2 --no-c4 --no-concurrent 2>gpurun_out/env_ab.err | python -c "import json,sys;print(json.loads(sys.stdin.read())['value'])" 2>/dev/null) || v="FAILED: $(tail -c 400 gpurun_out/env_ab.err | tr '\n' ' ')"
    echo "round $round [$cfg]: $v"
  done
done
