#!/bin/bash
# un-profiled s/shape for a list of environment settings, two alternating rounds on one box:  tools/env_ab.sh "A=1" "A=2 B=3" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
  for cfg in "$@"; do
    v=$(env $cfg timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-c2 --no-c4 --no-concurrent 2>/dev/null | python -c "import json,sys;print(json.loads(sys.stdin.read())['value'])")
    echo "round $round [$cfg]: $v"
  done
done
