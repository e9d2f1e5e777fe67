#!/bin/bash
# un-profiled s/shape for group-local GroupNorm partitioning settings (two alternating rounds, same box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
  for cfg in "8 128" "8 256" "8 512" "4 128" "2 128" "8 64"; do
    set -- $cfg
    v=$(ISHAP_GN_PARTS=$1 ISHAP_GN_PART_ELEMS=$2 timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-c2 2>/dev/null | python -c "import json,sys;print(json.loads(sys.stdin.read())['value'])")
    echo "round $round parts<=$1 elems>=$2: $v"
  done
done
