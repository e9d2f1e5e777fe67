#!/bin/bash
# in-situ sweep of one policy constant: sweep_env.sh VAR v1 v2 ...   (prints s/shape of bench.py per value)
var=$1; shift
for v in "$@"; do
  r=$(env $var=$v timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  echo "$var=$v s/shape=$r"
done
