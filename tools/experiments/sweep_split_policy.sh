#!/bin/bash
# in-situ sweep of the split-K policy constants (workgroup fill target, minimum K-steps per slice)
for fill in ${FILLS:-96 128 160 224 320}; do for ms in ${MINSTEPS:-4 6 9 14}; do
  v=$(ISHAP_SPLIT_FILL=$fill ISHAP_SPLIT_MINSTEPS=$ms timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  echo "fill=$fill minsteps=$ms s/shape=$v"
done; done
