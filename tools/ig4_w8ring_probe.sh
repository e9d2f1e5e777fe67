#!/bin/bash
# ring depth of igemm4's sliced launches on the 8x8 maps (harness, HBM-cold weights): 6/3 (default), 5/3, 4/3, 4/2
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/ig4_w8ring.txt; : > $O
for shape in "8 1024 1024 0 16" "8 2048 1024 0 16" "8 1024 2048 0 8" "8 1024 1024 2048 16" "8 1792 1024 0 14"; do
  set -- $shape
  cold=$(( 400 * 1024 * 1024 / (($2 * 9 + $4) * $3 * 2) + 1 ))
  for v in r6 r5 r4 r42; do echo -n "$v | $shape | " >> $O; timeout -k 5 60 ./build/ig4_w8$v $1 $2 $3 0 $5 6 3 0 $cold $4 2>&1 | grep -E "^gen|tiled" | tr '\n' ' ' >> $O; echo >> $O; done
done
cat $O
