#!/bin/bash
# K slices of igemm4's sliced launches on the 16x16 maps (harness, HBM-cold weights)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/ig4_w16.txt; : > $O
for shape in "16 768 768 0" "16 1024 1024 0" "16 512 512 0" "16 1536 768 0" "16 768 1536 0" "16 768 768 768"; do
  set -- $shape
  cold=$(( 400 * 1024 * 1024 / (($2 * 9 + $4) * $3 * 2) + 1 ))
  for ks in 2 3 4 5 6 8 12 16; do echo -n "igemm4 ks=$ks | $shape | " >> $O; timeout -k 5 60 ./build/ig4_w8 $1 $2 $3 0 $ks 6 3 0 $cold $4 2>&1 | grep -E "^gen|tiled" | tr '\n' ' ' >> $O; echo >> $O; done
done
cat $O
