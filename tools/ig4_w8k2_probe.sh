#!/bin/bash
# igemm4's sliced launch on the 8x8 maps with the folded 1x1 second source, against conv3_small (harness, HBM-cold weights)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/ig4_w8k2.txt; : > $O
for shape in "8 1024 1024 2048" "8 1024 1024 1792" "8 1024 1024 768" "8 1792 1024 0" "8 1024 768 0"; do
  set -- $shape
  cold=$(( 400 * 1024 * 1024 / (($2 * 9 + $4) * $3 * 2) + 1 ))
  for ks in 4 6; do echo -n "small3 ks=$ks | $shape | " >> $O; timeout -k 5 60 ./build/ig4_w8 $1 $2 $3 0 $ks 5 3 0 $cold $4 2>&1 | grep -E "^gen" >> $O || echo >> $O; done
  for ks in 8 12 16; do echo -n "igemm4 ks=$ks | $shape | " >> $O; timeout -k 5 60 ./build/ig4_w8 $1 $2 $3 0 $ks 6 3 0 $cold $4 2>&1 | grep -E "^gen|tiled" | tr '\n' ' ' >> $O; echo >> $O; done
done
cat $O
