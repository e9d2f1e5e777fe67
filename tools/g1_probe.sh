#!/bin/bash
# 1x1 GEMMs on the 8x8 / 16x16 maps: the tiled kernel with K slices left pending against the one-launch skinny kernel (harness, HBM-cold weights)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/g1.txt; : > $O
for shape in "8 1024 1024" "8 3072 1024" "8 1024 3072" "16 768 768" "16 2304 768" "16 768 2304"; do
  set -- $shape
  cold=$(( 400 * 1024 * 1024 / ($2 * $3 * 2) + 1 ))
  for mt in 1 2 4; do echo -n "skinny mt=$mt | $shape | " >> $O; timeout -k 5 60 ./build/ig4_w8 $1 $2 $3 0 $mt 4 1 0 $cold 2>&1 | grep -E "^gen" >> $O || echo >> $O; done
  for ks in 1 2 4 8 16; do echo -n "igemm2 ks=$ks | $shape | " >> $O; timeout -k 5 60 ./build/ig4_w8 $1 $2 $3 0 $ks 2 1 0 $cold 2>&1 | grep -E "^gen" >> $O || echo >> $O; done
done
cat $O
