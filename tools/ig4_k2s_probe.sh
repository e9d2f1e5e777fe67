#!/bin/bash
# the folded 1x1 second source on igemm4's 64x64 tiles (now dealt out over the K slices) against igemm2 (harness, HBM-cold weights)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/ig4_k2s.txt; : > $O
for shape in "32 512 512 1280 2" "32 512 512 256 2" "64 256 256 512 1" "64 256 256 768 1" "32 512 512 1280 1" "32 512 512 1280 4"; do
  set -- $shape
  cold=$(( 400 * 1024 * 1024 / (($2 * 9 + $4) * $3 * 2) + 1 ))
  echo -n "igemm2 | $shape | " >> $O; timeout -k 5 60 ./build/ig4_w8 $1 $2 $3 0 $5 2 3 0 $cold $4 2>&1 | grep -E "^gen" >> $O || echo >> $O
  echo -n "igemm4 | $shape | " >> $O; ISHAP_IG4_K2_SMALL=1 timeout -k 5 60 ./build/ig4_w8 $1 $2 $3 0 $5 6 3 0 $cold $4 2>&1 | grep -E "^gen|tiled" | tr '\n' ' ' >> $O; echo >> $O
done
cat $O
