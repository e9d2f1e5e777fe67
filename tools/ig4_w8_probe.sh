#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/ig4_w8.txt; : > $O
for shape in "8 1024 1024" "8 2048 1024" "8 1024 2048"; do
  set -- $shape
  cold=$(( 400 * 1024 * 1024 / ($2 * $3 * 18) + 1 ))
  for ks in 3 4 6; do echo -n "small3 ks=$ks | $shape | " >> $O; timeout -k 5 60 ./build/ig4_w8 $1 $2 $3 0 $ks 5 3 0 $cold 2>&1 | grep -E "^gen" >> $O || echo >> $O; done
  for ks in 4 8 12 16; do echo -n "igemm4 ks=$ks | $shape | " >> $O; ISHAP_IG4_W8=1 timeout -k 5 60 ./build/ig4_w8 $1 $2 $3 0 $ks 6 3 0 $cold 2>&1 | grep -E "^gen|tiled" | tr '\n' ' ' >> $O; echo >> $O; done
done
cat $O
