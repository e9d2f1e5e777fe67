#!/bin/bash
# weight-ring depth of igemm4's 64x64 tiles by K-steps per slice (harness, HBM-cold weights): 6/3 (default), 5/3, 4/3
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/ig4_ring2.txt; : > $O
for shape in "16 768 768 0 8 0" "16 1024 1024 0 4 0" "16 512 512 0 8 0" "16 1536 768 0 8 0" "32 512 512 0 2 0" "32 256 512 0 4 0" "32 768 512 0 2 0" "64 256 256 0 1 1" "64 512 512 0 1 1" "64 256 512 0 1 1"; do
  set -- $shape
  cold=$(( 400 * 1024 * 1024 / (($2 * 9 + $4) * $3 * 2) + 1 ))
  for v in w8r6 s53 s43; do echo -n "$v | $shape | " >> $O; timeout -k 5 60 ./build/ig4_$v $1 $2 $3 0 $5 6 3 $6 $cold $4 2>&1 | grep -E "^gen" | tr '\n' ' ' >> $O; echo >> $O; done
done
cat $O
