#!/bin/bash
# where a small-map 3x3 launch (conv3_small_kernel, 8x8 maps) spends its time: stamps + ablations; output gpurun_out/s3_probe2.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/s3_probe2.txt; : > $O
for shape in "8 1024 1024 0 4" "8 2048 1024 0 6" "8 1024 2048 0 3" "8 1024 1024 0 2" "8 1024 1024 0 8"; do
  set -- $shape
  cold=$(( 400 * 1024 * 1024 / ($2 * $3 * 18) + 1 ))
  for v in s3 s3_nox s3_now; do echo -n "$v | $shape | " >> $O; timeout -k 5 60 ./build/$v $1 $2 $3 $4 $5 5 3 0 $cold 2>&1 | grep "^gen" >> $O; done
  echo "== stamps $shape" >> $O; timeout -k 5 60 ./build/s3_st $1 $2 $3 $4 $5 5 3 0 $cold 2>&1 | grep -v "^one" >> $O
done
cat $O
