#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/ig4_tall.txt; : > $O
for shape in "64 512 512" "64 768 512" "64 256 512" "64 1024 512"; do
  set -- $shape
  cold=$(( 300 * 256 * 256 / ($2 * $3) + 1 ))
  for tall in 0 1; do
    echo -n "tall=$tall | $shape | " >> $O
    ISHAP_IG4_TALL=$tall timeout -k 5 60 ./build/ig4_tall $1 $2 $3 0 1 6 3 1 $cold 2>&1 | grep -E "^gen|tiled" | tr '\n' ' ' >> $O; echo >> $O
  done
done
cat $O
