#!/bin/bash
# igemm4 with the fragment reads interleaved into the MFMA blocks (-DIG4_INTERLEAVE) against the burst form (harness, HBM-cold weights)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/ig4_il.txt; : > $O
run() { # name binary H Cin Cout big ks stats k2
  cold=$(( 400 * 1024 * 1024 / (($4 * 9 + $9) * $5 * 2) + 1 ))
  echo -n "$1 | $3 $4 $5 big=$6 ks=$7 k2=$9 | " >> $O
  timeout -k 5 60 ./build/$2 $3 $4 $5 $6 $7 6 3 $8 $cold $9 2>&1 | grep -E "^gen|tiled" | tr '\n' ' ' >> $O; echo >> $O
}
for v in base il il_noload; do
  run $v ig4_$v 128 256 256 1 1 1 0
  run $v ig4_$v 128 512 256 1 1 1 0
  run $v ig4_$v 128 256 256 1 1 1 512
  run $v ig4_$v 64 256 256 0 1 1 0
  run $v ig4_$v 64 512 512 0 1 1 0
  run $v ig4_$v 32 512 512 0 2 0 0
  run $v ig4_$v 16 768 768 0 8 0 0
  run $v ig4_$v 8 1024 1024 0 16 0 0
done
cat $O
