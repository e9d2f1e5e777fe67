#!/bin/bash
# igemm2 (gen 2) vs the dx-reuse kernel igemm4 (gen 6) in the stand-alone harness; output gpurun_out/ig4_probe.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/ig4_probe.txt
mkdir -p gpurun_out; : > $O
run() { v=$1; shift; echo -n "$v teams=${ISHAP_IG4_TEAMS:-2} | $@ | " >> $O; timeout -k 5 60 ./build/ig4_$v "$@" 2>&1 | grep -E "^gen|tiled" | tr '\n' ' ' >> $O; echo >> $O; }
for shape in "128 256 256 1 1" "128 512 256 1 1" "128 256 128 1 1" "64 256 256 0 1" "64 512 256 0 1" "64 512 512 0 1" "64 768 256 0 1" "32 512 512 0 2" "32 768 512 0 2" "32 1024 512 0 2" "32 256 512 0 2"; do
  set -- $shape
  cold=$(( 300 * 256 * 256 / ($2 * $3) + 1 ))
  run d $1 $2 $3 $4 $5 2 3 1 $cold
  for v in d noramp pro1 pro3 deep shal; do run $v $1 $2 $3 $4 $5 6 3 1 $cold; done
  if [ $4 = 0 ]; then ISHAP_IG4_TEAMS=0 run d $1 $2 $3 $4 $5 6 3 1 $cold; fi
done
for shape in "128 256 256 1 1" "64 256 256 0 1"; do
  set -- $shape
  echo "== stamps gen6 $shape" >> $O; timeout -k 5 60 ./build/ig4_st $1 $2 $3 $4 $5 6 3 1 300 >> $O 2>&1
done
tail -3 $O
