#!/bin/bash
# halo-reuse kernel (gen 3) against the LDS-DMA kernel (gen 2), warm harness, cold weights (4 copies), statistics on
cd ${GRAFT_REPO_ROOT:-/root/repo}
for shape in "128 256 256 1" "128 512 256 1" "128 256 128 1" "64 256 256 0" "64 512 256 0" "64 512 512 0" "32 512 512 0"; do
  for gen in 2 3; do
    ./build/bi_cur $shape 1 $gen 3 1 4 | grep gen
  done
done
