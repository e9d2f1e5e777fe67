#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for shp in "128 256 256 1" "128 128 256 1" "128 512 256 1" "64 256 256 0" "64 512 256 0" "64 512 512 0" "32 512 512 0" "32 1024 512 0"; do
  set -- $shp
  for gen in 2 3; do
    timeout -k 5 60 ./build/bi2 $1 $2 $3 $4 1 $gen 3 0 4 | grep gen
  done
done
