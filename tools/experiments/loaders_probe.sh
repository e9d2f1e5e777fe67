#!/bin/bash
# 4 vs 8 loader waves on the 64x64-tile kernel (single team: ISHAP_HALVES=1), harness
cd ${GRAFT_REPO_ROOT:-/root/repo}
export ISHAP_HALVES=1
for b in bi2 bi3DIG2_LOADERS8; do
  for shp in "64 256 256 0 3" "64 512 256 0 3" "32 512 512 0 3" "32 512 512 0 1" "32 1536 512 0 1" "16 768 768 0 1"; do
    set -- $shp
    echo -n "$b "; timeout -k 5 60 ./build/$b $1 $2 $3 $4 1 2 $5 0 4 2>&1 | grep gen | cut -c1-100
  done
done
