#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for b in bi2 bi3DIG2_LOADERS8 bi3DABL_NOMFMA bi3DIG2_LOADERS8DABL_NOMFMA; do
  for shp in "128 256 256 1" "128 512 256 1" "64 256 256 0" "64 512 256 0" "32 512 512 0"; do
    set -- $shp
    echo -n "$b "; timeout -k 5 60 ./build/$b $1 $2 $3 $4 1 2 3 0 4 | grep gen
  done
done
