#!/bin/bash
# warm (4 weight copies) vs HBM-cold (copies cycling through > 300 MB) weights, harness
cd ${GRAFT_REPO_ROOT:-/root/repo}
export ISHAP_HALVES=2
for shp in "32 512 512 0 1 1 600" "32 1536 512 0 1 1 200" "16 768 768 0 1 1 300" "16 768 768 0 8 3 30" "32 512 512 0 2 3 70" "64 256 256 0 1 3 300" "128 256 256 1 1 3 300" "8 1024 1024 0 4 3 20"; do
  set -- $shp
  for nb in 4 $7; do
    gen=2; [ "$1" = "8" ] && gen=5
    echo -n "nbuf=$nb "; timeout -k 5 60 ./build/bi2 $1 $2 $3 $4 $5 $gen $6 0 $nb 2>&1 | grep "^gen" | cut -c1-100
  done
done
