#!/bin/bash
# small-map 3x3 kernel probes: K slices sweep (z), ablations, tiled kernel for comparison; HBM-cold weights (nbuf copies cycled)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for shp in "8 1024 1024 16" "8 2048 1024 8" "16 768 768 24" "16 1792 768 12" "32 512 512 40" "32 768 512 30"; do
  set -- $shp
  nb=$4
  echo "== H=$1 Cin=$2 Cout=$3 nbuf=$nb"
  for z in 1 2 3 4 6 8; do
    printf "small3 z=%-2s " $z; timeout -k 5 60 ./build/bi $1 $2 $3 0 $z 5 3 0 $nb | tr '\n' ' ' | sed 's/one-launch vs tiled: //; s/gen5.*stats=0 ://'; echo
  done
  for b in biDS3_ABL_NOX biDS3_ABL_NOW biDS3_ABL_NOXDS3_ABL_NOW; do
    printf "%-26s z=4 " $b; timeout -k 5 60 ./build/$b $1 $2 $3 0 4 5 3 0 $nb | tail -1 | sed 's/gen5.*stats=0 ://'
  done
  printf "tiled: "; for ks in 2 4 8 16; do timeout -k 5 60 ./build/bi $1 $2 $3 0 $ks 2 3 0 $nb | tail -1 | awk '{printf "%s us(ks=%s)  ", $(NF-3), "'$ks'"}'; done; echo
done
