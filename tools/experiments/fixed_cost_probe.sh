#!/bin/bash
# Fixed vs per-K-step cost of the tiled conv kernel on the large maps: same map, growing Cin, statistics epilogue on/off.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for H in 128 64 32; do
  for cin in 64 128 256 512 1024; do
    for st in 0 1; do
      for big in 0 1; do
        timeout -k 5 60 ./build/bi $H $cin 256 $big 1 2 3 $st 4 | grep gen
      done
    done
  done
done
