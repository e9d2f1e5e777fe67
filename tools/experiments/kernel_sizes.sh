#!/bin/bash
# Code bytes of every kernel in the library (instruction-cache footprint: a launch re-fetches what it executes).
cd $(dirname $0)/../..
T=$(mktemp -d)
for f in ishapediting_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only --no-gpu-bundle-output -c $f -o $T/$b.co 2>/dev/null || continue
  /opt/rocm/lib/llvm/bin/llvm-readelf -s --wide $T/$b.co | awk -v f=$b '$4=="FUNC" {printf "%-14s %7d B  %s\n", f, $3, $8}'
done | sort -k2 -n -r | c++filt | cut -c1-150
rm -rf $T
