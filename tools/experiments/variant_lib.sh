#!/bin/bash
# build/lib_NAME.so = the in-tree library with some translation units recompiled under extra -D flags (same-box A/B: tools/ab_libs.sh).
#   tools/experiments/variant_lib.sh NAME "-DGN_HOIST_PIX=0" norm.hip norm_bwd.hip
# The objects of the other sources are taken from ishapediting_amd/csrc/build/ (run __graft_entry__.build() first).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; defs=$2; shift 2
T=$(mktemp -d)
objs=""
for o in $R/ishapediting_amd/csrc/build/*.o; do
  b=$(basename $o .o)
  hit=0; for f in "$@"; do [ "$b.hip" = "$f" ] && hit=1; done
  if [ $hit = 1 ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=14 $defs -c $R/ishapediting_amd/csrc/$b.hip -o $T/$b.o
    objs="$objs $T/$b.o"
  else
    objs="$objs $o"
  fi
done
mkdir -p $R/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/lib_$name.so $objs
rm -rf $T
echo "$R/build/lib_$name.so"
