#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...]: the whole library with extra compiler flags -> build/lib_NAME.so
# (for same-box A/B runs with tools/ab_libs.sh; the product build is ishapediting_amd/build.py)
set -e
cd $(dirname $0)/..
name=$1; shift
O=/tmp/ishap_variant_$name
mkdir -p $O build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=14"
srcs=$(python -c "import ishapediting_amd.build as b; print(' '.join(b.SOURCES))")
for s in $srcs; do
  echo "/opt/rocm/bin/hipcc $FLAGS $@ -c ishapediting_amd/csrc/$s -o $O/${s%.hip}.o"
done | xargs -P 6 -I{} bash -c "{}"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/lib_$name.so $O/*.o
ls -la build/lib_$name.so
