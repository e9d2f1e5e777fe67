#!/bin/bash
# harness builds for tools/ig4_probe.sh (ring depths of the dx-reuse conv kernel igemm4.hip)
cd $(dirname $0)/..
mkdir -p build
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=14"
b() { n=$1; shift; echo "/opt/rocm/bin/hipcc $F $@ tools/bench_igemm.hip -o build/ig4_$n"; }
{
b d -DIG4_BIG_W=6 -DIG4_BIG_X=3 -DIG4_SMALL_W=8 -DIG4_SMALL_X=4
b s -DIG4_BIG_W=5 -DIG4_BIG_X=3 -DIG4_SMALL_W=6 -DIG4_SMALL_X=3
b t -DIG4_BIG_W=4 -DIG4_BIG_X=2 -DIG4_SMALL_W=4 -DIG4_SMALL_X=2
b st -DIG_STAMPS
b nomfma -DABL_NOMFMA
} | xargs -P 5 -I{} bash -c "{}" 2>&1 | grep -E "error" 
ls -la build/ig4_*
