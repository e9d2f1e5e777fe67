#!/bin/bash
# harness builds for tools/ig4_probe.sh (ring depths / prologue forms of the dx-reuse conv kernel igemm4.hip)
cd $(dirname $0)/..
mkdir -p build
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=14"
b() { n=$1; shift; echo "/opt/rocm/bin/hipcc $F $@ tools/bench_igemm.hip -o build/ig4_$n"; }
{
b d
b noramp -DIG4_PRO=9
b pro1 -DIG4_PRO=1
b pro3 -DIG4_PRO=3
b deep -DIG4_BIG_W=6 -DIG4_BIG_X=3 -DIG4_SMALL_W=8 -DIG4_SMALL_X=4 -DIG4_TEAM_W=5 -DIG4_TEAM_X=3
b shal -DIG4_BIG_W=4 -DIG4_BIG_X=2 -DIG4_SMALL_W=4 -DIG4_SMALL_X=2 -DIG4_TEAM_W=4 -DIG4_TEAM_X=2
b st -DIG_STAMPS
b noload -DABL_NOLOAD
b nomfma -DABL_NOMFMA
} | xargs -P 7 -I{} bash -c "{}" 2>&1 | grep -E "error" 
ls -la build/ig4_*
