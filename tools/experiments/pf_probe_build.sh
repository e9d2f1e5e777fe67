#!/bin/bash
# builds the conv harness in the variants tools/experiments/pf_probe.sh runs (here, on the CPU box; the binaries travel under build/)
cd $(dirname $0)/../..
mkdir -p build
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=14"
b() { n=$1; shift; echo "/opt/rocm/bin/hipcc $F $@ tools/bench_igemm.hip -o build/pf_$n"; }
{
b base
b pf8 -DIG2_PF_DIST=8
b pf16 -DIG2_PF_DIST=16
b pf99 -DIG2_PF_DIST=999
b pf16s -DIG2_PF_DIST=16 -DIG2_PF_SHARE=1
b pf99s -DIG2_PF_DIST=999 -DIG2_PF_SHARE=1
b now -DABL_NOW
b nox -DABL_NOX
b samex -DABL_SAMEX
b nst8 -DIG2_SMALL_NST=8
b nst5 -DIG2_BIG_NST=5
b nst8pf -DIG2_SMALL_NST=8 -DIG2_PF_DIST=999
b st_base -DIG_STAMPS
b st_pf99 -DIG_STAMPS -DIG2_PF_DIST=999
} | xargs -P 7 -I{} bash -c "{}"
ls -la build/pf_*
