#!/bin/bash
# what a GroupNorm apply + SiLU of the activation slabs, done in LDS by igemm4's loader waves, would cost the K loop
# (tools/bench_igemm.hip -DABL_XFORM: timing probe, wrong results), against the gn_apply launch + boundary it would remove (8.6 + 1.5 us
# on 128^2 x 256).  Build on the CPU side first:
#   F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=14"
#   hipcc $F tools/bench_igemm.hip -o build/ig_base; hipcc $F -DABL_XFORM tools/bench_igemm.hip -o build/ig_xform
#   hipcc $F -DABL_XFORM -DABL_XFORM_FILM tools/bench_igemm.hip -o build/ig_xform_film; hipcc $F -DIG_STAMPS tools/bench_igemm.hip -o build/ig_stamps
# args of the harness: H Cin Cout big ksplit gen ksize stats nbuf
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for shape in "128 256 256 1" "128 512 256 1" "64 512 512 0" "64 256 256 0" "64 768 256 0" "32 512 512 0"; do
  for v in base xform xform_film; do
    echo "== $v: $shape"
    timeout -k 5 60 build/ig_$v $shape 1 6 3 1 8 | tail -1
  done
done
echo "== stamps: 64^2 256 -> 256 (M=4096, N=256, K=2304), statistics epilogue, HBM-cold weights"
timeout -k 5 60 build/ig_stamps 64 256 256 0 1 6 3 1 8 | tail -14
echo "== stamps: 128^2 256 -> 256"
timeout -k 5 60 build/ig_stamps 128 256 256 1 1 6 3 1 8 | tail -14
