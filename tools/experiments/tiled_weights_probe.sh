#!/bin/bash
# Round 6: row-major [Cout][9 Cin] weight operand vs the tile-packed one (IgemmArgs::w_tiled; csrc/igemm4.hip) in the conv harness,
# HBM-cold weights (16 rotating copies).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -mllvm
# -amdgpu-kernarg-preload-count=14 tools/bench_igemm.hip -o build/ig_base.   Harness args: H Cin Cout big ksplit gen ksize stats nbuf k2 tiled
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { for t in 0 1; do echo -n "tiled=$t  "; timeout -k 5 60 build/ig_base $@ $t | tail -1; done; }
echo "== 8x8 1024 -> 1024, 16 slices";            run 8 1024 1024 0 16 6 3 0 16 0
echo "== 8x8 2048 -> 1024, 16 slices";            run 8 2048 1024 0 16 6 3 0 8 0
echo "== 8x8 1024 -> 1024 + 2048 folded skip";    run 8 1024 1024 0 16 6 3 0 8 2048
echo "== 8x8 1024 -> 2048 (dgrad shape)";         run 8 1024 2048 0 8 6 3 0 8 0
echo "== 16x16 768 -> 768, 4 slices";             run 16 768 768 0 4 6 3 0 16 0
echo "== 16x16 768 -> 768, 8 slices";             run 16 768 768 0 8 6 3 0 16 0
echo "== 16x16 1536 -> 768, 8 slices";            run 16 1536 768 0 8 6 3 0 8 0
echo "== 32x32 512 -> 512, 2 slices";             run 32 512 512 0 2 6 3 0 16 0
echo "== 32x32 512 -> 512, 1 slice, statistics";  run 32 512 512 0 1 6 3 1 16 0
echo "== 64x64 256 -> 256 (two-team), statistics"; run 64 256 256 0 1 6 3 1 16 0
echo "== 64x64 512 -> 512 (128x64 tiles)";        run 64 512 512 0 1 6 3 1 16 0
echo "== 128x128 256 -> 256 (128x128 tiles)";     run 128 256 256 1 1 6 3 1 16 0
echo "== 128x128 512 -> 256 (128x128 tiles)";     run 128 512 256 1 1 6 3 1 16 0
