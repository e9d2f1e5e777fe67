#!/bin/bash
# Round 6: 128-pixel x 64-channel tiles (igemm4_kernel<128,64,W,...>) on the SLICED launches of the 32^2 / 16^2 maps: a K-step stages
# 8 KB of weights + 6.2 KB of activations for twice the FLOPs of the 64x64 tile's 8 + 3.1 KB; with half as many tiles the launch takes
# twice the K slices to fill the chip (the consumer adds them up).  build/ig_tall = tools/bench_igemm.hip -DIG4_TALL_PROBE.
# args: H Cin Cout big ksplit gen ksize stats nbuf ; HBM-cold weights (16 rotating copies)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { echo -n "$1 ks=$3  "; timeout -k 5 60 build/ig_$1 $2 0 $3 6 3 0 16 | tail -1; }
for shape in "32 512 512" "32 1024 512" "32 768 512" "32 256 512"; do
  run base "$shape" 2; run base "$shape" 4; run tall "$shape" 2; run tall "$shape" 4; run tall "$shape" 8
done
for shape in "16 768 768" "16 1536 768" "16 512 768" "16 1280 768"; do
  run base "$shape" 4; run base "$shape" 8; run tall "$shape" 8; run tall "$shape" 16
done
