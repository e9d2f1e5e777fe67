#!/bin/bash
# Round 6: ring depth of the 64x64 GEMM kernel (igemm2_kernel<64,64,NST,false,1>) on the short-K 1x1 GEMMs of the attention blocks:
# 4 slots (3 K-steps in flight) against 6 / 8 (the whole K of a K = 512 launch in flight at once).  Build: tools/bench_igemm.hip with
# -DIG2_SMALL_NST=6|8 -> build/ig_nst6|8.  HBM-cold weights; args: H Cin Cout big ksplit gen ksize stats nbuf
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for shape in "32 512 512" "32 512 1536" "32 1536 512" "16 768 768" "16 768 2304" "16 2304 768" "64 256 256"; do
  for v in base nst6 nst8; do echo -n "$v  "; timeout -k 5 60 build/ig_$v $shape 0 1 2 1 0 16 | tail -1; done
done
