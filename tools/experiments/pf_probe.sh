#!/bin/bash
# Round 4: K-step ablation table + L2 weight-prefetch / ring-depth variants of the LDS-DMA conv kernel (harness builds from
# tools/experiments/pf_probe_build.sh).  Output: gpurun_out/pf_probe.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/pf_probe.txt
mkdir -p gpurun_out; : > $O
run() { # variant, args...
  v=$1; shift
  echo -n "$v | $@ | " >> $O
  timeout -k 5 60 ./build/pf_$v "$@" 2>&1 | grep "^gen" >> $O || echo "FAILED" >> $O
}
for cold in 1 300; do
 for v in base pf8 pf16 pf99 pf16s pf99s now nox samex nst8 nst5 nst8pf; do
  run $v 128 256 256 1 1 2 3 1 $cold
  run $v 64 256 256 0 1 2 3 1 $cold
 done
done
for v in base pf16 pf99 pf99s nst8 nst8pf now nox; do
  run $v 64 512 512 0 1 2 3 1 64
  run $v 128 512 256 1 1 2 3 1 150
  run $v 32 512 512 0 2 2 3 0 64
  run $v 16 768 768 0 8 2 3 0 32
  run $v 32 512 512 0 1 2 1 1 600
  run $v 64 512 256 0 1 2 3 1 150
done
echo "== stamps base 64 256 256" >> $O; timeout -k 5 60 ./build/pf_st_base 64 256 256 0 1 2 3 1 300 >> $O 2>&1
echo "== stamps pf99 64 256 256" >> $O; timeout -k 5 60 ./build/pf_st_pf99 64 256 256 0 1 2 3 1 300 >> $O 2>&1
echo "== stamps base 128 256 256" >> $O; timeout -k 5 60 ./build/pf_st_base 128 256 256 1 1 2 3 1 300 >> $O 2>&1
echo "== stamps pf99 128 256 256" >> $O; timeout -k 5 60 ./build/pf_st_pf99 128 256 256 1 1 2 3 1 300 >> $O 2>&1
tail -5 $O
