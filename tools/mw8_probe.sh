#!/bin/bash
# 128x128 conv kernel with four vs eight MFMA waves (ISHAP_MW8=0|1), warm harness, cold weights (4 copies), statistics on
cd ${GRAFT_REPO_ROOT:-/root/repo}
for shape in "128 256 256 1" "128 512 256 1" "128 256 128 1" "64 512 512 1"; do
  for mw in 0 1; do
    echo -n "ISHAP_MW8=$mw "; ISHAP_MW8=$mw ./build/bi_cur $shape 1 2 3 1 4 | grep gen
  done
done
