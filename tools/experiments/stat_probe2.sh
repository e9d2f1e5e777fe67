#!/bin/bash
# statistics cost of the staged epilogue: no statistics / one table / 8 / 32 copies of the table (harness, warm)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for b in bs bsDABL_STAT_COPIES8 bsDABL_STAT_COPIES32; do
  for st in 0 1; do
    echo -n "$b stats=$st "; ./build/$b 128 256 256 1 1 2 3 $st 4 | grep gen
    echo -n "$b stats=$st "; ./build/$b 64 256 256 0 1 2 3 $st 4 | grep gen
    echo -n "$b stats=$st "; ./build/$b 64 512 256 0 1 2 3 $st 4 | grep gen
  done
done
