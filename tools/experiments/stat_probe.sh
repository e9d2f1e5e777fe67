#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for b in bi4 bi4DABL_STAT_COPIES8 bi4DABL_STAT_COPIES32; do
  for st in 0 1; do
    echo -n "$b "; ./build/$b 128 256 256 1 1 2 3 $st 4 | grep gen
    echo -n "$b "; ./build/$b 64 256 256 0 1 2 3 $st 4 | grep gen
  done
done
