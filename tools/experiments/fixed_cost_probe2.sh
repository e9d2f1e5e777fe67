#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
BI_EMPTY=0 ./build/bi2 128 64 256 1 1 2 3 0 4 | grep empty
BI_EMPTY=131072 ./build/bi2 128 64 256 1 1 2 3 0 4 | grep empty
for b in bi2 bi2DABL_NOLOAD bi2DABL_NOMFMA bi2DABL_NOEPI bi2DABL_NOLOADDABL_NOEPI; do
  for cin in 64 256 512; do
    echo -n "$b "; timeout -k 5 60 ./build/$b 128 $cin 256 1 1 2 3 0 4 | grep gen
  done
  echo -n "$b "; timeout -k 5 60 ./build/$b 64 256 256 0 1 2 3 0 4 | grep gen
  echo -n "$b "; timeout -k 5 60 ./build/$b 64 64 256 0 1 2 3 0 4 | grep gen
done
