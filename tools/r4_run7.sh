#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
cp build/lib_fuse.so ishapediting_amd/libishap_hip.so
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputests7.log 2>&1; rc=$?
tail -4 gpurun_out/r4_gputests7.log
[ $rc -ne 0 ] && exit $rc
AB_GREP="gn_apply|igemm4" AB_LINES=14 bash tools/ab_libs.sh build/lib_gnact.so build/lib_fuse.so > gpurun_out/r4_ab_fuse.txt 2>&1
tail -40 gpurun_out/r4_ab_fuse.txt
cp build/lib_fuse.so ishapediting_amd/libishap_hip.so
