#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
cp build/lib_gnact.so ishapediting_amd/libishap_hip.so
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputests5.log 2>&1; rc=$?
tail -4 gpurun_out/r4_gputests5.log
[ $rc -ne 0 ] && exit $rc
AB_GREP=gn_ AB_LINES=24 bash tools/ab_libs.sh build/lib_ig4d.so build/lib_gnact.so > gpurun_out/r4_ab_gnact.txt 2>&1
tail -54 gpurun_out/r4_ab_gnact.txt
cp build/lib_gnact.so ishapediting_amd/libishap_hip.so
