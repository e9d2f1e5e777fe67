#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/ig4_probe2.sh
cp build/lib_ig4c.so ishapediting_amd/libishap_hip.so
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputests4.log 2>&1; rc=$?
tail -4 gpurun_out/r4_gputests4.log
[ $rc -ne 0 ] && exit $rc
AB_GREP=igemm bash tools/ab_libs.sh build/lib_base.so build/lib_ig4.so build/lib_ig4c.so > gpurun_out/r4_ab_ig4c.txt 2>&1
tail -34 gpurun_out/r4_ab_ig4c.txt
cp build/lib_ig4c.so ishapediting_amd/libishap_hip.so
