#!/bin/bash
# round 4, GPU call 3: GPU suite with the generic igemm4 (second source, W = 16), harness probe 2, same-box A/B of three builds
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
cp build/lib_ig4b.so ishapediting_amd/libishap_hip.so
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputests3.log 2>&1; rc=$?
tail -4 gpurun_out/r4_gputests3.log
[ $rc -ne 0 ] && exit $rc
bash tools/ig4_probe2.sh
AB_GREP=igemm bash tools/ab_libs.sh build/lib_base.so build/lib_ig4.so build/lib_ig4b.so > gpurun_out/r4_ab_ig4b.txt 2>&1
tail -30 gpurun_out/r4_ab_ig4b.txt
cp build/lib_ig4b.so ishapediting_amd/libishap_hip.so
