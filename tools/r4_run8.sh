#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
cp build/lib_final.so ishapediting_amd/libishap_hip.so
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > gpurun_out/r4_gputests8.log 2>&1; rc=$?
tail -4 gpurun_out/r4_gputests8.log; grep -E "^igemm2:|^oneteam:" gpurun_out/r4_gputests8.log
exit $rc
