#!/bin/bash
# igemm4 on the 8x8 maps, in situ: un-profiled s/shape with the route off / on (with and without the folded-skip launches), then the GPU tests
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/env_ab.sh "ISHAP_IG4_W8=0" "ISHAP_IG4_W8_K2=0" "ISHAP_IG4_W8_K2=1" > gpurun_out/w8_ab.txt 2>&1
cat gpurun_out/w8_ab.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/w8_gputests.log 2>&1; rc=$?
tail -4 gpurun_out/w8_gputests.log
exit $rc
