#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > gpurun_out/r4_gputests9.log 2>&1; rc=$?
tail -4 gpurun_out/r4_gputests9.log; grep -E "^igemm2:|^oneteam:|^small3:|^skinny:|^ISHAP" gpurun_out/r4_gputests9.log
exit $rc
