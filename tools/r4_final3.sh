#!/bin/bash
# last pass of round 4: the measurement set of r4_final2.sh plus the two-rank rehearsal on one GPU
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/r4_final2.sh || exit 1
timeout -k 10 300 python bench.py --gpus 2 --rehearse --steps 2 --warmup 1 --no-cpu-baseline --no-c2 --no-c4 --no-concurrent > gpurun_out/final/rehearse.json 2> gpurun_out/final/rehearse.err
echo "rehearse rc=$?"; tail -c 600 gpurun_out/final/rehearse.json
