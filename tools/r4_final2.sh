#!/bin/bash
# final pass of round 4: full-length C3 parity report with the final library, then the measurement set
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
cp build/lib_final.so ishapediting_amd/libishap_hip.so
timeout -k 10 900 python tools/parity_report.py --T 200 --W 40 --res 256 --out gpurun_out/r4_parity_c3_full.json > gpurun_out/r4_parity.log 2> gpurun_out/r4_parity.err; tail -2 gpurun_out/r4_parity.err; cut -c1-400 gpurun_out/r4_parity.log
bash tools/r4_final1.sh
