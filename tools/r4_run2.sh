#!/bin/bash
# round 4, GPU call 2: GPU suite with igemm4 in the library, the igemm4 harness probe, same-box A/B (igemm2-only vs igemm4), parity report
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
cp build/lib_ig4.so ishapediting_amd/libishap_hip.so
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputests2.log 2>&1; rc=$?
tail -4 gpurun_out/r4_gputests2.log
[ $rc -ne 0 ] && exit $rc
bash tools/ig4_probe.sh
AB_GREP=igemm bash tools/ab_libs.sh build/lib_base.so build/lib_ig4.so > gpurun_out/r4_ab_ig4.txt 2>&1
tail -24 gpurun_out/r4_ab_ig4.txt
cp build/lib_ig4.so ishapediting_amd/libishap_hip.so
timeout -k 10 900 python tools/parity_report.py --T 200 --W 40 --res 256 --out gpurun_out/r4_parity_c3_full.json > gpurun_out/r4_parity.log 2> gpurun_out/r4_parity.err; tail -3 gpurun_out/r4_parity.err; cut -c1-800 gpurun_out/r4_parity.log
