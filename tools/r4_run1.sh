#!/bin/bash
# round 4, GPU call: the whole GPU suite, then the bench line with the C4 / C2 legs, then the in-situ A/B of the weight prefetch
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputests.log 2>&1; rc=$?
tail -5 gpurun_out/r4_gputests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py --c4-shape-profile gpurun_out/r4_c4_shapes.csv --shape-profile gpurun_out/r4_c3_shapes.csv > gpurun_out/r4_bench_v1.json 2> gpurun_out/r4_bench_v1.err; rc=$?
tail -3 gpurun_out/r4_bench_v1.err; cat gpurun_out/r4_bench_v1.json | cut -c1-600
[ $rc -ne 0 ] && exit $rc
AB_GREP=igemm2 bash tools/ab_libs.sh build/lib_base.so build/lib_pf16s.so > gpurun_out/r4_ab_pf.txt 2>&1
cat gpurun_out/r4_ab_pf.txt | tail -20
cp build/lib_base.so ishapediting_amd/libishap_hip.so
