#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp build/lib_gnact.so ishapediting_amd/libishap_hip.so
bash tools/env_ab.sh "X=1" "ISHAP_IG4_TEAM_STEPS=36" "ISHAP_IG4_TEAM_STEPS=24" "ISHAP_IG4_TEAMS=0" "ISHAP_IGEMM4=1" > gpurun_out/r4_env_ab.txt 2>&1
cat gpurun_out/r4_env_ab.txt
AB_GREP=igemm4 bash tools/ab_libs.sh build/lib_gnact.so build/lib_gnact_shal.so build/lib_gnact_big4.so > gpurun_out/r4_ab_rings.txt 2>&1
grep -E "round|==" gpurun_out/r4_ab_rings.txt
cp build/lib_gnact.so ishapediting_amd/libishap_hip.so
