#!/bin/bash
# round 4 measurement set: PMC passes first (they feed profiles/pmc_traffic.json, which bench.py's roofline.traffic reads), then the
# bench line, per-shape tables, kernel trace; everything under gpurun_out/final/
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/final; mkdir -p gpurun_out/final
bash tools/pmc_only.sh || exit 1
python3 tools/pmc_traffic_json.py gpurun_out/final/pmc_FETCH_SIZE.txt gpurun_out/final/pmc_WRITE_SIZE.txt profiles/pmc_traffic.json || exit 1
cp profiles/pmc_traffic.json gpurun_out/final/pmc_traffic.json
bash tools/final_profile.sh || exit 1
bash tools/pmc_mfma.sh
ls gpurun_out/final
