#!/bin/bash
# group-local GroupNorm: workgroups ("parts") per (image, group) -- total kernel time of the local GN kernels per setting
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for cfg in "8 256 256" "16 256 512" "16 512 512" "16 128 512" "8 128 256"; do
  set -- $cfg
  export ISHAP_GN_PARTS=$1 ISHAP_GN_PART_ELEMS=$2 ISHAP_GN_MAX_WGS=$3
  rm -rf /tmp/gp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/gp.json 2>/dev/null || exit 1
  python3 - "$1 $2 $3" $(find /tmp/gp -name "*kernel_stats.csv") <<'PY'
import csv, sys
f = b = nf = nb = 0
for r in csv.DictReader(open(sys.argv[2])):
    if "gn_local_kernel" in r["Name"]: f += float(r["TotalDurationNs"]); nf += int(r["Calls"])
    if "gn_bwd_local_kernel" in r["Name"]: b += float(r["TotalDurationNs"]); nb += int(r["Calls"])
print(f"parts<= {sys.argv[1]}: gn_local {f/1e6:.2f} ms / {nf} = {f/nf/1e3:.2f} us, gn_bwd_local {b/1e6:.2f} ms / {nb} = {b/nb/1e3:.2f} us, total {(f+b)/1e6:.2f} ms")
PY
done
