"""profiles/pmc_traffic.json from the pmc_summary.py outputs of the FETCH_SIZE and WRITE_SIZE passes (bytes per launch per kernel
symbol, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) and, when given, the pmc_mfma_summary.py output of the
MFMA pass (mfma_util_pct per symbol).  tools/final_profile.sh writes it from the SAME run as the round's pmc_*.txt files.
Usage: pmc_traffic_json.py FETCH.txt WRITE.txt out.json [MFMA.txt]"""
import json
import re
import sys

out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES ... (separate passes, tools/final_profile.sh) over "
                 "tools/pmc_step.py in the plain launch sequence (ISHAP_OVERLAP_TAIL=0); FETCH_SIZE doubled (gfx950 counts 128-B requests of "
                 "wide reads at 64 B, MI355X_MICROARCH.md HBM section); bytes per launch, mean over dispatches", "kernels": {}}
for path in sys.argv[1:3]:
    for line in open(path):
        m = re.match(r"(.+?)\s+(FETCH_SIZE|WRITE_SIZE)\s+dispatches=\s*(\d+)\s+mean=\s*([\d.]+) KiB\s+bytes_corrected=\s*(\d+)", line)
        if m:
            out["kernels"].setdefault(m.group(1).strip(), {})[m.group(2)] = {"dispatches": int(m.group(3)), "bytes_per_launch": int(m.group(5))}
if len(sys.argv) > 4:
    for line in open(sys.argv[4]):
        m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m and m.group(1).strip() in out["kernels"]:
            out["kernels"][m.group(1).strip()]["mfma_util_pct"] = float(m.group(3))
# the HBM-class group of bench.py's `roofline_small_maps` (conv / GEMM launches with M <= 256 rows): the symbols that ONLY such launches
# use at batch 1 -- igemm4's 8- and 16-pixel-wide tiles and the skinny GEMM kernel; the group's tiled 1x1 GEMMs share
# igemm2_kernel<64, 64, 4, false, 1> with the 32^2 / 64^2 maps and cannot be told apart in a counter pass
grp = [k for name, k in out["kernels"].items() if re.match(r"igemm4_kernel<64, 64, (8|16), ", name) or name.startswith("igemm_skinny_kernel")]
grp = [k for k in grp if "FETCH_SIZE" in k and "WRITE_SIZE" in k]
if grp:
    nd = sum(k["FETCH_SIZE"]["dispatches"] for k in grp)
    out["small_maps"] = {"bytes_per_launch": int(sum((k["FETCH_SIZE"]["bytes_per_launch"] + k["WRITE_SIZE"]["bytes_per_launch"]) * k["FETCH_SIZE"]["dispatches"] for k in grp) / nd),
                         "dispatches": nd,
                         "what": "FETCH_SIZE (doubled) + WRITE_SIZE per launch, mean over the dispatches of igemm4_kernel<64, 64, 8 | 16, ...> and igemm_skinny_kernel "
                                 "(the symbols only the 8x8 / 16x16 maps use; their tiled 1x1 GEMMs share a symbol with larger maps and are not included)"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(f"{len(out['kernels'])} kernels -> {sys.argv[3]}")
