#!/usr/bin/env python3
"""Dependent memory round trips the compiler put into a kernel: per kernel symbol, the number of vector-memory loads, of
`s_waitcnt vmcnt(0)`, and of those waits that stand within a few instructions BEHIND a load (a load the wave then sits on).

Why (round 6): hipcc closes every conditional block that holds a load -- `if (p) v = p[i];`, `c ? p[i] : 0.f` with a run-time c -- with
an `s_waitcnt vmcnt(0)`, and it hoists loop-invariant arithmetic on freshly loaded operands above a loop, waiting for them there.  In
the latency-bound kernels of the small maps each such wait is a full memory round trip on the critical path of a 7 us launch:
gn_bwd_local_kernel fetched gamma / beta / FiLM / the saved input only after its 16 slice loads had come back (-0.3 ... -1.0 us per
launch, 60 launches per guided step, once fixed), drag_terms_kernel had 41 of its 46 waits directly behind one of its 45 texel loads.
Remedies used in csrc/: unconditional loads at clamped / substituted addresses with the VALUE selected afterwards, or inline-asm loads
the compiler cannot see (norm_local.hip PfVec -- with the wait on every path before the registers can be re-allocated).

    python tools/isa_wait_audit.py [file.hip ...]        (default: every csrc/*.hip; compiles device-only to assembly, no GPU needed)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-kernarg-preload-count=14", "--cuda-device-only", "-S"]
NEAR = 6          # instructions between a load and a vmcnt(0) for the wait to count as "behind the load"


def audit(path, tmp):
    out = os.path.join(tmp, os.path.basename(path) + ".s")
    r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *FLAGS, path, "-o", out], capture_output=True, text=True)
    if r.returncode != 0:
        print(f"{path}: hipcc failed\n{r.stderr[-400:]}")
        return {}
    stats, name, last = {}, None, None
    for i, l in enumerate(open(out)):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, last = m.group(1), None
            stats[name] = [0, 0, 0]
            continue
        if name is None:
            continue
        if "global_load" in l or "buffer_load" in l or "scratch_load" in l:
            stats[name][0] += 1
            last = i
        elif "s_waitcnt" in l and "vmcnt(0)" in l:
            stats[name][1] += 1
            if last is not None and i - last <= NEAR:
                stats[name][2] += 1
        elif "s_endpgm" in l:
            name = None
    return stats


def main():
    files = sys.argv[1:] or sorted(os.path.join(ROOT, "ishapediting_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "ishapediting_amd", "csrc")) if f.endswith(".hip"))
    print(f"{'loads':>6s} {'vmcnt(0)':>9s} {'behind a load':>14s}  kernel")
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            for k, v in audit(f, tmp).items():
                if v[0]:
                    print(f"{v[0]:6d} {v[1]:9d} {v[2]:14d}  {os.path.basename(f)}: {k[:110]}")


if __name__ == "__main__":
    main()
