"""Build libishap_hip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the tree)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libishap_hip.so")
SOURCES = ["igemm.hip", "igemm2.hip", "igemm4.hip", "igemm_skinny.hip", "norm.hip", "norm_bwd.hip", "norm_local.hip", "norm_api.hip", "attention.hip", "misc.hip", "ddpm.hip", "decode.hip", "decode_bwd.hip", "drag.hip", "surface.hip", "unet.hip", "backward.hip", "api.hip"]
# -amdgpu-kernarg-preload-count: the first 14 dwords of a kernel's SCALAR leading parameters arrive in SGPRs at dispatch
# (gfx940+; csrc/common.h IgemmHot) instead of behind an s_load round trip at kernel entry
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-mllvm", "-amdgpu-kernarg-preload-count=14"]


def _newer(src: str, dst: str) -> bool:
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(HERE, "..", "include", "ishap.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _newer(src, obj):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
