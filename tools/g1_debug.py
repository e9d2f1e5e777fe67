"""Does ISHAP_G1_SLICES change the full-size input gradient?  (debug helper for tests/test_gpu_fullsize.py)"""
import os, subprocess, sys, tempfile, numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_fullsize as T
from pathlib import Path
tmp = Path(tempfile.mkdtemp())
ref = T._run_fullsize_worker(tmp, "default", {})
for name, env in (("g1off", {"ISHAP_G1_SLICES": "0"}), ("skinnyoff", {"ISHAP_SKINNY": "0"}), ("pendsplit", {"ISHAP_PEND_NOSPLIT": "36", "ISHAP_PEND_MINSTEPS": "6"})):
    got = T._run_fullsize_worker(tmp, name, env)
    print(name, {k: float(np.abs(got[k].astype(np.float64) - ref[k].astype(np.float64)).max()) for k in ("out", "tap", "gx")})
