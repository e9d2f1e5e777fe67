"""Does ISHAP_G1_SLICES change the full-size input gradient?  (debug helper for tests/test_gpu_fullsize.py: with the worker's
1e-2 cotangent and with a unit one)"""
import os, sys, tempfile, numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_fullsize as T
from pathlib import Path
for scale in ("1e-2", "1.0"):
    T._FUSE_WORKER = T._FUSE_WORKER.replace("* 1e-2)", "* %s)" % scale).replace("* 1.0)", "* %s)" % scale)
    tmp = Path(tempfile.mkdtemp())
    ref = T._run_fullsize_worker(tmp, "default", {})
    got = T._run_fullsize_worker(tmp, "g1off", {"ISHAP_G1_SLICES": "0"})
    print("cotangent scale", scale, "max |gx|", float(np.abs(ref["gx"]).max()),
          {k: float(np.abs(got[k].astype(np.float64) - ref[k].astype(np.float64)).max()) for k in ("out", "tap", "gx")})
