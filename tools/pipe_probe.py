"""Per-call times of icikt_pairs_f64 on the c4 matrix with the pipelined host path on / off and the caller's memory staged
(default) or page-locked for the call (h2d=register); verbose=1 prints the stream timeline of one call."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icikendalltau_amd import _lib
from bench import make_matrix
X = make_matrix(10000, 1024, 1000, 4)
ctx = _lib.Context(0)
P = 1024*1023//2
out = np.empty((P,4)); rsn = np.zeros(P, np.int32)
L = _lib.lib()
for plan in ("pipe=1", "pipe=1,verbose=1", "pipe=1,h2d=register", "pipe=1,h2d=register,verbose=1", "pipe=0", "pipe=1", "pipe=1,h2d=register"):
    ctx.debug_set_plan(plan)
    ts=[]
    for _ in range(4 if "verbose" not in plan else 1):
        t0=time.perf_counter()
        rc = L.icikt_pairs_f64(ctx._h, X.ctypes.data, 10000, 1024, 10000, None, None, 0, 1, 0, 0, 0, out.ctypes.data, None, rsn.ctypes.data)
        ts.append(time.perf_counter()-t0)
    print(plan, " ".join("%.2f" % (t*1e3) for t in ts), flush=True)
