"""Per-call times of icikt_pairs_f64 on the c4 matrix with the pipelined host path on / off and the caller's memory staged
(default) or in memory the caller page-locked (ICIKT_FLAG_HOST_PINNED: "pin" below); verbose=1 prints the stream timeline of one call."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icikendalltau_amd import _lib
from bench import make_matrix
X = make_matrix(10000, 1024, 1000, 4)
ctx = _lib.Context(0)
P = 1024*1023//2
out = np.empty((P,4)); rsn = np.zeros(P, np.int32)
Xp = _lib.pinned_empty(X.shape, order="F"); Xp[...] = X
outp = _lib.pinned_empty((P,4)); rsnp = _lib.pinned_empty(P, dtype=np.int32)
L = _lib.lib()
for plan, pin in (("pipe=1", 0), ("pipe=1,verbose=1", 0), ("pipe=1", 1), ("pipe=1,verbose=1", 1), ("pipe=0", 0), ("pipe=1", 0), ("pipe=1", 1)):
    ctx.debug_set_plan(plan)
    ts=[]
    for _ in range(4 if "verbose" not in plan else 1):
        t0=time.perf_counter()
        rc = L.icikt_pairs_f64(ctx._h, (Xp if pin else X).ctypes.data, 10000, 1024, 10000, None, None, 0, 1, 0, 0, 8 if pin else 0,
                               (outp if pin else out).ctypes.data, None, (rsnp if pin else rsn).ctypes.data)
        ts.append(time.perf_counter()-t0)
    print(plan, "pin" if pin else "staged", " ".join("%.2f" % (t*1e3) for t in ts), flush=True)
