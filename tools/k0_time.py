"""K0 (the per-column pre-pass) alone on the c4 and c3 matrices and longer / shorter columns: device-resident timing
(development aid).  argv[1]: a debug plan, e.g. k0=1 (the 256-thread shape)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icikendalltau_amd import _lib
from bench import make_matrix
for (n, S, na, seed) in ((10000, 1024, 1000, 4), (10000, 256, 500, 3), (50000, 512, 1000, 5), (6887, 96, 300, 2), (3000, 1024, 100, 6), (20000, 512, 500, 7)):
    X = make_matrix(n, S, na, seed)
    ctx = _lib.Context(0)
    if len(sys.argv) > 1:
        ctx.debug_set_plan(sys.argv[1])     # e.g. k0=1: the 256-thread shape
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    ts = []
    for _ in range(6):
        ctx.reset_timers(); ctx.prepare_dev(dX.data_ptr(), n, S, n, _lib.FLAG_TIMING); ctx.sync()
        ts.append(ctx.kernel_ms(_lib.K_PREPARE)[0])
    print(n, S, "K0 ms", " ".join("%.3f" % t for t in ts), flush=True)
