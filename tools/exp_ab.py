"""A/B of two builds of the library on the c4 workload in one call (two processes, interleaved twice)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r)
from icikendalltau_amd import _lib
if sys.argv[1] != "cur": _lib.LIB_PATH = sys.argv[1]; _lib.needs_build = lambda: False
import numpy as np, torch
from bench import make_matrix
n, S, na, seed = 10000, 1024, 1000, 4
X = make_matrix(n, S, na, seed)
ctx = _lib.Context(0)
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
P = S * (S - 1) // 2
ctx.set_pairs_combn(S, 0, P)
out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
ctx.prepare_dev(dX.data_ptr(), n, S, n, 0); ctx.sync()
ts = []
for _ in range(8):
    ctx.reset_timers(); ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
    ts.append(ctx.kernel_ms(_lib.K_PAIRS)[0])
print(sys.argv[1], "K1 ms", " ".join("%%.2f" %% t for t in ts), flush=True)
''' % ROOT
for which in ("cur", os.path.join(ROOT, "tools", "exp_libA.so"), "cur", os.path.join(ROOT, "tools", "exp_libA.so")):
    subprocess.run([sys.executable, "-c", code, which])
