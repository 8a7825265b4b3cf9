"""K1 time vs number of missing values per column (how much the open-group steps cost)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icikendalltau_amd import _lib
from bench import make_matrix
n, S = 10000, 512
P = S * (S - 1) // 2
ctx = _lib.Context(0)
out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
for na in (0, 64, 1000, 3000):
    X = make_matrix(n, S, na, 4)
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    ctx.set_pairs_combn(S, 0, P)
    ctx.prepare_dev(dX.data_ptr(), n, S, n, 0); ctx.sync()
    ts = []
    for _ in range(3):
        ctx.reset_timers(); ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
        ts.append(ctx.kernel_ms(_lib.K_PAIRS)[0])
    print(f"missing per column {na:5d}: K1 {min(ts):.2f} ms -> {P / (min(ts) / 1e3):.3e} pairs/s", flush=True)
