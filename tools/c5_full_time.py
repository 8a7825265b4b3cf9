"""K1 on the full c5 matrix (50 000 x 2 048) for a few plans and pair counts (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icikendalltau_amd import _lib
from bench import make_matrix
n, S = 50000, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
X = make_matrix(n, S, 1000, 5)
ctx = _lib.Context(0)
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
ctx.prepare_dev(dX.data_ptr(), n, S, n, 0); ctx.sync()
Pfull = S * (S - 1) // 2
for P in (20000, 200000, Pfull):
    P = min(P, Pfull)
    ctx.set_pairs_combn(S, 0, P)
    out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
    for plan in ({}, {"gridmult": 2}, {"pend": "l"}):
        ctx.debug_set_plan(dict(plan, verbose=1 if P == Pfull else 0))
        ts = []
        for _ in range(2):
            ctx.reset_timers(); ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
            ts.append(ctx.kernel_ms(_lib.K_PAIRS)[0])
        print(f"S={S} P={P} plan={plan or 'default'}: K1 {min(ts):9.2f} ms -> {P / (min(ts) / 1e3):.3e} pairs/s", flush=True)
