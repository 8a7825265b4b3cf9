"""Does K1 slow down under sustained load?  Back-to-back launches of the same work, per-launch time (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icikendalltau_amd import _lib
from bench import make_matrix
n, S, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
plan = sys.argv[4] if len(sys.argv) > 4 else ""
X = make_matrix(n, S, max(1, n // 50), 5)
ctx = _lib.Context(0)
ctx.debug_set_plan(plan)
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
ctx.prepare_dev(dX.data_ptr(), n, S, n, 0); ctx.sync()
P = min(P, S * (S - 1) // 2)
ctx.set_pairs_combn(S, 0, P)
out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
ts = []
for i in range(int(sys.argv[5]) if len(sys.argv) > 5 else 24):
    ctx.reset_timers(); ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
    ts.append(ctx.kernel_ms(_lib.K_PAIRS)[0])
print(f"n={n} S={S} P={P} plan={plan!r}: per-launch ms:", " ".join(f"{t:.1f}" for t in ts), flush=True)
