"""One prepare + K1 launches on a tied matrix (LEVELS distinct values per column) for rocprofv3 counter passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icikendalltau_amd import _lib
n, S = 10000, int(os.environ.get("N_SAMP", "256"))
levels = int(os.environ.get("LEVELS", "1000"))
rng = np.random.default_rng(3)
base = rng.standard_normal((n, S))
X = base.copy() if levels == 0 else np.round(base * (levels / 6.0))
X[rng.random(X.shape) < 0.05] = np.nan
ctx = _lib.Context(0)
ctx.debug_set_plan(os.environ.get("PLAN", ""))
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
P = S * (S - 1) // 2
ctx.set_pairs_combn(S, 0, P)
out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
ctx.prepare_dev(dX.data_ptr(), n, S, n, 0); ctx.sync()
for _ in range(int(os.environ.get("REPS", "2"))):
    ctx.run_dev(1, 0, False, 0, out.data_ptr()); ctx.sync()
