"""One prepare + a few K1 launches (default: the c4 workload; N_FEAT / N_SAMP / MAX_PAIRS override) for
rocprofv3 counter passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icikendalltau_amd import _lib
from bench import make_matrix
n = int(os.environ.get("N_FEAT", "10000"))
S = int(os.environ.get("N_SAMP", "1024"))
na, seed = n // 10, 4
X = make_matrix(n, S, na, seed)
ctx = _lib.Context(0)
ctx.debug_set_plan(os.environ.get("PLAN", ""))  # e.g. PLAN="np=1,wpb=4"
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
P = min(S * (S - 1) // 2, int(os.environ.get("MAX_PAIRS", "1000000000")))
ctx.set_pairs_combn(S, 0, P)
out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
ctx.prepare_dev(dX.data_ptr(), n, S, n, 0); ctx.sync()
for _ in range(int(os.environ.get("REPS", "2"))):
    ctx.run_dev(1, 0, False, 0, out.data_ptr()); ctx.sync()
