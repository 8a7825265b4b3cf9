"""Config c2: the yeast matrix (6 887 x 96, zeros -> missing), all 4 560 pairs, device-resident timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icikendalltau_amd import _lib

z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "yeast_missing.npz"))
X = np.asfortranarray(z[z.files[0]].astype(np.float64))
X[X == 0] = np.nan
if len(sys.argv) > 2:
    X = np.asfortranarray(X[:, :int(sys.argv[2])])   # fewer columns: fewer pairs
n, S = X.shape
nd = [len(np.unique(X[~np.isnan(X[:, c]), c])) for c in range(S)]
print(f"yeast {n} x {S}: missing per column {np.isnan(X).sum(0).min()}..{np.isnan(X).sum(0).max()}, distinct values per column {min(nd)}..{max(nd)}")
ctx = _lib.Context(0)
if len(sys.argv) > 1:
    ctx.debug_set_plan(sys.argv[1])   # e.g. "np=2"
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
P = S * (S - 1) // 2
ctx.set_pairs_combn(S, 0, P)
out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
for _ in range(4):
    ctx.reset_timers()
    ctx.prepare_dev(dX.data_ptr(), n, S, n, _lib.FLAG_TIMING)
    ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
    k = [ctx.kernel_ms(i)[0] for i in range(3)]
    print(f"K0 {k[0]:.3f} ms  K1 {k[1]:.3f} ms  K2 {k[2]:.3f} ms -> {P / (sum(k) / 1e3):.3e} pairs/s")
