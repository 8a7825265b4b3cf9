"""Count-like data (negative binomial counts with per-feature means spread over four decades, zeros -> missing): K0 / K1 / K2
for a shape (development aid): argv n_feat n_samp [plan]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icikendalltau_amd import _lib
n, S = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(21)
mu = np.exp(rng.normal(2.0, 2.0, size=n))[:, None] * np.exp(rng.normal(0.0, 0.3, size=S))[None, :]
r = 2.0
X = rng.negative_binomial(r, r / (r + mu)).astype(np.float64)
X[X == 0] = np.nan
c = X[:, 0]; c = c[~np.isnan(c)]; u, cnt = np.unique(c, return_counts=True)
print(f"{n} x {S}: column 0 has {len(c)} present rows, {len(u)} distinct values, {(cnt >= 2).sum()} tie groups, largest {cnt.max()}, rows in groups > 32: {cnt[cnt > 32].sum()}")
X = np.asfortranarray(X)
ctx = _lib.Context(0)
if len(sys.argv) > 3: ctx.debug_set_plan(sys.argv[3])
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
P = S * (S - 1) // 2
ctx.set_pairs_combn(S, 0, P)
out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
for _ in range(3):
    ctx.reset_timers()
    ctx.prepare_dev(dX.data_ptr(), n, S, n, _lib.FLAG_TIMING)
    ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
    k = [ctx.kernel_ms(i)[0] for i in range(3)]
print(f"   K0 {k[0]:.3f} K1 {k[1]:.3f} K2 {k[2]:.3f} ms")
