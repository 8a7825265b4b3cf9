"""Wide columns (65 535 < n <= 262 144): K0 / K1 time of the plain 32-bit path (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icikendalltau_amd import _lib
ctx = _lib.Context(0)
for n, S in ((70000, 64), (100000, 64), (262144, 32)):
    rng = np.random.default_rng(n)
    base = rng.standard_normal(n)
    X = base[:, None] + 0.5 * rng.standard_normal((n, S))
    X[rng.random((n, S)) < 0.02] = np.nan
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    P = S * (S - 1) // 2
    ctx.set_pairs_combn(S, 0, P)
    out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
    for _ in range(2):
        ctx.reset_timers()
        ctx.prepare_dev(dX.data_ptr(), n, S, n, _lib.FLAG_TIMING)
        ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
        k = [ctx.kernel_ms(i)[0] for i in range(3)]
    print(f"n={n} S={S} P={P}: K0 {k[0]:.2f} ms  K1 {k[1]:.2f} ms  K2 {k[2]:.3f} ms -> {P / (k[1] / 1e3):.3e} pairs/s (K1)", flush=True)
