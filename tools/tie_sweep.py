"""K1 time against the number of distinct values per column (development aid): continuous data runs the hot
steps, heavily tied data the general (mixed-group) steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icikendalltau_amd import _lib

n, S = int(os.environ.get("N_FEAT", "10000")), int(os.environ.get("N_SAMP", "256"))
rng = np.random.default_rng(3)
base = rng.standard_normal((n, S))
ctx = _lib.Context(0)
if len(sys.argv) > 1:
    ctx.debug_set_plan(sys.argv[1])   # e.g. tgmax=1000
scale = n / 10000.0   # the same group SIZES at any column length
P = S * (S - 1) // 2
ctx.set_pairs_combn(S, 0, P)
out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
for levels in (0, 5000, 1000, 200, 50, 10, 3):
    X = base.copy() if levels == 0 else np.round(base * (levels * scale / 6.0))
    X[rng.random(X.shape) < 0.05] = np.nan
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    ts = []
    for _ in range(3):
        ctx.reset_timers()
        ctx.prepare_dev(dX.data_ptr(), n, S, n, _lib.FLAG_TIMING)
        ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
        ts.append(ctx.kernel_ms(_lib.K_PAIRS)[0])
    print(f"~{levels or 'continuous'} distinct values: K1 {min(ts):8.2f} ms -> {P / (min(ts) / 1e3):.3e} pairs/s", flush=True)
