import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icikendalltau_amd import _lib
from bench import make_matrix
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
S, na, seed = 512, max(1, n // 50), 5
X = make_matrix(n, S, na, seed)
ctx = _lib.Context(0)
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
P = 60000
ctx.set_pairs_combn(S, 0, P)
out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
ctx.prepare_dev(dX.data_ptr(), n, S, n, _lib.FLAG_TIMING); ctx.sync()
print("n", n, "K0 ms for", S, "columns:", ctx.kernel_ms(_lib.K_PREPARE)[0])
ref = None
for v in [dict(pend="l", np="1"), dict(pend="l", np="2"), dict(pend="g", np="1"), dict(pend="g", np="2"), dict()]:
    ctx.debug_set_plan(v)
    ts = []
    for _ in range(2):
        ctx.reset_timers(); ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
        ts.append(ctx.kernel_ms(_lib.K_PAIRS)[0])
    o = out.cpu().numpy()
    if ref is None: ref = o.copy()
    print(v or "default plan", "K1 %.1f ms -> %.3e pairs/s" % (min(ts), P / (min(ts) / 1e3)), "same" if np.array_equal(o, ref, equal_nan=True) else "DIFF", flush=True)
