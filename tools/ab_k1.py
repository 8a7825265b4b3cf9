"""A/B timing of K1 launch plans in ONE process (interleaved rounds), c4 workload by default."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icikendalltau_amd import _lib
from bench import make_matrix

n, S, na, seed = 10000, 1024, 1000, 4
if len(sys.argv) > 1 and sys.argv[1] == "c3": S, na, seed = 256, 500, 3
X = make_matrix(n, S, na, seed)
ctx = _lib.Context(0)
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
P = S * (S - 1) // 2
ctx.set_pairs_combn(S, 0, P)
out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
ctx.prepare_dev(dX.data_ptr(), n, S, n, 0); ctx.sync()
variants = [dict(np=str(np_), wpb=str(w)) for np_ in (1, 2) for w in (2, 4, 8)]
ref = None
res = {i: [] for i in range(len(variants))}
for rnd in range(3):
    for i, v in enumerate(variants):
        ctx.debug_set_plan(v)
        ctx.reset_timers()
        ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
        res[i].append(ctx.kernel_ms(_lib.K_PAIRS)[0])
        o = out.cpu().numpy()
        if ref is None: ref = o.copy()
        assert np.array_equal(o, ref, equal_nan=True), v
for i, v in enumerate(variants):
    print(v, "K1 ms min %.2f med %.2f -> %.3e pairs/s" % (min(res[i]), sorted(res[i])[1], P / (min(res[i]) / 1e3)), flush=True)
