"""c2 (yeast): K1 time of the pairs that STREAM column j, per j, and of the pairs that GATHER column i, per i -- where a
launch that never fills the chip spends its time (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icikendalltau_amd import _lib
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "yeast_missing.npz"))
X = np.asfortranarray(z[z.files[0]].astype(np.float64)); X[X == 0] = np.nan
n, S = X.shape
ctx = _lib.Context(0)
if len(sys.argv) > 1: ctx.debug_set_plan(sys.argv[1])
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
ctx.prepare_dev(dX.data_ptr(), n, S, n, 0); ctx.sync()
pi_all, pj_all = np.triu_indices(S, k=1)
out = torch.empty((len(pi_all), 4), dtype=torch.float64, device="cuda")
def t_of(pi, pj):
    ctx.set_pairs(pi.astype(np.int32), pj.astype(np.int32))
    ts = []
    for _ in range(3):
        ctx.reset_timers(); ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
        ts.append(ctx.kernel_ms(_lib.K_PAIRS)[0])
    return min(ts)
print("all pairs: %.3f ms" % t_of(pi_all, pj_all))
res = []
for c in range(S):
    m = pj_all == c
    if m.sum() == 0: continue
    v = X[:, c]; v = v[~np.isnan(v)]
    u, cnt = np.unique(v, return_counts=True)
    res.append((t_of(pi_all[m], pj_all[m]), c, int(m.sum()), len(v), len(u), int(cnt.max()), int((cnt > 32).sum()), int(cnt[cnt > 32].sum())))
res.sort(reverse=True)
print("streamed column j: ms, j, pairs, rows, distinct, largest group, groups > 32 rows, rows in them")
for r in res[:12] + res[-4:]: print("  %.3f  j=%d pairs=%d rows=%d distinct=%d maxgroup=%d big=%d bigrows=%d" % r)
