import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icikendalltau_amd import _lib
from oracle import oracle as O
ctx = _lib.Context(0)
ctx.debug_set_plan("verbose=1")
rng = np.random.default_rng(43)
n = 30000
X = rng.standard_normal((n, 5))
X[rng.random(X.shape) < 0.05] = np.nan
X[:, 1] = np.round(X[:, 1] * 20)
X[:, 4] = np.where(rng.random(n) < 0.5, np.nan, X[:, 4])
pi, pj = np.triu_indices(5, k=1)
for p in ("global", "local"):
    for k in range(len(pi)):
        print("pair", pi[k], pj[k], p, flush=True)
        out, cnt, rsn = ctx.pairs(X, pi[k:k+1].astype(np.int32), pj[k:k+1].astype(np.int32), p)
        ref, rcnt, rr = O.ici_pairs(X, pi[k:k+1], pj[k:k+1], p)
        print("   ", np.array_equal(cnt, rcnt[:, :cnt.shape[1]]), float(np.nanmax(np.abs(out - ref))), flush=True)
    out, cnt, rsn = ctx.pairs(X, perspective=p)
    print("all pairs", p, "ok", flush=True)
