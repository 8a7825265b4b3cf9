"""Timing ablations of K1 (results are wrong when a debug bit is set): 1 = no gather, 2 = no in-step
all-pairs, 4 = no prefix rebuild."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icikendalltau_amd import _lib
from bench import make_matrix
n, S, na, seed = 10000, 1024, 1000, 4
X = make_matrix(n, S, na, seed)
ctx = _lib.Context(0)
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
P = S * (S - 1) // 2
ctx.set_pairs_combn(S, 0, P)
out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
ctx.prepare_dev(dX.data_ptr(), n, S, n, 0); ctx.sync()
for np_ in (1, 2):
    for dbg in (0, 1, 2, 4, 3, 5, 6, 7):
        os.environ.update(ICIKT_K1_NP=str(np_), ICIKT_K1_WPB="4", ICIKT_K1_DEBUG=str(dbg))
        ts = []
        for _ in range(3):
            ctx.reset_timers()
            ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
            ts.append(ctx.kernel_ms(_lib.K_PAIRS)[0])
        print(f"NP={np_} dbg={dbg} (nogather={dbg&1} noallpairs={(dbg>>1)&1} norebuild={(dbg>>2)&1}): K1 {min(ts):.2f} ms", flush=True)
