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
variants = [dict(ICIKT_K1_NP="2", ICIKT_K1_WPB="4", ICIKT_K1_DEBUG=d) for d in ("0", "32", "64")] + \
           [dict(ICIKT_K1_NP="4", ICIKT_K1_WPB="2", ICIKT_K1_DEBUG=d) for d in ("0", "32", "64")]
ref = None
for rnd in range(2):
    for v in variants:
        os.environ.update(v)
        ctx.reset_timers(); ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
        o = out.cpu().numpy()
        if ref is None: ref = o.copy()
        print(v, "K1 %.2f ms" % ctx.kernel_ms(_lib.K_PAIRS)[0], "same" if np.array_equal(o, ref, equal_nan=True) else "DIFFERENT", flush=True)
