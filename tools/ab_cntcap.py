"""c4 K1 time against the entries of the pairs' counter tables (count mode): the table only takes LDS on continuous data,
so the times show what the footprint costs (development aid).  argv: plans, e.g. tgmax=128 tgmax=512 ''"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icikendalltau_amd import _lib
from bench import make_matrix
n, S = int(os.environ.get("N_FEAT", "10000")), int(os.environ.get("N_SAMP", "1024"))
X = make_matrix(n, S, n // 10, 4)
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
P = S * (S - 1) // 2
ctx = _lib.Context(0)
ctx.set_pairs_combn(S, 0, P)
out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
plans = sys.argv[1:] or ["", "tgmax=128", "tgmax=256", "tgmax=512", "tgmax=640", "tgmax=768", ""]
for rnd in range(2):
    for plan in plans:
        ctx.debug_set_plan(plan + (",verbose=1" if rnd == 0 and plan else ("verbose=1" if rnd == 0 else "")) if True else plan)
        ts = []
        for _ in range(4):
            ctx.reset_timers()
            ctx.prepare_dev(dX.data_ptr(), n, S, n, _lib.FLAG_TIMING)
            ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
            ts.append(ctx.kernel_ms(_lib.K_PAIRS)[0])
        print(f"plan '{plan}': K1 " + " ".join("%.3f" % t for t in ts), flush=True)
