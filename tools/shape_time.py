"""K0 / K1 / K2 and the host-buffer call for a data-set SHAPE (development aid): argv n_feat n_samp [levels] [plan]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icikendalltau_amd import _lib
from bench import make_matrix
n, S = int(sys.argv[1]), int(sys.argv[2])
levels = int(sys.argv[3]) if len(sys.argv) > 3 else 0
X = make_matrix(n, S, n // 10, 11, levels)
ctx = _lib.Context(0)
if len(sys.argv) > 4: ctx.debug_set_plan(sys.argv[4])
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
P = S * (S - 1) // 2
ctx.set_pairs_combn(S, 0, P)
out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
for _ in range(4):
    ctx.reset_timers()
    t0 = time.perf_counter()
    ctx.prepare_dev(dX.data_ptr(), n, S, n, _lib.FLAG_TIMING)
    ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
    dt = time.perf_counter() - t0
    k = [ctx.kernel_ms(i)[0] for i in range(3)]
print(f"{n} x {S} levels={levels}: K0 {k[0]:.3f} K1 {k[1]:.3f} K2 {k[2]:.3f} ms, resident wall {dt * 1e3:.3f} ms")
t = []
for _ in range(5):
    t0 = time.perf_counter(); r = ctx.pairs(X, perspective="global", want_counts=False); t.append(time.perf_counter() - t0)
print(f"   host-buffer pairs call: {min(t) * 1e3:.3f} ms")
