import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icikendalltau_amd import _lib
from bench import make_matrix
n, S = 10000, 1024
X = make_matrix(n, S, 1000, 4)
ctx = _lib.Context(0)
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
ctx.prepare_dev(dX.data_ptr(), n, S, n, 0); ctx.sync()
for _ in range(3):
    ctx.reset_timers()
    ctx.expand_cols_dev(0, S, _lib.FLAG_TIMING); ctx.sync()
    print("expand 1024 cols: %.3f ms" % ctx.kernel_ms(_lib.K_PREPARE)[0])
