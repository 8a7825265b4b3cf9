"""A/B of two builds of the library on the c4 pair kernel in one call: ICIKT_LIB=<other .so> selects the second
build (processes alternate twice)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np, torch
from icikendalltau_amd import _lib
from bench import make_matrix
X = make_matrix(10000, 1024, 1000, 4)
ctx = _lib.Context(0)
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
P = 1024 * 1023 // 2
ctx.set_pairs_combn(1024, 0, P)
out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
ctx.prepare_dev(dX.data_ptr(), 10000, 1024, 10000, 0); ctx.sync()
ts = []
for _ in range(10):
    ctx.reset_timers(); ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
    ts.append(ctx.kernel_ms(_lib.K_PAIRS)[0])
print(os.environ.get("ICIKT_LIB", "current"), "K1 ms", " ".join("%%.2f" %% t for t in ts), flush=True)
''' % ROOT
other = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "exp_libA.so")
for which in (None, other, None, other):
    env = dict(os.environ)
    env.pop("ICIKT_LIB", None)
    if which:
        env["ICIKT_LIB"] = which
    subprocess.run([sys.executable, "-c", code], env=env)
