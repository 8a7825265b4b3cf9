"""A/B inside one process: the pipelined host entry on c4 with the pre-pass of the later chunks in its small shape (the
library's choice) against the large shape everywhere (plan k0=0); staged and caller-pinned memory; alternating rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icikendalltau_amd import _lib
from bench import make_matrix
X = make_matrix(10000, 1024, 1000, 4)
ctx = _lib.Context(0)
P = 1024 * 1023 // 2
out = np.empty((P, 4)); rsn = np.zeros(P, np.int32)
Xp = _lib.pinned_empty(X.shape, order="F"); Xp[...] = X
outp = _lib.pinned_empty((P, 4)); rsnp = _lib.pinned_empty(P, dtype=np.int32)
L = _lib.lib()
def call(pin):
    t0 = time.perf_counter()
    rc = L.icikt_pairs_f64(ctx._h, (Xp if pin else X).ctypes.data, 10000, 1024, 10000, None, None, 0, 1, 0, 0, 8 if pin else 0,
                           (outp if pin else out).ctypes.data, None, (rsnp if pin else rsn).ctypes.data)
    assert rc == 0
    return (time.perf_counter() - t0) * 1e3
for _ in range(3): call(0); call(1)
res = {}
for rnd in range(4):
    for plan in ("", "k0=0"):
        ctx.debug_set_plan(plan)
        for pin in (0, 1):
            ts = [call(pin) for _ in range(8)]
            res.setdefault((plan or "default", pin), []).extend(ts)
for k, v in sorted(res.items()):
    v = sorted(v)
    print(f"plan {k[0]:8s} {'pinned' if k[1] else 'staged'}: median {v[len(v)//2]:.2f} ms  mean {sum(v)/len(v):.2f}  min {v[0]:.2f}  max {v[-1]:.2f}  (n={len(v)})", flush=True)
