"""PCIe-inclusive timing of the host-buffer entries (development aid): icikt_pairs_f64 staged (default) and from caller-pinned memory (ICIKT_FLAG_HOST_PINNED), and
icikt_pairs_multi_f64 on the devices given (default: one rank over RCCL, then two ranks on device 0 with copies)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icikendalltau_amd import _lib
from bench import make_matrix

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
n, S, na, seed = {"c3": (10000, 256, 500, 3), "c4": (10000, 1024, 1000, 4)}[cfg]
X = make_matrix(n, S, na, seed)
P = S * (S - 1) // 2
ctx = _lib.Context(0)
ref = None
Xp = _lib.pinned_empty(X.shape, order="F")
Xp[...] = X
for mode in ("staged", "pinned"):
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        out, _c, rsn = ctx.pairs(Xp if mode == "pinned" else X, perspective="global", want_counts=False,
                                 flags=_lib.FLAG_HOST_PINNED if mode == "pinned" else 0)
        ts.append(time.perf_counter() - t0)
    if ref is None: ref = out
    print(f"{cfg} icikt_pairs_f64 {mode:8s} best {min(ts)*1e3:7.2f} ms  -> {P/min(ts):.3e} pairs/s (PCIe-inclusive)"
          f"  {'same' if np.array_equal(out, ref) else 'DIFF'}", flush=True)
for devices, ex in (([0], "rccl"), ([0, 0], "copy"), ([0, 0, 0, 0], "copy")):
    m = _lib.MultiContext(devices, exchange=ex)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        out, _c, rsn = m.pairs(X, perspective="global", want_counts=False)
        ts.append(time.perf_counter() - t0)
    m.pairs(X, perspective="global", want_counts=False, flags=_lib.FLAG_TIMING)
    print(f"{cfg} icikt_pairs_multi_f64 devices={devices} {ex}: best {min(ts)*1e3:7.2f} ms -> {P/min(ts):.3e} pairs/s; phases (synced) "
          f"{ {k: round(v, 2) for k, v in m.phase_ms().items()} }  {'same' if np.array_equal(out, ref) else 'DIFF'}", flush=True)
    m.close()
