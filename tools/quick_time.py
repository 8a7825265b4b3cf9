"""Quick device-resident timing of the three kernels (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from icikendalltau_amd import _lib

def make(n, S, na_per_col, seed):
    rng = np.random.default_rng(seed)
    X = np.asfortranarray(rng.standard_normal((n, S)))
    if na_per_col:
        idx = np.argpartition(X, na_per_col, axis=0)[:na_per_col]
        np.put_along_axis(X, idx, np.nan, axis=0)
    return X

def run(n, S, na, seed, reps=3, max_pairs=None):
    X = make(n, S, na, seed)
    ctx = _lib.Context(0)
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()  # (S, n) row-major == column-major n x S
    P = S * (S - 1) // 2
    if max_pairs: P = min(P, max_pairs)
    ctx.set_pairs_combn(S, 0, P)
    out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for r in range(reps):
        ctx.reset_timers()
        t0 = time.perf_counter()
        ctx.prepare_dev(dX.data_ptr(), n, S, n, _lib.FLAG_TIMING)
        ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr())
        ctx.sync()
        dt = time.perf_counter() - t0
        k = [ctx.kernel_ms(i)[0] for i in range(3)]
        print(f"n={n} S={S} P={P} wall={dt*1e3:.2f} ms  K0={k[0]:.2f} K1={k[1]:.2f} K2={k[2]:.3f} ms  -> {P/dt:.3e} pairs/s (K1 only {P/(k[1]/1e3):.3e})", flush=True)
    return out

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "c3"
    if which == "c3": run(10000, 256, 500, 3)
    elif which == "c4": run(10000, 1024, 1000, 4)
    elif which == "c5": run(50000, 2048, 1000, 5, reps=1, max_pairs=int(sys.argv[2]) if len(sys.argv) > 2 else 200000)
    elif which == "yeast":
        z = np.load("tests/golden/yeast_missing.npz"); X = z["X"].copy(); X[X == 0] = np.nan
        ctx = _lib.Context(0)
        for _ in range(3):
            t0 = time.perf_counter(); ctx.pairs(X, perspective="global", want_counts=False); print("yeast host path", time.perf_counter() - t0)
