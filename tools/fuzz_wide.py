"""Randomised parity sweep of the wide-column path (65 535 < n <= 262 144) against the oracle's exact mode
(development aid, GPU box).  Columns are kept positively correlated so that `dis` stays far below 2^31, where the
reference's (and the oracle's) int accumulator stops being meaningful."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icikendalltau_amd import _lib
from oracle import oracle as O

def make(rng, n, S):
    base = rng.standard_normal(n)
    X = base[:, None] + rng.choice([0.05, 0.1, 0.2]) * rng.standard_normal((n, S))
    for c in range(S):
        kind = rng.integers(0, 6)
        if kind == 1:
            X[:, c] = np.round(X[:, c] * rng.choice([3, 40, 64, 1000]))      # groups from thousands of rows to a few
        elif kind == 2:   # exact group sizes around the 64-row window
            k = int(rng.choice([63, 64, 65, 128, 129]))
            order = np.argsort(X[:, c], kind="stable")
            vals = np.repeat(np.arange(n // k + 1), k)[:n].astype(np.float64)
            X[order, c] = vals
        elif kind == 3:   # left-censored: the missing rows are the last, long group
            m = int(rng.integers(1, n // 4))
            X[np.argsort(X[:, c])[:m], c] = np.nan
        elif kind == 4:
            X[rng.random(n) < rng.choice([0.001, 0.05]), c] = np.nan
    return np.asfortranarray(X)

def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = _lib.Context(0)
    bad = 0
    for case in range(cases):
        n = int(rng.choice([65536, 65537, 70000, 99999, 131072, 200000]))
        S = int(rng.integers(2, 5))
        X = make(rng, n, S)
        pi, pj = (a.astype(np.int32) for a in np.triu_indices(S, k=1))
        persp = rng.choice(["global", "local"])
        out, cnt, rsn = ctx.pairs(X, perspective=persp)
        ref, rcnt, rrsn = O.ici_pairs(X, pi, pj, persp, int32_compat=False)
        trust = rcnt[:, 2] < 2**30
        ok = np.array_equal(rsn, rrsn) and np.array_equal(cnt[trust], rcnt[trust][:, :cnt.shape[1]])
        ok = ok and (not trust.any() or float(np.nanmax(np.abs(out[trust] - ref[trust]))) <= 1e-10)
        bad += 0 if ok else 1
        print(f"case {case}: n={n} S={S} {persp} trusted {int(trust.sum())}/{len(trust)} -> {'ok' if ok else 'FAIL'}", flush=True)
    print("done:", "FAIL" if bad else "ok", bad)
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
