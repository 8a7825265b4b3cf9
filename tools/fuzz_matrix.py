"""Randomised parity sweep of the FRONT-END entries on the HIP engine against the same front-end on the CPU oracle
(development aid, GPU box only): ici_kendalltau through the one-call matrix entry (icikt_matrix_f64: exclusion rule,
pair kernels, scale_and_reshape on the device) and through the pair-list route (return_matrix = False), kt_fast with
its NA policies, pairwise_completeness.  Shapes, value models, global_na rules, include_only forms, diag_good,
scale_max, perspective, alternative and continuity are drawn at random; doubles within 1e-10, NaN patterns, `keep`,
pair order and `core` equal.

    python tools/fuzz_matrix.py [cases] [seed] [multi2|multi3]

`multi2` / `multi3`: the HIP side is the multi-device engine with two / three ranks on device 0 (icikt_pairs_multi_f64 /
icikt_matrix_multi_f64: column-sharded pre-pass, exchange, pair blocks, gather).
"""
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np

import fuzz_gpu
from icikendalltau_amd import api
from tests.oracle_engine import OracleEngine

ATOL = 1e-10
KEYS = ("cor", "raw", "pvalue", "taumax", "completeness")


def close(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape or not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    fin = ~np.isnan(a)
    same_inf = np.array_equal(np.isinf(a[fin]), np.isinf(b[fin])) and np.array_equal(a[fin][np.isinf(a[fin])], b[fin][np.isinf(b[fin])])
    ok = fin & ~np.isinf(a)
    return same_inf and (not ok.any() or np.max(np.abs(a[ok] - b[ok])) <= ATOL)


def draw_matrix(rng):
    r = rng.random()
    n = int(rng.integers(2, 60)) if r < 0.3 else int(rng.integers(60, 900)) if r < 0.85 else int(rng.integers(900, 9000))
    S = int(rng.integers(2, 20)) if n < 900 else int(rng.integers(2, 8))
    X = np.stack([fuzz_gpu.make_column(rng, n) for _ in range(S)], axis=1)
    if rng.random() < 0.5:
        X[rng.random((n, S)) < rng.choice([0.02, 0.2])] = 0.0
    if rng.random() < 0.3:
        X[rng.random((n, S)) < 0.02] = rng.choice([np.inf, -np.inf])
    return np.asfortranarray(X), [f"s{i}" for i in range(S)]


def draw_kw(rng, names):
    kw = {}
    if rng.random() < 0.5:
        kw["perspective"] = "local"
    if rng.random() < 0.3:
        kw["scale_max"] = False
    if rng.random() < 0.3:
        kw["diag_good"] = False
    g = rng.integers(0, 5)
    if g == 1:
        kw["global_na"] = (0,)
    elif g == 2:
        kw["global_na"] = None
    elif g == 3:
        kw["global_na"] = (float("nan"), 0, -1.0, 2.0)
    elif g == 4:
        kw["global_na"] = (float("nan"), float("inf"))
    if rng.random() < 0.3:
        kw["alternative"] = str(rng.choice(["less", "greater"]))
    if rng.random() < 0.3:
        kw["continuity"] = True
    io = rng.integers(0, 4)
    S = len(names)
    if io == 1:
        kw["include_only"] = [names[i] for i in rng.choice(S, int(rng.integers(1, S + 1)), replace=False)] + (["nope"] if rng.random() < 0.3 else [])
    elif io == 2 and S > 2:
        m = int(rng.integers(1, 2 * S))
        kw["include_only"] = [[names[i] for i in rng.integers(0, S, m)], [names[i] for i in rng.integers(0, S, m)]]
    return kw


def one_case(rng, case, hip, orc):
    X, names = draw_matrix(rng)
    kw = draw_kw(rng, names)
    what = rng.integers(0, 10)
    desc = f"case {case}: n={X.shape[0]} S={X.shape[1]} kind={what} kw={kw}"

    def both(fn):
        res = []
        for eng in (hip, orc):
            try:
                with warnings.catch_warnings(record=True) as w:
                    warnings.simplefilter("always")
                    res.append(("ok", fn(eng), sorted(str(m.message) for m in w if "NA returned" in str(m.message))))
            except ValueError as e:      # e.g. "No comparisons to do."
                res.append(("err", str(e), None))
        return res

    if what < 5:      # the one-call matrix entry
        a, b = both(lambda eng: api.ici_kendalltau(X, colnames=names, engine=eng, **kw))
        if a[0] != b[0] or (a[0] == "err" and a[1] != b[1]):
            return "FAIL", desc + f" -> {a[0]} {a[1] if a[0] == 'err' else ''} vs {b[0]} {b[1] if b[0] == 'err' else ''}"
        if a[0] == "ok":
            for k in KEYS:
                if not close(a[1][k], b[1][k]):
                    x, y = np.asarray(a[1][k], dtype=np.float64), np.asarray(b[1][k], dtype=np.float64)
                    bad = np.argwhere(~((x == y) | (np.isnan(x) & np.isnan(y)) | (np.abs(x - y) <= ATOL)))
                    i, j = bad[0]
                    return "FAIL", desc + (f" -> {k} differs in {len(bad)} cells, first [{i},{j}]: {x[i, j]!r} vs {y[i, j]!r}; "
                                           f"raw {np.asarray(a[1]['raw'])[i, j]!r} vs {np.asarray(b[1]['raw'])[i, j]!r}, max taumax "
                                           f"{np.nanmax(np.asarray(a[1]['taumax']))!r} vs {np.nanmax(np.asarray(b[1]['taumax']))!r}")
            if not np.array_equal(np.asarray(a[1]["keep"]), np.asarray(b[1]["keep"])):
                return "FAIL", desc + " -> keep differs"
            if a[2] != b[2]:
                return "FAIL", desc + f" -> warnings {len(a[2])} vs {len(b[2])}"
    elif what < 7:    # the pair-list route, data-frame result
        a, b = both(lambda eng: api.ici_kendalltau(X, colnames=names, engine=eng, return_matrix=False, **kw))
        if a[0] != b[0]:
            return "FAIL", desc + f" -> {a[0]} vs {b[0]}"
        if a[0] == "ok":
            da, db = a[1]["cor"], b[1]["cor"]
            for k in ("s1", "s2"):
                if list(da[k]) != list(db[k]):
                    return "FAIL", desc + f" -> column {k} differs"
            for k in ("core", "raw", "pvalue", "taumax", "completeness", "cor"):
                if not close(da[k], db[k]):
                    return "FAIL", desc + f" -> column {k} differs"
    elif what < 9:    # kt_fast
        use = str(rng.choice(["everything", "complete.obs", "pairwise.complete.obs", "all.obs"]))
        Xn = X.copy()
        if rng.random() < 0.4:
            Xn = np.nan_to_num(Xn, nan=1.5)
        desc += f" use={use}"
        a, b = both(lambda eng: api.kt_fast(Xn, use=use, colnames=names, engine=eng))
        if a[0] != b[0]:
            return "FAIL", desc + f" -> {a[0]} vs {b[0]}"
        if a[0] == "ok":
            for k in ("tau", "pvalue"):
                if not close(a[1][k], b[1][k]):
                    return "FAIL", desc + f" -> {k} differs"
    else:
        kwc = {k: v for k, v in kw.items() if k in ("global_na", "include_only")}
        a, b = both(lambda eng: api.pairwise_completeness(X, colnames=names, engine=eng, **kwc))
        if a[0] != b[0]:
            return "FAIL", desc + f" -> {a[0]} vs {b[0]}"
        if a[0] == "ok" and not close(a[1], b[1]):
            return "FAIL", desc + " -> completeness differs"
    return "ok", desc


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    mode = sys.argv[3] if len(sys.argv) > 3 else ""
    hip = api.MultiHipEngine(devices=[0] * int(mode[5:])) if mode.startswith("multi") else api.HipEngine()
    orc = OracleEngine()
    os.makedirs("gpurun_out", exist_ok=True)
    tally = {"ok": 0, "FAIL": 0}
    t0 = time.time()
    for c in range(cases):
        state = rng.bit_generator.state
        status, desc = one_case(rng, c, hip, orc)
        tally[status] += 1
        if status != "ok":
            print(status, desc, flush=True)
            np.save(f"gpurun_out/fuzz_matrix_fail_{c}.npy", np.array([repr(state)], dtype=object), allow_pickle=True)
        if c % 100 == 99:
            print(f"[{time.time() - t0:6.1f}s] {c + 1} cases: {tally}", flush=True)
    print("done:", tally, flush=True)
    sys.exit(1 if tally["FAIL"] else 0)


if __name__ == "__main__":
    main()
