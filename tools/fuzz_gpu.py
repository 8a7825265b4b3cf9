"""Randomised parity sweep of the HIP path against the CPU oracle (development aid, GPU box only).

Each case draws a column length, a column count, a per-column value model (continuous, rounded to a few
levels, blocks of equal values, constants), a missingness pattern (random, left-censored, whole column) and
a K1 launch plan override (pairs per wave, pend placement, half-wave step, joint-tie mode), then compares
counts bit-exactly and the four doubles within 1e-10 for both perspectives.

    python tools/fuzz_gpu.py [cases] [seed] [big|mid|ext|r4|fam]

`mid`: 16 000 .. 31 000 rows (both kernel families of 18 337 .. 30 656 rows, the windows of the tie program) with
two more value models: many small tie groups (n / 2 .. n / 40 distinct values) and a tied region beside a
continuous one.  `ext`: the default lengths with those two models.  `fam`: 15 200 .. 31 000 rows x 48 columns (a task list
that fills the chip: the library chooses the kernel family from the columns' tie structure), half of the matrices continuous.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from icikendalltau_amd import _lib
from oracle import oracle as O

ATOL = 1e-10
BIG = False  # set by main(): argv[3] == 'big'
MID = False  # set by main(): argv[3] == 'mid'
EXT = False  # set by main(): argv[3] == 'ext' (the default lengths with the two extra value models)
FAM = False  # set by main(): argv[3] == 'fam'
R4 = False   # set by main(): argv[3] == 'r4': `ext` + values that collide in the pre-pass's one-word sort (its second pass)
             # + the pre-pass shape (plan key k0) drawn at random


def make_column(rng, n):
    kind = rng.integers(0, 12 if R4 else 10 if (MID or EXT) else 8)
    if kind >= 10:  # values that share the top 48 bits of their sortable keys: distinct low bits, ties among them, either sign
        base = float(rng.choice([1.0, -3.0, 1e-300, 7.5e200]))
        k = rng.integers(0, int(rng.choice([3, 50, 60000])), n)
        x = base * (1.0 + k * 2.0 ** -52) if kind == 10 else np.where(rng.random(n) < 0.5, rng.standard_normal(n), base * (1.0 + k * 2.0 ** -50))
    elif kind == 8:  # many small tie groups
        x = np.round(rng.standard_normal(n) * n / (2.5 * float(rng.choice([2, 4, 8, 24, 40]))))
    elif kind == 9:  # half of the column tied in small groups, half continuous
        x = rng.standard_normal(n)
        lowhalf = x < np.median(x)
        x[lowhalf] = np.round(x[lowhalf] * float(rng.choice([50, 500, 3000])))
    elif kind == 0:
        x = rng.standard_normal(n)
    elif kind == 1:
        x = np.round(rng.standard_normal(n) * rng.choice([0.5, 2, 10, 100]))
    elif kind == 2:  # long runs of equal values
        k = int(rng.integers(1, 6))
        x = rng.integers(0, k + 1, n).astype(np.float64)
    elif kind == 3:  # a few big blocks plus continuous rest
        x = rng.standard_normal(n)
        for _ in range(int(rng.integers(1, 4))):
            m = int(rng.integers(1, max(2, n // 2)))
            x[rng.choice(n, m, replace=False)] = rng.standard_normal()
    elif kind == 4:
        x = np.full(n, rng.standard_normal())
        if n > 1 and rng.random() < 0.7:
            x[rng.choice(n, int(rng.integers(1, n)), replace=False)] = rng.standard_normal()
    elif kind == 5:  # lognormal, as intensities are
        x = np.exp(rng.standard_normal(n) * 2)
    elif kind == 6:  # integers with many ties, including +-0 and infinities
        x = rng.integers(-3, 4, n).astype(np.float64)
        x[x == 0] = rng.choice([0.0, -0.0], int((x == 0).sum()))
        if n > 4 and rng.random() < 0.3:
            x[rng.choice(n, 2, replace=False)] = [np.inf, -np.inf]
    else:  # value exactly 0.1 above the minimum region so that min - 0.1 can collide
        x = np.round(rng.random(n) * 20) / 10.0
    na = rng.integers(0, 6)
    if na == 1:
        x[rng.random(n) < rng.choice([0.01, 0.1, 0.5, 0.9])] = np.nan
    elif na == 2 and n > 1:  # left censored
        k = int(rng.integers(1, n))
        x[np.argsort(x, kind="stable")[:k]] = np.nan
    elif na == 3 and rng.random() < 0.2:
        x[:] = np.nan
    return x


def one_case(ctx, rng, case):
    r = rng.random()
    if FAM:
        n = int(rng.integers(15200, 18337)) if rng.random() < 0.7 else int(rng.integers(18337, 31000))
    elif BIG:  # long columns only: the one-pair-per-wave kernels, pend in LDS and in global memory
        n = int(rng.integers(10000, 65536))
    elif MID:
        n = int(rng.integers(16000, 31000))
    elif r < 0.55:
        n = int(rng.integers(1, 700))
    elif r < 0.9:
        n = int(rng.integers(700, 6000))
    else:
        n = int(rng.integers(6000, 40000))
    S = int(rng.integers(2, 9)) if n < 6000 else int(rng.integers(2, 5))
    if BIG or MID:
        S = int(rng.integers(2, 4))
    if FAM:
        S = 48
        if rng.random() < 0.5:   # continuous columns (a few ties at most), missing values at random
            X = rng.standard_normal((n, S))
            X[rng.random(X.shape) < rng.choice([0.0, 0.02, 0.3])] = np.nan
            if rng.random() < 0.5:
                X[:, int(rng.integers(0, S))] = np.round(X[:, 0] * 3)
            X = np.asfortranarray(X)
        else:
            X = np.asfortranarray(np.stack([make_column(rng, n) for _ in range(S)], axis=1))
    else:
        X = np.asfortranarray(np.stack([make_column(rng, n) for _ in range(S)], axis=1))
    if rng.random() < 0.3 and S > 2:  # correlated columns: shared rows missing, shared ties
        X[:, 1] = np.where(rng.random(n) < 0.7, X[:, 0], X[:, 1])
    env = {  # launch-plan overrides (icikt_debug_set_plan); "" = the library's choice
        "np": rng.choice(["", "1", "2"]),
        "pend": rng.choice(["", "lds", "global"]),
        "half": rng.choice(["", "0", "1"]),
        "tgmax": rng.choice(["", "-1", "2", "1000000"]),
    }
    if FAM:
        env = {"np": "", "pend": "", "half": rng.choice(["", "", "", "0", "1"]), "tgmax": rng.choice(["", "", "-1", "300"])}
    if R4:
        env["k0"] = rng.choice(["", "0", "1"])
        # round 4, second half: the joint-tie modes of long tie groups (list / count / row) and SOLO steps, forced apart
        env["list"] = rng.choice(["", "", "0", "2", "20"])
        env["solo"] = rng.choice(["", "", "0"])
        env["waves"] = rng.choice(["", "", "4", "12"])
        env["split"] = rng.choice(["", "", "1", "2", "4"])   # tasks cut in segments whatever the launch's size
        env["tgmax"] = rng.choice(["", "-1", "2", "40", "300", "1000000"])
    ctx.debug_set_plan(env)
    flags = int(rng.random() < 0.25)  # exact int64 sums
    persp = rng.choice(["global", "local"])
    alt = rng.choice(["two.sided", "less", "greater"])
    cont = bool(rng.random() < 0.3)
    desc = f"case {case}: n={n} S={S} {persp} {alt} cont={cont} flags={flags} plan={ {k: v for k, v in env.items() if v} }"
    try:
        out, cnt, rsn = ctx.pairs(X, None, None, persp, alt, cont, flags)
    except Exception as e:  # an invalid override (e.g. LDS pend for a long column) is a refusal, not a failure
        if "ICIKT" in str(e) or "icikt" in str(e):
            return "refused", desc + f" -> {e}"
        raise
    pi, pj = np.triu_indices(S, k=1)
    ref, rcnt, rrsn = O.ici_pairs(X, pi, pj, persp, alt, cont, int32_compat=not (flags & 1))
    ok = rrsn == 0
    bad = None
    if not np.array_equal(rsn, rrsn):
        bad = f"reasons {rsn} vs {rrsn}"
    elif not np.array_equal(cnt[ok], rcnt[ok][:, :cnt.shape[1]]):
        w = np.argwhere(cnt[ok] != rcnt[ok][:, :cnt.shape[1]])[0]
        bad = f"counts differ at pair {w[0]} field {w[1]}: {cnt[ok][w[0]]} vs {rcnt[ok][w[0]]}"
    elif not np.array_equal(np.isnan(out), np.isnan(ref)):
        bad = "NaN pattern"
    elif np.any(~np.isnan(ref)) and np.nanmax(np.abs(out - ref)) > ATOL:
        bad = f"max abs diff {np.nanmax(np.abs(out - ref))}"
    if bad:
        np.save(f"gpurun_out/fuzz_fail_{case}.npy", X)
        return "FAIL", desc + " -> " + bad
    return "ok", desc


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    global BIG, MID, EXT, R4, FAM
    FAM = len(sys.argv) > 3 and sys.argv[3] == "fam"
    R4 = len(sys.argv) > 3 and sys.argv[3] == "r4"
    BIG = len(sys.argv) > 3 and sys.argv[3] == "big"
    MID = len(sys.argv) > 3 and sys.argv[3] == "mid"
    EXT = len(sys.argv) > 3 and sys.argv[3] == "ext"
    rng = np.random.default_rng(seed)
    ctx = _lib.Context(0)
    os.makedirs("gpurun_out", exist_ok=True)
    tally = {"ok": 0, "FAIL": 0, "refused": 0}
    t0 = time.time()
    for c in range(cases):
        status, desc = one_case(ctx, rng, c)
        tally[status] += 1
        if status != "ok":
            print(status, desc, flush=True)
        if c % 25 == 24:
            print(f"[{time.time() - t0:6.1f}s] {c + 1} cases: {tally}", flush=True)
    print("done:", tally, flush=True)
    sys.exit(1 if tally["FAIL"] else 0)


if __name__ == "__main__":
    main()
