"""The CPU oracle against the reference's own known answers, snapshots and documented outputs.

This is what pins the oracle (and therefore every GPU parity claim) to the reference:
tests/testthat/test-kendall-tau.R, tests/testthat/_snaps/kendall-tau.md, README.md, all reproduced
without R through the RNG emulation in oracle/rrng.py (inputs committed under tests/golden/).
"""
import numpy as np
import pytest
from scipy import stats

from oracle import oracle as O
from oracle.rrng import RRandom


def test_rrng_matches_r():
    # set.seed(1234); rnorm(5)  (R >= 3.6 defaults)
    r = RRandom(1234)
    assert np.allclose(r.rnorm(5), [-1.2070657, 0.2774292, 1.0844412, -2.3456977, 0.4291247], atol=5e-8)
    # golden arrays are what the emulation produces
    r = RRandom(1234)
    z = np.load("tests/golden/snapshot_50000.npz")
    assert np.array_equal(r.rnorm(50000), z["x"]) and np.array_equal(r.rnorm(50000), z["y"])


def test_basic_known_answers():
    # test-kendall-tau.R:5-32: ici_kt == cor(method = "kendall") and cor.test(exact = FALSE)
    x = np.arange(1, 11, dtype=float)
    ys = [np.arange(1, 11, dtype=float)]
    y = ys[0].copy(); y[1] = 15; ys.append(y)
    ys.append(np.arange(10, 0, -1, dtype=float))
    y = ys[2].copy(); y[1] = 15; ys.append(y)
    for y in ys:
        out, cnt, rsn = O.ici_kt(x, y)
        ref = stats.kendalltau(x, y, method="asymptotic")
        assert rsn == 0 and out[0] == pytest.approx(ref.statistic, abs=1e-15)
    assert O.ici_kt(x, ys[0])[0][0] == 1.0 and O.ici_kt(x, ys[2])[0][0] == -1.0
    y = ys[3]
    for alt in ("two-sided", "less", "greater"):
        ref = stats.kendalltau(x, y, method="asymptotic", alternative=alt)
        out, _, _ = O.ici_kt(x, y, alternative=alt.replace("-", "."))
        assert out[1] == pytest.approx(ref.pvalue, rel=1e-12)
    # completeness: :24-31
    y = y.copy(); y[1] = np.nan
    assert O.ici_kt(x, y)[0][3] == pytest.approx(0.9)
    x = x.copy(); x[7] = np.nan
    assert O.ici_kt(x, y)[0][3] == pytest.approx(0.8)
    x[1] = np.nan
    assert O.ici_kt(x, y)[0][3] == pytest.approx(1 - 1 / 9)
    assert O.ici_kt(x, y, perspective="global")[0][3] == pytest.approx(0.8)


def test_na_and_error_contract():
    # test-kendall-tau.R:42-59
    rng = np.random.default_rng(0)
    x = np.sort(rng.standard_normal(100))
    out, _, rsn = O.ici_kt(x, np.full(100, np.nan))
    assert rsn == O.REASON_ALL_MISSING and np.all(np.isnan(out))
    assert np.all(out.view(np.uint64) == np.uint64(0x7FF00000000007A2))  # NA_real_
    with pytest.raises(ValueError, match="not the same length"):
        O.ici_kt(x, x[:99])
    assert O.ici_kt(x[1:2], x[1:2])[2] == O.REASON_SHORT
    assert O.ici_kt(x, np.ones(100))[2] == O.REASON_SINGLE_UNIQUE


def test_snapshot_50000(expected):
    z = np.load("tests/golden/snapshot_50000.npz")
    out, cnt, rsn = O.ici_kt(z["x"], z["y"], "global")
    e = expected["snapshot_50000"]
    assert round(out[0], 8) == e["tau"] and round(out[1], 8) == e["pvalue"]
    assert out[2] == 1.0 and out[3] == 1.0


def test_readme_example(expected):
    z = np.load("tests/golden/readme_s1_s4.npz")
    e = expected["readme"]["r_4"]
    out, cnt, rsn = O.ici_kt(z["s3"], z["s4"], "global")
    assert round(out[0], 7) == e["raw"] and round(out[2], 6) == e["taumax"] and out[3] == pytest.approx(e["completeness"])
    assert out[1] == e["pvalue"]
    assert round(out[0] / out[2], 7) == e["cor"]
    assert O.ici_kt(z["s1"], z["s2"], "global")[0][0] == 1.0
    assert 1 - np.isnan(z["s3"]).mean() == pytest.approx(e["diag_completeness"])


def test_ktfast_snapshots(expected):
    # kt_fast(use = "complete.obs" / "pairwise.complete.obs") reduce to ici_kt on NA-free rows
    X = np.load("tests/golden/ktfast_100x4.npz")["X"]
    e = expected["ktfast"]
    keep = np.ones(100, bool)
    keep[e["complete_obs_drop_row"] - 1] = False

    def sig(v, d):
        return float(f"{v:.{d}g}")

    for key, tau in e["complete_tau"].items():
        i, j = map(int, key.split("-"))
        out, _, _ = O.ici_kt(X[keep, i], X[keep, j], "global")
        assert out[0] == pytest.approx(tau, rel=2e-7) and sig(out[1], 7) == pytest.approx(e["complete_p"][key], rel=1e-6)
    out, _, _ = O.ici_kt(X[keep, 0], X[keep, 0], "global")
    assert sig(out[1], 7) == e["self_p_n99"]
    for key, tau in e["pairwise_tau"].items():
        i, j = map(int, key.split("-"))
        out, _, _ = O.ici_kt(X[:, i], X[:, j], "global")
        assert out[0] == pytest.approx(tau, rel=2e-7) and sig(out[1], 7) == pytest.approx(e["pairwise_p"][key], rel=1e-6)
    out, _, _ = O.ici_kt(X[:, 1], X[:, 1], "global")
    assert sig(out[1], 7) == e["self_p_n100"]


def test_completeness_snapshot(expected):
    X = np.load("tests/golden/completeness_50x100.npz")["X"]
    assert np.isnan(X).sum() == 40
    for row in expected["completeness"]["rows_4_6"]:
        i, j = int(row["s1"][1:]) - 1, int(row["s2"][1:]) - 1
        miss = (np.isnan(X[:, i]) | np.isnan(X[:, j])).sum()
        assert miss == row["missingness"] and 1 - miss / 50 == pytest.approx(row["completeness"])
        out, _, _ = O.ici_kt(X[:, i], X[:, j], "global")
        assert out[3] == pytest.approx(row["completeness"])


@pytest.mark.parametrize("perspective", ["global", "local"])
def test_fast_equals_bruteforce(perspective):
    # test-kendall-tau.R:34-40 in spirit: the O(n log n) path against O(n^2) enumeration
    rng = np.random.default_rng(1)
    for trial in range(30):
        n = int(rng.integers(2, 200))
        x = np.round(rng.standard_normal(n) * rng.choice([1, 3, 50]))
        y = np.round(rng.standard_normal(n) * rng.choice([1, 3, 50]))
        x[rng.random(n) < 0.2] = np.nan
        y[rng.random(n) < 0.2] = np.nan
        out, cnt, rsn = O.ici_kt(x, y, perspective, int32_compat=False)
        if rsn not in (0, 4):
            continue
        bf = O.bruteforce(x, y, perspective)
        assert (cnt["dis"], cnt["ntie"], cnt["xtie"], cnt["ytie"]) == (bf["dis"], bf["ntie"], bf["xtie"], bf["ytie"])
        assert bf["con"] + bf["dis"] + bf["xtie"] + bf["ytie"] - bf["ntie"] == cnt["tot"]


def test_agrees_with_scipy_without_quirks():
    # scipy.stats.kendalltau is the algorithm the reference was translated from (README.md:157-160);
    # identical when at most one column has tie groups >= 3 (no Q1) and no group >= 1024 (no Q2)
    rng = np.random.default_rng(2)
    x = rng.standard_normal(500)
    y = x + rng.standard_normal(500)
    y[:60] = np.nan
    out, _, _ = O.ici_kt(x, y, "global")
    yf = np.where(np.isnan(y), np.nanmin(y) - 0.1, y)
    ref = stats.kendalltau(x, yf, method="asymptotic")
    assert out[0] == pytest.approx(ref.statistic, abs=1e-14) and out[1] == pytest.approx(ref.pvalue, rel=1e-10)


def test_quirk_q1_t0_halved():
    # count_rank_tie: t0 = sum t(t-1)(t-2) / 2 (src/kendallc.cpp:113); scipy has no "/2".  With tie
    # groups >= 3 in BOTH columns the p-value therefore differs from scipy while tau does not.
    rng = np.random.default_rng(3)
    x = rng.integers(0, 5, 200).astype(float)
    y = rng.integers(0, 5, 200).astype(float)
    out, cnt, _ = O.ici_kt(x, y, "global")
    ref = stats.kendalltau(x, y, method="asymptotic")
    assert out[0] == pytest.approx(ref.statistic, abs=1e-14)
    assert abs(out[1] - ref.pvalue) > 1e-6
    t = np.bincount(x.astype(int))
    assert cnt["x0"] == int((t * (t - 1) * (t - 2)).sum() // 2)


def test_quirk_q2_int32_wrap_only_above_1024():
    n = 3000
    rng = np.random.default_rng(5)
    x = rng.standard_normal(n)
    y = 0.02 * x + rng.standard_normal(n)
    y[:1100] = np.nan  # tie group of 1100 > 1023: t(t-1)(2t+5) wraps in int32
    a, ca, _ = O.ici_kt(x, y, "global", int32_compat=True)
    b, cb, _ = O.ici_kt(x, y, "global", int32_compat=False)
    assert ca["y1"] != cb["y1"] and cb["y1"] == 1100 * 1099 * 2205
    assert ca["y1"] == ((1100 * 1099 * 2205 + 2**31) % 2**32) - 2**31
    assert a[0] == b[0] and a[1] != b[1]
    # the NA_INTEGER propagation rule of Rcpp sugar cannot fire below n = 70000
    t = np.arange(2, 70000, dtype=np.int64)
    w = lambda v: ((v + 2**31) % 2**32) - 2**31
    p1 = w(t * (t - 1))
    for arr in (p1, w(p1 * (t - 2)), w(p1 * w(2 * t + 5))):
        assert not np.any(arr == -2**31)


def test_local_derivation_identities():
    # what the GPU epilogue relies on: local = global minus the both-missing rows
    rng = np.random.default_rng(4)
    for _ in range(50):
        n = int(rng.integers(5, 300))
        x = np.round(rng.standard_normal(n) * 2)
        y = np.round(rng.standard_normal(n) * 2)
        x[rng.random(n) < 0.3] = np.nan
        y[rng.random(n) < 0.3] = np.nan
        g, cg, rg = O.ici_kt(x, y, "global", int32_compat=False)
        l, cl, rl = O.ici_kt(x, y, "local", int32_compat=False)
        if rg or rl:
            continue
        cb = int((np.isnan(x) & np.isnan(y)).sum())
        assert cl["dis"] == cg["dis"] and cl["n"] == cg["n"] - cb
        assert cl["ntie"] == cg["ntie"] - cb * (cb - 1) // 2
