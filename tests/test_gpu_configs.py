"""The BASELINE.json configurations themselves under -m gpu (generators = bench.make_matrix):
full-size HIP runs through the C ABI, the oracle on a seeded sample of pairs with counts bit-exact and doubles
within 1e-10, and size-independent properties on every pair.  The property mirrored at size is the reference's
"matrix path equals single-pair path" (tests/testthat/test-kendall-tau.R:61-70)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL = 1e-10


def _oracle():
    from oracle import oracle as O
    return O


def _sample_check(X, pi, pj, out, cnt, rsn, perspective, n_sample, seed):
    O = _oracle()
    rng = np.random.default_rng(seed)
    sel = rng.choice(len(pi), size=min(n_sample, len(pi)), replace=False)
    ref, rcnt, rrsn = O.ici_pairs(X, pi[sel], pj[sel], perspective)
    assert np.array_equal(rsn[sel], rrsn)
    assert np.array_equal(cnt[sel], rcnt[:, :cnt.shape[1]]), "integer counts differ from the oracle"
    assert np.array_equal(np.isnan(out[sel]), np.isnan(ref))
    d = float(np.nanmax(np.abs(out[sel] - ref)))
    assert d <= ATOL, d
    return sel


def _properties(n, n_na, out, cnt, rsn, perspective):
    """Checks that need no oracle: every pair of a BASELINE matrix is computable; the counts obey their identities."""
    assert np.all(rsn == 0)
    assert not np.isnan(out).any()
    tau, p, tmax, comp = out.T
    assert np.all(np.abs(tau) <= 1.0)
    assert np.all((p >= 0) & (p <= 1))
    f = {k: cnt[:, i] for i, k in enumerate(("n", "missing", "dis", "ntie", "xtie", "ytie", "x0", "x1", "y0", "y1", "tot"))}
    if perspective == "global":
        c_both = 2 * n_na - f["missing"]                       # |A u B| = |A| + |B| - |A n B|
        assert np.all(f["n"] == n)
        assert np.all(f["xtie"] == n_na * (n_na - 1) // 2) and np.all(f["ytie"] == f["xtie"])
        assert np.allclose(comp, 1.0 - f["missing"] / n, rtol=0, atol=1e-15)
    else:
        # rows missing in both columns are dropped: n_local = n - c_both, missing_local = missing_global - c_both
        c_both = n - f["n"]
        assert np.all(f["missing"] == 2 * n_na - 2 * c_both)
    assert np.all((c_both >= 0) & (c_both <= n_na))
    g = c_both if perspective == "global" else np.zeros_like(c_both)
    assert np.all(f["ntie"] >= g * (g - 1) // 2)               # the both-missing rows are joint ties
    assert np.all(f["tot"] == f["n"] * (f["n"] - 1) // 2)
    assert np.all((f["dis"] >= 0) & (f["dis"] <= f["tot"]))
    # tau from the counts (kendallc.cpp:300-303) agrees with the reported tau
    den = np.sqrt((f["tot"] - f["xtie"]).astype(np.float64) * (f["tot"] - f["ytie"]).astype(np.float64))
    tau_c = (f["tot"] - f["xtie"] - f["ytie"] + f["ntie"] - 2 * f["dis"]) / den
    assert np.allclose(np.clip(tau_c, -1, 1), tau, rtol=0, atol=1e-12)


def _swap_symmetry(ctx, X, pi, pj, out, cnt, perspective):
    """ici_kt(y, x): tau, tau_max, completeness and the joint counts are symmetric; x and y tie sums trade places."""
    out_s, cnt_s, rsn_s = ctx.pairs(X, pj, pi, perspective)
    assert np.all(rsn_s == 0)
    assert np.allclose(out_s[:, [0, 2, 3]], out[:, [0, 2, 3]], rtol=0, atol=1e-12)
    assert np.allclose(out_s[:, 1], out[:, 1], rtol=0, atol=1e-12)
    names = ("n", "missing", "dis", "ntie", "xtie", "ytie", "x0", "x1", "y0", "y1", "tot")
    ix = [names.index(k) for k in ("n", "missing", "dis", "ntie", "ytie", "xtie", "y0", "y1", "x0", "x1", "tot")]
    assert np.array_equal(cnt_s, cnt[:, ix])


def _single_pair_path(ctx, X, pi, pj, out, perspective, picks):
    for p in picks:
        o, _c, r = ctx.pair(X[:, pi[p]], X[:, pj[p]], perspective)
        assert r == 0 and np.array_equal(o, out[p]), (p, o, out[p])


@pytest.mark.parametrize("perspective", ["global", "local"])
def test_config_c3(hip_ctx, perspective):
    """c3: 10 000 x 256, the 500 smallest values of every column missing (5 %), 32 640 pairs."""
    from bench import make_matrix
    n, S, n_na = 10000, 256, 500
    X = make_matrix(n, S, n_na, seed=3)
    pi, pj = (a.astype(np.int32) for a in np.triu_indices(S, k=1))
    out, cnt, rsn = hip_ctx.pairs(X, perspective=perspective)
    assert out.shape == (32640, 4)
    _properties(n, n_na, out, cnt, rsn, perspective)
    _sample_check(X, pi, pj, out, cnt, rsn, perspective, 2000, seed=33)
    _swap_symmetry(hip_ctx, X, pi, pj, out, cnt, perspective)
    _single_pair_path(hip_ctx, X, pi, pj, out, perspective, [0, 255, 20000, 32639])


def test_config_c4(hip_ctx):
    """c4 (the configuration the metric is quoted on): 10 000 x 1 024, 1 000 smallest per column missing,
    perspective global, 523 776 pairs -- k1_pairs<2, false, 5> with the closed-form last group."""
    from bench import make_matrix
    n, S, n_na = 10000, 1024, 1000
    X = make_matrix(n, S, n_na, seed=4)
    pi, pj = (a.astype(np.int32) for a in np.triu_indices(S, k=1))
    out, cnt, rsn = hip_ctx.pairs(X, perspective="global")
    assert out.shape == (523776, 4)
    _properties(n, n_na, out, cnt, rsn, "global")
    sel = _sample_check(X, pi, pj, out, cnt, rsn, "global", 2500, seed=44)
    _swap_symmetry(hip_ctx, X, pi, pj, out, cnt, "global")
    _single_pair_path(hip_ctx, X, pi, pj, out, "global", [0, 1023, 300000, 523775])
    # the explicit-list entry and the combn entry agree on the sampled pairs (unsorted list: other task grouping)
    out_l, cnt_l, _ = hip_ctx.pairs(X, pi[sel], pj[sel], "global")
    assert np.array_equal(out_l, out[sel]) and np.array_equal(cnt_l, cnt[sel])
    # local from the same matrix on a column block: derived in K2 from the same K1 counts
    sub = np.arange(0, 1024, 16)
    Xs = np.asfortranarray(X[:, sub])
    out_loc, cnt_loc, rsn_loc = hip_ctx.pairs(Xs, perspective="local")
    qi, qj = (a.astype(np.int32) for a in np.triu_indices(len(sub), k=1))
    _properties(n, n_na, out_loc, cnt_loc, rsn_loc, "local")
    _sample_check(Xs, qi, qj, out_loc, cnt_loc, rsn_loc, "local", 600, seed=45)


@pytest.mark.parametrize("n", [8137, 9000, 10000, 10176])
@pytest.mark.parametrize("na_frac", [0.0, 0.1])
def test_half_wave_five_words_per_lane(plan_ctx, n, na_frac):
    """Every column length window of k1_pairs<2, false, 5> (8 137 .. 10 176 rows: five bitset words per lane of a
    half) by construction, with and without the closed-form last tie group (missing values), continuous and tied
    columns; forced to two pairs per wave (the plan would pick one pair per wave for so few pairs), then the
    library's own plan."""
    rng = np.random.default_rng(n)
    S = 10
    X = rng.standard_normal((n, S))
    X[:, 3] = np.round(X[:, 3] * 300)          # ~1 800 distinct values: tie groups inside steps
    X[:, 6] = np.round(X[:, 6] * 3)            # ~20 distinct values: groups longer than a step
    if na_frac:
        k = int(n * na_frac)
        idx = np.argpartition(X, k, axis=0)[:k]
        np.put_along_axis(X, idx, np.nan, axis=0)   # left-censored: the missing rows are the last tie group
        X[rng.random(n) < 0.3, 8] = np.nan          # and one column with scattered missing values on top
    O = _oracle()
    pi, pj = (a.astype(np.int32) for a in np.triu_indices(S, k=1))
    for plan in ({"np": 2}, None):
        plan_ctx.debug_set_plan(plan)
        for perspective in ("global", "local"):
            out, cnt, rsn = plan_ctx.pairs(X, perspective=perspective)
            ref, rcnt, rrsn = O.ici_pairs(X, pi, pj, perspective)
            assert np.array_equal(rsn, rrsn)
            assert np.array_equal(cnt, rcnt[:, :cnt.shape[1]])
            assert float(np.nanmax(np.abs(out - ref))) <= ATOL


@pytest.mark.parametrize("n", [10177, 12224, 12225, 14000, 14272, 14273])
def test_half_wave_six_and_seven_words_per_lane(plan_ctx, n):
    """k1_pairs<2, false, 6> (10 177 .. 12 224 rows) and <2, false, 7> (12 225 .. 14 272 rows), and the first length
    past them (one pair per wave): every window edge, missing values, tie groups inside and longer than a step; two pairs per wave forced, then the
    library's own plan."""
    rng = np.random.default_rng(n)
    S = 8
    X = rng.standard_normal((n, S))
    X[:, 2] = np.round(X[:, 2] * 400)
    X[:, 5] = np.round(X[:, 5] * 3)
    k = n // 12
    idx = np.argpartition(X, k, axis=0)[:k]
    np.put_along_axis(X, idx, np.nan, axis=0)
    X[rng.random(n) < 0.3, 7] = np.nan
    O = _oracle()
    pi, pj = (a.astype(np.int32) for a in np.triu_indices(S, k=1))
    for plan in ({"np": 2}, None):
        plan_ctx.debug_set_plan(plan)
        for perspective in ("global", "local"):
            out, cnt, rsn = plan_ctx.pairs(X, perspective=perspective)
            ref, rcnt, rrsn = O.ici_pairs(X, pi, pj, perspective)
            assert np.array_equal(rsn, rrsn)
            assert np.array_equal(cnt, rcnt[:, :cnt.shape[1]])
            assert float(np.nanmax(np.abs(out - ref))) <= ATOL


@pytest.mark.parametrize("n", [14273, 16320, 16321, 17000, 18336, 18337])
def test_half_wave_nine_words_per_lane(plan_ctx, n):
    """k1_pairs<2, false, 9> (14 273 .. 18 336 rows; eight words per lane run the nine-word kernel), 15-bit fields in
    the packed in-step compares, 32-byte prefix slots; and the first length past it (one pair per wave)."""
    rng = np.random.default_rng(n)
    S = 6
    X = rng.standard_normal((n, S))
    X[:, 2] = np.round(X[:, 2] * 500)
    X[:, 4] = np.round(X[:, 4] * 3)
    k = n // 15
    idx = np.argpartition(X, k, axis=0)[:k]
    np.put_along_axis(X, idx, np.nan, axis=0)
    X[rng.random(n) < 0.25, 5] = np.nan
    O = _oracle()
    pi, pj = (a.astype(np.int32) for a in np.triu_indices(S, k=1))
    for plan in ({"np": 2}, None):
        plan_ctx.debug_set_plan(plan)
        for perspective in ("global", "local"):
            out, cnt, rsn = plan_ctx.pairs(X, perspective=perspective)
            ref, rcnt, rrsn = O.ici_pairs(X, pi, pj, perspective)
            assert np.array_equal(rsn, rrsn)
            assert np.array_equal(cnt, rcnt[:, :cnt.shape[1]])
            assert float(np.nanmax(np.abs(out - ref))) <= ATOL


def test_half_wave_five_words_natural_plan(hip_ctx):
    """10 000 x 128 = 8 128 pairs: enough pairs for the plan to choose two pairs per wave by itself."""
    from bench import make_matrix
    n, S, n_na = 10000, 128, 1000
    X = make_matrix(n, S, n_na, seed=9)
    pi, pj = (a.astype(np.int32) for a in np.triu_indices(S, k=1))
    out, cnt, rsn = hip_ctx.pairs(X, perspective="global")
    _properties(n, n_na, out, cnt, rsn, "global")
    _sample_check(X, pi, pj, out, cnt, rsn, "global", 1500, seed=99)


def test_config_c5_columns(hip_ctx):
    """c5 shape at full column length: 50 000 rows, 1 000 smallest per column missing, include_only = the first
    columns (vector semantics: pairs with s1 OR s2 in the set), both perspectives, pairwise_completeness on the
    same subset.  The full 2 048-column run is bench.py --config c5 (profiles/); here 96 columns keep the oracle
    affordable."""
    from bench import make_matrix
    from icikendalltau_amd import api
    n, S, n_na = 50000, 96, 1000
    X = make_matrix(n, S, n_na, seed=5)
    names = [f"s{i}" for i in range(S)]
    pi, pj, _core = api.setup_comparisons(names, include_only=names[:8], diag_good=True, ncore=1)
    assert len(pi) == 8 * (S - 1) - 8 * 7 // 2
    O = _oracle()
    rng = np.random.default_rng(55)
    sel = rng.choice(len(pi), size=60, replace=False)
    for perspective in ("global", "local"):
        out, cnt, rsn = hip_ctx.pairs(X, pi, pj, perspective)
        _properties(n, n_na, out, cnt, rsn, perspective)
        ref, rcnt, rrsn = O.ici_pairs(X, pi[sel], pj[sel], perspective)
        assert np.array_equal(cnt[sel], rcnt[:, :cnt.shape[1]])
        assert float(np.nanmax(np.abs(out[sel] - ref))) <= ATOL
    miss = hip_ctx.missingness(X, pi, pj)
    m = np.isnan(X)
    want = np.array([(m[:, i] | m[:, j]).sum() for i, j in zip(pi[:200], pj[:200])])
    assert np.array_equal(miss[:200], want)
