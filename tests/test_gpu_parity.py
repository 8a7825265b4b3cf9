"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bars (BASELINE.json north_star): integer pair counts bit-exact; tau / p-value / tau_max / completeness
within 1e-10 absolute (double).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL = 1e-10  # north_star tolerance for the four doubles


def _oracle():
    from oracle import oracle as O
    return O


def _check(ctx, X, pi=None, pj=None, perspective="global", alternative="two.sided", continuity=False, flags=0):
    O = _oracle()
    S = X.shape[1]
    if pi is None:
        pi_o, pj_o = np.triu_indices(S, k=1)
    else:
        pi_o, pj_o = pi, pj
    out, cnt, rsn = ctx.pairs(X, pi, pj, perspective, alternative, continuity, flags)
    ref, rcnt, rrsn = O.ici_pairs(X, pi_o, pj_o, perspective, alternative, continuity, int32_compat=not (flags & 1))
    assert np.array_equal(rsn, rrsn), (rsn[:20], rrsn[:20])
    ok = rrsn == 0
    assert np.array_equal(cnt[ok], rcnt[ok][:, :cnt.shape[1]]), "integer counts differ"
    assert np.array_equal(np.isnan(out), np.isnan(ref))
    d = np.nanmax(np.abs(out - ref)) if np.any(~np.isnan(ref)) else 0.0
    assert d <= ATOL, d
    # NA pairs carry R's NA_real_ payload
    na = np.isnan(out[:, 0]) & (rrsn != 0)
    if na.any():
        assert np.all(out[na].view(np.uint64) == np.uint64(0x7FF00000000007A2))
    return out, cnt, rsn


def test_selftest(hip_ctx):
    hip_ctx.selftest()


@pytest.mark.parametrize("n", [2, 3, 5, 63, 64, 65, 127, 128, 129, 1000, 4097])
@pytest.mark.parametrize("perspective", ["global", "local"])
def test_random_with_na(hip_ctx, n, perspective):
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, 6))
    X[rng.random(X.shape) < 0.15] = np.nan
    _check(hip_ctx, X, perspective=perspective)


@pytest.mark.parametrize("n", [700, 5000, 12000, 40000])
def test_values_that_differ_only_in_their_low_bits(hip_ctx, n):
    """The pre-pass sorts one-word elements first -- the top 48 bits of a value's sortable key | its row -- and repeats a
    column with (full key, row) elements when the full keys show an inversion (icikt_kernels.hip: kv_gt, k0_prepare).
    Columns whose values share their top 48 key bits and differ below them (relative differences under 2^-36) must take
    that second pass: distinct low bits in random order, with ties among them, beside ordinary columns, negative values
    (keys complemented), +-0 and missing values; and the boundary case where only SOME neighbours collide."""
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, 8))
    ulp = 2.0 ** -52
    k = rng.permutation(n) % 60000
    X[:, 0] = 1.0 + k * ulp                                   # all top-48 prefixes equal: any order but the rows' is an inversion
    X[:, 1] = -(3.0 + (rng.integers(0, 5000, n) * 4) * ulp)    # negative, with ties: 5 000 distinct values below the 48 bits
    X[:, 2] = np.where(rng.random(n) < 0.5, X[:, 2], 7.0 + rng.integers(0, 3, n) * ulp * 8)   # half ordinary, half three close values
    X[:, 3] = np.round(X[:, 3], 2) + rng.integers(0, 2, n) * 2.0 ** -45                     # pairs of near-equal values around each level
    X[:, 4] = np.where(rng.random(n) < 0.3, 0.0, rng.integers(-2, 3, n) * 2.0 ** -1070)      # zeros and subnormals: prefixes all equal
    X[rng.random(X.shape) < 0.05] = np.nan
    X[::97, 4] = -0.0
    for p in ("global", "local"):
        _check(hip_ctx, X, perspective=p)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 4095, 4096, 4097, 8200, 10000, 16384, 16385, 30000, 50000, 65535])
def test_prepass_small_shape(plan_ctx, n):
    """The pre-pass has two shapes of one body: 1 024 threads x 4 elements (default) and 256 threads x 16 (what the
    later chunks of the pipelined host path run beside the pair kernel).  Forced by the plan key k0 = 1 / 0 they must
    prepare the SAME state -- the same results bit for bit, counts included -- on short and long columns, with ties, with
    values that collide in the one-word sort (second pass), and each equals the oracle."""
    rng = np.random.default_rng(n + 5)
    S = 6
    X = rng.standard_normal((n, S))
    if n > 8:
        X[:, 1] = np.round(X[:, 1] * 3)
        X[:, 2] = np.round(X[:, 2] * 200)
        X[:, 3] = 1.0 + (rng.permutation(n) % 50000) * 2.0 ** -52        # one-word collisions everywhere
        X[rng.random(X.shape) < 0.07] = np.nan
        X[:, 5] = np.where(rng.random(n) < 0.6, np.nan, X[:, 5])
    res = {}
    for shape in (0, 1):
        plan_ctx.debug_set_plan({"k0": shape})
        res[shape] = _check(plan_ctx, X, perspective="global") if n <= 20000 else plan_ctx.pairs(X, perspective="global")
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b, equal_nan=True)
    if n > 20000:   # (the oracle on a few pairs only: long columns)
        O = _oracle()
        pi, pj = np.triu_indices(S, k=1)
        ref, rcnt, rrsn = O.ici_pairs(X, pi[:5], pj[:5], "global")
        assert np.array_equal(res[1][1][:5], rcnt[:, :res[1][1].shape[1]]) and np.array_equal(res[1][2][:5], rrsn)
        assert float(np.nanmax(np.abs(res[1][0][:5] - ref))) <= ATOL


@pytest.mark.parametrize("levels", [2, 3, 7, 50])
@pytest.mark.parametrize("perspective", ["global", "local"])
def test_heavy_ties(hip_ctx, levels, perspective):
    # both columns with tie groups >= 3 everywhere (exercises the reference's t0/2 quirk, Q1)
    rng = np.random.default_rng(levels)
    X = rng.integers(0, levels, size=(777, 8)).astype(np.float64)
    X[rng.random(X.shape) < 0.1] = np.nan
    _check(hip_ctx, X, perspective=perspective)
    _check(hip_ctx, X, perspective=perspective, flags=1)


def test_long_tie_groups_span_steps(hip_ctx):
    # tie groups far longer than one 64-row step in both columns, sorted and shuffled
    rng = np.random.default_rng(5)
    n = 3000
    a = np.repeat(np.arange(6.0), n // 6)
    b = np.tile(np.repeat(np.arange(3.0), 100), n // 300)
    c = rng.permutation(a)
    d = rng.standard_normal(n)
    d[:900] = np.nan
    e = a.copy()
    e[rng.random(n) < 0.3] = np.nan
    X = np.stack([a, b, c, d, e, -a], axis=1)
    for p in ("global", "local"):
        _check(hip_ctx, X, perspective=p)


def test_degenerate_columns(hip_ctx):
    rng = np.random.default_rng(11)
    n = 200
    X = rng.standard_normal((n, 7))
    X[:, 1] = np.nan                      # all missing -> NA x4, silent
    X[:, 2] = 3.25                        # constant -> single unique value
    X[5:, 3] = np.nan                     # 5 values, rest missing
    X[:, 4] = np.where(np.arange(n) < 100, np.nan, 1.0)   # NA + constant: two groups
    X[:, 5] = np.where(np.arange(n) < 100, 1.0, np.nan)   # complementary to col 4 in local mode
    X[0, 6] = -0.0
    X[1, 6] = 0.0
    for p in ("global", "local"):
        out, cnt, rsn = _check(hip_ctx, X, perspective=p)
        assert set(rsn.tolist()) >= {0, 1, 3}


def test_short_vectors(hip_ctx):
    O = _oracle()
    for n in (0, 1, 2):
        x = np.arange(n, dtype=np.float64)
        y = x[::-1].copy()
        out, cnt, rsn = hip_ctx.pair(x, y, "global")
        ref, rcnt, rr = O.ici_kt(x, y, "global")
        assert rsn == rr
        assert np.array_equal(np.isnan(out), np.isnan(ref))
    # local: one surviving row
    x = np.array([np.nan, np.nan, 1.0])
    y = np.array([np.nan, np.nan, 2.0])
    out, cnt, rsn = hip_ctx.pair(x, y, "local")
    assert rsn == 2 and np.all(np.isnan(out))


def test_fill_collides_with_min(hip_ctx):
    # fl(min - 0.1) == min: missing values tie with the minimum (kendallc.cpp:214-219 taken literally)
    rng = np.random.default_rng(3)
    n = 500
    X = rng.standard_normal((n, 4)) * 1e18
    X[:, 0] = np.abs(X[:, 0]) + 1e18
    X[rng.random(X.shape) < 0.2] = np.nan
    X[:10, 0] = 1e18
    X[:, 3] = np.where(np.isnan(X[:, 3]), np.nan, -np.inf)
    X[::3, 3] = 7.0
    for p in ("global", "local"):
        _check(hip_ctx, X, perspective=p)


@pytest.mark.parametrize("alternative", ["two.sided", "less", "greater", "bogus"])
@pytest.mark.parametrize("continuity", [False, True])
def test_alternatives(hip_ctx, alternative, continuity):
    rng = np.random.default_rng(17)
    X = rng.standard_normal((300, 5))
    X[:, 1] = X[:, 0] + 0.5 * X[:, 1]
    X[:, 2] = -X[:, 0] + 0.1 * X[:, 2]
    X[rng.random(X.shape) < 0.05] = np.nan
    _check(hip_ctx, X, alternative=alternative, continuity=continuity)


def test_explicit_pair_list_and_self_pairs(hip_ctx):
    rng = np.random.default_rng(23)
    X = rng.standard_normal((400, 9))
    X[rng.random(X.shape) < 0.1] = np.nan
    pi = np.array([0, 0, 5, 8, 3, 3, 3, 2, 7, 1, 4], dtype=np.int32)
    pj = np.array([1, 0, 2, 8, 4, 5, 6, 2, 0, 0, 4], dtype=np.int32)
    _check(hip_ctx, X, pi, pj, perspective="global")
    _check(hip_ctx, X, pi, pj, perspective="local")


def test_unsorted_pair_lists_group_into_tasks(hip_ctx):
    """Arbitrary pair lists (shuffled, duplicates, pi > pj, self pairs, odd column counts): the host groups
    pairs that share pj and whose pi are the two columns of one rec block into one task; every pair's
    result must still land at its own position."""
    rng = np.random.default_rng(61)
    n, S = 1500, 9
    X = rng.standard_normal((n, S))
    X[rng.random(X.shape) < 0.15] = np.nan
    X[:, 4] = np.round(X[:, 4] * 2)
    pi, pj = np.meshgrid(np.arange(S), np.arange(S), indexing="ij")
    pi, pj = pi.ravel().astype(np.int32), pj.ravel().astype(np.int32)     # all 81 ordered pairs incl. self
    for trial in range(3):
        perm = rng.permutation(len(pi))
        sel = np.concatenate([perm, perm[:17]])                              # duplicates
        _check(hip_ctx, X, pi[sel], pj[sel], perspective="global")
    # sorted by (pi, pj) but with gaps: rows without a partner row, partner rows with different pj sets
    keep = rng.random(len(pi)) < 0.6
    _check(hip_ctx, X, pi[keep], pj[keep], perspective="local")


def test_golden_snapshot_50000(hip_ctx, golden_dir, expected):
    # reference snapshot: tests/testthat/_snaps/kendall-tau.md:6-7 (n = 50 000, no ties)
    z = np.load(f"{golden_dir}/snapshot_50000.npz")
    out, cnt, rsn = hip_ctx.pair(z["x"], z["y"], "global")
    e = expected["snapshot_50000"]
    assert abs(out[0] - e["tau"]) < 5e-9 and abs(out[1] - e["pvalue"]) < 5e-9
    assert out[2] == 1.0 and out[3] == 1.0
    ref, rcnt, _ = _oracle().ici_kt(z["x"], z["y"], "global")
    assert cnt["dis"] == rcnt["dis"] and np.max(np.abs(out - ref)) <= ATOL


def test_golden_readme(hip_ctx, golden_dir, expected):
    z = np.load(f"{golden_dir}/readme_s1_s4.npz")
    e = expected["readme"]["r_4"]
    out, cnt, rsn = hip_ctx.pair(z["s3"], z["s4"], "global")
    assert abs(out[0] - e["raw"]) < 5e-8 and abs(out[2] - e["taumax"]) < 5e-7 and abs(out[3] - e["completeness"]) < 1e-12
    assert abs(out[0] / out[2] - e["cor"]) < 5e-8
    out, _, _ = hip_ctx.pair(z["s1"], z["s2"], "global")
    assert out[0] == 1.0


def test_yeast_config_c2(hip_ctx, golden_dir):
    # BASELINE config c2: yeast_missing 6887 x 96, zeros -> NA, perspective = global (4 560 pairs);
    # the oracle is checked on a pair subsample to keep the CPU side short.
    z = np.load(f"{golden_dir}/yeast_missing.npz")
    X = z["X"].copy()
    X[X == 0] = np.nan
    out, cnt, rsn = hip_ctx.pairs(X, perspective="global")
    iu, ju = np.triu_indices(X.shape[1], k=1)
    rng = np.random.default_rng(0)
    sel = rng.choice(len(iu), 400, replace=False)
    ref, rcnt, rrsn = _oracle().ici_pairs(X, iu[sel], ju[sel], "global")
    assert np.array_equal(rsn[sel], rrsn)
    assert np.array_equal(cnt[sel], rcnt[:, :cnt.shape[1]])
    assert np.nanmax(np.abs(out[sel] - ref)) <= ATOL
    out_l, cnt_l, rsn_l = hip_ctx.pairs(X, iu[sel], ju[sel], perspective="local")
    ref_l, rcnt_l, rrsn_l = _oracle().ici_pairs(X, iu[sel], ju[sel], "local")
    assert np.array_equal(cnt_l, rcnt_l[:, :cnt_l.shape[1]])
    assert np.nanmax(np.abs(out_l - ref_l)) <= ATOL


def test_missingness(hip_ctx):
    rng = np.random.default_rng(2)
    X = rng.standard_normal((333, 10))
    X[rng.random(X.shape) < 0.2] = np.nan
    pi, pj = np.triu_indices(10, k=0)
    m = hip_ctx.missingness(X, pi, pj)
    ref = np.array([(np.isnan(X[:, i]) | np.isnan(X[:, j])).sum() for i, j in zip(pi, pj)])
    assert np.array_equal(m, ref)


def test_combn_ranges_concatenate_to_full(hip_ctx):
    """The multi-GPU sharding unit: contiguous [begin, end) blocks of combn order (bench.py, one per rank)."""
    import torch
    rng = np.random.default_rng(31)
    n, S = 700, 23
    X = np.asfortranarray(rng.standard_normal((n, S)))
    X[rng.random(X.shape) < 0.1] = np.nan
    full, _, _ = hip_ctx.pairs(X, perspective="global", want_counts=False)
    P = S * (S - 1) // 2
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    hip_ctx.prepare_dev(dX.data_ptr(), n, S, n)
    parts = []
    world = 3
    n_each = -(-P // world)
    for r in range(world):
        b, e = min(P, r * n_each), min(P, (r + 1) * n_each)
        hip_ctx.set_pairs_combn(S, b, e)
        assert hip_ctx.num_pairs() == e - b
        out = torch.empty((e - b, 4), dtype=torch.float64, device="cuda")
        hip_ctx.run_dev(1, 0, False, 0, out.data_ptr())
        hip_ctx.sync()
        parts.append(out.cpu().numpy())
    assert np.array_equal(np.concatenate(parts), full, equal_nan=True)


def test_api_through_hip_engine(hip_ctx, golden_dir, expected):
    """ici_kendalltau() / ici_kt() / pairwise_completeness() with the default (MI355X) engine."""
    from icikendalltau_amd import api
    from tests.oracle_engine import OracleEngine
    z = np.load(f"{golden_dir}/readme_s1_s4.npz")
    m2 = np.c_[z["s3"], z["s4"]]
    r4 = api.ici_kendalltau(m2, colnames=["s3", "s4"], return_matrix=False)["cor"]
    e = expected["readme"]["r_4"]
    assert round(r4["raw"][0], 7) == e["raw"] and round(r4["cor"][0], 7) == e["cor"]
    X = np.load(f"{golden_dir}/completeness_50x100.npz")["X"]
    names = [f"s{i}" for i in range(1, 101)]
    g = api.ici_kendalltau(X, colnames=names)
    o = api.ici_kendalltau(X, colnames=names, engine=OracleEngine())
    for k in ("cor", "raw", "pvalue", "taumax", "completeness"):
        assert np.nanmax(np.abs(g[k].to_numpy() - o[k].to_numpy())) <= ATOL
    assert np.array_equal(api.pairwise_completeness(X, colnames=names).to_numpy(),
                          api.pairwise_completeness(X, colnames=names, engine=OracleEngine()).to_numpy())
    res = api.ici_kt(z["s3"], z["s4"], "global")
    assert round(res.tau, 7) == e["raw"]
    with pytest.warns(RuntimeWarning, match="single unique value"):
        api.ici_kt(np.arange(10.0), np.ones(10))


def test_c5_shape_include_only_subset(hip_ctx):
    """Config c5's shape at a column count the oracle finishes in seconds: n = 50 000, 1 000 smallest values per
    column missing, include_only = the first four names (vector form: every pair with one of them),
    perspective local and global, plus pairwise_completeness on the same subset."""
    from icikendalltau_amd import api
    from tests.oracle_engine import OracleEngine
    rng = np.random.default_rng(5)
    n, S = 50000, 40
    X = rng.standard_normal((n, S))
    idx = np.argpartition(X, 1000, axis=0)[:1000]
    np.put_along_axis(X, idx, np.nan, axis=0)
    names = [f"s{i}" for i in range(S)]
    for persp in ("global", "local"):
        g = api.ici_kendalltau(X, colnames=names, include_only=names[:4], perspective=persp, return_matrix=False)["cor"]
        o = api.ici_kendalltau(X, colnames=names, include_only=names[:4], perspective=persp, return_matrix=False,
                               engine=OracleEngine())["cor"]
        assert len(g) == 4 * (S - 1) - 6 + S and list(g["s1"]) == list(o["s1"]) and list(g["s2"]) == list(o["s2"])
        for k in ("raw", "pvalue", "taumax", "completeness", "cor"):
            assert np.nanmax(np.abs(g[k].to_numpy() - o[k].to_numpy())) <= ATOL, (persp, k)
    pc_g = api.pairwise_completeness(X, colnames=names, include_only=names[:4], return_matrix=False)
    pc_o = api.pairwise_completeness(X, colnames=names, include_only=names[:4], return_matrix=False, engine=OracleEngine())
    assert np.array_equal(pc_g["missingness"].to_numpy(), pc_o["missingness"].to_numpy())
    assert np.allclose(pc_g["completeness"].to_numpy(), pc_o["completeness"].to_numpy(), rtol=0, atol=1e-15)


@pytest.mark.parametrize("np_", ["1", "2"])
def test_pairs_per_wave_variants(plan_ctx, np_):
    """K1 launch plans: one pair per wave or two (one per half); odd run lengths leave partly filled waves."""
    hip_ctx = plan_ctx
    hip_ctx.debug_set_plan({"np": np_})
    rng = np.random.default_rng(41)
    X = rng.standard_normal((2500, 11))
    X[rng.random(X.shape) < 0.12] = np.nan
    X[:, 2] = np.round(X[:, 2] * 3)          # ties inside the streamed/gathered columns
    X[:, 7] = np.round(X[:, 7])
    for p in ("global", "local"):
        _check(hip_ctx, X, perspective=p)
    pi = np.array([0, 0, 0, 1, 3, 3, 3, 3, 3, 9, 2], dtype=np.int32)
    pj = np.array([4, 5, 6, 1, 0, 1, 2, 4, 5, 10, 7], dtype=np.int32)
    _check(hip_ctx, X, pi, pj, perspective="global")


def test_long_columns_generic_rebuild(hip_ctx):
    """n = 30 000: more than 4 bitset words per lane (generic prefix rebuild), multi-tile K0 merges."""
    rng = np.random.default_rng(43)
    n = 30000
    X = rng.standard_normal((n, 5))
    X[rng.random(X.shape) < 0.05] = np.nan
    X[:, 1] = np.round(X[:, 1] * 20)         # ~100 distinct values: tie groups of hundreds of rows
    X[:, 4] = np.where(rng.random(n) < 0.5, np.nan, X[:, 4])
    for p in ("global", "local"):
        _check(hip_ctx, X, perspective=p)


@pytest.mark.parametrize("n", [10177, 10300, 16321, 17000, 21000, 24900, 26000, 41000])
def test_one_pair_per_wave_plans(hip_ctx, n):
    """Column lengths on either side of every K1 plan boundary above the half-wave range: one pair per wave
    with the four-row all-pairs step, <= 4 and > 4 bitset words per lane (two-words-at-a-time rebuild with
    padded arrays), pend in LDS and in global memory; continuous, tied and half-missing columns."""
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, 4))
    X[rng.random(X.shape) < 0.04] = np.nan
    X[:, 1] = np.round(X[:, 1] * 4)            # ~30 distinct values: tie groups spanning many steps
    X[:, 3] = np.where(rng.random(n) < 0.5, np.nan, 0.5 * X[:, 0] + X[:, 3])
    for p in ("global", "local"):
        _check(hip_ctx, X, perspective=p)


@pytest.mark.parametrize("n", [15200, 17000])
@pytest.mark.parametrize("tied", [False, True])
def test_kernel_family_by_tie_structure_nine_words(plan_ctx, n, tied):
    """15 200 .. 18 336 rows with a task list that fills the chip: continuous columns take the long-column kernel, tied
    ones the half-wave kernels (plan_k1 / matrix_tied) -- both against the oracle, and against either family forced."""
    rng = np.random.default_rng(n + int(tied))
    X = rng.standard_normal((n, 48))                   # 1 128 pairs: more than four per CU
    if tied:
        X = np.round(X * 150)                          # ~900 distinct values
    X[rng.random(X.shape) < 0.03] = np.nan
    auto = _check(plan_ctx, X, perspective="global")
    for half in ("0", "1"):
        plan_ctx.debug_set_plan({"half": half})
        forced = plan_ctx.pairs(X, perspective="global")
        assert np.array_equal(forced[1], auto[1]) and np.array_equal(forced[0], auto[0], equal_nan=True)


def test_max_length_65535(hip_ctx):
    rng = np.random.default_rng(47)
    n = 65535
    x = rng.standard_normal(n)
    y = 0.3 * x + rng.standard_normal(n)
    y[rng.random(n) < 0.03] = np.nan
    O = _oracle()
    for p in ("global", "local"):
        out, cnt, rsn = hip_ctx.pair(x, y, p)
        ref, rcnt, rr = O.ici_kt(x, y, p)
        assert rsn == rr == 0
        assert all(cnt[k] == rcnt[k] for k in cnt)
        assert np.max(np.abs(out - ref)) <= ATOL
    from icikendalltau_amd._lib import IciktError
    with pytest.raises(IciktError, match="262144"):
        hip_ctx.pair(np.zeros(262145), np.zeros(262145))


def test_wide_columns_longest(hip_ctx):
    """n = ICIKT_MAX_FEATURES_WIDE = 262 144 rows: one wave per CU (98 KB of LDS state), three columns."""
    n = 262144
    rng = np.random.default_rng(5)
    base = rng.standard_normal(n)
    X = base[:, None] + 0.04 * rng.standard_normal((n, 3))   # tau ~ 0.97: `dis` must fit the reference's int
    X[:, 1] = np.round(X[:, 1] * 300)
    X[np.argsort(X[:, 2])[:5000], 2] = np.nan
    X = np.asfortranarray(X)
    O = _oracle()
    pi, pj = (a.astype(np.int32) for a in np.triu_indices(3, k=1))
    out, cnt, rsn = hip_ctx.pairs(X, perspective="global")
    ref, rcnt, rrsn = O.ici_pairs(X, pi, pj, "global", int32_compat=False)
    assert int(rcnt[:, 2].max()) < 2**30               # far from where the oracle's int dis saturates
    assert np.array_equal(rsn, rrsn) and np.array_equal(cnt, rcnt[:, :cnt.shape[1]])
    assert float(np.nanmax(np.abs(out - ref))) <= ATOL


@pytest.mark.parametrize("n", [65536, 70001, 140000])
def test_wide_columns(hip_ctx, n):
    """65 535 < n <= 262 144 rows: the plain 32-bit path (k0_prepare<true>, k1_wide), exact integer arithmetic --
    compared with the oracle in its exact mode (int32_compat = False; the reference's int32 sums wrap at these
    lengths).  Continuous columns, small tie groups (batches of complete groups), a few huge groups (the two-pass
    long-group path), left-censored and scattered missing values, both perspectives; then the host entry for a
    matrix, its explicit-list form, REUSE_COUNTS and missingness on the same shape."""
    rng = np.random.default_rng(n)
    S = 5
    # (the columns are positively correlated: the reference keeps `dis` in an int, src/kendallc.cpp:78, and so does
    #  the oracle -- with independent columns of 140 000 rows the count itself, ~n^2 / 4, no longer fits)
    base = rng.standard_normal(n)
    X = base[:, None] + (0.35 if n < 100000 else 0.1) * rng.standard_normal((n, S))
    X[:, 1] = np.round(X[:, 1] * 2000)            # groups of a few rows up to ~50
    X[:, 2] = np.round(X[:, 2] * 1.5)             # ~10 groups of thousands of rows
    k = n // 20
    idx = np.argpartition(X[:, [0, 3]], k, axis=0)[:k]
    for c, col in enumerate((0, 3)):
        X[idx[:, c], col] = np.nan                # left-censored: the missing rows are the last (long) tie group
    X[rng.random(n) < 0.02, 4] = np.nan
    X = np.asfortranarray(X)
    O = _oracle()
    pi, pj = (a.astype(np.int32) for a in np.triu_indices(S, k=1))
    for perspective in ("global", "local"):
        out, cnt, rsn = hip_ctx.pairs(X, perspective=perspective)
        ref, rcnt, rrsn = O.ici_pairs(X, pi, pj, perspective, int32_compat=False)
        assert np.array_equal(rsn, rrsn)
        assert int(rcnt[:, 2].max()) < 2**30              # dis far from where the reference's int saturates
        assert np.array_equal(cnt, rcnt[:, :cnt.shape[1]])
        assert float(np.nanmax(np.abs(out - ref))) <= ATOL
    # explicit list in another order, pairs in both orientations; a single pair through the vector entry
    sel = np.array([9, 0, 4, 4, 7], dtype=np.int64)
    qi = np.where(np.arange(len(sel)) % 2 == 0, pi[sel], pj[sel]).astype(np.int32)
    qj = np.where(np.arange(len(sel)) % 2 == 0, pj[sel], pi[sel]).astype(np.int32)
    out_l, cnt_l, rsn_l = hip_ctx.pairs(X, qi, qj, "global")
    ref_l, rcnt_l, rrsn_l = O.ici_pairs(X, qi, qj, "global", int32_compat=False)
    assert np.array_equal(rsn_l, rrsn_l) and np.array_equal(cnt_l, rcnt_l[:, :cnt_l.shape[1]])
    assert float(np.nanmax(np.abs(out_l - ref_l))) <= ATOL
    o1, c1, r1 = hip_ctx.pair(X[:, 1], X[:, 2], "local")
    ro, rc_, rr = O.ici_kt(X[:, 1], X[:, 2], "local", int32_compat=False)
    assert r1 == rr and all(c1[f] == rc_[f] for f in c1) and np.max(np.abs(o1 - ro)) <= ATOL
    miss = hip_ctx.missingness(X, pi, pj)
    M = np.isnan(X)
    assert np.array_equal(miss, np.array([(M[:, a] | M[:, b]).sum() for a, b in zip(pi, pj)]))


def test_int32_wrap_of_a_large_joint_cell(hip_ctx):
    """Q2 at its far end: a joint (x group, y group) cell of >= 46 342 rows makes the reference's int32
    cnt * (cnt - 1) wrap in `ntie`.  The (fill, fill) cell, a (value, fill) cell and a (value, value) cell,
    against the oracle's int32 and exact modes."""
    rng = np.random.default_rng(71)
    n = 60000
    big = rng.permutation(n)[:50000]
    X = rng.standard_normal((n, 4))
    X[big, 0] = 1.25            # one value 50 000 times
    X[big[:48000], 1] = np.nan  # 48 000 missing, all inside column 0's big group: a (value, fill) cell
    X[big[:47000], 2] = -3.0    # (value, value) cell with column 0
    X[big[:49000], 3] = np.nan  # with column 1: a (fill, fill) cell of 48 000 rows
    for p in ("global", "local"):
        _check(hip_ctx, X, perspective=p, flags=0)
        _check(hip_ctx, X, perspective=p, flags=1)


def test_int32_wrap_regime_matches_oracle(hip_ctx):
    """Q2: tie groups >= 1024 rows make the reference's int32 tie sums wrap; default mode reproduces the
    wrap, ICIKT_FLAG_EXACT_INT64 does not (both against the oracle's two modes)."""
    rng = np.random.default_rng(53)
    n = 6000
    X = rng.standard_normal((n, 4))
    X[:1500, 0] = np.nan
    X[rng.random(n) < 0.25, 1] = np.nan
    X[:, 2] = np.round(X[:, 2])              # a few huge non-missing tie groups
    for p in ("global", "local"):
        _check(hip_ctx, X, perspective=p, flags=0)
        _check(hip_ctx, X, perspective=p, flags=1)


@pytest.mark.parametrize("S,world", [(10, 4), (5, 4), (3, 3)])
def test_sharded_prepass_matches_full(hip_ctx, S, world):
    """icikt_prepare_cols_dev over column ranges (what ranks do before the all-gather) fills the same
    prepared state as one full icikt_prepare_dev.  (5, 4) and (3, 3): the last rank's range is EMPTY and starts at an
    odd column -- it used to be refused, which sent every rank back to the replicated pre-pass.)"""
    import torch
    rng = np.random.default_rng(59)
    n = 900
    X = np.asfortranarray(rng.standard_normal((n, S)))
    X[rng.random(X.shape) < 0.1] = np.nan
    X[:, S - 1] = np.round(X[:, S - 1] * 2)
    full, _, _ = hip_ctx.pairs(X, perspective="global", want_counts=False)
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    cols_per = 2 * -(-S // (2 * world))  # even: the rec table interleaves column pairs
    alloc = cols_per * world
    for r in range(world):  # one context plays all ranks in turn: the slices land in the same arrays
        c0, c1 = min(S, r * cols_per), min(S, (r + 1) * cols_per)
        hip_ctx.prepare_cols_dev(dX.data_ptr(), n, S, n, c0, c1, alloc)
    arrays = hip_ctx.prep_arrays()
    assert len(arrays) == 5 and all(ptr and bpc > 0 for ptr, bpc in arrays)
    P = S * (S - 1) // 2
    hip_ctx.set_pairs_combn(S, 0, P)
    out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
    hip_ctx.run_dev(1, 0, False, 0, out.data_ptr())
    hip_ctx.sync()
    assert np.array_equal(out.cpu().numpy(), full, equal_nan=True)


def test_expand_cols_rebuilds_rec_hirow_tgroups(hip_ctx):
    """icikt_expand_cols_dev: rec / hirow / tgroups rebuilt from order + gflag equal what the pre-pass wrote
    (ranks exchange only order, the bitsets and stats), for continuous, tied, constant and all-missing columns."""
    import torch
    from icikendalltau_amd import _lib
    rng = np.random.default_rng(67)
    for n in (1, 63, 64, 65, 900, 5000, 20000):
        S = 7
        X = np.asfortranarray(rng.standard_normal((n, S)))
        X[rng.random(X.shape) < 0.1] = np.nan
        X[:, 1] = np.round(X[:, 1] * 2)
        X[:, 2] = 3.0
        X[:, 3] = np.nan
        X[: n // 2, 4] = np.nan
        X[:, 5] = np.round(X[:, 5] * 50)
        dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
        hip_ctx.prepare_dev(dX.data_ptr(), n, S, n)
        hip_ctx.sync()
        arrays = hip_ctx.prep_arrays()

        def view(i, cols):
            ptr, bpc = arrays[i]
            class _D:  # zero-copy view of the library's buffer
                __cuda_array_interface__ = {"shape": (bpc * cols,), "typestr": "|u1", "data": (ptr, False), "version": 2}
            return torch.as_tensor(_D(), device="cuda")

        alloc = S + (S & 1)
        derived = {i: view(i, alloc if i in (1, 2) else S) for i in (1, 2, 4)}   # rec and hirow interleave column pairs
        want = {i: t.clone() for i, t in derived.items()}
        # tgroups beyond a column's ntg entries and the pad rows of rec / hirow are unspecified: compare what K1 reads
        for t in derived.values():
            t.fill_(0xEE)
        hip_ctx.expand_cols_dev(0, S)
        hip_ctx.sync()
        meta = view(3, S).cpu().numpy().view(np.uint32).reshape(S, -1)   # per column: 3 bitsets, then 16 words of stats
        ntg = meta[:, -16 + 7]
        n_pad = arrays[2][1] // 2            # rows of a rec / hirow block: the column's rows, padding, and the guard row
        guard = (n + 63) // 64 * 64        # PrepView::rec_rows: row 64 ceil(n / 64) of every column
        rec_w = want[1].cpu().numpy().view(np.uint32).reshape(-1, n_pad, 2)
        rec_g = derived[1].cpu().numpy().view(np.uint32).reshape(-1, n_pad, 2)
        hi_w = want[2].cpu().numpy().view(np.uint16).reshape(-1, n_pad, 2)
        hi_g = derived[2].cpu().numpy().view(np.uint16).reshape(-1, n_pad, 2)
        tg_w = want[4].cpu().numpy().view(np.uint32).reshape(S, -1)
        tg_g = derived[4].cpu().numpy().view(np.uint32).reshape(S, -1)
        for c in range(S):
            assert np.array_equal(rec_g[c >> 1, :n, c & 1], rec_w[c >> 1, :n, c & 1]), (n, c)
            assert np.array_equal(hi_g[c >> 1, :n, c & 1], hi_w[c >> 1, :n, c & 1]), (n, c)
            assert rec_g[c >> 1, guard, c & 1] == rec_w[c >> 1, guard, c & 1] == guard, (n, c)   # q = 64 W, lo = 0
            assert hi_g[c >> 1, guard, c & 1] == hi_w[c >> 1, guard, c & 1] == 0, (n, c)
            assert np.array_equal(tg_g[c, :ntg[c]], tg_w[c, :ntg[c]]), (n, c)


def test_kt_fast_through_hip_engine(hip_ctx, golden_dir, expected):
    """kt_fast() NA policies on the MI355X engine against the reference's snapshots and the checker engine."""
    from icikendalltau_amd import api
    from tests.oracle_engine import OracleEngine
    X = np.load(f"{golden_dir}/ktfast_100x4.npz")["X"]
    names = ["s1", "s2", "s3", "s4"]
    e = expected["ktfast"]
    x_na2 = X.copy()
    x_na2[9, 0] = np.nan
    for use in ("complete.obs", "pairwise.complete.obs", "everything"):
        g = api.kt_fast(x_na2, use=use, colnames=names)
        o = api.kt_fast(x_na2, use=use, colnames=names, engine=OracleEngine())
        for k in ("tau", "pvalue"):
            a, b = g[k].to_numpy(), o[k].to_numpy()
            assert np.array_equal(np.isnan(a), np.isnan(b))
            assert np.all(np.isnan(a)) or np.nanmax(np.abs(a - b)) <= ATOL
    mp = api.kt_fast(x_na2, use="pairwise.complete.obs", colnames=names)
    assert float(f"{mp['tau'].iloc[1, 2]:.7g}") == e["pairwise_tau"]["1-2"]
    assert float(f"{mp['pvalue'].iloc[1, 1]:.7g}") == e["self_p_n100"]
    # a larger pairwise case with scattered NAs (every pair drops different rows)
    rng = np.random.default_rng(61)
    Y = rng.standard_normal((400, 12))
    Y[rng.random(Y.shape) < 0.1] = np.nan
    Y[:, 5] = np.round(Y[:, 5] * 2)
    nm = [f"c{i}" for i in range(12)]
    g = api.kt_fast(Y, use="pairwise.complete.obs", colnames=nm)
    o = api.kt_fast(Y, use="pairwise.complete.obs", colnames=nm, engine=OracleEngine())
    assert np.nanmax(np.abs(g["tau"].to_numpy() - o["tau"].to_numpy())) <= ATOL
    assert np.nanmax(np.abs(g["pvalue"].to_numpy() - o["pvalue"].to_numpy())) <= ATOL


def test_c_abi_error_contract(hip_ctx):
    """Status codes + messages instead of exceptions across the boundary (include/icikt.h)."""
    import ctypes
    import torch
    from icikendalltau_amd import _lib
    L = _lib.lib()
    h = hip_ctx._h
    X = np.asfortranarray(np.random.default_rng(0).standard_normal((50, 4)))
    out = np.empty((1, 4))
    bad = np.array([9], dtype=np.int32)
    ok = np.array([0], dtype=np.int32)
    rc = L.icikt_pairs_f64(h, X.ctypes.data, 50, 4, 50, bad.ctypes.data, ok.ctypes.data, 1, 1, 0, 0, 0,
                           out.ctypes.data, None, None)
    assert rc == -1 and b"out of range" in L.icikt_last_error(h)
    rc = L.icikt_pairs_f64(h, X.ctypes.data, 50, 4, 10, ok.ctypes.data, ok.ctypes.data, 1, 1, 0, 0, 0,
                           out.ctypes.data, None, None)
    assert rc == -1                                    # ld < n_feat
    rc = L.icikt_pairs_f64(h, X.ctypes.data, 50, 4, 50, ok.ctypes.data, ok.ctypes.data, 1, 7, 0, 0, 0,
                           out.ctypes.data, None, None)
    assert rc == -1 and b"perspective" in L.icikt_last_error(h)
    c2 = _lib.Context(0)
    d = torch.empty(4, dtype=torch.float64, device="cuda")
    with pytest.raises(_lib.IciktError, match="prepare"):
        c2.run_dev(1, 0, False, 0, d.data_ptr())       # run before prepare
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    c2.prepare_dev(dX.data_ptr(), 50, 4, 50)
    with pytest.raises(_lib.IciktError, match="pair list"):
        c2.run_dev(1, 0, False, 0, d.data_ptr())       # run before set_pairs
    c2.set_pairs(np.array([0, 5], np.int32), np.array([1, 2], np.int32))
    with pytest.raises(_lib.IciktError, match="column"):
        c2.run_dev(1, 0, False, 0, d.data_ptr())       # pair list refers to column 5 of 4
    with pytest.raises(_lib.IciktError, match="range"):
        c2.set_pairs_combn(4, 0, 7)                    # C(4,2) = 6
    c2.close()
    # zero pairs / zero rows are not errors
    o, c, r = hip_ctx.pairs(X, np.array([], np.int32), np.array([], np.int32))
    assert o.shape == (0, 4)
    o, c, r = hip_ctx.pairs(np.empty((0, 3)), perspective="global")
    assert o.shape == (3, 4) and np.all(r == 1) and np.all(np.isnan(o))


@pytest.mark.parametrize("plan", [{"tgmax": "-1"}, {"tgmax": "3"}, {"tgmax": "40"}, {"tgmax": "1000000"}, {"tgmax": "1000000", "list": "0"},
                                  {"tgmax": "60", "list": "2"}, None])
def test_joint_tie_counting_modes(plan_ctx, plan):
    """Joint ties of tie groups of more than 32 rows, by the tie groups of the gathered column: range counts per listed
    group (list mode: up to `list` <= 128 groups), a counter per group in the pair's table (count mode: up to tgmax groups,
    as many as the LDS holds) or row by row; every mode forced on columns with 2 ... 1 500 tie groups, long and short.
    tgmax = 3 / 40 / 60: the two pairs of a task differ in their mode (columns 0 | 1, 2 | 3, 4 | 5, 6 | 7); list = 0: no
    list mode, every same-address atomic of count mode on the three-group column; None: the library's own choice."""
    hip_ctx = plan_ctx
    hip_ctx.debug_set_plan(plan)
    rng = np.random.default_rng(71)
    n = 4000
    X = np.empty((n, 8))
    X[:, 0] = rng.standard_normal(n)                         # one tie group after masking (missing only)
    X[:, 1] = np.round(rng.standard_normal(n) * 400)         # ~1 500 small tie groups
    X[:, 2] = rng.integers(0, 3, n)                          # three huge groups
    X[:, 3] = np.round(rng.standard_normal(n) * 5)           # ~30 groups of ~100+
    X[:, 4] = np.sort(X[:, 3])                               # the same values, sorted: groups contiguous in row order
    X[:, 5] = np.where(rng.random(n) < 0.6, np.nan, rng.standard_normal(n))
    X[:, 6] = np.repeat(np.arange(n // 80), 80)              # groups of 80: every group spans two steps
    X[:, 7] = -X[:, 6]
    X[rng.random(X.shape) < 0.07] = np.nan
    for p in ("global", "local"):
        _check(hip_ctx, X, perspective=p)


@pytest.mark.parametrize("n", [20000, 33000, 50000, 65535])
@pytest.mark.parametrize("plan", [None, {"np": "1"}, {"tgmax": "1000000"}, {"tgmax": "300", "half": "0"}, {"tgmax": "-1"}])
def test_whole_wave_count_mode(plan_ctx, n, plan):
    """Count mode of the whole-wave kernels (columns too long for the half-wave ones, or `half=0`): a counter per tie group of
    the gathered column -- the pre-pass writes girow for long columns of more than eight tie groups -- against list and row mode,
    on streamed groups of one step, two (rows kept in registers), and many (counters read back by tie group), with gathered
    columns of ~10, ~150 (list), ~300, ~900 (count) and thousands (row mode, or count mode when the table is forced) of groups."""
    plan_ctx.debug_set_plan(plan)
    rng = np.random.default_rng(n + 5)
    base = rng.standard_normal((n, 8))
    X = np.empty((n, 8))
    X[:, 0] = np.round(base[:, 0] * 1.5)          # ~10 groups of thousands of rows
    X[:, 1] = np.round(base[:, 1] * 25)           # ~150 groups
    X[:, 2] = np.round(base[:, 2] * 50)           # ~300
    X[:, 3] = np.round(base[:, 3] * 150)          # ~900
    X[:, 4] = np.round(base[:, 4] * 1000)         # thousands of short groups
    X[:, 5] = base[:, 5]                          # continuous
    X[:, 6] = np.round(0.7 * base[:, 2] * 50 + 0.3 * base[:, 6] * 50)   # correlated with column 2: joint ties
    X[:, 7] = np.repeat(np.arange((n + 99) // 100), 100)[:n]           # groups of 100 rows, contiguous in row order
    X[rng.random(X.shape) < 0.04] = np.nan
    _check(plan_ctx, X, perspective="global")
    if n <= 33000:
        _check(plan_ctx, X, perspective="local")


@pytest.mark.parametrize("n", [900, 6887, 10000, 18000, 30000])
@pytest.mark.parametrize("plan", [None, {"solo": "0"}, {"list": "0", "tgmax": "1000000"}, {"waves": "4"}])
def test_tie_program_step_kinds(plan_ctx, n, plan):
    """Every step kind of the tie program against every joint-tie mode: streamed columns whose tie groups have 1, 2-5,
    6-16, 17-32 (SOLO: one group per 32-row sub-step), 33-64, 65-128 (two steps, rows kept in registers) and several
    hundred rows (streamed again at their close), in runs and interleaved, against gathered columns with one tie group, a
    handful, ~100 (list mode), several hundred (count mode) and thousands (row mode, or count mode where the launch is
    small enough for a large table: `waves`).  Both orders of every pair are in the list: each column streams and is
    gathered."""
    rng = np.random.default_rng(n + 17)

    def grouped(sizes):   # a column made of tie groups of the given sizes, values in random row order
        v = np.repeat(np.arange(len(sizes), dtype=np.float64), sizes)[:n]
        v = np.concatenate([v, np.arange(n - len(v)) + len(sizes)]) if len(v) < n else v
        return rng.permutation(v)

    def sizes_of(lo, hi):
        out, tot = [], 0
        while tot < n:
            out.append(int(rng.integers(lo, hi + 1)))
            tot += out[-1]
        return out

    cols = [rng.standard_normal(n),                          # continuous: one tie group after masking
            grouped(sizes_of(1, 5)), grouped(sizes_of(6, 16)), grouped(sizes_of(17, 32)), grouped(sizes_of(33, 64)),
            grouped(sizes_of(65, 128)), grouped(sizes_of(200, 700)),
            grouped([int(x) for x in rng.choice([1, 1, 1, 3, 20, 30, 50, 100, 400], size=n)]),   # every kind interleaved
            np.round(rng.standard_normal(n) * 15), np.round(rng.standard_normal(n) * 60)]       # ~100 / ~400 tie groups
    X = np.asfortranarray(np.stack(cols, axis=1))
    X[rng.random(X.shape) < 0.04] = np.nan
    S = X.shape[1]
    pi, pj = np.triu_indices(S, k=1)
    pi, pj = np.concatenate([pi, pj]).astype(np.int32), np.concatenate([pj, pi]).astype(np.int32)
    plan_ctx.debug_set_plan(plan)
    _check(plan_ctx, X, pi, pj, perspective="global")


@pytest.mark.parametrize("n", [3400, 10000, 18336, 30600])
def test_tie_program_most_steps_per_row(hip_ctx, n):
    """The step records of a column are sized for the most steps its tie program can have, n / 17 + a few: a MIXED step of
    fewer than 33 rows is followed by a group of more than 32 rows, and a closing GROUP step of fewer than 33 rows follows a
    64-row piece of its group.  The patterns that come closest: tie groups of 1 and 33 rows in turn (two steps per 34
    rows), 1 / 65 (three per 66), 1 / 33 / 65 mixed -- as streamed columns of every pair, against each other and a
    continuous column."""
    rng = np.random.default_rng(n)

    def pattern(sizes):
        reps = n // sum(sizes) + 1
        v = np.repeat(np.arange(reps * len(sizes), dtype=np.float64), np.tile(sizes, reps))[:n]
        return rng.permutation(v)

    X = np.asfortranarray(np.stack([pattern([1, 33]), pattern([33, 1]), pattern([1, 65]), pattern([1, 33, 1, 65, 1, 34, 2, 64]),
                                    rng.standard_normal(n)], axis=1))
    X[rng.random(X.shape) < 0.01] = np.nan
    S = X.shape[1]
    pi, pj = np.triu_indices(S, k=1)
    pi, pj = np.concatenate([pi, pj]).astype(np.int32), np.concatenate([pj, pi]).astype(np.int32)
    _check(hip_ctx, X, pi, pj, perspective="global")


@pytest.mark.parametrize("split", ["1", "2", "4", None])
@pytest.mark.parametrize("n", [70, 900, 2500, 6887, 17000, 30000])
def test_task_segments(plan_ctx, n, split):
    """A task list that leaves the chip half empty is cut in segments (a wave per segment: it inserts the rows in front of
    its segment uncounted, walks its part, the segments add their counts up).  1 / 2 / 4 segments forced, and the library's
    own choice, on columns whose segment marks fall in the singleton region (continuous data: the only tie group is the
    missing rows'), inside the tie program (tied data), behind a long first tie group (a mark cannot cut a group: marks
    coincide, segments are empty), on one group only, and on columns too short for any mark."""
    rng = np.random.default_rng(n + 5)
    cols = [rng.standard_normal(n), rng.standard_normal(n),                       # continuous
            np.round(rng.standard_normal(n) * 40), np.round(rng.standard_normal(n) * 3),   # tied, many / few groups
            np.where(np.arange(n) < n // 2, 7.0, rng.standard_normal(n) - 9.0),   # the LARGEST value ties half the rows: the walk starts inside a long group
            np.full(n, 1.5),                                                        # one group
            np.repeat(np.arange((n + 39) // 40, dtype=np.float64), 40)[:n],         # groups of 40 rows, in row order
            rng.permutation(np.repeat(np.arange((n + 69) // 70, dtype=np.float64), 70)[:n])]
    X = np.asfortranarray(np.stack(cols, axis=1))
    X[rng.random(X.shape) < 0.08] = np.nan
    X[: n // 3, 1] = np.nan                                                          # a long closed-form tail
    S = X.shape[1]
    pi, pj = np.triu_indices(S, k=1)
    pi, pj = np.concatenate([pi, pj]).astype(np.int32), np.concatenate([pj, pi]).astype(np.int32)
    plan_ctx.debug_set_plan({"split": split, "np": "2"} if split is not None else None)
    for persp in ("global", "local"):
        _check(plan_ctx, X, pi, pj, perspective=persp)


@pytest.mark.parametrize("mode", ["staged", "pinned"])
def test_host_upload_modes(plan_ctx, mode):
    """The host-buffer entry copies the matrix in column chunks that overlap the pre-pass; staged through the library's
    pinned buffers (default) or, with ICIKT_FLAG_HOST_PINNED, straight out of memory the caller has page-locked -- the
    same results, also with a leading dimension larger than n_feat and for a matrix of several chunks.  (There is no
    mode that hands pageable memory of this size to a copy, and none that page-locks it:
    tests/test_gpu_matrix.py::test_transfers_staged_and_caller_pinned.)"""
    from icikendalltau_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(5)
    n, S, ld = 9000, 300, 9016                       # 300 columns x 70 KB: three 8 MB chunks
    alloc = _lib.pinned_empty if mode == "pinned" else np.empty
    buf = alloc((S, ld))                             # row-major (S, ld) == column-major ld x S
    buf[...] = 7.0
    buf[:, :n] = rng.standard_normal((S, n))
    buf[:, :n][rng.random((S, n)) < 0.05] = np.nan
    X = np.asfortranarray(buf[:, :n].T)              # compact copy for the reference run
    plan_ctx.debug_set_plan(None)
    ref = plan_ctx.pairs(X, perspective="global", want_counts=False)
    P = S * (S - 1) // 2
    out = alloc((P, 4))
    rsn = alloc(P, dtype=np.int32)
    rsn[...] = 0
    rc = L.icikt_pairs_f64(plan_ctx._h, buf.ctypes.data, n, S, ld, None, None, 0, 1, 0, 0,
                           _lib.FLAG_HOST_PINNED if mode == "pinned" else 0, out.ctypes.data, None, rsn.ctypes.data)
    assert rc == 0, L.icikt_last_error(plan_ctx._h)
    assert np.array_equal(out, ref[0]) and np.array_equal(rsn, ref[2])
    with pytest.raises(_lib.IciktError, match="unknown key"):
        plan_ctx.debug_set_plan("nope=1")


def test_pairs_complete_on_device(hip_ctx):
    """icikt_pairs_complete_f64 (kt_fast use = "pairwise.complete.obs"): masking + per-pair sort + counting on the
    device, several chunks of pairs, self pairs included; against the oracle on host-masked vectors."""
    O = _oracle()
    rng = np.random.default_rng(73)
    n, S = 20000, 120                                   # 7 260 pairs x 2 columns x 160 KB = 2.3 GB of masked columns
    X = np.asfortranarray(rng.standard_normal((n, S)))
    X[rng.random(X.shape) < 0.08] = np.nan
    X[:, 7] = np.round(X[:, 7] * 3)
    pi, pj = (a.astype(np.int32) for a in np.triu_indices(S, k=0))   # self pairs too, as kt_fast has them
    out, cnt, rsn = hip_ctx.pairs_complete(X, pi, pj, want_counts=True)
    assert out.shape == (len(pi), 4) and np.all(rsn == 0)
    sel = rng.choice(len(pi), size=150, replace=False)
    for p in sel:
        either = np.isnan(X[:, pi[p]]) | np.isnan(X[:, pj[p]])
        x = np.where(either, np.nan, X[:, pi[p]])
        y = np.where(either, np.nan, X[:, pj[p]])
        ref, rcnt, rr = O.ici_kt(x, y, "local")
        assert rr == 0
        assert [int(v) for v in cnt[p]] == [rcnt[k] for k in ("n", "missing", "dis", "ntie", "xtie", "ytie", "x0", "x1", "y0", "y1", "tot")]
        assert float(np.max(np.abs(out[p] - ref))) <= ATOL
    assert np.all(cnt[:, 1] == 0)                        # nothing missing remains after the drop
    # the python front-end reaches it
    from icikendalltau_amd import api
    names = [f"c{i}" for i in range(6)]
    g = api.kt_fast(X[:3000, :6], use="pairwise.complete.obs", colnames=names)
    from tests.oracle_engine import OracleEngine
    o = api.kt_fast(X[:3000, :6], use="pairwise.complete.obs", colnames=names, engine=OracleEngine())
    assert np.nanmax(np.abs(g["tau"].to_numpy() - o["tau"].to_numpy())) <= ATOL
