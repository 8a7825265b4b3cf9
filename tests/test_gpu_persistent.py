"""The long-column pair kernels under -m gpu: pairs on the whole wave (k1_pairs<1, 0> / <2, 0>: one pair, or two one after
the other), a persistent grid whose waves FETCH many tasks in order from per-XCD counters and re-initialise their LDS
state per task (icikt_capi.cpp: launch_pair_tasks; icikt_kernels.hip: the task loop of k1_pairs).  The natural plan of
the small matrices an oracle can afford gives every wave at most one task, so the grid is capped through
icikt_debug_set_plan("gridcap=...") -- every wave then runs ten or more tasks back to back -- plus one natural-plan
case with more tasks than resident waves, the c4b configuration (Bernoulli(0.1) missingness,
a fifth of the columns in the reference's int32-wrap regime, src/kendallc.cpp:112-114) and the column lengths on
either side of the 15-bit packed two-step count (n = 32 768).  (Rounds 1-2 kept open tie groups in a second bitset
`pend`, for long columns in per-wave global slots: the plan key `pend` is still accepted and ignored.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL = 1e-10


def _oracle():
    from oracle import oracle as O
    return O


def _tied_matrix(n, S, seed):
    """Columns that exercise every step kind of the pair kernel: continuous, many small tie groups, a few huge ones,
    left-censored missing values (the closed-form last group), scattered missing values, half-missing."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, S))
    for c in range(S):
        kind = c % 6
        if kind == 1:
            X[:, c] = np.round(X[:, c] * 300)            # ~1 800 distinct values: groups inside a step
        elif kind == 2:
            X[:, c] = np.round(X[:, c] * 3)              # ~20 distinct values: groups of thousands of rows
        elif kind == 3:
            k = n // 25                                  # left-censored: the missing rows are the last tie group
            X[np.argpartition(X[:, c], k)[:k], c] = np.nan
        elif kind == 4:
            X[rng.random(n) < 0.3, c] = np.nan           # scattered
            X[:, c] = np.round(X[:, c] * 40)             # ... and tied
        elif kind == 5:
            X[rng.random(n) < 0.5, c] = np.nan           # half missing, otherwise continuous
    return np.asfortranarray(X)


def _all_pairs_vs_oracle(ctx, X, perspectives=("global", "local")):
    O = _oracle()
    S = X.shape[1]
    pi, pj = (a.astype(np.int32) for a in np.triu_indices(S, k=1))
    for perspective in perspectives:
        out, cnt, rsn = ctx.pairs(X, perspective=perspective)
        ref, rcnt, rrsn = O.ici_pairs(X, pi, pj, perspective)
        assert np.array_equal(rsn, rrsn)
        assert np.array_equal(cnt, rcnt[:, :cnt.shape[1]]), "integer counts differ from the oracle"
        assert np.array_equal(np.isnan(out), np.isnan(ref))
        assert float(np.nanmax(np.abs(out - ref))) <= ATOL


@pytest.mark.parametrize("n", [33000, 50000, 65535])
def test_persistent_waves_run_many_tasks(plan_ctx, n):
    """276 pairs on a grid capped at 8 workgroups = 32 waves: every wave fetches eight or nine tasks in a row from its
    XCD group's counter, after tasks that opened, extended and closed tie groups.  All pairs, both perspectives,
    counts bit-exact.  Then a matrix of another length on the SAME context."""
    plan_ctx.debug_set_plan({"np": 1, "pend": "g", "gridcap": 8})
    X = _tied_matrix(n, 24, seed=n)
    _all_pairs_vs_oracle(plan_ctx, X)
    plan_ctx.debug_set_plan({"np": 1, "pend": "g", "gridcap": 3})      # fewer groups than XCD counters
    X2 = _tied_matrix(41000 if n != 50000 else 36111, 12, seed=n + 1)
    _all_pairs_vs_oracle(plan_ctx, X2)
    # and the first matrix again with the library's own grid: same numbers as under the cap
    plan_ctx.debug_set_plan(None)
    capped = plan_ctx.pairs(X, perspective="global")
    plan_ctx.debug_set_plan({"gridcap": 1})                             # one workgroup, one counter: four waves do it all
    one = plan_ctx.pairs(X, perspective="global")
    assert np.array_equal(capped[0], one[0]) and np.array_equal(capped[1], one[1])


@pytest.mark.parametrize("n", [18400, 33000, 50000])
def test_two_pairs_per_wave_long_columns(plan_ctx, n):
    """k1_pairs<2, 0>: two pairs of a long column per wave, one after the other on the whole wave, sharing the
    streamed column and one 8-byte gather per row; the packed two-step chains of the two pairs take turns (pair 1 counts
    its first singleton step alone).  Tied, left-censored, scattered-missing and half-missing columns -- every general
    step of the pair kernel with two pairs in flight -- on a capped grid (every wave runs several tasks) and on the
    library's own grid; all pairs, both perspectives, counts bit-exact; an odd
    column count leaves tasks with a single pair."""
    X = _tied_matrix(n, 19, seed=n + 7)
    plan_ctx.debug_set_plan({"np": 2, "pend": "g", "gridcap": 4})
    _all_pairs_vs_oracle(plan_ctx, X)
    plan_ctx.debug_set_plan({"np": 2, "pend": "g"})
    two = plan_ctx.pairs(X, perspective="global")
    plan_ctx.debug_set_plan({"np": 1, "pend": "g"})
    one = plan_ctx.pairs(X, perspective="global")
    assert np.array_equal(two[0], one[0], equal_nan=True) and np.array_equal(two[1], one[1])


@pytest.mark.parametrize("n", [20000, 32767, 32768, 32769, 41000, 65535])
def test_long_column_half_layout_loop(plan_ctx, n):
    """The singleton region of k1_pairs<2, 0> in the half layout (one pair per 32-lane half, 32-row sub-steps over the
    two-level counts, a lane exchanging two histogram bins; positions above 2^15 compared halved) against the previous
    whole-wave loop (plan key hyb=0) and against the oracle: mostly continuous columns -- long singleton regions, with
    odd and even step counts in front of the fill group -- plus a tied and a half-missing column whose tie steps carry on
    from the state the loop leaves; a capped grid makes every wave run several tasks."""
    rng = np.random.default_rng(n)
    S = 11
    X = np.asfortranarray(rng.standard_normal((n, S)))
    for c in range(S):
        k = int(rng.integers(1, 400)) + 64 * (c & 1)
        X[np.argsort(X[:, c], kind="stable")[:k], c] = np.nan          # left-censored: the fill group closes the walk
    X[:, 3] = np.round(X[:, 3] * 40)
    X[rng.random(n) < 0.5, 6] = np.nan
    # (half=0: up to 30 656 rows the tied column would otherwise send the matrix to the half-wave kernels)
    plan_ctx.debug_set_plan({"np": 2, "half": 0, "gridcap": 4})
    _all_pairs_vs_oracle(plan_ctx, X)
    plan_ctx.debug_set_plan({"np": 2, "half": 0})
    new = plan_ctx.pairs(X, perspective="local")
    plan_ctx.debug_set_plan({"np": 2, "half": 0, "hyb": 0})
    old = plan_ctx.pairs(X, perspective="local")
    assert np.array_equal(new[0], old[0], equal_nan=True) and np.array_equal(new[1], old[1])
    plan_ctx.debug_set_plan({"np": 2, "half": 0, "hyb": 0, "gridcap": 4})
    _all_pairs_vs_oracle(plan_ctx, X)                                  # the previous loop stays covered


@pytest.mark.parametrize("n", [18337, 22529, 26624, 30656])
def test_half_wave_kernels_11_to_15_words(plan_ctx, n):
    """18 337 .. 30 656 rows: both kernel families fit -- the half-wave kernels with 11, 13 and 15 words per lane (the
    column's tie program cut in two windows of 16 384 positions by the pre-pass) and two pairs on the whole wave; the
    library chooses by the columns' tie groups (matrix_tied).  Each family forced, and the library's own choice, on tied /
    censored / scattered-missing columns: all pairs, both perspectives, against the oracle; the families agree bit for
    bit on the doubles too."""
    X = _tied_matrix(n, 20, seed=n)
    plan_ctx.debug_set_plan({"half": 1})
    _all_pairs_vs_oracle(plan_ctx, X)
    h = plan_ctx.pairs(X, perspective="global")
    plan_ctx.debug_set_plan({"half": 0})
    w = plan_ctx.pairs(X, perspective="global")
    plan_ctx.debug_set_plan(None)
    d = plan_ctx.pairs(X, perspective="global")
    for a, b in ((h, w), (h, d)):
        assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1])
    # ... and continuous columns (the library takes the whole-wave kernels), through the pipelined host path as well
    rng = np.random.default_rng(n)
    Y = np.asfortranarray(rng.standard_normal((n, 130)))
    Y[rng.random(Y.shape) < 0.02] = np.nan
    ref = plan_ctx.pairs(Y, perspective="global")
    plan_ctx.debug_set_plan({"half": 1})
    got = plan_ctx.pairs(Y, perspective="global")
    assert np.array_equal(ref[0], got[0], equal_nan=True) and np.array_equal(ref[1], got[1])


def test_persistent_grid_natural_plan(hip_ctx):
    """36 000 x 128 = 8 128 tasks against the ~5 000 waves the chip holds at that length: the library's own plan
    makes waves take a second task.  Oracle on 300 sampled pairs, count identities on all."""
    from bench import make_matrix
    from tests.test_gpu_configs import _properties, _sample_check
    n, S, n_na = 36000, 128, 720
    X = make_matrix(n, S, n_na, seed=36)
    pi, pj = (a.astype(np.int32) for a in np.triu_indices(S, k=1))
    for perspective in ("global", "local"):
        out, cnt, rsn = hip_ctx.pairs(X, perspective=perspective)
        _properties(n, n_na, out, cnt, rsn, perspective)
        _sample_check(X, pi, pj, out, cnt, rsn, perspective, 300, seed=360)


def test_config_c4b(hip_ctx):
    """c4b (BASELINE.md section 3 / SURVEY.md section 8(d)): the c4 matrix with Bernoulli(0.1) missingness instead of
    exactly 1 000 missing rows per column.  About a fifth of the columns then have a fill group of >= 1 024 rows, where
    the reference's int32 t(t-1)(2t+5) wraps (src/kendallc.cpp:112-114): the default mode must reproduce the wrap, the
    ICIKT_FLAG_EXACT_INT64 mode must not.  Full size (523 776 pairs), 2 000 sampled pairs per mode against the
    oracle's two modes."""
    O = _oracle()
    n, S = 10000, 1024
    rng = np.random.default_rng(4)
    X = np.asfortranarray(rng.standard_normal((n, S)))
    X[rng.random((n, S)) < 0.1] = np.nan
    nna = np.isnan(X).sum(axis=0)
    wrap_cols = int((nna >= 1024).sum())
    assert 100 < wrap_cols < 400, wrap_cols               # ~21 % of the columns are in the wrap regime
    pi, pj = (a.astype(np.int32) for a in np.triu_indices(S, k=1))
    sel = np.random.default_rng(404).choice(len(pi), size=2000, replace=False)
    in_wrap = (nna[pi[sel]] >= 1024) | (nna[pj[sel]] >= 1024)
    assert in_wrap.sum() > 400
    res = {}
    for flags in (0, 1):
        out, cnt, rsn = hip_ctx.pairs(X, perspective="global", flags=flags)
        assert out.shape == (len(pi), 4) and np.all(rsn == 0) and not np.isnan(out).any()
        ref, rcnt, rrsn = O.ici_pairs(X, pi[sel], pj[sel], "global", int32_compat=(flags == 0))
        assert np.all(rrsn == 0)
        assert np.array_equal(cnt[sel], rcnt[:, :cnt.shape[1]]), "integer counts differ from the oracle"
        assert float(np.max(np.abs(out[sel] - ref))) <= ATOL
        res[flags] = (out, cnt)
    # tau, tau_max and completeness do not depend on the mode (the wrap reaches x1 / y1 only, i.e. the variance)
    assert np.array_equal(res[0][0][:, [0, 2, 3]], res[1][0][:, [0, 2, 3]])
    differs = np.any(res[0][1] != res[1][1], axis=1)
    col_wrap = nna >= 1024
    assert np.array_equal(differs, col_wrap[pi] | col_wrap[pj])      # exactly the pairs that touch a wrapping column
    # local on a column block that holds wrapping columns: K2 re-derives the shrunken fill group in the same arithmetic
    sub = np.concatenate([np.nonzero(col_wrap)[0][:12], np.nonzero(~col_wrap)[0][:12]])
    Xs = np.asfortranarray(X[:, sub])
    qi, qj = (a.astype(np.int32) for a in np.triu_indices(len(sub), k=1))
    for flags in (0, 1):
        out, cnt, rsn = hip_ctx.pairs(Xs, perspective="local", flags=flags)
        ref, rcnt, rrsn = O.ici_pairs(Xs, qi, qj, "local", int32_compat=(flags == 0))
        assert np.array_equal(rsn, rrsn) and np.array_equal(cnt, rcnt[:, :cnt.shape[1]])
        assert float(np.max(np.abs(out - ref))) <= ATOL


@pytest.mark.parametrize("n", [32704, 32767, 32768, 32769])
def test_packed_two_step_count_at_its_limit(plan_ctx, n):
    """One pair per wave counts the in-step pairs of two hot steps in one packed chain while positions fit 15 bits
    (n <= 32 768: wave_allpairs_packed2, A = 0x7FFF - q, B = lo).  Columns without ties or missing values, so that q
    and lo reach n - 1 (A = 0 and B = 0x7FFF at n = 32 768) and the hot loop covers the whole column -- 511 steps at
    32 767 (an odd step out), 512 at 32 768, the unpacked compare at 32 769; plus columns whose hot region ends early
    (missing rows) at an odd and an even step count.  pend in LDS and in global memory."""
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, 6))
    X[:, 1] = -X[:, 0] + 1e-3 * X[:, 1]                     # strongly discordant with column 0: many flags set
    X[:, 2] = np.arange(n, dtype=np.float64)                # sorted: q = row
    X[np.argpartition(X[:, 3], 64 * 3 + 5)[:64 * 3 + 5], 3] = np.nan     # hot region n - 197 rows
    X[np.argpartition(X[:, 4], 64 * 4 + 9)[:64 * 4 + 9], 4] = np.nan     # hot region n - 265 rows
    X[:, 5] = np.round(X[:, 5] * 1000)                      # tie groups inside steps: the packed loop is left early
    X = np.asfortranarray(X)
    for plan in ({"np": 1, "pend": "l"}, {"np": 1, "pend": "g", "gridcap": 2}):
        plan_ctx.debug_set_plan(plan)
        _all_pairs_vs_oracle(plan_ctx, X)


@pytest.mark.parametrize("n", [32770, 40001, 65535])
def test_halved_positions_in_the_packed_chain(plan_ctx, n):
    """Above 32 768 rows the one-pair kernels pack two hot steps with HALVED positions (q >> 1 against (lo + 1) >> 1),
    exact as long as no tie group of the gathered column starts at an odd position (K0: COL_ODD_TIE).  Gathered columns:
    continuous (flag clear: halved chain), continuous with missing rows (fill group at position 0: clear), tie groups of
    two rows that ALL start at even positions (clear: the chain meets q == lo with lo even), the same shifted by one row
    (every group starts at an odd position: flag set, the unpacked compare), strongly correlated columns (rows adjacent
    in both orders share steps: q = lo - 1 and q = lo + 1 inside a step)."""
    rng = np.random.default_rng(n)
    base = rng.standard_normal(n)
    X = np.empty((n, 7))
    X[:, 0] = base
    X[:, 1] = base + 1e-4 * rng.standard_normal(n)               # near-identical order: neighbours in A are neighbours in B
    X[:, 2] = -base + 1e-4 * rng.standard_normal(n)              # ... and reversed
    r = np.argsort(np.argsort(rng.standard_normal(n)))           # ranks 0 .. n-1
    X[:, 3] = r // 2                                             # pairs (0,1), (2,3), ...: every tie group starts even
    X[:, 4] = (r + 1) // 2                                       # (1,2), (3,4), ...: every tie group starts odd
    X[:, 5] = rng.standard_normal(n)
    X[np.argpartition(X[:, 5], 777)[:777], 5] = np.nan           # left-censored: the fill group starts at 0
    X[:, 6] = np.where(rng.random(n) < 0.4, np.nan, base + 0.5 * rng.standard_normal(n))
    X = np.asfortranarray(X)
    for plan in ({"np": 1, "pend": "g"}, {"np": 1, "pend": "g", "gridcap": 2}, {"np": 1, "pend": "l"}):
        plan_ctx.debug_set_plan(plan)
        _all_pairs_vs_oracle(plan_ctx, X)
        # both orientations of every pair: each column is the gathered one somewhere
        S = X.shape[1]
        pi, pj = (a.astype(np.int32) for a in np.triu_indices(S, k=1))
        out, cnt, rsn = plan_ctx.pairs(X, pj, pi, "global")
        ref, rcnt, rrsn = _oracle().ici_pairs(X, pj, pi, "global")
        assert np.array_equal(cnt, rcnt[:, :cnt.shape[1]]) and float(np.nanmax(np.abs(out - ref))) <= ATOL
