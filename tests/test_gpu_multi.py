"""icikt_pairs_multi_f64 -- several GPUs behind one C call (include/icikt.h; replaces the reference's worker
fan-out, R/kendalltau.R:158,250-255).  On the one-GPU test box:
  * n_gpu = 1 with exchange "rccl": the real RCCL calls (ncclCommInitAll, ncclAllGather in place, ncclGather) on a
    one-rank communicator;
  * devices [0, 0] / [0, 0, 0] with exchange "copy": every rank thread, the column-sharded pre-pass, the exchange
    of order + meta, the local rebuild, the block partition of the pair list and the gather, with device copies in
    place of the collectives (RCCL refuses a device listed twice).
Results must be identical to the single-device entry: same kernels, same integer counts, same doubles."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _matrix(n, S, seed, na=0.08):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, S))
    X[rng.random(X.shape) < na] = np.nan
    X[:, 1] = np.round(X[:, 1] * 5)
    X[:, S - 1] = np.round(X[:, S - 1] * 100)
    return np.asfortranarray(X)


def _same(a, b):
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)


@pytest.fixture(scope="module")
def single(hip_ctx):
    return hip_ctx


def test_one_rank_rccl_matches_single_device(single):
    from icikendalltau_amd import _lib
    m = _lib.MultiContext([0], exchange="rccl")
    try:
        assert m.uses_rccl
        X = _matrix(3000, 40, 1)
        for persp in ("global", "local"):
            _same(m.pairs(X, perspective=persp), single.pairs(X, perspective=persp))
        pi = np.array([0, 0, 5, 7, 7, 39, 3] * 20, dtype=np.int32)
        pj = np.array([1, 2, 6, 8, 9, 0, 3] * 20, dtype=np.int32)
        _same(m.pairs(X, pi, pj, "global"), single.pairs(X, pi, pj, "global"))
        ph = m.phase_ms()
        assert set(ph) == {"prepare", "exchange", "pairs", "gather"} and all(v >= 0 for v in ph.values())
    finally:
        m.close()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0], [0, 0, 0, 0, 0]])
def test_rank_threads_with_copy_exchange(single, devices):
    from icikendalltau_amd import _lib
    from oracle import oracle as O
    m = _lib.MultiContext(devices, exchange="auto")   # duplicates -> "copy"
    try:
        assert not m.uses_rccl
        # S = 37: an odd number of columns, a ragged last column shard and a ragged last pair block
        X = _matrix(2600, 37, len(devices))
        for persp in ("global", "local"):
            got = m.pairs(X, perspective=persp)
            _same(got, single.pairs(X, perspective=persp))
        pi, pj = (a.astype(np.int32) for a in np.triu_indices(37, k=1))
        ref, rcnt, rrsn = O.ici_pairs(X, pi, pj, "local")
        assert np.array_equal(got[2], rrsn) and np.array_equal(got[1], rcnt[:, :got[1].shape[1]])
        assert float(np.nanmax(np.abs(got[0] - ref))) <= 1e-10
        # an explicit list in a caller's order (include_only-style), without counts
        rng = np.random.default_rng(7)
        sel = rng.permutation(len(pi))[:500]
        a = m.pairs(X, pi[sel], pj[sel], "global", want_counts=False)
        b = single.pairs(X, pi[sel], pj[sel], "global", want_counts=False)
        assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[2], b[2])
        # timing flag: phases end with a synchronisation
        m.pairs(X, perspective="global", flags=_lib.FLAG_TIMING)
        assert m.phase_ms()["pairs"] > 0
        # repeated calls reuse every buffer
        _same(m.pairs(X, perspective="global"), single.pairs(X, perspective="global"))
    finally:
        m.close()


def test_multi_longer_columns_and_degenerate(single):
    from icikendalltau_amd import _lib
    m = _lib.MultiContext([0, 0], exchange="copy")
    try:
        X = _matrix(12000, 24, 3, na=0.1)       # one pair per wave plan
        X[:, 4] = np.nan                         # an all-missing column: reason 1 pairs in both blocks
        X[:, 9] = 2.5                            # constant: reason 3
        _same(m.pairs(X, perspective="global"), single.pairs(X, perspective="global"))
        # too little work to split: the single-device path inside the same entry
        Xs = _matrix(300, 3, 4)
        _same(m.pairs(Xs, perspective="local"), single.pairs(Xs, perspective="local"))
        o, c, r = m.pairs(np.empty((0, 30)), perspective="global")
        assert o.shape == (435, 4) and np.all(r == 1)
    finally:
        m.close()


def test_multi_long_tied_columns_count_mode(single):
    """Columns too long for the half-wave kernels, tied: the rows' tie-group indices that the whole-wave kernels' count mode
    reads are rebuilt on the receiving rank (the expand kernel) -- two ranks against one device, and the oracle on a few pairs."""
    from icikendalltau_amd import _lib
    from oracle import oracle as O
    m = _lib.MultiContext([0, 0], exchange="copy")
    try:
        rng = np.random.default_rng(77)
        n, S = 36000, 18                                 # 153 pairs: enough for the entry to use both ranks
        X = np.round(rng.standard_normal((n, S)) * np.array([2, 30, 60, 150, 60, 30, 2, 500, 60, 60, 30, 150, 60, 30, 150, 60, 2, 30]))
        X[rng.random(X.shape) < 0.05] = np.nan
        X = np.asfortranarray(X)
        got = m.pairs(X, perspective="global")
        assert m.ranks_used == 2
        _same(got, single.pairs(X, perspective="global"))
        pi = np.array([0, 2, 3, 7, 12], dtype=np.int32)
        pj = np.array([2, 4, 11, 8, 15], dtype=np.int32)
        ref, rcnt, rrsn = O.ici_pairs(X, pi, pj, "global")
        idx = [int(i * S - i * (i + 1) // 2 + (j - i - 1)) for i, j in zip(pi, pj)]
        assert np.array_equal(got[1][idx], rcnt[:, :got[1].shape[1]]) and np.allclose(got[0][idx], ref, rtol=0, atol=1e-10, equal_nan=True)
    finally:
        m.close()


def test_multi_error_contract():
    import ctypes
    from icikendalltau_amd import _lib
    L = _lib.lib()
    h = ctypes.c_void_p()
    bad = (ctypes.c_int * 2)(0, 99)
    assert L.icikt_multi_create(bad, 2, 0, ctypes.byref(h)) == -1 and not h.value        # no such device
    dup = (ctypes.c_int * 2)(0, 0)
    assert L.icikt_multi_create(dup, 2, _lib.EXCHANGE["rccl"], ctypes.byref(h)) == -1      # RCCL refuses duplicates
    assert L.icikt_multi_create(dup, 0, 0, ctypes.byref(h)) == -1
    m = _lib.MultiContext([0, 0])
    try:
        X = _matrix(500, 20, 5)
        with pytest.raises(_lib.IciktError, match="column index out of range"):
            m.pairs(X, np.array([0] * 200, np.int32), np.array([20] * 200, np.int32))
        with pytest.raises(_lib.IciktError, match="262144"):
            m.pairs(np.zeros((262145, 4)))
        # wide columns (65 535 < n <= 262 144) run on the first device alone, through the same entry
        Xw = _matrix(66000, 4, 6)
        from icikendalltau_amd import _lib as L2
        c1 = L2.Context(0)
        try:
            _same(m.pairs(Xw, perspective="global"), c1.pairs(Xw, perspective="global"))
        finally:
            c1.close()
        # and still usable afterwards
        o, _c, r = m.pairs(X, perspective="global")
        assert np.all(r == 0) and not np.isnan(o).any()
    finally:
        m.close()


def test_api_n_gpu_argument(single):
    """api.ici_kendalltau(n_gpu = / devices =): the Python mirror of the R wrapper's n_gpu -- `core` numbers the GPUs'
    pair blocks (ceiling(P / n_gpu) consecutive pairs, R/kendalltau.R:250-255), results equal the one-GPU call."""
    from icikendalltau_amd import api
    X = _matrix(1500, 21, 9)
    names = [f"s{i}" for i in range(21)]
    one = api.ici_kendalltau(X, colnames=names, return_matrix=False)["cor"]
    two = api.ici_kendalltau(X, colnames=names, return_matrix=False, devices=[0, 0, 0])["cor"]
    for col in ("s1", "s2"):
        assert list(one[col]) == list(two[col])
    for col in ("raw", "pvalue", "taumax", "completeness", "cor"):
        assert np.array_equal(one[col].to_numpy(), two[col].to_numpy(), equal_nan=True)
    assert list(two["core"]) == [1] * 70 + [2] * 70 + [3] * 70 + [0] * 21     # 210 pairs over 3 GPUs
    assert list(one["core"]) == [1] * 210 + [0] * 21
    m = api.ici_kendalltau(X, colnames=names, n_gpu=1, perspective="local")
    assert m["cor"].shape == (21, 21)


@pytest.mark.parametrize("devices,exchange", [([0, 0, 0, 0], "copy"), ([0, 0, 0], "copy"), ([0], "rccl")])
def test_cost_weighted_blocks(single, devices, exchange):
    """ICIKT_FLAG_BALANCE_COST: the pair list is cut into consecutive blocks of equal COST -- the pre-pass leaves with
    every column what streaming it costs the pair kernel (hot / MIXED / GROUP steps) -- instead of the reference's
    ceiling(n_todo / ncore) pairs per chunk (R/kendalltau.R:250-255).  The second half of this matrix is tied (its
    columns cost about twice the continuous ones), so the cut moves; the results, their order, counts and reasons are
    those of one device, for all pairs of the triangle, an explicit list and the full-matrix entry."""
    from icikendalltau_amd import _lib
    rng = np.random.default_rng(17)
    n, S = 6000, 48
    X = rng.standard_normal((n, S))
    X[:, S // 2:] = np.round(X[:, S // 2:] * 300)          # ~2 000 distinct values: MIXED steps when streamed
    X[rng.random(X.shape) < 0.04] = np.nan
    X = np.asfortranarray(X)
    P = S * (S - 1) // 2
    G = len(devices)
    m = _lib.MultiContext(devices, exchange=exchange)
    try:
        ref = single.pairs(X, perspective="global")
        eq = m.pairs(X, perspective="global")
        b_eq = m.block_bounds()
        _same(eq, ref)
        got = m.pairs(X, perspective="global", flags=_lib.FLAG_BALANCE_COST)
        b = m.block_bounds()
        _same(got, ref)
        assert m.ranks_used == G and len(b) == G + 1 and b[0] == 0 and b[-1] == P and all(x <= y for x, y in zip(b, b[1:]))
        each = -(-P // G)
        assert b_eq == [min(P, r * each) for r in range(G + 1)]
        assert all(y - x <= 2 * each for x, y in zip(b, b[1:]))
        if G > 1:
            # pairs (i, j) stream column j: the late rows of combn order hold tied columns only, the early ones both kinds --
            # equal-cost blocks are therefore not equal-count blocks
            assert b != b_eq
            pi, pj = np.triu_indices(S, k=1)
            cost = np.where(pj >= S // 2, 2.0, 1.0)          # a model of it: tied columns about twice the cost
            share = [cost[x:y].sum() / cost.sum() for x, y in zip(b, b[1:])]
            share_eq = [cost[x:y].sum() / cost.sum() for x, y in zip(b_eq, b_eq[1:])]
            assert max(share) - min(share) < max(share_eq) - min(share_eq)
        # an explicit list (unsorted, both orientations) and "local"
        sel = rng.permutation(P)[:700]
        iu, ju = np.triu_indices(S, k=1)
        qi = np.where(rng.random(700) < 0.5, iu[sel], ju[sel]).astype(np.int32)
        qj = np.where(qi == iu[sel], ju[sel], iu[sel]).astype(np.int32)
        _same(m.pairs(X, qi, qj, "local", flags=_lib.FLAG_BALANCE_COST), single.pairs(X, qi, qj, "local"))
        # the full-matrix entry: the first device assembles from blocks of different lengths
        o5, k5, r5 = m.matrix(X, (float("nan"),), flags=_lib.FLAG_BALANCE_COST)
        e5, ek5, er5 = single.matrix(X, (float("nan"),))
        assert np.array_equal(o5, e5, equal_nan=True) and np.array_equal(k5, ek5) and np.array_equal(r5, er5)
    finally:
        m.close()
