"""The full-matrix entry icikt_matrix_f64 / icikt_matrix_multi_f64 under -m gpu: setup_missing_matrix (R/utils.R:1-23)
applied inside the pre-pass, the pair kernels, and scale_and_reshape (R/kendalltau.R:357-421) on the device, against
 (a) the host assembly of the same pair results (api.ici_kendalltau's pair-list path: bit for bit), and
 (b) the whole front-end run on the oracle engine (1e-10, the north_star tolerance).
Also: the staged transfers (the library's own pinned buffers) and caller-pinned memory (ICIKT_FLAG_HOST_PINNED), and the
per-rank figures of the multi-device driver."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL = 1e-10
KEYS = ("cor", "raw", "pvalue", "taumax", "completeness")


def _data(n, S, seed, with_nan=True):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, S))
    X[rng.random((n, S)) < 0.07] = 0.0                    # the default global_na excludes zeros ...
    X[rng.random((n, S)) < 0.01] = np.inf                 # ... infinities (either sign) ...
    X[rng.random((n, S)) < 0.01] = -np.inf
    if with_nan:
        X[rng.random((n, S)) < 0.03] = np.nan             # ... and NA
    X[:, 2] = np.round(X[:, 2] * 3)                       # ties (and more zeros)
    X[:, 5] = np.where(rng.random(n) < 0.6, 0.0, X[:, 5])  # a mostly excluded column
    return np.asfortranarray(X)


def _host_path(X, names, **kw):
    """The pair-list path of the front-end on the HIP engine: host masking, icikt_pairs_f64, host scale_and_reshape."""
    from icikendalltau_amd import api

    class PairsOnly(api.HipEngine):       # no .matrix attribute -> api takes the pair-list route
        def __getattribute__(self, name):
            if name == "matrix":
                raise AttributeError(name)
            return super().__getattribute__(name)

    return api.ici_kendalltau(X, colnames=names, engine=PairsOnly(), **kw)


def _same(a, b, exact):
    for k in KEYS:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        assert x.shape == y.shape
        assert np.array_equal(np.isnan(x), np.isnan(y)), k
        if exact:
            assert np.array_equal(x, y, equal_nan=True), (k, np.nanmax(np.abs(x - y)))
        else:
            d = np.nanmax(np.abs(x - y)) if np.any(~np.isnan(x)) else 0.0
            assert d <= ATOL, (k, d)
    assert np.array_equal(np.asarray(a["keep"]), np.asarray(b["keep"]))


@pytest.mark.parametrize("kw", [
    dict(),
    dict(perspective="local"),
    dict(scale_max=False),
    dict(diag_good=False),
    dict(global_na=(0,)),                       # NaN in the data stays missing for ici_kt but is NOT excluded: n_good differs
    dict(global_na=None),
    dict(global_na=(float("nan"), 0, -1.0, 2.0)),
    dict(alternative="greater", continuity=True),
    dict(include_only=["s1", "s4", "nope"]),
    dict(include_only=[["s0", "s3", "s7"], ["s2", "s9", "s1"]], diag_good=False),
])
def test_matrix_entry_equals_host_assembly_and_oracle(hip_ctx, kw):
    from icikendalltau_amd import api
    from tests.oracle_engine import OracleEngine
    n, S = 1200, 12
    X = _data(n, S, seed=len(repr(kw)))
    X[X == 2.0] = 2.5
    X[5, 3], X[7, 4] = -1.0, 2.0                           # values the fourth rule excludes
    names = [f"s{i}" for i in range(S)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fast = api.ici_kendalltau(X, colnames=names, engine=api.HipEngine(), **kw)       # icikt_matrix_f64
        host = _host_path(X, names, **kw)
        orac = api.ici_kendalltau(X, colnames=names, engine=OracleEngine(), **kw)
    _same(fast, host, exact=True)
    _same(fast, orac, exact=False)
    assert fast["run_time"] > 0


@pytest.mark.parametrize("n_vals", [7, 32, 33, 50])
def test_matrix_entry_many_global_na_values(hip_ctx, n_vals):
    """The reference loops over ANY number of global_na values (R/utils.R:16-20).  The device rule of icikt_matrix_f64
    holds _lib.MASK_VALS = 32 distinct finite ones (round 3: six, and a seventh made ici_kendalltau() raise); a longer
    list is masked by the front-end (setup_missing_matrix) and takes the pair-list route.  Both equal the oracle
    front-end, and the C entry refuses the 33rd value with ICIKT_E_INVALID instead of ignoring it."""
    from icikendalltau_amd import _lib, api
    from tests.oracle_engine import OracleEngine
    rng = np.random.default_rng(n_vals)
    n, S = 900, 8
    X = np.asfortranarray(rng.integers(0, 80, size=(n, S)).astype(float))
    X[rng.random((n, S)) < 0.02] = np.nan
    gna = [float("nan"), float("inf")] + [float(v) for v in range(n_vals)] + [3.0, 3.0]     # duplicates do not count
    names = [f"s{i}" for i in range(S)]
    fast = api.ici_kendalltau(X, global_na=gna, colnames=names, engine=api.HipEngine())
    orac = api.ici_kendalltau(X, global_na=gna, colnames=names, engine=OracleEngine())
    _same(fast, orac, exact=False)
    assert np.array_equal(np.asarray(fast["keep"]).T, ~(np.isnan(X) | (X < n_vals)))
    if n_vals > _lib.MASK_VALS:
        with pytest.raises(_lib.IciktError, match="more than 32 distinct finite values"):
            hip_ctx.matrix(X, gna)
    else:
        out5, keep, _rc = hip_ctx.matrix(X, gna)
        assert np.array_equal(out5[1], np.asarray(fast["raw"]), equal_nan=True)


def test_matrix_entry_degenerate_columns_and_warnings(hip_ctx):
    """All-excluded and constant columns: NA cells, the reference's warning once per offending pair, max(taumax)
    with na.rm = TRUE over what is left."""
    from icikendalltau_amd import api
    n, S = 300, 7
    rng = np.random.default_rng(2)
    X = rng.standard_normal((n, S)) + 3.0
    X[:, 1] = 0.0                                          # excluded everywhere: silent NA (reason 1)
    X[:, 2] = 4.25                                         # constant: "single unique value" (reason 3)
    X[rng.random(n) < 0.2, 4] = np.nan
    names = [f"s{i}" for i in range(S)]
    with warnings.catch_warnings(record=True) as wf:
        warnings.simplefilter("always")
        fast = api.ici_kendalltau(X, colnames=names, engine=api.HipEngine())
    with warnings.catch_warnings(record=True) as wh:
        warnings.simplefilter("always")
        host = _host_path(X, names)
    mf = sorted(str(w.message) for w in wf if "NA returned" in str(w.message))
    mh = sorted(str(w.message) for w in wh if "NA returned" in str(w.message))
    assert len(mf) == S - 2 and mf == mh, (mf, [str(w.message) for w in wh])   # column 2 against every column but itself and the all-NA one
    _same(fast, host, exact=True)
    assert np.isnan(np.asarray(fast["raw"])[1, 3]) and np.asarray(fast["raw"])[1, 1] == 0.0   # n_good = 0 on the diagonal


def test_matrix_entry_c3_shape(hip_ctx):
    """c3's shape through the one-call entry: every cell equal to the host assembly of icikt_pairs_f64's results."""
    from bench import make_matrix
    from icikendalltau_amd import api
    n, S = 10000, 256
    X = make_matrix(n, S, 500, seed=3)
    names = [f"s{i}" for i in range(S)]
    fast = api.ici_kendalltau(X, colnames=names, engine=api.HipEngine())
    host = _host_path(X, names)
    _same(fast, host, exact=True)
    assert np.asarray(fast["keep"]).shape == (S, n)
    assert float(np.abs(np.asarray(fast["cor"])).max()) <= 1.0 + 1e-12


def test_matrix_c_abi_contract(hip_ctx):
    import ctypes
    from icikendalltau_amd import _lib
    L = _lib.lib()
    X = np.asfortranarray(np.random.default_rng(0).standard_normal((50, 4)))
    out5 = np.empty((5, 4, 4))
    gna = np.array([np.nan, 0.0])
    args = lambda **o: [hip_ctx._h, o.get("X", X.ctypes.data), 50, 4, o.get("ld", 50), o.get("gna", gna.ctypes.data),
                        o.get("ngna", 2), None, None, 0, o.get("persp", 1), 0, 0, 0, 1, 1, o.get("out", out5.ctypes.data),
                        None, None]
    assert L.icikt_matrix_f64(*args()) == 0
    assert L.icikt_matrix_f64(*args(X=None)) == -1 and b"null matrix" in L.icikt_last_error(hip_ctx._h)
    assert L.icikt_matrix_f64(*args(out=None)) == -1
    assert L.icikt_matrix_f64(*args(ld=49)) == -1
    assert L.icikt_matrix_f64(*args(persp=7)) == -1
    many = np.arange(1.0, 35.0)                          # 34 distinct finite values: one more than the device rule holds
    assert L.icikt_matrix_f64(*args(gna=many.ctypes.data, ngna=32)) == 0
    assert L.icikt_matrix_f64(*args(gna=many.ctypes.data, ngna=34)) == -1 and b"global_na" in L.icikt_last_error(hip_ctx._h)
    assert L.icikt_matrix_f64(*args(gna=None, ngna=0)) == 0


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_matrix_multi_equals_single(hip_ctx, devices):
    """icikt_matrix_multi_f64: rank threads apply the exclusion rule to their own columns, return their rows of keep,
    the first device assembles.  One rank = real RCCL on a one-rank communicator; a device listed more than once =
    device copies in place of the collectives (ragged column shards and pair blocks)."""
    from icikendalltau_amd import _lib
    n, S = 2000, 26
    X = _data(n, S, seed=77)
    gna = (float("nan"), float("inf"), 0.0)
    ref5, refk, refc = hip_ctx.matrix(X, gna)
    m = _lib.MultiContext(devices)
    try:
        out5, keep, rc = m.matrix(X, gna)
        assert m.ranks_used == len(devices)
        assert np.array_equal(out5, ref5, equal_nan=True) and np.array_equal(keep, refk) and np.array_equal(rc, refc)
        per_rank = m.rank_phase_ms()
        assert len(per_rank) == len(devices) and all(set(p) == {"prepare", "exchange", "pairs", "gather", "wait"} for p in per_rank)
        mx = m.phase_ms()
        assert all(abs(mx[k] - max(p[k] for p in per_rank)) < 1e-9 for k in mx)
        # an explicit list (self pairs, no diagonal rows)
        iu, ju = np.triu_indices(S, k=0)
        o2, k2, _ = m.matrix(X, gna, iu.astype(np.int32), ju.astype(np.int32), diag_good=False)
        r2, rk2, _ = hip_ctx.matrix(X, gna, iu.astype(np.int32), ju.astype(np.int32), diag_good=False)
        assert np.array_equal(o2, r2, equal_nan=True) and np.array_equal(k2, rk2)
        # too small to split: the first device alone, and the caller can tell
        o3, _k, _c = m.matrix(X[:, :3], gna)
        assert m.ranks_used == (1 if len(devices) > 1 else len(devices))
        r3, _k, _c = hip_ctx.matrix(np.asfortranarray(X[:, :3]), gna)
        assert np.array_equal(o3, r3, equal_nan=True)
    finally:
        m.close()


def test_transfers_staged_and_caller_pinned(plan_ctx):
    """The library never page-locks caller memory (no hipHostRegister in it since round 4): the matrix (21 MB), the pair
    and task lists and the result arrays cross its own pinned buffers (staging double buffer, bounce pieces, one pinned
    buffer per result array).  A caller that HAS page-locked its matrix and result arrays says so with
    ICIKT_FLAG_HOST_PINNED and they are copied from / into directly: hipHostMalloc memory (torch's pinned allocator) and
    a numpy array the caller registered itself (and can still unregister afterwards).  Same bytes on every route; both
    host entries and the multi-device one."""
    import ctypes
    from icikendalltau_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(5)
    n, S = 9000, 300
    X = np.asfortranarray(rng.standard_normal((n, S)))
    X[rng.random((n, S)) < 0.05] = np.nan
    plan_ctx.debug_set_plan({"pipe": 0})
    ref = plan_ctx.pairs(X, perspective="global")
    ref5 = plan_ctx.matrix(X, (float("nan"),))
    plan_ctx.debug_set_plan(None)
    got = plan_ctx.pairs(X, perspective="global")
    got5 = plan_ctx.matrix(X, (float("nan"),))
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref, got))
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref5, got5))
    # (1) everything in hipHostMalloc memory
    Xp = _lib.pinned_empty((n, S), order="F")
    Xp[...] = X
    assert Xp.flags.f_contiguous
    for plan in ({"pipe": 0}, {"pipe": 1}):
        plan_ctx.debug_set_plan(plan)
        got = plan_ctx.pairs(Xp, perspective="global", flags=_lib.FLAG_HOST_PINNED)
        assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref, got)), plan
    out5 = _lib.pinned_empty((5, S, S))
    keep = _lib.pinned_empty((S, n), dtype=np.uint8)
    rc5 = np.zeros(5, dtype=np.int64)
    gna = np.array([np.nan])
    rc = L.icikt_matrix_f64(plan_ctx._h, Xp.ctypes.data, n, S, n, gna.ctypes.data, 1, None, None, 0, 1, 0, 0,
                            _lib.FLAG_HOST_PINNED, 1, 1, out5.ctypes.data, keep.ctypes.data, rc5.ctypes.data)
    assert rc == 0, L.icikt_last_error(plan_ctx._h)
    assert np.array_equal(out5, ref5[0], equal_nan=True) and np.array_equal(keep.view(np.bool_), ref5[1])
    # (2) a matrix the caller registered itself; results staged (no flag: the library does not look at the memory)
    import torch
    cudart = torch.cuda.cudart()
    plan_ctx.debug_set_plan(None)
    assert int(cudart.cudaHostRegister(X.ctypes.data, X.nbytes, 0)) == 0
    try:
        got = plan_ctx.pairs(X, perspective="global")
        P = S * (S - 1) // 2
        out = _lib.pinned_empty((P, 4))
        rsn = _lib.pinned_empty(P, dtype=np.int32)
        rc = L.icikt_pairs_f64(plan_ctx._h, X.ctypes.data, n, S, n, None, None, 0, 1, 0, 0, _lib.FLAG_HOST_PINNED,
                               out.ctypes.data, None, rsn.ctypes.data)
        assert rc == 0, L.icikt_last_error(plan_ctx._h)
    finally:
        rc_unreg = int(cudart.cudaHostUnregister(X.ctypes.data))
    assert rc_unreg == 0
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref, got))
    assert np.array_equal(out, ref[0], equal_nan=True) and np.array_equal(rsn, ref[2])
    # (3) the single-pair entry copies its two vectors itself: the flag must not reach that copy
    o1, c1, r1 = plan_ctx.pair(X[:, 0], X[:, 1], "global", flags=_lib.FLAG_HOST_PINNED)
    assert np.array_equal(o1, ref[0][0], equal_nan=True)
    m = _lib.MultiContext([0, 0])
    try:
        got = m.pairs(X, perspective="global")
        assert m.ranks_used == 2
        assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref, got))
        got = m.pairs(Xp, perspective="global", flags=_lib.FLAG_HOST_PINNED)
        assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref, got))
    finally:
        m.close()
    with pytest.raises(_lib.IciktError, match="unknown key"):
        plan_ctx.debug_set_plan("h2d=register")      # removed with the mode
    with pytest.raises(_lib.IciktError, match="unknown key"):
        plan_ctx.debug_set_plan("regfail=1")


def test_pipelined_host_path(plan_ctx):
    """The host entries pipeline a matrix of several 8 MB chunks: the pre-pass of a chunk runs as it arrives and the pair
    kernel is launched once per chunk, over the tasks whose last column lies in it (icikt::host::upload_prepare_pairs).
    Same bytes out as the one-launch path: all pairs, an unsorted explicit list with both orientations and self pairs,
    the matrix entry, and on long columns (persistent pair kernel)."""
    rng = np.random.default_rng(12)
    n, S = 9000, 300                                     # three chunks of 116 columns
    X = np.asfortranarray(rng.standard_normal((n, S)))
    X[rng.random((n, S)) < 0.05] = np.nan
    X[:, 7] = np.round(X[:, 7] * 50)
    iu, ju = np.triu_indices(S, k=0)
    sel = rng.permutation(len(iu))[:5000]
    qi = np.where(rng.random(len(sel)) < 0.5, iu[sel], ju[sel]).astype(np.int32)
    qj = np.where(qi == iu[sel], ju[sel], iu[sel]).astype(np.int32)
    plan_ctx.debug_set_plan({"pipe": 0})
    ref_all = plan_ctx.pairs(X, perspective="global")
    ref_lst = plan_ctx.pairs(X, qi, qj, "local")
    ref_mat = plan_ctx.matrix(X, (float("nan"),))
    for plan in ({"pipe": 1}, {"pipe": 1, "np": 1}):
        plan_ctx.debug_set_plan(plan)
        got = plan_ctx.pairs(X, perspective="global")
        assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref_all, got)), plan
        got = plan_ctx.pairs(X, qi, qj, "local")
        assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref_lst, got)), plan
        got = plan_ctx.matrix(X, (float("nan"),))
        assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref_mat, got)), plan
    # long columns: the persistent kernel (pend in global memory) launched once per chunk
    n2, S2 = 40000, 60                                   # 19 MB in chunks of 26 columns
    X2 = np.asfortranarray(rng.standard_normal((n2, S2)))
    X2[rng.random((n2, S2)) < 0.03] = np.nan
    X2[:, 3] = np.round(X2[:, 3] * 20)
    plan_ctx.debug_set_plan({"pipe": 0})
    ref = plan_ctx.pairs(X2, perspective="global")
    plan_ctx.debug_set_plan({"pipe": 1})
    got = plan_ctx.pairs(X2, perspective="global")
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref, got))
