"""Invalid arguments at every buffer-taking entry point of the C ABI (include/icikt.h): a negative ICIKT_E_* code and a
message, never a crash, never a write through a bad pointer -- and the context still computes afterwards.  The
reference's equivalents are R errors (src/kendallc.cpp:168-170, BEGIN_RCPP / END_RCPP in src/RcppExports.cpp:84,95)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BIG = 1 << 40


def _p(a):
    return None if a is None else a.ctypes.data


def test_every_entry_refuses_bad_arguments(hip_ctx):
    from icikendalltau_amd import _lib
    L, h = _lib.lib(), hip_ctx._h
    rng = np.random.default_rng(5)
    n, S = 60, 5
    X = np.asfortranarray(rng.standard_normal((n, S)))
    pi = np.array([0, 1, 2], dtype=np.int32); pj = np.array([1, 2, 4], dtype=np.int32)
    P = 3
    out = np.full((S * (S - 1) // 2, 4), 7.0); cnt = np.zeros((S * (S - 1) // 2, len(_lib.CNT_FIELDS)), dtype=np.int64)
    rsn = np.zeros(S * (S - 1) // 2, dtype=np.int32)
    bad_i = np.array([0, -1, 2], dtype=np.int32); big_j = np.array([1, 2, S], dtype=np.int32)
    tried = 0

    def refused(rc, who):
        nonlocal tried
        tried += 1
        assert rc < 0, (who, rc)
        assert rc in (-1, -4, -5), (who, rc)                       # INVALID / TOO_LONG / STATE: never a HIP error
        assert len(L.icikt_last_error(h)) > 0, who
        assert np.all(out == 7.0), who                              # nothing was written

    # ---- icikt_pairs_f64 -----------------------------------------------------------------------------------------
    def pairs(X_=X, n_=n, S_=S, ld_=n, pi_=pi, pj_=pj, P_=P, persp=1, alt=0, out_=out, cnt_=cnt, rsn_=rsn):
        return L.icikt_pairs_f64(h, _p(X_), n_, S_, ld_, _p(pi_), _p(pj_), P_, persp, alt, 0, 0, _p(out_), _p(cnt_), _p(rsn_))
    for kw in (dict(X_=None), dict(n_=-1), dict(S_=-1), dict(ld_=n - 1), dict(P_=-1), dict(pi_=None, pj_=pj), dict(pi_=pi, pj_=None),
               dict(pi_=bad_i), dict(pj_=big_j), dict(persp=2), dict(persp=-1), dict(alt=9), dict(alt=-1), dict(out_=None),
               dict(n_=BIG, ld_=BIG)):
        refused(pairs(**kw), ("pairs", kw))
    # ---- icikt_matrix_f64 ----------------------------------------------------------------------------------------
    out5 = np.full((5, S, S), 7.0); keep = np.zeros((S, n), dtype=np.uint8); rc5 = np.zeros(5, dtype=np.int64)
    gna = np.array([np.nan, 0.0])

    def matrix(X_=X, n_=n, S_=S, ld_=n, gna_=gna, ng=2, pi_=pi, pj_=pj, P_=P, persp=1, alt=0, out_=out5):
        return L.icikt_matrix_f64(h, _p(X_), n_, S_, ld_, _p(gna_), ng, _p(pi_), _p(pj_), P_, persp, alt, 0, 0, 1, 1, _p(out_),
                                  _p(keep), _p(rc5))
    for kw in (dict(X_=None), dict(n_=-1), dict(S_=-1), dict(ld_=n - 1), dict(P_=-1), dict(pi_=None, pj_=pj), dict(pi_=bad_i),
               dict(pj_=big_j), dict(persp=5), dict(alt=17), dict(out_=None), dict(gna_=None, ng=2), dict(ng=-1), dict(ng=1000),
               dict(n_=BIG, ld_=BIG)):
        refused(matrix(**kw), ("matrix", kw))
        assert np.all(out5 == 7.0), kw
    # ---- icikt_pair_f64 ------------------------------------------------------------------------------------------
    x, y = X[:, 0].copy(), X[:, 1].copy()
    o4 = np.full(4, 7.0); c11 = np.zeros(len(_lib.CNT_FIELDS), dtype=np.int64); r1 = np.zeros(1, dtype=np.int32)

    def pair(x_=x, y_=y, n_=n, persp=0, alt=0):
        return L.icikt_pair_f64(h, _p(x_), _p(y_), n_, persp, alt, 0, 0, _p(o4), _p(c11), _p(r1))
    for kw in (dict(x_=None), dict(y_=None), dict(n_=-1), dict(n_=BIG), dict(persp=3), dict(alt=-2)):
        refused(pair(**kw), ("pair", kw))
        assert np.all(o4 == 7.0), kw
    # ---- icikt_pairs_complete_f64 / icikt_missingness_f64 ------------------------------------------------------------
    m = np.full(P, -7, dtype=np.int64)
    for kw in (dict(X_=None), dict(n_=-1), dict(ld_=n - 1), dict(P_=-1), dict(pi_=None), dict(pj_=big_j), dict(out_=None), dict(alt=9),
               dict(n_=100000, ld_=100000)):                          # (no wide path behind this entry)
        a = dict(X_=X, n_=n, S_=S, ld_=n, pi_=pi, pj_=pj, P_=P, alt=0, out_=out)
        a.update(kw)
        refused(L.icikt_pairs_complete_f64(h, _p(a["X_"]), a["n_"], a["S_"], a["ld_"], _p(a["pi_"]), _p(a["pj_"]), a["P_"], a["alt"],
                                           0, 0, _p(a["out_"]), None, None), ("pairs_complete", kw))
    for kw in (dict(X_=None), dict(n_=-1), dict(ld_=n - 1), dict(P_=-1), dict(pi_=None), dict(pi_=bad_i), dict(pj_=big_j), dict(m_=None)):
        a = dict(X_=X, n_=n, S_=S, ld_=n, pi_=pi, pj_=pj, P_=P, m_=m)
        a.update(kw)
        refused(L.icikt_missingness_f64(h, _p(a["X_"]), a["n_"], a["S_"], a["ld_"], _p(a["pi_"]), _p(a["pj_"]), a["P_"], _p(a["m_"])),
                ("missingness", kw))
        assert np.all(m == -7), kw
    # ---- the device-resident sequence: order of calls and ranges ---------------------------------------------------------
    import torch
    fresh = _lib.Context(0)
    try:
        hf = fresh._h
        dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
        dout = torch.full((10, 4), 7.0, dtype=torch.float64, device="cuda")

        def refused_f(rc, who):
            nonlocal tried
            tried += 1
            assert rc in (-1, -4, -5) and len(L.icikt_last_error(hf)) > 0, (who, rc)
        refused_f(L.icikt_run_dev(hf, 1, 0, 0, 0, dout.data_ptr(), None, None), "run before prepare")
        refused_f(L.icikt_expand_cols_dev(hf, 0, 2, 0), "expand before prepare")
        ptrs = (ctypes.c_void_p * 5)(); bpc = (ctypes.c_int64 * 5)()
        refused_f(L.icikt_prep_arrays(hf, ptrs, bpc), "prep_arrays before prepare")
        refused_f(L.icikt_prepare_dev(hf, None, n, S, n, 0), "prepare null")
        refused_f(L.icikt_prepare_dev(hf, dX.data_ptr(), n, S, n - 1, 0), "prepare ld")
        refused_f(L.icikt_prepare_cols_dev(hf, dX.data_ptr(), n, S, n, 1, 3, S + 1, 0), "odd range start")
        refused_f(L.icikt_prepare_cols_dev(hf, dX.data_ptr(), n, S, n, 0, 3, S + 1, 0), "odd range end")
        refused_f(L.icikt_prepare_cols_dev(hf, dX.data_ptr(), n, S, n, 2, 0, S + 1, 0), "reversed range")
        refused_f(L.icikt_prepare_cols_dev(hf, dX.data_ptr(), n, S, n, 0, S + 2, S + 2, 0), "range past n_samp")
        refused_f(L.icikt_prepare_cols_dev(hf, dX.data_ptr(), n, S, n, 0, S, S - 1, 0), "alloc_cols < n_samp")
        refused_f(L.icikt_set_pairs(hf, _p(bad_i), _p(pj), 3), "negative index")
        refused_f(L.icikt_set_pairs(hf, None, _p(pj), 3), "null list")
        refused_f(L.icikt_set_pairs_combn(hf, S, 5, 2), "combn range reversed")
        refused_f(L.icikt_set_pairs_combn(hf, S, 0, S * S), "combn range too long")
        assert L.icikt_prepare_dev(hf, dX.data_ptr(), n, S, n, 0) == 0
        refused_f(L.icikt_run_dev(hf, 1, 0, 0, 0, dout.data_ptr(), None, None), "run without pairs")
        assert L.icikt_set_pairs(hf, _p(pi), _p(big_j), 3) in (0, -1)          # an index beyond n_samp is caught here or at run
        if L.icikt_num_pairs(hf) == 3:
            refused_f(L.icikt_run_dev(hf, 1, 0, 0, 0, dout.data_ptr(), None, None), "pair beyond the prepared columns")
        assert L.icikt_set_pairs(hf, _p(pi), _p(pj), 3) == 0
        refused_f(L.icikt_run_dev(hf, 1, 0, 0, 0, None, None, None), "run null out")
        refused_f(L.icikt_run_dev(hf, 4, 0, 0, 0, dout.data_ptr(), None, None), "run perspective")
        refused_f(L.icikt_expand_cols_dev(hf, 2, 9, 0), "expand past n_samp")
        assert torch.all(dout == 7.0)
        assert L.icikt_run_dev(hf, 1, 0, 0, 0, dout.data_ptr(), None, None) == 0 and L.icikt_sync(hf) == 0
        ref, _c, _r = hip_ctx.pairs(X, pi, pj, "global", want_counts=False)
        assert np.array_equal(dout.cpu().numpy()[:3], ref, equal_nan=True)
    finally:
        fresh.close()
    # ---- null handles ----------------------------------------------------------------------------------------------
    assert L.icikt_pairs_f64(None, _p(X), n, S, n, None, None, 0, 1, 0, 0, 0, _p(out), None, None) == -1
    assert L.icikt_matrix_f64(None, _p(X), n, S, n, _p(gna), 2, None, None, 0, 1, 0, 0, 0, 1, 1, _p(out5), None, None) == -1
    assert L.icikt_sync(None) == -1 and L.icikt_run_dev(None, 1, 0, 0, 0, None, None, None) == -1
    L.icikt_ctx_destroy(None)                                           # a no-op, as free(NULL)
    assert tried >= 70
    # and the context still works
    got, _c, _r = hip_ctx.pairs(X, perspective="global", want_counts=False)
    from oracle import oracle as O
    i, j = np.triu_indices(S, k=1)
    ref, _c2, _r2 = O.ici_pairs(X, i, j, "global", "two.sided", False, want_counts=False)
    assert np.max(np.abs(got - ref)) <= 1e-10
