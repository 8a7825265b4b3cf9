"""Host-side mirror of the reference's R front-end (R/kendalltau.R:96-308,357-421,563-629), exercised on
the CPU with the checker engine (tests/oracle_engine.py).  Expected values are the reference's own test
assertions and README output (tests/golden/expected.json)."""
import warnings

import numpy as np
import pandas as pd
import pytest

from icikendalltau_amd import api
from tests.oracle_engine import OracleEngine

ENG = OracleEngine()


def _names(n):
    return [f"s{i}" for i in range(1, n + 1)]


def test_readme_matrix_and_dataframe_outputs(expected):
    z = np.load("tests/golden/readme_s1_s4.npz")
    e = expected["readme"]
    r1 = api.ici_kendalltau(np.c_[z["s1"], z["s2"]], colnames=["s1", "s2"], engine=ENG)
    assert list(r1.keys()) == ["cor", "raw", "pvalue", "taumax", "completeness", "keep", "run_time"]
    assert np.all(r1["cor"].to_numpy() == 1.0)
    m2 = np.c_[z["s3"], z["s4"]]
    r2 = api.ici_kendalltau(m2, colnames=["s3", "s4"], engine=ENG)
    assert round(r2["cor"].loc["s3", "s4"], 7) == e["r_2_cor"] and r2["cor"].loc["s4", "s3"] == r2["cor"].loc["s3", "s4"]
    assert r2["keep"].shape == (2, 1000) and r2["keep"].dtype == bool
    r4 = api.ici_kendalltau(m2, colnames=["s3", "s4"], return_matrix=False, engine=ENG)["cor"]
    assert list(r4.columns) == ["s1", "s2", "core", "raw", "pvalue", "taumax", "completeness", "cor"]
    assert len(r4) == 3 and list(r4["s1"]) == ["s3", "s3", "s4"] and list(r4["core"]) == [1, 0, 0]
    row = r4.iloc[0]
    assert round(row["raw"], 7) == e["r_4"]["raw"] and round(row["taumax"], 6) == e["r_4"]["taumax"]
    assert row["completeness"] == pytest.approx(e["r_4"]["completeness"]) and round(row["cor"], 7) == e["r_4"]["cor"]
    assert r4.iloc[1]["completeness"] == pytest.approx(e["r_4"]["diag_completeness"]) and r4.iloc[1]["raw"] == 1.0


def test_matrix_equals_single_pair():
    # test-kendall-tau.R:61-70, 223-239
    rng = np.random.default_rng(0)
    x = np.sort(rng.standard_normal(100))
    y = x + 1
    y[:20] = np.nan
    m = np.c_[x, y]
    mc = api.ici_kendalltau(m, global_na=[np.nan], perspective="global", scale_max=False, colnames=["x", "y"], engine=ENG)
    single = api.ici_kt(x, y, "global", engine=ENG)
    assert single[0] == mc["raw"].iloc[1, 0] and single["tau"] == single.tau
    lc = api.ici_kendalltau(m, global_na=[np.nan], perspective="global", scale_max=False, return_matrix=False,
                            colnames=["x", "y"], engine=ENG)["cor"]
    assert len(lc) == 3 and lc["raw"][0] == mc["raw"].iloc[1, 0] and lc["raw"][2] == mc["raw"].iloc[1, 1]


def test_include_only_semantics(expected):
    # test-kendall-tau.R:102-136
    X = np.load("tests/golden/include_only_50x100.npz")["X"]
    names = _names(100)
    z = expected["include_only"]["zeros"]
    r = api.ici_kendalltau(X, include_only="s1", colnames=names, engine=ENG)
    assert int((r["cor"].to_numpy() == 0).sum()) == z["s1"]
    r = api.ici_kendalltau(X, include_only=["s1", "s3"], colnames=names, engine=ENG)
    assert int((r["cor"].to_numpy() == 0).sum()) == z["s1,s3"]
    inc = {"s1": "s1", "s2": ["s2", "s3"]}
    r3 = api.ici_kendalltau(X, include_only=inc, colnames=names, engine=ENG)
    assert int((r3["cor"].to_numpy() == 0).sum()) == z["list"]
    r4 = api.ici_kendalltau(X, include_only=pd.DataFrame({"s1": ["s1", "s1"], "s2": ["s2", "s3"]}), colnames=names, engine=ENG)
    assert np.array_equal(r4["cor"].to_numpy(), r3["cor"].to_numpy())
    r5 = api.ici_kendalltau(X, include_only=inc, diag_good=False, colnames=names, engine=ENG)
    assert int((r5["cor"].to_numpy() == 0).sum()) == z["list_nodiag"]
    r6 = api.ici_kendalltau(X, include_only=inc, diag_good=False, return_matrix=False, colnames=names, engine=ENG)
    assert len(r6["cor"]) == expected["include_only"]["n_rows_nodiag_df"]
    with pytest.raises(ValueError, match="list of two vectors"):
        api.ici_kendalltau(X, include_only=[["s1"], ["s2", "s3"], ["s4"]], diag_good=False, colnames=names, engine=ENG)
    with pytest.raises(ValueError, match="No comparisons to do."):
        api.ici_kendalltau(X, include_only=[["s102"], ["s105"]], diag_good=False, colnames=names, engine=ENG)


def test_completeness_matches_pairwise_completeness(expected):
    # test-kendall-tau.R:138-151
    X = np.load("tests/golden/completeness_50x100.npz")["X"]
    names = _names(100)
    xc = api.ici_kendalltau(X, perspective="global", return_matrix=False, colnames=names, engine=ENG)["cor"]
    comp = api.pairwise_completeness(X, return_matrix=False, colnames=names, engine=ENG)
    assert len(xc) == len(comp) == 100 * 99 // 2 + 100
    assert np.allclose(xc["completeness"].to_numpy(), comp["completeness"].to_numpy(), atol=1e-15)
    for k, row in enumerate(expected["completeness"]["rows_4_6"]):
        got = comp.iloc[3 + k]
        assert (got["s1"], got["s2"], got["missingness"]) == (row["s1"], row["s2"], row["missingness"])
        assert got["completeness"] == pytest.approx(row["completeness"])
    cm = api.pairwise_completeness(X, colnames=names, engine=ENG)
    assert cm.shape == (100, 100) and cm.loc["s1", "s5"] == pytest.approx(0.98)


def test_input_validation_messages():
    # test-kendall-tau.R:241-254
    x = np.random.default_rng(1).standard_normal((20, 10))
    with pytest.raises(ValueError, match="Colnames of `data_matrix` must be be specified."):
        api.ici_kendalltau(x, engine=ENG)
    with pytest.raises(ValueError, match="Colnames of `data_matrix` must be be specified."):
        api.pairwise_completeness(x, engine=ENG)
    with pytest.raises(TypeError, match="must be a numeric type"):
        api.ici_kendalltau(x.astype(str), colnames=_names(10), engine=ENG)
    df = pd.DataFrame(x, columns=_names(10))
    r = api.ici_kendalltau(df, engine=ENG)
    assert list(r["cor"].columns) == _names(10)


def test_check_timing(expected):
    X = np.random.default_rng(2).standard_normal((100, 40))
    res = api.ici_kendalltau(X, check_timing=True, colnames=_names(40), engine=ENG)
    assert res["value"][0] == expected["check_timing"]["n_tested"] and res["value"][1] == expected["check_timing"]["n_todo"]
    assert list(res["which"])[-1] == "time_days"


def test_global_na_and_scaling():
    rng = np.random.default_rng(3)
    X = np.abs(rng.standard_normal((60, 5)))
    X[rng.random(X.shape) < 0.2] = 0.0       # zeros are missing by default (global_na = c(NA, Inf, 0))
    X[3, 2] = np.inf
    ex = api.setup_missing_matrix(X, (np.nan, np.inf, 0))
    assert ex.sum() == (X == 0).sum() + 1
    r = api.ici_kendalltau(X, colnames=_names(5), engine=ENG)
    raw, cor, tm = r["raw"].to_numpy(), r["cor"].to_numpy(), r["taumax"].to_numpy()
    iu = np.triu_indices(5, 1)
    assert np.allclose(cor[iu], raw[iu] / tm[iu].max())
    n_good = (~ex).sum(0)
    assert np.allclose(np.diag(raw), n_good / n_good.max()) and np.all(np.diag(r["pvalue"].to_numpy()) == 0)
    r2 = api.ici_kendalltau(X, global_na=[np.nan], colnames=_names(5), engine=ENG)
    assert not np.allclose(r2["raw"].to_numpy()[iu], raw[iu])


def test_warnings_and_na_rows():
    X = np.random.default_rng(4).standard_normal((30, 4))
    X[:, 1] = 1.0
    X[:, 2] = np.nan
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        r = api.ici_kendalltau(X, global_na=[np.nan], colnames=_names(4), return_matrix=False, engine=ENG)["cor"]
    assert any("single unique value" in str(m.message) for m in w)
    assert np.isnan(r["raw"][0]) and np.isnan(r["raw"][1])
    with pytest.warns(RuntimeWarning, match="only have a single value"):
        api.ici_kt([1.0], [2.0], engine=ENG)
    with pytest.raises(ValueError, match="not the same length"):
        api.ici_kt([1.0, 2.0], [2.0], engine=ENG)


def test_scaling_skips_na_pairs_wherever_they_sit():
    """max(taumax, na.rm = TRUE) (R/kendalltau.R:368-373) over results that hold R's NA_real_ -- a SIGNALLING NaN, for
    which numpy's NaN-skipping reductions are not safe (C fmax answers NaN for one): NA pairs in the middle and at the
    end of the pair list, and the maximum in front of them."""
    rng = np.random.default_rng(8)
    X = rng.standard_normal((60, 4))
    X[:5, 0] = np.nan                  # taumax < 1 for every pair but (s2, s3)
    X[:, 3] = 4.25                     # a constant column: its three pairs are NA, the last two of the list among them
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = api.ici_kendalltau(X, global_na=[np.nan], colnames=_names(4), return_matrix=False, engine=ENG)["cor"]
    tm, raw, cor = r["taumax"].to_numpy()[:6], r["raw"].to_numpy()[:6], r["cor"].to_numpy()[:6]
    assert np.isnan(tm).tolist() == [False, False, True, False, True, True]
    assert tm.view(np.uint64)[2] == 0x7FF00000000007A2        # NA_real_, not a quiet NaN
    best = max(t for t in tm if t == t)
    assert best == tm[3] == 1.0 and tm[0] < 1.0
    ok = ~np.isnan(tm)
    assert np.array_equal(cor[ok], raw[ok] / best) and np.all(np.isnan(cor[~ok]))


def test_all_pairs_degenerate_scales_by_minus_inf():
    """max(taumax, na.rm = TRUE) over no value at all is -Inf in R (max(numeric(0)), R/kendalltau.R:368-373), so
    cor = raw / -Inf: NA pairs stay NA.  The device assembly (k_assemble) does the same; the host assembly used NaN."""
    X = np.tile(np.array([1.0, 2.0, 3.0])[None, :], (20, 1))       # every column constant: every pair is NA
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = api.ici_kendalltau(X, global_na=[np.nan], colnames=_names(3), return_matrix=False, engine=ENG)["cor"]
    assert np.all(np.isnan(r["cor"].to_numpy()[:3])) and np.all(r["cor"].to_numpy()[3:] == 1.0)


def test_many_global_na_values_take_the_host_masking_route():
    """The reference loops over ANY number of global_na values (R/utils.R:16-20).  The device-side rule of the one-call
    matrix entry holds _lib.MASK_VALS distinct finite ones; a longer list must not fail: the front-end masks the matrix
    itself (setup_missing_matrix) and runs the pair-list route.  `_Matrixless` has a `matrix` method that must not be
    reached with the long list."""
    from icikendalltau_amd import _lib
    rng = np.random.default_rng(21)
    X = rng.integers(0, 60, size=(80, 5)).astype(float)
    gna = [float("nan")] + [float(v) for v in range(40)]           # 40 distinct finite values

    class _Matrixless(type(ENG)):
        def matrix(self, *a, **k):
            raise AssertionError("the device rule cannot hold this list")

    eng = _Matrixless()
    assert len(gna) - 1 > _lib.MASK_VALS
    r = api.ici_kendalltau(X, global_na=gna, colnames=_names(5), engine=eng)
    Xm = X.copy()
    Xm[X < 40] = np.nan
    ref = api.ici_kendalltau(Xm, global_na=[np.nan], colnames=_names(5), engine=ENG)
    for k in ("cor", "raw", "pvalue", "taumax", "completeness"):
        assert np.array_equal(r[k].to_numpy(), ref[k].to_numpy(), equal_nan=True), k
    assert np.array_equal(r["keep"], ref["keep"])


def test_setup_comparisons_order_and_cores():
    pi, pj, core = api.setup_comparisons(_names(5), ncore=3)
    assert list(zip(pi, pj))[:5] == [(0, 1), (0, 2), (0, 3), (0, 4), (1, 2)]  # utils::combn order
    assert list(core) == [1, 1, 1, 1, 2, 2, 2, 2, 3, 3]                      # ceiling(10 / 3) = 4 per core
    pi, pj, _ = api.setup_comparisons(_names(3), diag_good=False)
    assert list(zip(pi, pj)) == [(0, 1), (0, 2), (1, 2), (0, 0), (1, 1), (2, 2)]


def _sig7(v):
    return float(f"{v:.7g}")


def test_kt_fast_snapshots(expected):
    # tests/testthat/test-kendall-tau.R:153-205 and _snaps/kendall-tau.md:19-88
    X = np.load("tests/golden/ktfast_100x4.npz")["X"]
    names = _names(4)
    e = expected["ktfast"]
    from scipy import stats
    fast = api.kt_fast(X, colnames=names, engine=ENG)
    base = np.array([[stats.kendalltau(X[:, i], X[:, j]).statistic for j in range(4)] for i in range(4)])
    assert np.allclose(fast["tau"].to_numpy(), base, atol=1e-14)
    x_na = X.copy()
    x_na[:, 0] = np.nan
    assert np.all(np.isnan(api.kt_fast(x_na, use="complete.obs", colnames=names, engine=ENG)["tau"].to_numpy()))
    x_na2 = X.copy()
    x_na2[9, 0] = np.nan
    ev = api.kt_fast(x_na2[:, 0], x_na2[:, 1], engine=ENG)
    assert np.all(np.isnan(ev["tau"].to_numpy())) and np.all(np.isnan(ev["pvalue"].to_numpy()))
    pc = api.kt_fast(x_na2[:, 0], x_na2[:, 1], use="complete.obs", engine=ENG)
    assert _sig7(pc["tau"].iloc[0, 1]) == e["complete_tau"]["0-1"] and _sig7(pc["pvalue"].iloc[0, 1]) == e["complete_p"]["0-1"]
    assert pc["tau"].iloc[0, 0] == 1.0 and _sig7(pc["pvalue"].iloc[0, 0]) == e["self_p_n99"]
    pp = api.kt_fast(x_na2[:, 0], x_na2[:, 1], use="pairwise.complete.obs", engine=ENG)
    # the reference's expect_equal() passes on all.equal's mean relative difference; entry by entry only the
    # second self comparison differs (100 rows instead of 99: 3.48e-49 vs 1.08e-48)
    assert np.array_equal(pp["tau"].to_numpy(), pc["tau"].to_numpy())
    assert np.allclose(pp["pvalue"].to_numpy(), pc["pvalue"].to_numpy(), rtol=0, atol=1.5e-8)
    assert _sig7(pp["pvalue"].iloc[1, 1]) == e["self_p_n100"] and _sig7(pp["pvalue"].iloc[0, 0]) == e["self_p_n99"]
    assert np.all(np.isnan(api.kt_fast(x_na2, use="everything", colnames=names, engine=ENG)["tau"].to_numpy()))
    mc = api.kt_fast(x_na2, use="complete.obs", colnames=names, engine=ENG)
    for key, t in e["complete_tau"].items():
        i, j = map(int, key.split("-"))
        assert mc["tau"].iloc[i, j] == pytest.approx(t, rel=2e-7) and _sig7(mc["pvalue"].iloc[i, j]) == pytest.approx(e["complete_p"][key], rel=1e-6)
    mp = api.kt_fast(x_na2, use="pairwise.complete.obs", colnames=names, engine=ENG)
    for key, t in e["pairwise_tau"].items():
        i, j = map(int, key.split("-"))
        assert mp["tau"].iloc[i, j] == pytest.approx(t, rel=2e-7) and _sig7(mp["pvalue"].iloc[i, j]) == pytest.approx(e["pairwise_p"][key], rel=1e-6)
    assert _sig7(mp["pvalue"].iloc[1, 1]) == e["self_p_n100"] and _sig7(mp["pvalue"].iloc[0, 0]) == e["self_p_n99"]
    assert np.array_equal(mp["tau"].to_numpy()[:, 0], mc["tau"].to_numpy()[:, 0])
    df = api.kt_fast(X, return_matrix=False, colnames=names, engine=ENG)["tau"]
    assert list(df.columns) == ["s1", "s2", "core", "tau", "pvalue"] and len(df) == 10
    assert df["tau"][3] == fast["tau"].loc["s2", "s3"]
    with pytest.raises(ValueError, match="is not a supported"):
        api.kt_fast(X, use="na.or.complete", colnames=names, engine=ENG)
    with pytest.raises(ValueError, match="Colnames of `x` must be be specified."):
        api.kt_fast(X, engine=ENG)
    with pytest.raises(ValueError, match="should both be provided as vectors"):
        api.kt_fast(X[:, 0], engine=ENG)
    with pytest.raises(ValueError, match="must be vectors"):
        api.kt_fast(X, X, colnames=names, engine=ENG)


def test_ici_kt_report_output(capsys):
    """ici_kt(output != "simple") prints the report of src/kendallc.cpp:342-363: same labels in the same order,
    integers as integers, doubles in std::to_string's "%f" form; nothing is printed for NA results (the reference
    returns before the report) or for output = "simple"."""
    from icikendalltau_amd import api
    from tests.oracle_engine import OracleEngine
    x = np.arange(1.0, 11.0)
    y = x.copy()
    y[1] = 15.0                                           # tests/testthat/test-kendall-tau.R:8-10
    res = api.ici_kt(x, y, "global", output="full", engine=OracleEngine())
    text = capsys.readouterr().out
    labels = [ln.split(":")[0] for ln in text.splitlines()]
    assert labels == ["min_x", "min_y", "n_entry", "missingness", "completeness", "tot", "sum_obs", "dis",
                      "con_minus_dis (k_numerator)", "n_tie", "m", "x_tie", "y_tie", "s_adjusted", "var", "z_b", "tau",
                      "tau_max", "pvalue"]
    d = dict(ln.split(":", 1) for ln in text.splitlines())
    assert d["min_x"].strip() == "0.900000" and d["n_entry"].strip() == "10" and d["tot"].strip() == "45"
    assert d["sum_obs"].strip() == "11" and d["dis"].strip() == "8" and d["m"].strip() == "90"
    assert d["con_minus_dis (k_numerator)"].strip() == "29.000000"
    assert abs(float(d["tau"]) - res.tau) < 1e-6 and abs(float(d["pvalue"]) - res.pvalue) < 1e-6
    assert float(d["var"]) == 125.0                       # n(n-1)(2n+5)/18 without ties
    api.ici_kt(x, y, "global", engine=OracleEngine())
    assert capsys.readouterr().out == ""
    with pytest.warns(RuntimeWarning):
        api.ici_kt(x, np.ones(10), "global", output="full", engine=OracleEngine())
    assert capsys.readouterr().out == ""


def test_bench_self_launch_command_line():
    """`python bench.py --gpus N` as typed (no WORLD_SIZE): the parent starts torch.distributed.run with one rank per GPU
    of one node, rendezvous on 127.0.0.1, and hands its own arguments on -- and does so without importing torch."""
    import subprocess
    import sys
    import bench
    cmd = bench.child_command(["--gpus", "8", "--steps", "3", "--warmup", "1"], 8, 29611)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[3:10] == ["--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29611"]
    assert cmd[10].endswith("bench.py") and cmd[11:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]
    assert bench.should_spawn(8, "torchrun", {}) and bench.should_spawn(2, "torchrun", {"RANK": ""})
    assert not bench.should_spawn(1, "torchrun", {})                       # N = 1: this process IS the run
    assert not bench.should_spawn(8, "torchrun", {"WORLD_SIZE": "8"})      # already a rank (the driver's own launch)
    assert not bench.should_spawn(8, "inlib", {})                          # one process drives all GPUs
    assert bench.should_spawn(1, "torchrun", {"ICIKT_BENCH_FORCE_DIST": "1"})
    p = bench.free_port()
    assert 1024 < p < 65536
    # importing bench (what the parent has done when it spawns) must not have loaded torch
    code = "import sys, bench; assert 'torch' not in sys.modules, 'bench imported torch at module level'"
    r = subprocess.run([sys.executable, "-c", code], cwd=bench.ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # and a parent whose child fails relays the failure: a command that cannot start ranks (no such GPU count is checked
    # by torchrun itself; here the child is made to fail at once through an invalid argument)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--config", "nope"], cwd=bench.ROOT, capture_output=True, text=True)
    assert r.returncode != 0 and "invalid choice" in r.stderr


def test_cost_weighted_block_cut_arithmetic():
    """icikt_cost_blocks (the cut behind ICIKT_FLAG_BALANCE_COST; host arithmetic of the C library, no device): against a
    brute-force walk over the pairs -- block k begins with the first pair whose preceding pairs cost >= total k / G --
    for all pairs of the triangle (each pair streams its SECOND column) and for explicit lists; block-length limit; and
    with equal costs the cut is the reference's ceiling(n_todo / ncore) chunks wherever those divide evenly
    (R/kendalltau.R:250-255)."""
    from icikendalltau_amd import _lib
    rng = np.random.default_rng(0)

    def brute(costs, G):
        tot, run, k, b = sum(costs), 0, 1, [0] * (G + 1)
        for p, c in enumerate(costs):
            while k < G and run * G >= tot * k:
                b[k] = p
                k += 1
            run += c
        while k <= G:
            b[k] = len(costs)
            k += 1
        return b

    for _ in range(200):
        S, G = int(rng.integers(2, 45)), int(rng.integers(1, 9))
        cost = rng.integers(1, 60, S).astype(np.uint32)
        iu, ju = np.triu_indices(S, k=1)
        assert _lib.cost_blocks(cost, G) == brute([int(cost[j]) for j in ju], G)
        m = int(rng.integers(1, 400))
        pj = rng.integers(0, S, m).astype(np.int32)
        assert _lib.cost_blocks(cost, G, pj=pj) == brute([int(cost[j]) for j in pj], G)
    # a block-length limit: no block longer than it, all pairs covered, order kept
    cost = np.array([1] * 20 + [50] * 20, dtype=np.uint32)
    P = 40 * 39 // 2
    each = -(-P // 4)
    b = _lib.cost_blocks(cost, 4, max_block=each + 10)
    assert b[0] == 0 and b[-1] == P and all(0 <= y - x <= each + 10 for x, y in zip(b, b[1:]))
    with pytest.raises(_lib.IciktError):
        _lib.cost_blocks(cost, 4, max_block=10)            # four blocks of ten pairs cannot hold 780
    # equal costs, lengths that divide evenly: the reference's chunks
    assert _lib.cost_blocks(np.ones(33, dtype=np.uint32), 4) == [0, 132, 264, 396, 528]
    with pytest.raises(_lib.IciktError):
        _lib.cost_blocks(cost, 2, pj=np.array([0, 40], dtype=np.int32))   # a column index outside the matrix
