"""The R .Call glue (icikendalltau_amd/r/icikt_rglue.c) compiled and driven WITHOUT R: tests/r_mock/ is a test double of
the subset of R's C API the glue uses (R itself is absent here and on the GPU box).  On the CPU: the glue compiles with
warnings as errors, registers its three routines with the arities the R wrapper (icikt_mi355x.R) calls them with, and
rejects bad arguments before it touches a device.  Under -m gpu: the three entry points end to end against the Python
binding of the same C ABI (one-based indices, result shapes and names, NA payloads, the counts record, error texts)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "r_mock")
GLUE = os.path.join(ROOT, "icikendalltau_amd", "r", "icikt_rglue.c")
RWRAP = os.path.join(ROOT, "icikendalltau_amd", "r", "icikt_mi355x.R")
OUT = os.path.join(MOCK, "_build", "librglue_mock.so")

LGLSXP, INTSXP, REALSXP, VECSXP = 10, 13, 14, 19
NA_REAL_BITS = 0x7FF00000000007A2


def _build():
    from icikendalltau_amd import _lib
    if _lib.needs_build():
        _lib.build()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    srcs = [GLUE, os.path.join(MOCK, "r_mock.c")]
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in srcs + [_lib.LIB_PATH]):
        return OUT
    libdir = os.path.dirname(_lib.LIB_PATH)
    # (-Wno-cast-function-type: R's own DL_FUNC idiom casts every routine to void *(*)(void))
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Wno-cast-function-type", "-Werror", "-fPIC", "-shared",
           "-I", MOCK, "-I", os.path.join(ROOT, "include"), *srcs, "-L", libdir, "-licikt_hip",
           f"-Wl,-rpath,{libdir}", "-o", OUT]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return OUT


@pytest.fixture(scope="module")
def glue():
    from icikendalltau_amd import _lib
    _lib.lib()                       # the HIP library first (and torch's HIP runtime before it): see _lib.lib()
    L = ctypes.CDLL(_build())
    vp, ci, cl = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
    for name, res, args in [("mock_null", vp, []), ("mock_real_vector", vp, [vp, cl]), ("mock_real_matrix", vp, [vp, ci, ci]),
                            ("mock_int_vector", vp, [vp, cl]), ("mock_logical", vp, [ci]), ("mock_string", vp, [ctypes.c_char_p]),
                            ("mock_type", ci, [vp]), ("mock_length", cl, [vp]), ("mock_is_matrix", ci, [vp]),
                            ("mock_nrow", ci, [vp]), ("mock_ncol", ci, [vp]), ("mock_data", vp, [vp]),
                            ("mock_list_elt", vp, [vp, cl]), ("mock_list_name", ctypes.c_char_p, [vp, cl]),
                            ("mock_dotcall", vp, [ctypes.c_char_p, ci, ctypes.POINTER(vp)]),
                            ("mock_last_error", ctypes.c_char_p, []), ("mock_routine_name", ctypes.c_char_p, [ci]),
                            ("mock_routine_nargs", ci, [ci]), ("mock_n_routines", ci, []), ("mock_dynamic_symbols", ci, []),
                            ("mock_protect_depth", ci, []), ("mock_init", None, []), ("mock_unload", None, []),
                            ("mock_reset", None, [])]:
        getattr(L, name).restype = res
        getattr(L, name).argtypes = args
    L.mock_init()
    yield L
    L.mock_unload()
    L.mock_reset()


class R:
    """Argument builders and result readers over the mock's driver."""

    def __init__(self, L):
        self.L = L

    def matrix(self, X):
        X = np.asfortranarray(X, dtype=np.float64)
        return self.L.mock_real_matrix(X.ctypes.data, X.shape[0], X.shape[1])

    def real(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        return self.L.mock_real_vector(v.ctypes.data, len(v))

    def ints(self, v):
        v = np.ascontiguousarray(v, dtype=np.int32)
        return self.L.mock_int_vector(v.ctypes.data, len(v))

    def int1(self, v):
        return self.ints([v])

    def lgl(self, v):
        return self.L.mock_logical(int(v))

    def chr(self, s):
        return self.L.mock_string(s.encode())

    @property
    def null(self):
        return self.L.mock_null()

    def call(self, name, *args):
        arr = (ctypes.c_void_p * len(args))(*args)
        res = self.L.mock_dotcall(name.encode(), len(args), arr)
        if not res:
            raise RuntimeError(self.L.mock_last_error().decode())
        assert self.L.mock_protect_depth() == 0, "PROTECT / UNPROTECT imbalance"
        return res

    def value(self, s):
        t, n = self.L.mock_type(s), self.L.mock_length(s)
        if t == VECSXP:
            return {self.L.mock_list_name(s, i).decode(): self.value(self.L.mock_list_elt(s, i)) for i in range(n)}
        if t == 0:
            return None
        ct = {REALSXP: ctypes.c_double, INTSXP: ctypes.c_int, LGLSXP: ctypes.c_int}[t]
        a = np.ctypeslib.as_array(ctypes.cast(self.L.mock_data(s), ctypes.POINTER(ct)), shape=(n,)).copy() if n else np.empty(0, ct)
        if self.L.mock_is_matrix(s):
            a = a.reshape((self.L.mock_nrow(s), self.L.mock_ncol(s)), order="F")
        return a


def _call_arity(text, routine):
    """Number of arguments after the routine name in the wrapper's .Call("routine", ...) (top-level commas)."""
    start = text.index(f'.Call("{routine}"')
    i = text.index("(", start) + 1
    depth, commas = 1, 0
    while depth:
        ch = text[i]
        depth += ch in "(["
        depth -= ch in ")]"
        commas += (ch == "," and depth == 1)
        i += 1
    return commas


def test_glue_compiles_registers_and_checks_arguments(glue):
    L, r = glue, R(glue)
    table = {L.mock_routine_name(i).decode(): L.mock_routine_nargs(i) for i in range(L.mock_n_routines())}
    assert table == {"icikt_R_pairs": 9, "icikt_R_matrix": 11, "icikt_R_missingness": 4}
    assert L.mock_dynamic_symbols() == 0                      # R_useDynamicSymbols(dll, FALSE), as src/RcppExports.cpp:127
    wrapper = open(RWRAP).read()
    for name, nargs in table.items():                          # the R wrapper calls every routine with its registered arity
        if f'.Call("{name}"' in wrapper:
            assert _call_arity(wrapper, name) == nargs, name
    assert re.search(r'\.Call\("icikt_R_pairs"', wrapper) and re.search(r'\.Call\("icikt_R_matrix"', wrapper)
    # argument checks that come before any device is touched
    with pytest.raises(RuntimeError, match="Incorrect number of arguments"):
        r.call("icikt_R_pairs", r.null)
    with pytest.raises(RuntimeError, match="not in load table"):
        r.call("icikt_R_nothing", r.null)
    args = [r.null, r.null, r.chr("global"), r.chr("two.sided"), r.lgl(0), r.int1(0), r.int1(1), r.lgl(0)]
    with pytest.raises(RuntimeError, match="exclude_data must be a double matrix"):
        r.call("icikt_R_pairs", r.real([1.0, 2.0]), *args)        # a vector, not a matrix
    with pytest.raises(RuntimeError, match="exclude_data must be a double matrix"):
        r.call("icikt_R_pairs", r.ints([1, 2]), *args)
    with pytest.raises(RuntimeError, match="data_matrix must be a double matrix"):
        r.call("icikt_R_matrix", r.real([1.0]), r.real([0.0]), r.null, r.null, r.chr("global"), r.chr("two.sided"), r.lgl(0),
               r.lgl(1), r.lgl(1), r.int1(0), r.int1(1))
    L.mock_reset()


@pytest.mark.gpu
def test_glue_entry_points_end_to_end(glue, hip_ctx):
    from icikendalltau_amd import _lib
    L, r = glue, R(glue)
    rng = np.random.default_rng(77)
    n, S = 700, 9
    X = np.asfortranarray(rng.standard_normal((n, S)))
    X[rng.random((n, S)) < 0.08] = np.nan
    X[:, 2] = np.round(X[:, 2] * 3)
    X[:, 5] = 1.25                                               # constant: reason 3, NA_real_ x 4
    common = lambda persp="global", alt="two.sided", cont=0: [r.chr(persp), r.chr(alt), r.lgl(cont)]

    # all pairs (pi = pj = NULL) with the counts record
    got = r.value(r.call("icikt_R_pairs", r.matrix(X), r.null, r.null, *common(), r.int1(0), r.int1(1), r.lgl(1)))
    out, cnt, rsn = hip_ctx.pairs(X, perspective="global")
    assert list(got) == ["raw", "pvalue", "taumax", "completeness", "reason", "counts"]
    for f, k in enumerate(("raw", "pvalue", "taumax", "completeness")):
        assert np.array_equal(got[k].view(np.uint64), out[:, f].copy().view(np.uint64)), k     # NA payloads included
    assert np.array_equal(got["reason"], rsn) and (rsn == 3).sum() == S - 1
    assert got["raw"].view(np.uint64)[rsn == 3].tolist() == [NA_REAL_BITS] * (S - 1)
    assert got["counts"].shape == (len(rsn), len(_lib.CNT_FIELDS)) and np.array_equal(got["counts"], cnt.astype(np.float64))

    # a one-based pair list, both orientations and a self pair; local / greater / continuity; no counts
    pi = np.array([1, 4, 9, 3, 7, 2]); pj = np.array([2, 1, 3, 3, 6, 8])
    got = r.value(r.call("icikt_R_pairs", r.matrix(X), r.ints(pi), r.ints(pj), *common("local", "greater", 1), r.int1(0),
                         r.int1(1), r.lgl(0)))
    out, _c, rsn = hip_ctx.pairs(X, pi - 1, pj - 1, "local", "greater", True, want_counts=False)
    assert got["counts"] is None
    assert np.array_equal(got["pvalue"], out[:, 1], equal_nan=True) and np.array_equal(got["reason"], rsn)
    # an alternative the reference does not know leaves p = 0 (src/kendallc.cpp:323-332)
    got = r.value(r.call("icikt_R_pairs", r.matrix(X), r.ints([1]), r.ints([2]), *common("global", "neither"), r.int1(0),
                         r.int1(1), r.lgl(0)))
    assert got["pvalue"][0] == 0.0

    # the whole matrix in one call: raw data in, global_na = c(NA, Inf, 0)
    Xr = X.copy()
    Xr[rng.random((n, S)) < 0.05] = 0.0
    Xr[3, 4] = np.inf
    gna = np.array([np.nan, np.inf, 0.0])
    got = r.value(r.call("icikt_R_matrix", r.matrix(Xr), r.real(gna), r.null, r.null, *common(), r.lgl(1), r.lgl(1), r.int1(0),
                         r.int1(1)))
    out5, keep, rc5 = hip_ctx.matrix(Xr, tuple(gna))
    assert list(got) == ["cor", "raw", "pvalue", "taumax", "completeness", "keep", "reason_counts"]
    for f, k in enumerate(("cor", "raw", "pvalue", "taumax", "completeness")):
        assert got[k].shape == (S, S) and np.array_equal(got[k], out5[f], equal_nan=True), k
    assert got["keep"].shape == (S, n) and np.array_equal(got["keep"].astype(bool), np.asarray(keep, dtype=bool))
    assert np.array_equal(got["reason_counts"], np.asarray(rc5, dtype=np.float64))
    # ... with a pair list, diag_good = FALSE, scale_max = FALSE
    got = r.value(r.call("icikt_R_matrix", r.matrix(Xr), r.real(gna), r.ints(pi), r.ints(pj), *common("local"), r.lgl(0), r.lgl(0),
                         r.int1(0), r.int1(1)))
    out5, _k, _r = hip_ctx.matrix(Xr, tuple(gna), pi - 1, pj - 1, "local", scale_max=False, diag_good=False)
    assert np.array_equal(got["cor"], out5[0], equal_nan=True) and np.array_equal(got["raw"], out5[1], equal_nan=True)

    # missingness
    got = r.value(r.call("icikt_R_missingness", r.matrix(X), r.ints(pi), r.ints(pj), r.int1(0)))
    assert np.array_equal(got, hip_ctx.missingness(X, pi - 1, pj - 1).astype(np.float64))

    # n_gpu = 2: devices 0 and 1 behind one call (icikt_pairs_multi_f64) where they exist, else an R error, never a fallback
    two = [r.matrix(X), r.ints(pi), r.ints(pj), *common(), r.int1(0), r.int1(2), r.lgl(0)]
    if _lib.device_count() >= 2:
        got = r.value(r.call("icikt_R_pairs", *two))
        out, _c, rsn = hip_ctx.pairs(X, pi - 1, pj - 1, "global", want_counts=False)
        assert np.array_equal(got["raw"], out[:, 0], equal_nan=True) and np.array_equal(got["reason"], rsn)
    else:
        with pytest.raises(RuntimeError, match="icikt"):
            r.call("icikt_R_pairs", *two)

    # errors of the library arrive as R errors with the library's message
    with pytest.raises(RuntimeError, match="out of range"):
        r.call("icikt_R_pairs", r.matrix(X), r.ints([1]), r.ints([S + 1]), *common(), r.int1(0), r.int1(1), r.lgl(0))
    with pytest.raises(RuntimeError, match="pi and pj differ in length"):
        r.call("icikt_R_pairs", r.matrix(X), r.ints([1, 2]), r.ints([2]), *common(), r.int1(0), r.int1(1), r.lgl(0))
    L.mock_reset()
