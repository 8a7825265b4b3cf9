"""ctypes binding of oracle/icikt_oracle.c (TEST INFRASTRUCTURE ONLY).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg and
nowhere else; the product package icikendalltau_amd never imports it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libicikt_oracle.so")

PERSPECTIVES = {"local": 0, "global": 1}
ALTERNATIVES = {"two.sided": 0, "less": 1, "greater": 2}
COUNT_FIELDS = ("n", "missing", "dis", "ntie", "xtie", "ytie", "x0", "x1", "y0", "y1", "tot", "sum_obs")

REASON_OK = 0
REASON_ALL_MISSING = 1
REASON_SHORT = 2
REASON_SINGLE_UNIQUE = 3
REASON_TIES_EQ_TOTAL = 4

_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "icikt_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        dp = ctypes.POINTER(ctypes.c_double)
        i64p = ctypes.POINTER(ctypes.c_int64)
        i32p = ctypes.POINTER(ctypes.c_int32)
        L.icikt_oracle_pair.argtypes = [dp, dp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, dp, i64p, i32p]
        L.icikt_oracle_pair.restype = ctypes.c_int
        L.icikt_oracle_pairs.argtypes = [dp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, i32p, i32p,
                                         ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         dp, i64p, i32p]
        L.icikt_oracle_pairs.restype = ctypes.c_int
        L.icikt_oracle_bruteforce.argtypes = [dp, dp, ctypes.c_int64, ctypes.c_int, i64p]
        L.icikt_oracle_bruteforce.restype = ctypes.c_int
        L.icikt_oracle_pnorm.argtypes = [ctypes.c_double, ctypes.c_int]
        L.icikt_oracle_pnorm.restype = ctypes.c_double
        L.icikt_oracle_counts_fields.restype = ctypes.c_int
        assert L.icikt_oracle_counts_fields() == len(COUNT_FIELDS)
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _alt_code(alternative: str) -> int:
    return ALTERNATIVES.get(alternative, 3)


def ici_kt(x, y, perspective="local", alternative="two.sided", continuity=False, int32_compat=True):
    """Oracle for the reference's ici_kt(); returns (out4, counts dict, reason)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    if x.shape[0] != y.shape[0]:
        raise ValueError("'X' and 'Y' are not the same length!")  # kendallc.cpp:168-170
    out = np.empty(4, dtype=np.float64)
    cnt = np.zeros(len(COUNT_FIELDS), dtype=np.int64)
    reason = ctypes.c_int32(0)
    rc = lib().icikt_oracle_pair(_dp(x), _dp(y), x.shape[0], PERSPECTIVES[perspective], _alt_code(alternative),
                                 int(bool(continuity)), int(bool(int32_compat)), _dp(out),
                                 cnt.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.byref(reason))
    if rc:
        raise MemoryError("oracle allocation failed")
    return out, dict(zip(COUNT_FIELDS, cnt.tolist())), reason.value


def ici_pairs(X, pi, pj, perspective="global", alternative="two.sided", continuity=False, int32_compat=True,
              want_counts=True):
    """Oracle for ici_split(): X is (n_feat, n_samp) in any layout; pairs are 0-based."""
    Xf = np.asfortranarray(X, dtype=np.float64)
    n_feat, n_samp = Xf.shape
    pi = np.ascontiguousarray(pi, dtype=np.int32)
    pj = np.ascontiguousarray(pj, dtype=np.int32)
    P = pi.shape[0]
    out = np.empty((P, 4), dtype=np.float64)
    cnt = np.zeros((P, len(COUNT_FIELDS)), dtype=np.int64) if want_counts else None
    reasons = np.zeros(P, dtype=np.int32)
    i32p = ctypes.POINTER(ctypes.c_int32)
    rc = lib().icikt_oracle_pairs(_dp(Xf), n_feat, n_samp, max(n_feat, 1), pi.ctypes.data_as(i32p),
                                  pj.ctypes.data_as(i32p), P, PERSPECTIVES[perspective], _alt_code(alternative),
                                  int(bool(continuity)), int(bool(int32_compat)), _dp(out),
                                  cnt.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)) if want_counts else None,
                                  reasons.ctypes.data_as(i32p))
    if rc:
        raise RuntimeError(f"oracle failed rc={rc}")
    return out, cnt, reasons


def bruteforce(x, y, perspective="global"):
    """O(n^2) enumeration: dict(con, dis, xtie, ytie, ntie) with exact int64 counts."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    cnt = np.zeros(5, dtype=np.int64)
    rc = lib().icikt_oracle_bruteforce(_dp(x), _dp(y), x.shape[0], PERSPECTIVES[perspective],
                                       cnt.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    if rc:
        raise MemoryError
    return dict(zip(("con", "dis", "xtie", "ytie", "ntie"), cnt.tolist()))


def pnorm(z, lower_tail=True):
    return lib().icikt_oracle_pnorm(float(z), int(bool(lower_tail)))
