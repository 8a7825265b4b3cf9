#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz and tests/golden/expected.json.

Run in the BUILD container (needs /root/reference for the yeast data file only):

    python tests/golden/make_golden.py

Inputs are recreated with an exact emulation of R's RNG (oracle/rrng.py), following
the R statements quoted beside each case; EXPECTED VALUES ARE TRANSCRIBED from the
reference's own files (snapshot, README, rendered vignette, test assertions) -- they
are not produced by any code in this repository.  The yeast matrix is the reference's
data file data/yeast_missing.rda (bzip2 + R XDR serialisation), decoded as data.
"""
from __future__ import annotations

import bz2
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.rrng import RRandom  # noqa: E402

REF = "/root/reference"


def yeast():
    d = bz2.decompress(open(os.path.join(REF, "data", "yeast_missing.rda"), "rb").read())
    i = d.find(b"\x00\x00\x02\x0e")  # REALSXP with attributes
    n = struct.unpack(">i", d[i + 4:i + 8])[0]
    a = np.frombuffer(d[i + 8:i + 8 + 8 * n], dtype=">f8").astype(np.float64)
    rest = d[i + 8 + 8 * n:]
    k = rest.find(b"dim\x00\x00\x00\x0d\x00\x00\x00\x02")
    nrow, ncol = struct.unpack(">ii", rest[k + 11:k + 19])
    assert nrow * ncol == n
    # dimnames: VECSXP(2) of STRSXP(nrow), STRSXP(ncol)
    k = rest.find(b"dimnames")
    p = k + 8
    assert rest[p:p + 8] == b"\x00\x00\x00\x13\x00\x00\x00\x02"
    p += 8
    names = []
    for expect in (nrow, ncol):
        assert rest[p:p + 4] == b"\x00\x00\x00\x10", rest[p:p + 8]
        cnt = struct.unpack(">i", rest[p + 4:p + 8])[0]
        assert cnt == expect
        p += 8
        cur = []
        for _ in range(cnt):
            ln = struct.unpack(">i", rest[p + 4:p + 8])[0]
            cur.append(rest[p + 8:p + 8 + ln].decode())
            p += 8 + ln
        names.append(cur)
    X = a.reshape((ncol, nrow)).T  # column-major payload
    return np.asfortranarray(X), names[0], names[1]


def main():
    expected = {}

    # -- snapshot: tests/testthat/test-kendall-tau.R:72-78, _snaps/kendall-tau.md:1-8
    r = RRandom(1234)
    x = r.rnorm(50000)
    y = r.rnorm(50000)
    np.savez_compressed(os.path.join(HERE, "snapshot_50000.npz"), x=x, y=y)
    expected["snapshot_50000"] = {
        "source": "tests/testthat/_snaps/kendall-tau.md:6-7",
        "r_code": "set.seed(1234); x = rnorm(50000); y = rnorm(50000); ici_kt(x, y, perspective='global')",
        "perspective": "global",
        "tau": -0.00123518, "pvalue": 0.67867094, "tau_max": 1.0, "completeness": 1.0,
        "print_digits": 8,
    }

    # -- README.md:122-153 and :248-258
    r = RRandom(1234)
    s1 = np.sort(r.rnorm(1000, 100, 10))
    s2 = s1 + 10
    s3 = s1.copy()
    s3[r.sample(100, 50) - 1] = np.nan
    s4 = s2.copy()
    s4[r.sample(100, 50) - 1] = np.nan
    np.savez_compressed(os.path.join(HERE, "readme_s1_s4.npz"), s1=s1, s2=s2, s3=s3, s4=s4)
    expected["readme"] = {
        "source": "README.md:122-153,248-258",
        "r_code": "set.seed(1234); s1 = sort(rnorm(1000, 100, 10)); s2 = s1 + 10; s3 = s1; s3[sample(100, 50)] = NA; "
                  "s4 = s2; s4[sample(100, 50)] = NA; ici_kendalltau(cbind(s3, s4), return_matrix = FALSE)",
        "r_1_cor": 1.0,
        "r_2_cor": 0.9944616,
        "r_4": {"raw": 0.9924359, "pvalue": 0.0, "taumax": 0.997963, "completeness": 0.921, "cor": 0.9944616,
                "diag_completeness": 0.950, "diag_raw": 1.0},
    }

    # -- kt_fast snapshots (single pairs of ici_kt over complete rows):
    #    tests/testthat/test-kendall-tau.R:153-170, _snaps/kendall-tau.md:36-88
    r = RRandom(1234)
    X = r.rnorm(400).reshape(4, 100).T  # matrix(rnorm(400), nrow = 100, ncol = 4)
    np.savez_compressed(os.path.join(HERE, "ktfast_100x4.npz"), X=np.asfortranarray(X))
    expected["ktfast"] = {
        "source": "tests/testthat/_snaps/kendall-tau.md:56-88",
        "r_code": "set.seed(1234); x = matrix(rnorm(400), 100, 4); x_na2 = x; x_na2[10, 1] = NA; "
                  "kt_fast(x_na2, use = 'complete.obs') / 'pairwise.complete.obs'",
        "complete_obs_drop_row": 10,
        "complete_tau": {"0-1": 0.003092146, "0-2": 0.007215007, "0-3": 0.10904968,
                         "1-2": 0.0006184292, "1-3": 0.0455576170, "2-3": 0.0166975881},
        "complete_p": {"0-1": 9.638307e-01, "0-2": 9.157333e-01, "0-3": 1.097676e-01,
                       "1-2": 9.927638e-01, "1-3": 5.040615e-01, "2-3": 8.065540e-01},
        "self_p_n99": 1.076521e-48,
        "pairwise_tau": {"1-2": 0.002424242, "1-3": 0.044444444, "2-3": 0.010101010},
        "pairwise_p": {"1-2": 9.714917e-01, "1-3": 5.123482e-01, "2-3": 8.816279e-01},
        "self_p_n100": 3.480281e-49,
    }

    # -- completeness snapshot: test-kendall-tau.R:138-151, _snaps/kendall-tau.md:9-17
    r = RRandom(1234)
    Xc = r.rnorm(5000).reshape(50, 100)  # matrix(..., nrow = 50, ncol = 100, byrow = TRUE)
    idx = r.sample(5000, 40) - 1  # column-major linear index
    Xc = np.asfortranarray(Xc)
    Xc.reshape(-1, order="F")[idx] = np.nan
    np.savez_compressed(os.path.join(HERE, "completeness_50x100.npz"), X=Xc)
    expected["completeness"] = {
        "source": "tests/testthat/_snaps/kendall-tau.md:9-17",
        "r_code": "set.seed(1234); x = matrix(rnorm(5000), 50, 100, byrow = TRUE); x[sample(5000, 40)] = NA; "
                  "pairwise_completeness(x, return_matrix = FALSE)[4:6, ]",
        "rows_4_6": [{"s1": "s1", "s2": "s5", "missingness": 1, "completeness": 0.98},
                     {"s1": "s1", "s2": "s6", "missingness": 2, "completeness": 0.96},
                     {"s1": "s1", "s2": "s7", "missingness": 2, "completeness": 0.96}],
    }

    # -- include_only: test-kendall-tau.R:102-136
    r = RRandom(1234)
    Xi = np.asfortranarray(r.rnorm(5000).reshape(100, 50).T)  # matrix(rnorm(5000), nrow = 50, ncol = 100)
    np.savez_compressed(os.path.join(HERE, "include_only_50x100.npz"), X=Xi)
    expected["include_only"] = {
        "source": "tests/testthat/test-kendall-tau.R:102-136",
        "zeros": {"s1": 9702, "s1,s3": 9506, "list": 9896, "list_nodiag": 9996},
        "n_rows_nodiag_df": 2,
    }

    # -- check_timing: test-kendall-tau.R:256-265
    expected["check_timing"] = {"source": "tests/testthat/test-kendall-tau.R:256-265", "n_tested": 5, "n_todo": 780}

    # -- RNG-free known answers: test-kendall-tau.R:5-32 (vs base R cor / cor.test(exact = FALSE))
    expected["basic"] = {
        "source": "tests/testthat/test-kendall-tau.R:5-32",
        "note": "tau values are Kendall tau-b of tie-free vectors; p-values are cor.test(exact=FALSE) "
                "normal approximations, checked in tests against scipy.stats.kendalltau(method='asymptotic')",
        "completeness": [0.9, 0.8, 1 - 1 / 9, 0.8],
    }

    # -- yeast (BASELINE config c2 input; no golden outputs exist in the reference)
    if os.path.isdir(REF):
        Xy, rn, cn = yeast()
        assert Xy.shape == (6887, 96)
        np.savez_compressed(os.path.join(HERE, "yeast_missing.npz"), X=Xy, colnames=np.array(cn))
        expected["yeast"] = {"source": "data/yeast_missing.rda", "shape": [6887, 96], "n_zero": int((Xy == 0).sum())}

    with open(os.path.join(HERE, "expected.json"), "w") as f:
        json.dump(expected, f, indent=1)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
