"""Randomised sweep of the HOST entries' transfer paths (development aid, GPU box only): matrices of 9 .. 48 MB with a
random leading dimension, random tie structures and missingness per column, all pairs or an unsorted pair list with both
orientations and self pairs, through icikt_pairs_f64 / icikt_matrix_f64
  (a) in one piece (plan key pipe=0: one copy, one pre-pass, one pair-kernel launch) and
  (b) pipelined by chunks (pipe=1), staged through the library's pinned buffers or -- matrix and result arrays in
      hipHostMalloc memory, ICIKT_FLAG_HOST_PINNED -- copied directly, with a random pairs-per-wave override;
the two must give the same bytes (results, counts, reasons; the five matrices and `keep`).  The one-piece path is what
tools/fuzz_gpu.py checks against the oracle.

    python tools/fuzz_pipe.py [cases] [seed]
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np

import fuzz_gpu
from icikendalltau_amd import _lib


def call_pairs(ctx, buf, n, S, ld, pi, pj, persp, want_counts, pinned=False):
    L = _lib.lib()
    P = S * (S - 1) // 2 if pi is None else len(pi)
    alloc = _lib.pinned_empty if pinned else np.empty
    out = alloc((P, 4))
    cnt = alloc((P, len(_lib.CNT_FIELDS)), dtype=np.int64) if want_counts else None
    rsn = alloc(P, dtype=np.int32)
    if cnt is not None:
        cnt[...] = 0
    rsn[...] = 0
    p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    rc = L.icikt_pairs_f64(ctx._h, p(buf), n, S, ld, p(pi), p(pj), P, persp, 0, 0, _lib.FLAG_HOST_PINNED if pinned else 0,
                           p(out), p(cnt), p(rsn))
    assert rc == 0, (rc, ctx.last_error() if hasattr(ctx, "last_error") else "")
    return out, cnt, rsn


def call_matrix(ctx, buf, n, S, ld, gna, pi, pj, persp, scale_max, diag_good, pinned=False):
    L = _lib.lib()
    P = 0 if pi is None else len(pi)
    alloc = _lib.pinned_empty if pinned else np.empty
    out5 = alloc((5, S, S))
    keep = alloc((S, n), dtype=np.uint8)
    keep[...] = 0
    rc5 = np.zeros(5, dtype=np.int64)
    p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    rc = L.icikt_matrix_f64(ctx._h, p(buf), n, S, ld, p(gna), len(gna), p(pi), p(pj), P, persp, 0, 0,
                            _lib.FLAG_HOST_PINNED if pinned else 0, int(scale_max),
                            int(diag_good), p(out5), p(keep), p(rc5))
    assert rc == 0, rc
    return out5, keep, rc5


def one_case(ctx, rng, case):
    mb = float(rng.uniform(9, 48))
    n = int(rng.choice([int(rng.integers(1500, 9000)), int(rng.integers(9000, 20000)), int(rng.integers(20000, 45000))]))
    S = max(3, int(mb * 2 ** 20 / (8 * n)))
    ld = n + int(rng.choice([0, 0, 1, 7, 64, 1000]))
    fuzz_gpu.MID = True   # all ten value models
    buf = np.full((S, ld), 7.0)            # row-major (S, ld) == column-major ld x S
    plain = rng.random() < 0.5             # half of the cases: mostly continuous columns with some missing values
    for c in range(S):
        if plain and rng.random() < 0.9:
            x = rng.standard_normal(n)
            x[rng.random(n) < 0.03] = np.nan
        else:
            x = fuzz_gpu.make_column(rng, n)
        buf[c, :n] = x
    persp = int(rng.integers(0, 2))
    pi = pj = None
    if rng.random() < 0.4:
        m = int(rng.integers(1, 4000))
        pi = rng.integers(0, S, m).astype(np.int32)
        pj = rng.integers(0, S, m).astype(np.int32)
    what = "matrix" if rng.random() < 0.3 else "pairs"
    plan = {"pipe": 1, "np": str(rng.choice(["", "1", "2"]))}
    pinned = bool(rng.random() < 0.35)
    plan = {k: v for k, v in plan.items() if v != ""}
    desc = f"case {case}: n={n} S={S} ld={ld} {what} persp={persp} list={None if pi is None else len(pi)} plan={plan} pinned={pinned}"
    gna = np.array([np.nan, np.inf, 0.0])
    sm, dg = bool(rng.random() < 0.7), bool(rng.random() < 0.7)
    res = []
    pbuf = None
    if pinned:
        pbuf = _lib.pinned_empty(buf.shape)
        pbuf[...] = buf
    for pl, pin in (({"pipe": 0}, False), (plan, pinned)):
        ctx.debug_set_plan(pl)
        src = pbuf if pin else buf
        if what == "pairs":
            res.append(call_pairs(ctx, src, n, S, ld, pi, pj, persp, want_counts=True, pinned=pin))
        else:
            res.append(call_matrix(ctx, src, n, S, ld, gna, pi, pj, persp, sm, dg, pinned=pin))
    ctx.debug_set_plan(None)
    for a, b in zip(*res):
        if not np.array_equal(a.view(np.uint8), b.view(np.uint8)):
            np.save(f"gpurun_out/fuzz_pipe_fail_{case}.npy", buf)
            return "FAIL", desc + " -> bytes differ"
    return "ok", desc


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    ctx = _lib.Context(0)
    os.makedirs("gpurun_out", exist_ok=True)
    tally = {"ok": 0, "FAIL": 0}
    t0 = time.time()
    for c in range(cases):
        status, desc = one_case(ctx, rng, c)
        tally[status] += 1
        if status != "ok":
            print(status, desc, flush=True)
        if c % 10 == 9:
            print(f"[{time.time() - t0:6.1f}s] {c + 1} cases: {tally}", flush=True)
    print("done:", tally, flush=True)
    sys.exit(1 if tally["FAIL"] else 0)


if __name__ == "__main__":
    main()
