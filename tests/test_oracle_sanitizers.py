"""The CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on
the pool, so the CPU build is where they run).  The reference relies on int32 wrap-around in its tie sums
(count_rank_tie, src/kendallc.cpp:112-114, tie groups >= 1024 rows); the restatement performs that arithmetic
in unsigned, so UBSan has nothing to flag -- proven here on a wrapping input -- and ASan checks every buffer of
the oracle on ragged / empty / all-missing inputs."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _find_asan():
    out = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


@pytest.mark.timeout(600)
def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    asan = _find_asan()
    if asan is None:
        pytest.skip("gcc has no libasan in this image")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_build/libicikt_oracle_san.so"], check=True)
    lib = os.path.join(ROOT, "oracle", "_build", "libicikt_oracle_san.so")
    script = tmp_path / "drive.py"
    script.write_text(textwrap.dedent(f"""
        import ctypes, sys
        import numpy as np
        sys.path.insert(0, {ROOT!r})
        from oracle import oracle as O
        O._LIB_PATH = {lib!r}
        O._lib = None
        ref = ctypes.CDLL({os.path.join(ROOT, 'oracle', '_build', 'libicikt_oracle.so')!r})
        rng = np.random.default_rng(0)
        # (1) int32 wrap regime: tie groups of 1 500 and 2 000 rows in both columns (t(t-1)(2t+5) > INT_MAX)
        n = 6000
        x = rng.standard_normal(n); y = rng.standard_normal(n)
        x[:2000] = np.nan; y[1000:2500] = np.nan; x[3000:4500] = 1.0; y[4000:5600] = -2.0
        for persp in ("global", "local"):
            for compat in (True, False):
                out, cnt, rsn = O.ici_kt(x, y, persp, int32_compat=compat)
                assert rsn == 0 and np.isfinite(out).all()
        out_w, cnt_w, _ = O.ici_kt(x, y, "global", int32_compat=True)
        out_e, cnt_e, _ = O.ici_kt(x, y, "global", int32_compat=False)
        assert cnt_w["x1"] != cnt_e["x1"], "the input was meant to wrap the int32 tie sum"
        # (2) ragged / degenerate inputs
        for m in (0, 1, 2, 3, 64, 65, 1000):
            a = rng.standard_normal(m); b = rng.standard_normal(m)
            a[rng.random(m) < 0.3] = np.nan
            for persp in ("global", "local"):
                O.ici_kt(a, b, persp)
                O.ici_kt(a, np.full(m, np.nan), persp)
                O.ici_kt(a, np.ones(m), persp)
            if m:
                O.bruteforce(a, b)
        X = rng.standard_normal((300, 7)); X[rng.random(X.shape) < 0.2] = np.nan; X[:, 3] = np.round(X[:, 3])
        pi, pj = np.triu_indices(7, k=1)
        O.ici_pairs(X, pi, pj, "local")
        O.ici_pairs(X, pi[:0], pj[:0], "global")
        print("sanitized oracle ok")
    """))
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    res = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=540)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "sanitized oracle ok" in res.stdout
    assert "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr
