"""Randomised parity sweep: shapes, tie structures, missingness patterns and K1 launch-plan overrides drawn at
random (tools/fuzz_gpu.py), HIP path vs the CPU oracle; counts bit-exact, doubles within 1e-10."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

_ENV = ("ICIKT_K1_NP", "ICIKT_K1_PEND", "ICIKT_K1_HALF", "ICIKT_K1_TGMAX")


@pytest.mark.parametrize("seed,big,cases", [(11, False, 120), (12, False, 120), (13, False, 120), (14, True, 40)])
def test_random_sweep(hip_ctx, seed, big, cases, tmp_path, monkeypatch):
    import fuzz_gpu

    monkeypatch.setattr(fuzz_gpu, "BIG", big)  # long columns only (10 000 .. 65 535 rows)

    monkeypatch.chdir(tmp_path)          # failing inputs are dumped under ./gpurun_out
    os.makedirs("gpurun_out", exist_ok=True)
    saved = {k: os.environ.get(k) for k in _ENV}
    rng = np.random.default_rng(seed)
    try:
        bad = []
        for case in range(cases):
            status, desc = fuzz_gpu.one_case(hip_ctx, rng, case)
            if status == "FAIL":
                bad.append(desc)
        assert not bad, "\n".join(bad)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
