"""Randomised parity sweep: shapes, tie structures, missingness patterns and K1 launch-plan overrides drawn at
random (tools/fuzz_gpu.py), HIP path vs the CPU oracle; counts bit-exact, doubles within 1e-10."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("seed,mode,cases", [(11, "", 120), (12, "", 120), (13, "", 120), (14, "big", 40),
                                             (15, "mid", 60), (16, "ext", 300)])
def test_random_sweep(plan_ctx, seed, mode, cases, tmp_path, monkeypatch):
    import fuzz_gpu

    monkeypatch.setattr(fuzz_gpu, "BIG", mode == "big")  # long columns only (10 000 .. 65 535 rows)
    monkeypatch.setattr(fuzz_gpu, "MID", mode == "mid")  # 16 000 .. 31 000 rows, two more tie models
    monkeypatch.setattr(fuzz_gpu, "EXT", mode == "ext")  # the default lengths with those models

    monkeypatch.chdir(tmp_path)          # failing inputs are dumped under ./gpurun_out
    os.makedirs("gpurun_out", exist_ok=True)
    rng = np.random.default_rng(seed)
    bad = []
    for case in range(cases):
        status, desc = fuzz_gpu.one_case(plan_ctx, rng, case)
        if status == "FAIL":
            bad.append(desc)
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("mode,seed,cases", [("", 21, 400), ("multi2", 22, 150)])
def test_random_front_end_sweep(hip_ctx, mode, seed, cases, tmp_path, monkeypatch):
    """tools/fuzz_matrix.py: ici_kendalltau through the one-call matrix entry and through the pair-list route, kt_fast,
    pairwise_completeness on the HIP engine (one device; two ranks on it) against the same front-end on the oracle."""
    import fuzz_matrix
    from icikendalltau_amd import api
    from tests.oracle_engine import OracleEngine

    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(seed)
    hip = api.MultiHipEngine(devices=[0, 0]) if mode == "multi2" else api.HipEngine()
    orc = OracleEngine()
    bad = []
    for case in range(cases):
        status, desc = fuzz_matrix.one_case(rng, case, hip, orc)
        if status == "FAIL":
            bad.append(desc)
    assert not bad, "\n".join(bad)


def test_random_transfer_path_sweep(plan_ctx, tmp_path, monkeypatch):
    """tools/fuzz_pipe.py: matrices of 9 .. 48 MB with a random leading dimension through icikt_pairs_f64 /
    icikt_matrix_f64 in one piece and pipelined by chunks (random transfer mode and pairs per wave): the same bytes."""
    import fuzz_gpu
    import fuzz_pipe

    monkeypatch.chdir(tmp_path)
    os.makedirs("gpurun_out", exist_ok=True)
    monkeypatch.setattr(fuzz_gpu, "MID", False)   # (fuzz_pipe switches the ten value models on)
    rng = np.random.default_rng(31)
    bad = []
    for case in range(40):
        status, desc = fuzz_pipe.one_case(plan_ctx, rng, case)
        if status == "FAIL":
            bad.append(desc)
    assert not bad, "\n".join(bad)
