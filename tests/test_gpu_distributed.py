"""The N > 1 path that ships, rehearsed with two ranks on the one visible GPU (backend gloo: collectives through
host memory; RCCL refuses two ranks on one device).  Child processes are started with torch.distributed.run; the
test process itself only reads their outputs.
  * bench.py's own sequence (sharding.ShardedPrepass + gather) on config c3, checked against the oracle by bench.py;
  * api.ici_kendalltau under a process group with the HIP engine (api._run_sharded -> HipEngine.pairs_block_dev):
    rank -> device mapping, sharded pre-pass, ragged blocks, every rank returns the full result.
The RCCL transport itself (nccl backend, distinct devices) can only run on the driver's multi-GPU node."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(script_args, env_extra, timeout):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533"] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def _bench_as_typed(args, env_extra, timeout):
    """`python bench.py --gpus N ...` as the driver types it for N = 1: no torch.distributed.run on the command line, no
    WORLD_SIZE in the environment -- bench.py starts its ranks itself (bench.spawn_ranks) and relays their line."""
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, "bench.py"] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(900)
def test_bench_sequence_two_ranks_gloo(tmp_path):
    res = _bench_as_typed(["--gpus", "2", "--config", "c3", "--steps", "2", "--warmup", "1"], {"ICIKT_BENCH_BACKEND": "gloo"}, 840)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # ONE JSON line on stdout, whatever the ranks printed
    line = json.loads(lines[0])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_gloo2_c3.json"), "w") as f:
        json.dump(line, f)
    assert line["n_gpus"] == 2 and line["config"]["pre_pass"] == "sharded+allgather"
    assert line["launched_by"]["child_rc"] == 0 and "--nproc-per-node 2" in line["launched_by"]["cmd"]
    assert line["backend"] == "gloo" and line["rccl_ranks"] is None
    assert set(line["rank_phase_ms"]) == {"k0", "exchange", "k1", "gather"}
    assert all(v["max"] >= v["min"] >= 0 for v in line["rank_phase_ms"].values()) and line["rank_phase_ms"]["k1"]["min"] > 0
    assert line["check"]["ok"] and line["check"]["assembled_pairs"] == 32640 and line["check"]["nan_rows"] == 0
    assert line["check"]["max_abs_diff"] <= 1e-10


@pytest.mark.timeout(900)
def test_bench_sequence_one_rank_rccl(tmp_path):
    """The same sequence over the real nccl (= RCCL) backend with a process group of ONE rank: zero-copy torch views of
    the library's device arrays go through all_gather_into_tensor and gather on the GPU (the two-rank rehearsal
    above moves them through host memory).  Self-launched too: `python bench.py --gpus 1` with ICIKT_BENCH_FORCE_DIST=1."""
    res = _bench_as_typed(["--gpus", "1", "--config", "c3", "--steps", "2", "--warmup", "1"], {"ICIKT_BENCH_FORCE_DIST": "1"}, 840)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["pre_pass"] == "sharded+allgather"
    assert line["backend"] == "nccl" and line["rccl_ranks"] == 1
    assert line["launched_by"]["child_rc"] == 0
    assert line["check"]["ok"] and line["check"]["assembled_pairs"] == 32640 and line["check"]["nan_rows"] == 0
    assert line["check"]["max_abs_diff"] <= 1e-10


@pytest.mark.timeout(900)
def test_api_run_sharded_two_ranks_gloo(tmp_path):
    res = _torchrun(["tests/dist_worker.py", str(tmp_path)], {}, 840)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    ref = np.load(tmp_path / "ref.npz")
    for r in range(2):
        meta = json.load(open(tmp_path / f"r{r}.json"))
        assert meta["world"] == 2 and meta["pre_pass"] == "sharded+allgather"
        assert meta["engine_device"] == 0                       # LOCAL_RANK modulo the visible devices
        assert meta["core"] == [1] * 127 + [2] * 126 + [0] * 23  # ceiling(253 / 2) per rank, diagonal rows core 0
        got = np.load(tmp_path / f"r{r}.npz")
        for k in ("raw", "pvalue", "taumax", "completeness", "cor", "loc"):
            assert np.allclose(got[k], ref[k], rtol=0, atol=1e-10, equal_nan=True), k
