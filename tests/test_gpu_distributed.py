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


@pytest.mark.timeout(900)
def test_bench_sequence_two_ranks_gloo(tmp_path):
    res = _torchrun(["bench.py", "--gpus", "2", "--config", "c3", "--steps", "2", "--warmup", "1"],
                    {"ICIKT_BENCH_BACKEND": "gloo"}, 840)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_gloo2_c3.json"), "w") as f:
        json.dump(line, f)
    assert line["n_gpus"] == 2 and line["config"]["pre_pass"] == "sharded+allgather"
    assert line["check"]["ok"] and line["check"]["assembled_pairs"] == 32640 and line["check"]["nan_rows"] == 0
    assert line["check"]["max_abs_diff"] <= 1e-10


@pytest.mark.timeout(900)
def test_bench_sequence_one_rank_rccl(tmp_path):
    """The same sequence over the real nccl (= RCCL) backend with a process group of ONE rank: zero-copy torch views of
    the library's device arrays go through all_gather_into_tensor and gather on the GPU (the two-rank rehearsal
    above moves them through host memory)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ICIKT_BENCH_FORCE_DIST="1")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29534", "bench.py", "--gpus", "1", "--config", "c3", "--steps", "2",
           "--warmup", "1"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=840)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["pre_pass"] == "sharded+allgather"
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
