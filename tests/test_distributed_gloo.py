"""N > 1 path on CPU: world_size 2 over gloo.  The sharding (`core` chunks = ranks), the all_gather of
the per-rank results and the reassembly are the product's code (icikendalltau_amd/api.py); the arithmetic
is supplied by the checker engine because this box has no GPU."""
import os
import pickle
import socket
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, X, names, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from icikendalltau_amd import api
    from tests.oracle_engine import OracleEngine
    res = api.ici_kendalltau(X, perspective="global", return_matrix=False, colnames=names, engine=OracleEngine())
    inc = api.ici_kendalltau(X, include_only=["s2", "s5"], colnames=names, engine=OracleEngine())
    with open(os.path.join(outdir, f"r{rank}.pkl"), "wb") as f:
        pickle.dump({"df": res["cor"], "inc": inc["cor"].to_numpy()}, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world_size_2_matches_single_process():
    import torch.multiprocessing as mp
    from icikendalltau_amd import api
    from tests.oracle_engine import OracleEngine
    rng = np.random.default_rng(0)
    X = rng.standard_normal((200, 9))
    X[rng.random(X.shape) < 0.1] = np.nan
    names = [f"s{i}" for i in range(1, 10)]
    single = api.ici_kendalltau(X, perspective="global", return_matrix=False, colnames=names, engine=OracleEngine())["cor"]
    single_inc = api.ici_kendalltau(X, include_only=["s2", "s5"], colnames=names, engine=OracleEngine())["cor"].to_numpy()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), X, names, d), nprocs=2, join=True)
        outs = [pickle.load(open(os.path.join(d, f"r{r}.pkl"), "rb")) for r in range(2)]
    for o in outs:  # every rank ends with the full result, as the single-process call does
        df = o["df"]
        assert list(df["s1"]) == list(single["s1"]) and list(df["s2"]) == list(single["s2"])
        for col in ("raw", "pvalue", "taumax", "completeness", "cor"):
            assert np.array_equal(df[col].to_numpy(), single[col].to_numpy(), equal_nan=True)
        # 36 pairs over 2 ranks: ceiling(36 / 2) = 18 per core (R/kendalltau.R:250-255), diagonal rows core 0
        assert list(df["core"]) == [1] * 18 + [2] * 18 + [0] * 9
        assert np.array_equal(o["inc"], single_inc)
