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


# ------------------------------------------------------------------------------------------------------------
# The sharded pre-pass sequence that bench.py and api.HipEngine run at N > 1 (icikendalltau_amd/sharding.py:
# prepare own columns -> agree -> all-gather order + meta in place -> rebuild the received columns), on a
# recording context whose "device arrays" are host buffers: slice offsets, evenness, the ragged last shard, the
# complement ranges handed to expand_cols_dev, the all-ranks-or-none fallback and the ragged result gather.
# ------------------------------------------------------------------------------------------------------------
class _RecordingCtx:
    BPC = {0: 48, 3: 40}   # bytes per column of `order` and `meta` (any sizes do)

    def __init__(self, rank, fail_prepare=False, fail_expand=False):
        self.rank, self.fail_prepare, self.fail_expand = rank, fail_prepare, fail_expand
        self.arrays = {}
        self.expanded = []
        self.prepared = []

    @staticmethod
    def pattern(arr_id, col, nbytes):
        return (np.arange(nbytes, dtype=np.uint32) * 7 + col * 13 + arr_id * 101).astype(np.uint8)

    def prepare_cols(self, c0, c1, alloc_cols, flags):
        if self.fail_prepare:
            raise RuntimeError("no memory on this rank")
        self.prepared.append((c0, c1, alloc_cols))
        for i, bpc in self.BPC.items():
            if i not in self.arrays:    # like the library's buffers, allocated once and reused by later passes
                self.arrays[i] = np.empty(alloc_cols * bpc, dtype=np.uint8)
            a = self.arrays[i]
            a[:] = 0xEE                 # other ranks' columns: garbage until exchanged
            for c in range(c0, c1):
                a[c * bpc:(c + 1) * bpc] = self.pattern(i, c, bpc)
        self.alloc_cols = alloc_cols

    def prep_arrays(self):
        out = [(0, 1)] * 5
        for i, bpc in self.BPC.items():
            out[i] = (self.arrays[i].ctypes.data, bpc)
        return out

    def expand_cols_dev(self, a, b, flags):
        if self.fail_expand:
            raise RuntimeError("launch failure")
        for i, bpc in self.BPC.items():     # what it rebuilds from must have arrived
            for c in range(a, b):
                assert np.array_equal(self.arrays[i][c * bpc:(c + 1) * bpc], self.pattern(i, c, bpc)), (i, c)
        self.expanded.append((a, b))


def _sharding_worker(rank, world, port, outdir, scenario):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from icikendalltau_amd import sharding
    S = 7
    ctx = _RecordingCtx(rank, fail_prepare=(scenario == "prepare_fails_on_rank_1" and rank == 1),
                        fail_expand=(scenario == "trial_fails_on_rank_0" and rank == 0))
    sp = sharding.ShardedPrepass(ctx, dist, torch.device("cpu"), via_host=True, view=sharding.host_bytes)
    ok = sp.setup(S, ctx.prepare_cols)
    res = {"ok": ok, "mode": sp.mode, "expanded": list(ctx.expanded), "prepared": list(ctx.prepared)}
    if ok:
        sp.run(0)   # a timed step repeats the same sequence
        res["expanded2"] = ctx.expanded[len(res["expanded"]):]
        res["complete"] = all(np.array_equal(ctx.arrays[i][c * b:(c + 1) * b], ctx.pattern(i, c, b))
                              for i, b in ctx.BPC.items() for c in range(S))
    # ragged result gather: 11 pairs over 2 ranks = blocks of 6 and 5
    P = 11
    begin, end, n_each = sharding.pair_block(P, rank, world)
    blk = torch.full((n_each, 4), float("nan"), dtype=torch.float64)
    blk[:end - begin] = torch.arange(begin, end, dtype=torch.float64)[:, None] * torch.ones(4, dtype=torch.float64)
    g0 = sharding.gather_blocks(dist, blk, n_each, torch.device("cpu"), True, to_all=False)
    gall = sharding.gather_blocks(dist, blk, n_each, torch.device("cpu"), True, to_all=True)
    res["gather_root"] = None if g0 is None else sharding.assemble(g0, P, n_each).numpy()
    res["gather_all"] = sharding.assemble(gall, P, n_each).numpy()
    with open(os.path.join(outdir, f"s{rank}.pkl"), "wb") as f:
        pickle.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("scenario", ["ok", "prepare_fails_on_rank_1", "trial_fails_on_rank_0"])
def test_sharded_prepass_sequence_world_2(scenario):
    import torch.multiprocessing as mp
    from icikendalltau_amd import sharding
    assert sharding.column_shard(7, 0, 2) == (0, 4, 4, 8) and sharding.column_shard(7, 1, 2) == (4, 7, 4, 8)
    assert sharding.column_shard(1024, 3, 8) == (384, 512, 128, 1024)
    assert sharding.column_shard(5, 3, 4) == (5, 5, 2, 8)               # more ranks than column pairs: empty shard
    assert sharding.pair_block(11, 0, 2) == (0, 6, 6) and sharding.pair_block(11, 1, 2) == (6, 11, 6)
    assert sharding.pair_block(523776, 7, 8) == (458304, 523776, 65472)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_sharding_worker, args=(2, _free_port(), d, scenario), nprocs=2, join=True)
        outs = [pickle.load(open(os.path.join(d, f"s{r}.pkl"), "rb")) for r in range(2)]
    for r, o in enumerate(outs):
        # the gathers work whatever the pre-pass decided
        assert np.array_equal(o["gather_all"][:, 0], np.arange(11.0))
        if r == 0:
            assert np.array_equal(o["gather_root"][:, 2], np.arange(11.0))
        else:
            assert o["gather_root"] is None
    if scenario == "ok":
        assert all(o["ok"] and o["mode"] == "sharded+allgather" and o["complete"] for o in outs)
        assert outs[0]["prepared"][0] == (0, 4, 8) and outs[1]["prepared"][0] == (4, 7, 8)
        assert outs[0]["expanded"] == [(4, 7)] and outs[1]["expanded"] == [(0, 4)]     # the complement of its shard
        assert outs[0]["expanded2"] == [(4, 7)] and outs[1]["expanded2"] == [(0, 4)]
    else:
        # one rank cannot: BOTH fall back (no rank is left alone inside a collective) and say so
        assert all((not o["ok"]) and o["mode"] == "replicated" for o in outs)


@pytest.mark.timeout(300)
def test_sharded_prepass_sequence_world_4():
    """Four ranks, seven columns: shards of two columns, the last one ragged (one column), and a rank's complement in
    two pieces (columns before and after its shard); 11 pairs in blocks of 3, the last block short."""
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_sharding_worker, args=(4, _free_port(), d, "ok"), nprocs=4, join=True)
        outs = [pickle.load(open(os.path.join(d, f"s{r}.pkl"), "rb")) for r in range(4)]
    assert all(o["ok"] and o["mode"] == "sharded+allgather" and o["complete"] for o in outs)
    assert [o["prepared"][0] for o in outs] == [(0, 2, 8), (2, 4, 8), (4, 6, 8), (6, 7, 8)]
    assert outs[0]["expanded"] == [(2, 7)] and outs[3]["expanded"] == [(0, 6)]
    assert outs[1]["expanded"] == [(0, 2), (4, 7)] and outs[2]["expanded"] == [(0, 4), (6, 7)]
    for r, o in enumerate(outs):
        assert np.array_equal(o["gather_all"][:, 0], np.arange(11.0))
        assert (o["gather_root"] is None) == (r != 0)
    assert np.array_equal(outs[0]["gather_root"][:, 3], np.arange(11.0))
