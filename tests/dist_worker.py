"""Worker of tests/test_gpu_distributed.py: one rank of a two-rank gloo rehearsal on ONE GPU (every rank uses
the visible device; collectives go through host memory).  It runs the product's N > 1 path -- api.ici_kendalltau
under an initialised process group with the default HIP engine: column-sharded pre-pass, all-gather of order +
meta, pair block, all-gather of the padded result blocks -- and compares with the oracle engine run in-process."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    outdir = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from icikendalltau_amd import api
    from tests.oracle_engine import OracleEngine
    rng = np.random.default_rng(3)
    X = rng.standard_normal((4000, 23))                    # 23 columns: ragged column shards, 253 pairs: ragged blocks
    X[rng.random(X.shape) < 0.07] = np.nan
    X[:, 2] = np.round(X[:, 2] * 4)
    names = [f"s{i}" for i in range(X.shape[1])]
    eng = api.HipEngine()                                  # device from LOCAL_RANK modulo the visible devices
    res = api.ici_kendalltau(X, perspective="global", return_matrix=False, colnames=names, engine=eng)
    loc = api.ici_kendalltau(X, perspective="local", include_only=["s3", "s9"], colnames=names, engine=eng)
    dist.barrier()
    # the reference for both: the single-process path with the checker engine (no process group consulted: it is
    # computed by rank 0 alone after the group is gone)
    out = {"rank": rank, "world": world, "pre_pass": eng.pre_pass, "engine_device": eng.ctx.device,
           "current_device": torch.cuda.current_device(), "core": res["cor"]["core"].tolist()}
    np.savez(os.path.join(outdir, f"r{rank}.npz"), raw=res["cor"]["raw"].to_numpy(), pvalue=res["cor"]["pvalue"].to_numpy(),
             taumax=res["cor"]["taumax"].to_numpy(), completeness=res["cor"]["completeness"].to_numpy(),
             cor=res["cor"]["cor"].to_numpy(), loc=loc["cor"].to_numpy())
    dist.destroy_process_group()
    if rank == 0:
        ref = api.ici_kendalltau(X, perspective="global", return_matrix=False, colnames=names, engine=OracleEngine())
        refl = api.ici_kendalltau(X, perspective="local", include_only=["s3", "s9"], colnames=names, engine=OracleEngine())
        np.savez(os.path.join(outdir, "ref.npz"), raw=ref["cor"]["raw"].to_numpy(), pvalue=ref["cor"]["pvalue"].to_numpy(),
                 taumax=ref["cor"]["taumax"].to_numpy(), completeness=ref["cor"]["completeness"].to_numpy(),
                 cor=ref["cor"]["cor"].to_numpy(), loc=refl["cor"].to_numpy())
    with open(os.path.join(outdir, f"r{rank}.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
