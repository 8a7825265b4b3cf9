"""Randomised parity sweep of the N > 1 path (development aid, GPU box only): every rank of a torch.distributed group
(gloo rehearsal: all ranks share the one visible GPU, collectives through host memory) runs api.ici_kendalltau with the
HIP engine -- column-sharded pre-pass, all-gather of order + meta, the rebuild kernel for the received columns
(k0_expand: rec / hirow / tie-group lists / tie program), the rank's pair block, all-gather of the padded blocks -- on
random shapes, tie structures, missingness and pair filters, and compares what it ends with against the oracle run
in-process on the whole pair list.  Launch:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29517 \
        tools/fuzz_dist.py [cases] [seed]
"""
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch  # noqa: F401
import torch.distributed as dist

import fuzz_gpu
from oracle import oracle as O

ATOL = 1e-10


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from icikendalltau_amd import api
    eng = api.HipEngine()
    rng = np.random.default_rng(seed)          # the same stream on every rank: the same cases
    fuzz_gpu.MID = True                        # all ten value models
    os.makedirs("gpurun_out", exist_ok=True)
    bad = 0
    modes = {}
    t0 = time.time()
    for case in range(cases):
        r = rng.random()
        n = int(rng.integers(2, 300)) if r < 0.3 else int(rng.integers(300, 6000)) if r < 0.8 else int(rng.integers(6000, 40000))
        S = int(rng.integers(2, 40)) if n < 6000 else int(rng.integers(2, 12))
        X = np.asfortranarray(np.stack([fuzz_gpu.make_column(rng, n) for _ in range(S)], axis=1))
        names = [f"s{i}" for i in range(S)]
        kw = {"perspective": str(rng.choice(["global", "local"])), "global_na": [float("nan")]}
        if rng.random() < 0.3:
            kw["include_only"] = [names[i] for i in rng.choice(S, int(rng.integers(1, S + 1)), replace=False)]
        if rng.random() < 0.3:
            kw["diag_good"] = False
        desc = f"case {case}: n={n} S={S} {kw}"
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                res = api.ici_kendalltau(X, colnames=names, engine=eng, return_matrix=False, **kw)["cor"]
            except ValueError as e:
                if "No comparisons" in str(e):
                    continue
                raise
        modes[eng.pre_pass] = modes.get(eng.pre_pass, 0) + 1
        pi, pj, _core = api.setup_comparisons(names, kw.get("include_only"), kw.get("diag_good", True), ncore=world)
        P = len(pi)
        ref, _cnt, _rsn = O.ici_pairs(X, pi, pj, kw["perspective"], "two.sided", False, want_counts=False)
        got = np.stack([res[k].to_numpy()[:P] for k in ("raw", "pvalue", "taumax", "completeness")], axis=1)
        why = None
        if not np.array_equal(np.isnan(got), np.isnan(ref)):
            why = "NaN pattern"
        else:
            fin = ~np.isnan(ref) & ~np.isinf(ref)
            if fin.any() and np.max(np.abs(got[fin] - ref[fin])) > ATOL:
                why = f"max abs diff {np.max(np.abs(got[fin] - ref[fin]))}"
        if why:
            bad += 1
            print(f"[rank {rank}] FAIL {desc} -> {why}", flush=True)
            if rank == 0:
                np.save(f"gpurun_out/fuzz_dist_fail_{case}.npy", X)
        if rank == 0 and case % 25 == 24:
            print(f"[{time.time() - t0:6.1f}s] {case + 1} cases, {bad} failures on rank 0; pre-pass modes {modes}", flush=True)
    t = torch.tensor([bad], dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        print(f"done: world {world}, {cases} cases, failures over all ranks: {int(t.item())}; pre-pass modes {modes}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(1 if int(t.item()) else 0)


if __name__ == "__main__":
    main()
