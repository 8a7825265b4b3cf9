#!/usr/bin/env python3
"""bench.py -- all-pairs ICI-Kendall-tau throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (the configuration the metric is quoted on): synthetic 10 000 features x
1 024 samples, numpy default_rng(4).standard_normal, per column the 1 000 smallest values missing
(10 % left-censored; tie group < 1024 so the reference's int32 tie sums do not wrap),
perspective = "global": 523 776 column pairs, each a pair of length-10 000 vectors.

One "step" = one full pass of the hot path over the matrix, input already resident in HBM:
  K0 per-column pre-pass (N > 1: each rank sorts S/N columns, then one RCCL all-gather of the prepared state;
  falls back to every rank sorting all columns) -> K1 pair kernel over this rank's contiguous block of the
  combn-ordered pair list (the reference's `core` chunks, R/kendalltau.R:250-255) -> K2 epilogue ->
  (N > 1) RCCL gather of the P/N x 4 results to rank 0.
value = pairs of the whole job / wall time (max over ranks); total work is fixed, so scaling = strong.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def make_matrix(n_feat: int, n_samp: int, n_na: int, seed: int) -> np.ndarray:
    """BASELINE.md synthetic config: column-major float64, per column the n_na smallest -> NaN."""
    rng = np.random.default_rng(seed)
    X = np.asfortranarray(rng.standard_normal((n_feat, n_samp)))
    if n_na:
        idx = np.argpartition(X, n_na, axis=0)[:n_na]
        np.put_along_axis(X, idx, np.nan, axis=0)
    return X


def host_cores() -> int:
    """Cores this process may actually use: the cgroup CPU quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    env = os.environ.get("ICIKT_CPU_BASELINE_CORES")
    return int(env) if env else n


def cpu_baseline(X: np.ndarray, P_total: int, sample_pairs: int, seed: int = 0):
    """The CPU restatement (oracle/, kind = "port") on this box's host cores over a bounded random
    sample of the same pair list; one C thread per core (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle as O
    O.lib()
    S = X.shape[1]
    rng = np.random.default_rng(seed)
    iu, ju = np.triu_indices(S, k=1)
    sel = rng.choice(P_total, size=min(sample_pairs, P_total), replace=False)
    pi, pj = iu[sel].astype(np.int32), ju[sel].astype(np.int32)
    cores = host_cores()
    chunks = np.array_split(np.arange(len(sel)), cores * 4)

    def work(ix):
        return O.ici_pairs(X, pi[ix], pj[ix], "global", want_counts=True)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        res = list(ex.map(work, chunks))
    dt = time.perf_counter() - t0
    out = np.concatenate([r[0] for r in res])
    cnt = np.concatenate([r[1] for r in res])
    return {"value": len(sel) / dt, "unit": "column-pairs/s", "cores": cores, "kind": "port",
            "sample": f"{len(sel)} random pairs of the same 10000x1024 matrix, {dt:.2f} s wall on {cores} threads "
                      f"({dt * cores / len(sel) * 1e3:.2f} core-ms per pair)"}, sel, out, cnt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-feat", type=int, default=10000)
    ap.add_argument("--n-samp", type=int, default=1024)
    ap.add_argument("--n-na", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=4)
    ap.add_argument("--cpu-sample", type=int, default=12000, help="pairs timed on the CPU baseline (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node "
                             f"{args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the product path has no CPU fallback)"
    # ICIKT_BENCH_BACKEND=gloo is a REHEARSAL mode for a one-GPU box: every rank uses the visible device
    # and the gather goes through host memory.  The driver's N > 1 runs use nccl (= RCCL over xGMI).
    backend = os.environ.get("ICIKT_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from icikendalltau_amd import _lib
    ctx = _lib.Context(dev_index)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    n, S = args.n_feat, args.n_samp
    X = make_matrix(n, S, args.n_na, args.seed)
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)  # (S, n) row-major == n x S column-major
    P_total = S * (S - 1) // 2
    n_each = -(-P_total // world)  # ceiling(n_todo / ncore), R/kendalltau.R:250
    begin, end = min(P_total, rank * n_each), min(P_total, (rank + 1) * n_each)
    ctx.set_pairs_combn(S, begin, end)
    P_local = end - begin
    out_local = torch.full((n_each, 4), float("nan"), dtype=torch.float64, device=dev)
    comm_dev = dev if backend == "nccl" else torch.device("cpu")
    gathered = ([torch.empty((n_each, 4), dtype=torch.float64, device=comm_dev) for _ in range(world)]
                if (world > 1 and rank == 0) else None)
    flags = _lib.FLAG_TIMING

    class _DevBytes:
        """Zero-copy view of a device allocation of the C library as a torch uint8 tensor."""

        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    # N > 1: each rank sorts only its S/N columns (K0); order, the bitsets and stats are all-gathered (RCCL over
    # xGMI) and the rest of the prepared state is rebuilt locally, instead of every rank repeating the whole
    # pre-pass; falls back to the replicated pre-pass if that cannot be set up.
    prep_mode = "single"
    cols_per = 2 * -(-S // (2 * world))  # even: the rec table interleaves column pairs
    alloc_cols = cols_per * world
    c0, c1 = min(S, rank * cols_per), min(S, (rank + 1) * cols_per)
    shards = None

    def prepare_sharded():
        ctx.prepare_cols_dev(dX.data_ptr(), n, S, n, c0, c1, alloc_cols, flags)
        exchange_shards()
        # rec / hirow / tgroups of the received columns are functions of their order and gflag: rebuilt here
        # instead of being sent (24 KB instead of 104 KB per column cross xGMI)
        if c0 > 0:
            ctx.expand_cols_dev(0, min(c0, S), flags)
        if c1 < S:
            ctx.expand_cols_dev(c1, S, flags)

    def exchange_shards():
        for full, nbytes in shards:
            mine = full[rank * nbytes:(rank + 1) * nbytes]
            if backend == "nccl":
                dist.all_gather_into_tensor(full, mine.clone())
            else:  # rehearsal: through host memory
                parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(parts, mine.cpu())
                full.copy_(torch.cat(parts))

    if world > 1 and os.environ.get("ICIKT_BENCH_PREP", "sharded") == "sharded":
        # local part first (no collective inside), then the ranks agree: either all of them run the sharded
        # pre-pass or none does -- a rank that falls back alone would leave the others inside a collective
        err = None
        try:
            ctx.prepare_cols_dev(dX.data_ptr(), n, S, n, c0, c1, alloc_cols, 0)
            arrays = ctx.prep_arrays()
            shards = [(torch.as_tensor(_DevBytes(arrays[i][0], arrays[i][1] * alloc_cols), device=dev),
                       arrays[i][1] * cols_per) for i in _lib.PREP_EXCHANGE]
        except Exception as e:  # noqa: BLE001
            err, shards = e, None
        ok = torch.tensor([0 if shards is None else 1], dtype=torch.int32, device=comm_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            try:
                prepare_sharded()  # one trial pass before the timed region
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                err = e
            ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=comm_dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            prep_mode = "sharded+allgather"
        else:
            print(f"[bench] rank {rank}: sharded pre-pass unavailable ({err!r}); every rank runs the whole pre-pass",
                  file=sys.stderr)
            shards = None
    if world > 1 and shards is None:
        prep_mode = "replicated"

    def prepare():
        if shards is None:
            ctx.prepare_dev(dX.data_ptr(), n, S, n, flags)
        else:
            prepare_sharded()

    def step():
        prepare()
        ctx.run_dev(_lib.PERSPECTIVE["global"], _lib.ALTERNATIVE["two.sided"], False, flags, out_local.data_ptr())
        if world > 1:
            # every rank's P/N x 4 results to rank 0 (RCCL over xGMI)
            dist.gather(out_local if backend == "nccl" else out_local.cpu(), gathered, dst=0)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.reset_timers()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    k_ms = {name: ctx.kernel_ms(k) for name, k in (("prepare", _lib.K_PREPARE), ("pairs", _lib.K_PAIRS),
                                                    ("epilogue", _lib.K_EPILOGUE))}
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = P_total * args.steps / elapsed
        # roofline of the dominant kernel (K1): algorithmic bytes per launch = pairs in the launch x (16 n + 32)
        # (two float64 columns in, four float64 out per pair: SURVEY.md section 8(d)) / avg launch duration
        k1_ms, k1_n = k_ms["pairs"]
        k1_avg_s = (k1_ms / max(k1_n, 1)) / 1e3
        alg_bytes = P_local * (16 * n + 32)
        achieved = alg_bytes / k1_avg_s / 1e9 if k1_avg_s > 0 else 0.0
        traffic = None
        prof = os.path.join(ROOT, "profiles", "k1_hbm_traffic.json")
        if os.path.exists(prof):
            try:
                with open(prof) as f:
                    pj = json.load(f)
                if pj.get("n_feat") == n and pj.get("n_samp") == S and pj.get("pairs_per_launch") == P_local:
                    traffic = pj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "column-pairs/s (+ full-matrix wall time) at 10k feat x 1k samp",
            "value": value, "unit": "column-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int32 rank/popcount counting + f64 epilogue", "data": "synthetic",
            "config": {"workload": f"c4: {n} features x {S} samples, {args.n_na} smallest per column missing, "
                                   f"perspective=global, {P_total} column pairs",
                       "pairs": P_total, "n_feat": n, "n_samp": S, "sharding": f"combn-order blocks over {world} rank(s)", "pre_pass": prep_mode},
            "full_matrix_wall_ms": ms_per_step,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": "k1_pairs",
                         "avg_launch_ms": k1_avg_s * 1e3, "algorithmic_bytes_per_launch": alg_bytes},
            "kernel_ms_per_step": {k: (v[0] / max(v[1], 1)) for k, v in k_ms.items()},
        }
        if world == 1 and args.cpu_sample > 0:
            base, sel, ref_out, ref_cnt = cpu_baseline(X, P_total, args.cpu_sample)
            line["cpu_baseline"] = base
            got = out_local[:P_local].cpu().numpy()[sel]
            line["check"] = {"pairs_checked_against_oracle": int(len(sel)),
                             "max_abs_diff": float(np.nanmax(np.abs(got - ref_out)))}
        if world > 1:
            # the assembled result: rank blocks concatenated in combn order (no NaN may be left in real pairs)
            full = torch.cat([g[:min(n_each, P_total - r * n_each)] for r, g in enumerate(gathered)]).cpu().numpy()
            line["check"] = {"assembled_pairs": int(full.shape[0]), "nan_rows": int(np.isnan(full[:, 0]).sum())}
            if args.cpu_sample > 0:
                from oracle import oracle as O
                rng = np.random.default_rng(1)
                sel = rng.choice(P_total, size=min(2000, P_total), replace=False)
                iu, ju = np.triu_indices(S, k=1)
                ref, _c, _r = O.ici_pairs(X, iu[sel], ju[sel], "global", want_counts=False)
                line["check"].update(pairs_checked_against_oracle=int(len(sel)),
                                     max_abs_diff=float(np.nanmax(np.abs(full[sel] - ref))))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
