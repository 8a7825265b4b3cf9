"""One-process-per-GPU sharding of the all-pairs path over torch.distributed (backend "nccl" = RCCL over xGMI).

The reference spreads `core` chunks of the pair list over furrr workers and ships the whole matrix to each
(R/kendalltau.R:158, 250-255; R/utils.R:68-80).  Here a rank is a GPU:

  * the pair list is cut into `world` consecutive blocks of ceiling(P / world) pairs (= the `core` chunks);
  * the per-column pre-pass is sharded too: rank r sorts columns [c0, c1) only, then the two prepared-state arrays
    that carry the information (`order`, `meta`) are all-gathered in place and the rest (`rec`, `hirow`, `girow`,
    `tgroups`, the tie program and its step records) is rebuilt locally for the received columns (icikt_expand_cols_dev);
  * results are gathered to rank 0 (bench) or all-gathered (api: every rank returns the full result).

bench.py and api.HipEngine both drive this module, and tests/test_distributed_gloo.py runs it with world_size 2
over gloo on a recording context, so the slice arithmetic, the all-ranks-or-none fallback decision and the
collective sequence that run at N > 1 on the GPUs are the ones that are tested on CPU.  The same flow inside one
process (host threads instead of ranks, no torch) is icikt_pairs_multi_f64 (csrc/icikt_multi.cpp).
"""
from __future__ import annotations

import ctypes
import math
import sys

from . import _lib


def pair_block(n_pairs: int, rank: int, world: int):
    """(begin, end, n_each): block `rank` of ceiling(n_pairs / world) consecutive pairs (R/kendalltau.R:250-255)."""
    n_each = int(math.ceil(n_pairs / world)) if n_pairs > 0 else 0
    return min(n_pairs, rank * n_each), min(n_pairs, (rank + 1) * n_each), n_each


def column_shard(n_samp: int, rank: int, world: int):
    """(c0, c1, cols_per, alloc_cols): an even number of columns per rank (the rec table interleaves column pairs);
    the prepared-state arrays are allocated for alloc_cols = cols_per * world >= n_samp columns so that every
    rank's slice has the same size."""
    cols_per = 2 * int(math.ceil(n_samp / (2 * world)))
    return min(n_samp, rank * cols_per), min(n_samp, (rank + 1) * cols_per), cols_per, cols_per * world


class _DevBytes:
    """Zero-copy view of a device allocation of the C library as a torch uint8 tensor."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def device_bytes(ptr: int, nbytes: int, device):
    import torch
    with torch.cuda.device(device):
        return torch.as_tensor(_DevBytes(ptr, nbytes), device=device)


def host_bytes(ptr: int, nbytes: int, device=None):
    """The same for a host allocation (the recording context of the CPU tests)."""
    import torch
    return torch.frombuffer((ctypes.c_char * nbytes).from_address(ptr), dtype=torch.uint8)


class ShardedPrepass:
    """Column-sharded pre-pass of one rank.

    ctx            the rank's context (prepare_cols via `prepare_local`, prep_arrays, expand_cols_dev)
    dist           torch.distributed (initialised), rank / world from it
    device         torch.device the context computes on: collectives run on buffers of THIS device
    via_host       True: collectives go through host memory (gloo rehearsal on a one-GPU box, CPU tests)
    view           maps (ptr, nbytes, device) of a library array to a uint8 tensor (device_bytes / host_bytes)
    """

    def __init__(self, ctx, dist, device, via_host: bool, view=device_bytes):
        self.ctx, self.dist, self.device, self.via_host, self.view = ctx, dist, device, via_host, view
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.shards = None
        self.n_samp = None
        self.mode = "replicated"
        self.error = None

    # -- collective agreement: every rank runs the sharded pre-pass, or none does ---------------------------------
    def _all_ok(self, ok: bool) -> bool:
        import torch
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if self.via_host else self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return int(t.item()) == 1

    def setup(self, n_samp: int, prepare_local, sync=None) -> bool:
        """prepare_local(c0, c1, alloc_cols, flags) runs K0 on this rank's columns (no collective inside).  One
        trial pass decides, for all ranks together, whether the sharded pre-pass is used; a rank that failed alone
        would otherwise leave the others inside a collective."""
        self.n_samp = n_samp
        self.c0, self.c1, self.cols_per, self.alloc_cols = column_shard(n_samp, self.rank, self.world)
        self.prepare_local = prepare_local
        try:
            prepare_local(self.c0, self.c1, self.alloc_cols, 0)
            arrays = self.ctx.prep_arrays()
            self.shards = [(self.view(arrays[i][0], arrays[i][1] * self.alloc_cols, self.device), arrays[i][1] * self.cols_per)
                           for i in _lib.PREP_EXCHANGE]
        except Exception as e:  # noqa: BLE001
            self.error, self.shards = e, None
        if self._all_ok(self.shards is not None):
            try:
                self.run(0)  # one trial pass outside any timed region
                if sync is not None:
                    sync()
            except Exception as e:  # noqa: BLE001
                self.error = e
            if self._all_ok(self.error is None):
                self.mode = "sharded+allgather"
                return True
        if self.error is not None:
            print(f"[icikt] rank {self.rank}: sharded pre-pass unavailable ({self.error!r}); every rank runs the whole "
                  "pre-pass", file=sys.stderr)
        self.shards = None
        self.mode = "replicated"
        return False

    def exchange(self):
        """All-gather, in place, of every exchanged array: rank r's columns are slice r of its own array."""
        import torch
        for full, nbytes in self.shards:
            mine = full[self.rank * nbytes:(self.rank + 1) * nbytes]
            if not self.via_host:
                # in place: the rank's slice already sits where the collective would put it (RCCL's in-place form,
                # sendbuff == recvbuff + rank * count: no copy of the slice per step)
                self.dist.all_gather_into_tensor(full, mine)
            else:
                parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
                self.dist.all_gather(parts, mine.cpu())
                full.copy_(torch.cat(parts))

    def run(self, flags: int = 0, on_phase=None):
        """K0 on this rank's columns, exchange, rebuild of rec / hirow / girow / tgroups / the tie program for the columns received.
        on_phase(name): called after "k0" and after "exchange" (all-gather + rebuild) have been enqueued -- bench.py
        records an event on the launch stream there (per-rank phase times)."""
        self.prepare_local(self.c0, self.c1, self.alloc_cols, flags)
        if on_phase is not None:
            on_phase("k0")
        self.exchange()
        if self.c0 > 0:
            self.ctx.expand_cols_dev(0, min(self.c0, self.n_samp), flags)
        if self.c1 < self.n_samp:
            self.ctx.expand_cols_dev(self.c1, self.n_samp, flags)
        if on_phase is not None:
            on_phase("exchange")


def gather_blocks(dist, out_local, n_each: int, device, via_host: bool, to_all: bool):
    """Every rank's padded block (n_each rows) to rank 0 (list of world tensors there, None elsewhere), or to every
    rank when to_all.  Buffers live on `device`, never on an ambient current device."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    comm_dev = torch.device("cpu") if via_host else device
    src = out_local.cpu() if via_host else out_local
    assert src.shape[0] == n_each
    if to_all:
        gathered = [torch.empty_like(src, device=comm_dev) for _ in range(world)]
        dist.all_gather(gathered, src)
        return gathered
    gathered = [torch.empty_like(src, device=comm_dev) for _ in range(world)] if rank == 0 else None
    dist.gather(src, gathered, dst=0)
    return gathered


def assemble(gathered, n_pairs: int, n_each: int):
    """Concatenate the ranks' blocks in pair order, dropping the padding of the (only) short last block."""
    import torch
    return torch.cat([g[:max(0, min(n_each, n_pairs - r * n_each))] for r, g in enumerate(gathered)])
