"""Multi-GPU path: row-range ("morsel") sharding of a scan and the group-by merge.

Each rank scans its own row range to a dense partial aggregate table laid out
[ n_min words | n_max words | n_sum words ] (include/resql_hip.h, rsq_query_execute_partial); the only exchange
step of the path is the merge of that table.  Integer min / max / sum are order-independent, so the merged table —
and the result — is bit-identical to the single-GPU run.
On GPUs the process group is RCCL over xGMI (backend "nccl"); the same code runs over gloo in the CPU tests.

Two merge strategies, chosen by table size:
  * small tables (Q1: 42 words) are latency-bound: ONE all-gather of every rank's table, then the three segment
    reductions locally (world x words is still tiny), instead of one all-reduce per segment;
  * large tables (2^20 groups: tens of MB) are bandwidth-bound: one ring all-reduce per non-empty segment, which moves
    2(N-1)/N of the table per rank instead of N-1 copies of it.
"""
from __future__ import annotations

from typing import Tuple

GATHER_LIMIT_WORDS = 1 << 14      # up to 128 KiB per rank goes through the single all-gather


def shard_rows(n_total: int, world: int, rank: int, tile: int = 128) -> Tuple[int, int]:
    """row range [row0, row0 + n) of `rank`: equal shards whose boundaries fall on 128-row wave tiles (the vector
    loads of the scan kernel need 16-byte aligned column offsets); the last rank takes the remainder"""
    per = (n_total // (tile * world)) * tile
    row0 = rank * per
    n = per if rank < world - 1 else n_total - per * (world - 1)
    return row0, n


def allreduce_partial(dist, partial, n_min: int, n_max: int, n_sum: int) -> None:
    """in-place merge of a partial aggregate table (1-D int64 tensor) across the ranks of `dist`:
    one all-reduce per non-empty segment"""
    if n_min:
        dist.all_reduce(partial[:n_min], op=dist.ReduceOp.MIN)
    if n_max:
        dist.all_reduce(partial[n_min:n_min + n_max], op=dist.ReduceOp.MAX)
    if n_sum:
        dist.all_reduce(partial[n_min + n_max:n_min + n_max + n_sum], op=dist.ReduceOp.SUM)


class PartialMerger:
    """Merges `partial` (this rank's table, a 1-D int64 tensor that stays bound to the query) across all ranks, in
    place.  Everything is enqueued on the current stream of `partial`'s device: no host synchronisation here."""

    def __init__(self, dist, partial, n_min: int, n_max: int, n_sum: int, world: int, always_collective: bool = False):
        import torch
        self.dist, self.partial = dist, partial
        self.n_min, self.n_max, self.n_sum, self.world = n_min, n_max, n_sum, world
        words = n_min + n_max + n_sum
        assert partial.numel() == words and partial.dtype == torch.int64
        self.collective = world > 1 or always_collective       # always_collective: run the exchange even with one rank (tests)
        self.gather = self.collective and words <= GATHER_LIMIT_WORDS
        self.all = torch.empty((world, words), dtype=torch.int64, device=partial.device) if self.gather else None
        self.strategy = "single rank" if not self.collective else ("one all-gather + local segment reductions" if self.gather
                                                         else "one all-reduce per segment (min | max | sum)")

    def merge(self) -> None:
        import torch
        if not self.collective:
            return
        if not self.gather:
            allreduce_partial(self.dist, self.partial, self.n_min, self.n_max, self.n_sum)
            return
        self.dist.all_gather_into_tensor(self.all.view(-1), self.partial)
        a, b = self.n_min, self.n_min + self.n_max
        if self.n_min:
            torch.amin(self.all[:, :a], dim=0, out=self.partial[:a])
        if self.n_max:
            torch.amax(self.all[:, a:b], dim=0, out=self.partial[a:b])
        if self.n_sum:
            torch.sum(self.all[:, b:], dim=0, out=self.partial[b:])
