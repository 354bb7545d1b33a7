"""Multi-GPU path: row-range ("morsel") sharding of a scan and the group-by merge.

Each rank scans its own row range to a dense partial aggregate table laid out
[ n_min words | n_max words | n_sum words ] (include/resql_hip.h, rsq_query_execute_partial); the only exchange
step of the path is the merge of that table: one all-reduce per non-empty segment.  Integer min / max / sum
are order-independent, so the merged table — and the result — is bit-identical to the single-GPU run.
On GPUs the process group is RCCL over xGMI (backend "nccl"); the same code runs over gloo in the CPU tests.
"""
from __future__ import annotations

from typing import Tuple


def shard_rows(n_total: int, world: int, rank: int, tile: int = 128) -> Tuple[int, int]:
    """row range [row0, row0 + n) of `rank`: equal shards whose boundaries fall on 128-row wave tiles (the vector
    loads of the scan kernel need 16-byte aligned column offsets); the last rank takes the remainder"""
    per = (n_total // (tile * world)) * tile
    row0 = rank * per
    n = per if rank < world - 1 else n_total - per * (world - 1)
    return row0, n


def allreduce_partial(dist, partial, n_min: int, n_max: int, n_sum: int) -> None:
    """in-place merge of a partial aggregate table (1-D int64 tensor) across the ranks of `dist`"""
    if n_min:
        dist.all_reduce(partial[:n_min], op=dist.ReduceOp.MIN)
    if n_max:
        dist.all_reduce(partial[n_min:n_min + n_max], op=dist.ReduceOp.MAX)
    if n_sum:
        dist.all_reduce(partial[n_min + n_max:n_min + n_max + n_sum], op=dist.ReduceOp.SUM)
