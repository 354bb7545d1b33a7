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


def unify_shard_stats(dist, table, world: int, device=None) -> None:
    """Every rank holds one shard (`table`, an engine.DeviceTable) of the same table: all-gather the shards' statistics blobs
    and make each shard plan as the whole table (include/resql_hip.h rsq_table_unify_shard_stats) — one dense group layout on
    all ranks whatever their rows hold, the whole table's row count where the reference's table sizes matter.  Call before
    compiling; a no-op for one rank."""
    import torch
    if dist is None or world == 1:
        return
    blob = table.stats_blob()
    mine = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
    if device is not None:
        mine = mine.to(device)
    every = torch.empty(world * len(blob), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(every, mine)
    raw = every.cpu().numpy().tobytes()
    table.unify_shard_stats([raw[i * len(blob):(i + 1) * len(blob)] for i in range(world)])


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
    place.  Everything is enqueued on the current stream of `partial`'s device: no host synchronisation here.
    With `query` (an engine.Query whose partial table is bound to `partial`, on a GPU) the small-table strategy is
    ONE all-gather + ONE fused merge kernel of the engine (rsq_query_merge_gathered, on the context's stream — which
    must be the current torch stream); without it (CPU tests over gloo) the segment reductions are torch ops."""

    def __init__(self, dist, partial, n_min: int, n_max: int, n_sum: int, world: int, always_collective: bool = False,
                 query=None):
        import torch
        self.dist, self.partial = dist, partial
        self.n_min, self.n_max, self.n_sum, self.world = n_min, n_max, n_sum, world
        words = n_min + n_max + n_sum
        assert partial.numel() == words and partial.dtype == torch.int64
        self.collective = dist is not None and (world > 1 or always_collective)   # always_collective: run the exchange even with one rank (tests)
        self.gather = self.collective and words <= GATHER_LIMIT_WORDS
        self.query = query if (query is not None and partial.is_cuda) else None
        self.all = torch.empty((world, words), dtype=torch.int64, device=partial.device) if self.gather else None
        self.strategy = "single rank" if not self.collective else (
            ("one all-gather + one fused merge kernel" if self.query is not None else "one all-gather + local segment reductions")
            if self.gather else "one all-reduce per segment (min | max | sum)")

    def merge(self) -> None:
        import torch
        if not self.collective:
            return
        if not self.gather:
            allreduce_partial(self.dist, self.partial, self.n_min, self.n_max, self.n_sum)
            return
        self.dist.all_gather_into_tensor(self.all.view(-1), self.partial)
        if self.query is not None:
            self.query.merge_gathered(self.all.data_ptr(), self.world)
            return
        a, b = self.n_min, self.n_min + self.n_max
        if self.n_min:
            torch.amin(self.all[:, :a], dim=0, out=self.partial[:a])
        if self.n_max:
            torch.amax(self.all[:, a:b], dim=0, out=self.partial[a:b])
        if self.n_sum:
            torch.sum(self.all[:, b:], dim=0, out=self.partial[b:])


# ------------------------------------------------------------------------------------------------
# Joins + high-cardinality groups across GPUs (TPC-H Q3): the small build sides are replicated (every
# rank builds its own hash tables from the full customer / orders tables), the big probe-side scan is sharded by row
# range with the boundaries moved to a change of the clustering key (lineitem is clustered by l_orderkey and the groups
# are keyed by it), so every group lives on exactly one rank.  Each rank then runs the WHOLE plan on its shard — same
# kernels, same host tail, ORDER BY ... LIMIT k included — and the only exchange is an all-gather of k result rows per
# rank, merged by the sort keys.  No partial-aggregate exchange at all.
# ------------------------------------------------------------------------------------------------
def shard_rows_on_key(n_total: int, world: int, rank: int, key_at, tile: int = 128) -> Tuple[int, int]:
    """like shard_rows, but every boundary is moved forward to the first row whose clustering key differs from the row
    before it (key_at(i) -> key of row i), so that no key value spans two shards.  Boundaries no longer fall on tile
    multiples; the engine's scan handles any row0 (rows are addressed relative to the shard's own columns)."""
    def snap(b: int) -> int:
        if b <= 0 or b >= n_total:
            return max(0, min(b, n_total))
        while b < n_total and key_at(b) == key_at(b - 1):
            b += 1
        return b
    per = (n_total // (tile * world)) * tile
    lo = snap(rank * per)
    hi = n_total if rank == world - 1 else snap((rank + 1) * per)
    return lo, max(0, hi - lo)


def merge_ordered_results(dist, result, order, limit, world: int, device=None):
    """All-gather the (at most `limit`) ordered result rows of every rank and merge them by `order`
    = [(column name, ascending)], keeping `limit` rows.  Returns a plan.Result on every rank.  Rows are ReSQL packed
    tuples (identical schema on all ranks); ties across ranks come out in rank order."""
    import functools
    import torch
    from . import plan as P
    ts = result.tuple_size
    rows = min(result.n_rows, limit)
    buf = torch.zeros(limit * ts + 8, dtype=torch.uint8, device=device)
    if rows:
        buf[:rows * ts] = torch.frombuffer(bytearray(result.tuples[:rows * ts]), dtype=torch.uint8).to(buf.device)
    buf[limit * ts:] = torch.frombuffer(bytearray(int(rows).to_bytes(8, "little")), dtype=torch.uint8).to(buf.device)
    if world > 1:
        every = torch.empty(world * buf.numel(), dtype=torch.uint8, device=buf.device)
        dist.all_gather_into_tensor(every, buf)
    else:
        every = buf
    every = every.cpu().numpy().tobytes()
    tuples = b""
    for r in range(world):
        chunk = every[r * (limit * ts + 8):(r + 1) * (limit * ts + 8)]
        n = int.from_bytes(chunk[limit * ts:], "little")
        tuples += chunk[:n * ts]
    merged = P.Result(result.names, result.types, result.offsets, ts, len(tuples) // ts, tuples)
    cols = [(merged.names.index(name), asc) for name, asc in order]

    def cmp(a, b):
        for c, asc in cols:
            va, vb = merged.value(a, c), merged.value(b, c)
            if va != vb:
                return (-1 if va < vb else 1) * (1 if asc else -1)
        return 0
    idx = sorted(range(merged.n_rows), key=functools.cmp_to_key(cmp))[:limit]
    out = b"".join(tuples[i * ts:(i + 1) * ts] for i in idx)
    res = P.Result(result.names, result.types, result.offsets, ts, len(idx), out)
    res.text = res.serialize()
    return res
