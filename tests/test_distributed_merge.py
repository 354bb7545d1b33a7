"""The N > 1 path on CPU: two gloo ranks shard lineitem by row range, build their partial aggregate tables, merge
them with the same merge code bench.py uses over RCCL (resql_amd/dist.py: PartialMerger), and rank 0 finalises through the engine's C ABI
(rsq_query_finalize_host).  The result must equal the oracle's on the unsharded table."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_ROWS = 40_001


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_path: str):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from resql_amd import datagen, engine, tpch
        from resql_amd import dist as rdist
        from resql_amd.dist import PartialMerger, shard_rows
        from test_engine_host import q1_partial_table_numpy

        row0, n = shard_rows(N_ROWS, world, rank)
        cols = datagen.lineitem_columns(row0, n, 0.01, columns=set(tpch.Q1_COLUMNS))
        partial = torch.from_numpy(q1_partial_table_numpy(cols, row0))
        merger = PartialMerger(dist, partial, 6, 0, 36, world)       # what bench.py does over RCCL
        assert merger.gather
        by_segments = partial.clone()
        merger.merge()
        rdist.allreduce_partial(dist, by_segments, 6, 0, 36)         # the large-table strategy must agree
        assert torch.equal(partial, by_segments)
        # a table with all three segments, through both strategies
        g = torch.Generator().manual_seed(7 + rank)
        t3 = torch.randint(-1000, 1000, (5 + 4 + 9,), generator=g, dtype=torch.int64)
        a3, b3 = t3.clone(), t3.clone()
        PartialMerger(dist, a3, 5, 4, 9, world).merge()
        saved = rdist.GATHER_LIMIT_WORDS
        rdist.GATHER_LIMIT_WORDS = 0
        m = PartialMerger(dist, b3, 5, 4, 9, world)
        assert not m.gather
        m.merge()
        rdist.GATHER_LIMIT_WORDS = saved
        assert torch.equal(a3, b3)
        every = [torch.empty_like(t3) for _ in range(world)]
        dist.all_gather(every, t3)
        st = torch.stack(every)
        assert torch.equal(a3, torch.cat([st[:, :5].amin(0), st[:, 5:9].amax(0), st[:, 9:].sum(0)]))
        if rank == 0:
            ctx = engine.Context(device=-1)
            li = tpch.lineitem_table(0.01, tpch.Q1_COLUMNS, n_rows=N_ROWS)
            q = ctx.compile(tpch.q1_plan(li), [ctx.table(li)])
            assert q.partial_layout() == (6, 0, 36)
            q.finalize_host(partial.numpy())
            with open(out_path, "w") as f:
                f.write(q.result().text)
    finally:
        dist.destroy_process_group()


def test_shards_cover_the_table_exactly():
    from resql_amd.dist import shard_rows
    for n in (0, 1, 127, 128, 1000, 59_999_996):
        for world in (1, 2, 3, 4, 8):
            ranges = [shard_rows(n, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and sum(c for _, c in ranges) == n
            for (a0, an), (b0, _) in zip(ranges, ranges[1:]):
                assert a0 + an == b0 and b0 % 128 == 0


def test_two_rank_merge_equals_single_run(tmp_path):
    from resql_amd import tpch
    from oracle import orc
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    li = tpch.lineitem_table(0.01, tpch.Q1_COLUMNS, n_rows=N_ROWS)
    assert open(out).read() == orc.execute(tpch.q1_plan(li)).text


# ---- joins + high-cardinality groups (TPC-H Q3) across ranks: key-aligned shards, all-gather of the top-k rows ----------
def _q3_shard_tables(sf, world, rank):
    from resql_amd import datagen, tpch, plan as P
    from resql_amd.dist import shard_rows_on_key
    n = datagen.n_lineitem(sf)
    keys = datagen.lineitem_columns(0, n, sf, columns={"l_orderkey"})["l_orderkey"]
    row0, cnt = shard_rows_on_key(n, world, rank, lambda i: int(keys[i]))
    cols = datagen.lineitem_columns(row0, cnt, sf, columns=set(tpch.Q3_LINEITEM_COLUMNS))
    li = tpch.make_table("lineitem", tpch.LINEITEM_SCHEMA, cols, cnt)
    return tpch.customer_table(sf), tpch.orders_table(sf), li, (row0, cnt)


def _q3_worker(rank: int, world: int, port: int, out_path: str):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from resql_amd import tpch
        from resql_amd.dist import merge_ordered_results
        from oracle import orc
        cu, od, li, _ = _q3_shard_tables(0.05, world, rank)
        local = orc.execute(tpch.q3_plan(cu, od, li))          # stands in for the engine: same plan on this rank's shard
        merged = merge_ordered_results(dist, local, [("revenue", False), ("o_orderdate", True)], 10, world)
        if rank == 0:
            with open(out_path, "w") as f:
                f.write(merged.text)
    finally:
        dist.destroy_process_group()


def test_key_aligned_shards_never_split_a_key():
    from resql_amd import datagen
    from resql_amd.dist import shard_rows_on_key
    sf = 0.02
    n = datagen.n_lineitem(sf)
    keys = datagen.lineitem_columns(0, n, sf, columns={"l_orderkey"})["l_orderkey"]
    for world in (1, 2, 3, 8):
        ranges = [shard_rows_on_key(n, world, r, lambda i: int(keys[i])) for r in range(world)]
        assert ranges[0][0] == 0 and sum(c for _, c in ranges) == n
        for (a0, an), (b0, _) in zip(ranges, ranges[1:]):
            assert a0 + an == b0
            if 0 < b0 < n:
                assert keys[b0] != keys[b0 - 1]


def test_two_rank_q3_equals_single_run(tmp_path):
    from resql_amd import tpch
    from oracle import orc
    out = str(tmp_path / "q3.txt")
    mp.spawn(_q3_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    sf = 0.05
    li = tpch.lineitem_table(sf, tpch.Q3_LINEITEM_COLUMNS)
    want = orc.execute(tpch.q3_plan(tpch.customer_table(sf), tpch.orders_table(sf), li)).text
    assert open(out).read() == want
