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
