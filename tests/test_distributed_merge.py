"""The N > 1 path on CPU: two gloo ranks shard lineitem by row range, build their partial aggregate tables, merge
them with the same all-reduce code bench.py uses over RCCL, and rank 0 finalises through the engine's C ABI
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
        from resql_amd.dist import allreduce_partial, shard_rows
        from test_engine_host import q1_partial_table_numpy

        row0, n = shard_rows(N_ROWS, world, rank)
        cols = datagen.lineitem_columns(row0, n, 0.01, columns=set(tpch.Q1_COLUMNS))
        partial = torch.from_numpy(q1_partial_table_numpy(cols, row0))
        allreduce_partial(dist, partial, 6, 0, 36)
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
