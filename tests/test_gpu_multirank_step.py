"""The multi-rank step of bench.py on one GPU: two row-range shards of lineitem are scanned one after the other by the
same device ("rank 0" and "rank 1" of a 2-way sharding), each with the asynchronous partial execution on torch's stream,
the partial tables are merged with resql_amd.dist.PartialMerger over an RCCL (nccl) process group of world size 1 whose
collective is forced to run, and the merged table is finalised.  Result == the oracle on the unsharded table.
(The 2-process merge itself is covered on CPU over gloo in tests/test_distributed_merge.py.)"""
import os
import socket

import pytest

from resql_amd import datagen, engine, tpch
from oracle import orc

pytestmark = pytest.mark.gpu


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def nccl_world1():
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("plan_of", [tpch.q1_plan, tpch.q6_plan])
def test_async_partial_merge_finalize(nccl_world1, plan_of):
    import torch
    from resql_amd.dist import PartialMerger, shard_rows
    dist = nccl_world1
    sf, world = 0.05, 2
    cols = tpch.Q1_COLUMNS if plan_of is tpch.q1_plan else tpch.Q6_COLUMNS
    n_total = datagen.n_lineitem(sf)
    dev = torch.device("cuda", 0)
    ctx = engine.Context(device=0)
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        ctx.set_stream(stream.cuda_stream)
        schema_only = tpch.lineitem_table(0.001, cols, n_rows=0)
        merged = None
        for rank in range(world):
            row0, n = shard_rows(n_total, world, rank)
            table = ctx.generate(engine.GEN_LINEITEM, n, sf, row0=row0)
            q = ctx.compile(plan_of(schema_only), [table])
            n_min, n_max, n_sum = q.partial_layout()
            partial = torch.zeros(n_min + n_max + n_sum, dtype=torch.int64, device=dev)
            q.bind_partial(partial.data_ptr(), partial.numel() * 8)
            merger = PartialMerger(dist, partial, n_min, n_max, n_sum, 1, always_collective=True)
            assert merger.gather
            for _ in range(3):                         # steps queue up behind each other without host syncs
                q.execute_partial_async()
                merger.merge()
            if merged is None:
                merged = partial.clone()
            else:                                      # what the all-gather + segment reductions do across real ranks
                a, b = n_min, n_min + n_max
                merged[:a] = torch.minimum(merged[:a], partial[:a])
                merged[a:b] = torch.maximum(merged[a:b], partial[a:b])
                merged[b:] += partial[b:]
                partial.copy_(merged)
                q.finalize()
                got = q.result()
                assert q.report().kernel_time_ms > 0
            if rank < world - 1:
                q.finalize()                           # accounts for the enqueued steps, checks the device error word
            q.close(); table.close()
    ctx.set_stream(None)
    ctx.close()
    want = orc.execute(plan_of(tpch.lineitem_table(sf, cols)))
    assert got.text == want.text


def test_shards_with_different_value_sets_through_the_rank_step(nccl_world1):
    """the dist-path step (asynchronous partial execution, RCCL collective of a world of one forced to run, merge kernel, finalize)
    over three shards whose statistics differ — shard 1 has no 'R' line, shard 2 no 'O' line — after the exchange bench.py's ranks
    make (stats_blob of every shard -> unify_shard_stats): one 6-group layout, the gathered tables reduce to the oracle's answer on
    the concatenated table.  The reference: one hash table all workers reach (aggregation.h:240-295)."""
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import shardcases
    from resql_amd.dist import PartialMerger
    dist = nccl_world1
    shards, row0, whole = shardcases.lineitem_shards(n_rows=30_000)
    dev = torch.device("cuda", 0)
    ctx = engine.Context(device=0)
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        ctx.set_stream(stream.cuda_stream)
        tabs = [ctx.table(shardcases.shard_table(c)) for c in shards]
        for t, r0 in zip(tabs, row0):
            t.set_row0(r0)
        blobs = [t.stats_blob() for t in tabs]
        schema_only = tpch.lineitem_table(0.001, tpch.Q1_COLUMNS, n_rows=0)
        queries, partials = [], []
        for t in tabs:
            t.unify_shard_stats(blobs)
            q = ctx.compile(tpch.q1_plan(schema_only), [t])
            n_min, n_max, n_sum = q.partial_layout()
            assert (n_min, n_max, n_sum) == (6, 0, 36)
            partial = torch.zeros(42, dtype=torch.int64, device=dev)
            q.bind_partial(partial.data_ptr(), partial.numel() * 8)
            queries.append(q); partials.append(partial)
        for q, partial in zip(queries, partials):
            q.execute_partial_async()
            PartialMerger(dist, partial, 6, 0, 36, 1, always_collective=True).merge()
        stream.synchronize()
        # what one all-gather delivers on three real ranks: the three tables back to back -> the engine's merge kernel on the root
        gathered = torch.cat(partials).contiguous()
        queries[0].merge_gathered(gathered.data_ptr(), 3)
        queries[0].finalize()
        got = queries[0].result()
        for q in queries[1:]:
            q.finalize()
        for q in queries:
            q.close()
        for t in tabs:
            t.close()
    ctx.set_stream(None)
    ctx.close()
    want = orc.execute(tpch.q1_plan(whole))
    assert got.text == want.text and got.tuples == want.tuples


def test_async_needs_a_plain_dense_plan(gpu_ctx):
    sf = 0.01
    li = tpch.lineitem_table(sf, tpch.Q3_LINEITEM_COLUMNS)
    cu, od = tpch.customer_table(sf), tpch.orders_table(sf)
    tabs = [gpu_ctx.table(t) for t in (cu, od, li)]
    q = gpu_ctx.compile(tpch.q3_plan(cu, od, li), tabs)
    with pytest.raises(engine.EngineError) as e:
        q.execute_partial_async()
    assert e.value.status == 3
    q.close()
    for t in tabs:
        t.close()


def test_q3_key_aligned_shards_through_the_engine(gpu_ctx):
    """TPC-H Q3 the multi-GPU way on one device: replicated customer / orders, lineitem in two shards cut at an
    l_orderkey change, the whole plan (joins, aggregation at the entry, ORDER BY ... LIMIT 10) per shard through the C ABI,
    then the ordered merge of the two top-10 lists.  Equals the oracle on the unsharded tables."""
    import os as _os, sys as _sys
    _sys.path.insert(0, _os.path.dirname(__file__))
    from resql_amd import plan as P
    from resql_amd.dist import merge_ordered_results, shard_rows_on_key
    sf, world = 0.1, 2
    n = datagen.n_lineitem(sf)
    keys = datagen.lineitem_columns(0, n, sf, columns={"l_orderkey"})["l_orderkey"]
    cu_h, od_h = tpch.customer_table(sf), tpch.orders_table(sf)
    cu, od = gpu_ctx.table(cu_h), gpu_ctx.table(od_h)
    schema_only = tpch.lineitem_table(0.001, tpch.Q3_LINEITEM_COLUMNS, n_rows=0)
    tuples, first = b"", None
    for rank in range(world):
        row0, cnt = shard_rows_on_key(n, world, rank, lambda i: int(keys[i]))
        assert row0 % 128 != 0 or rank == 0                      # the cut really is off the tile grid
        li = gpu_ctx.generate(engine.GEN_LINEITEM, cnt, sf, row0=row0, param=1)
        q = gpu_ctx.compile(tpch.q3_plan(cu_h, od_h, schema_only), [cu, od, li])
        q.execute()
        r = q.result()
        q.close(); li.close()
        first = first or r
        tuples += r.tuples
    both = P.Result(first.names, first.types, first.offsets, first.tuple_size, len(tuples) // first.tuple_size, tuples)
    # what the all-gather delivers on a real 2-rank run: both ranks' 10 rows; merge them all, keep 10
    merged = merge_ordered_results(None, both, [("revenue", False), ("o_orderdate", True)], 2 * 10, 1)
    cu.close(); od.close()
    want = orc.execute(tpch.q3_plan(cu_h, od_h, tpch.lineitem_table(sf, tpch.Q3_LINEITEM_COLUMNS)))
    assert merged.n_rows == 20
    assert merged.text.splitlines()[:11] == want.text.splitlines()
