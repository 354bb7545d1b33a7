"""Shards plan as the whole table (include/resql_hip.h rsq_table_stats_export / rsq_table_unify_shard_stats), on CPU.

The reference never sees a shard: one Relation, one hash table all workers reach (reference src/operators/aggregation.h:240-295,
src/JitContextFlounder.h:459-487) — any distribution of the rows over the workers gives the same relation.  Here: three shards of
lineitem, shard 1 without any 'R' line, shard 2 without any 'O' line.  Planned from their own statistics they derive three different
dense layouts (round 3 refused that); planned from the unified statistics they derive ONE, and the merged partial tables finalise
to the oracle's answer on the concatenated table — through plain calls, and through two gloo ranks with the exchange bench.py
uses (resql_amd/dist.py unify_shard_stats + PartialMerger)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from resql_amd import engine, plan as P, tpch  # noqa: E402
from oracle import orc  # noqa: E402
import shardcases  # noqa: E402
from test_engine_host import q1_partial_table_numpy  # noqa: E402

SCHEMA_ONLY = tpch.lineitem_table(0.001, tpch.Q1_COLUMNS, n_rows=0)


def _merge(tables):
    st = np.stack(tables)
    return np.concatenate([st[:, :6].min(0), st[:, 6:].sum(0)])


def test_shards_with_different_value_sets_plan_one_layout(compile_ctx):
    shards, row0, whole = shardcases.lineitem_shards()
    tabs = []
    for c, r0 in zip(shards, row0):
        t = compile_ctx.table(shardcases.shard_table(c))
        t.set_row0(r0)
        tabs.append(t)
    own = []
    for t in tabs:
        q = compile_ctx.compile(tpch.q1_plan(SCHEMA_ONLY), [t])
        own.append(shardcases.layout_line(q))
        q.close()
    assert "groups=6" in own[0] and "groups=4" in own[1] and "groups=3" in own[2]       # what round 3 refused to merge
    blobs = [t.stats_blob() for t in tabs]
    assert len({len(b) for b in blobs}) == 1
    for t in tabs:
        assert t.total_rows == t.n_rows
        t.unify_shard_stats(blobs)
        assert t.total_rows == whole.n_rows
    partials = []
    for t, c, r0 in zip(tabs, shards, row0):
        q = compile_ctx.compile(tpch.q1_plan(SCHEMA_ONLY), [t])
        assert shardcases.layout_line(q) == own[0] and q.partial_layout() == (6, 0, 36)
        partials.append(q1_partial_table_numpy(c, r0))
        q.close()
    want = orc.execute(tpch.q1_plan(whole))
    for t in tabs:                                            # whichever shard is the root finalises to the same relation
        q = compile_ctx.compile(tpch.q1_plan(SCHEMA_ONLY), [t])
        q.finalize_host(_merge(partials))
        got = q.result()
        assert got.text == want.text and got.tuples == want.tuples
        q.close()
    # unifying again (e.g. a second query over the same shards) changes nothing: the blobs carry the shards' OWN statistics
    assert [t.stats_blob() for t in tabs] == blobs
    for t in tabs:
        t.unify_shard_stats(blobs)
        assert t.total_rows == whole.n_rows
        t.close()


def test_an_empty_shard_plans_like_the_others(compile_ctx):
    shards, row0, whole = shardcases.lineitem_shards(drops=("", "ANR", ""))     # shard 1 keeps no row at all
    assert len(shards[1]["l_quantity"]) == 0
    tabs = []
    for c, r0 in zip(shards, row0):
        t = compile_ctx.table(shardcases.shard_table(c))
        t.set_row0(r0)
        tabs.append(t)
    blobs = [t.stats_blob() for t in tabs]
    lines = []
    for t in tabs:
        t.unify_shard_stats(blobs)
        q = compile_ctx.compile(tpch.q1_plan(SCHEMA_ONLY), [t])
        lines.append(shardcases.layout_line(q))
        q.close()
    assert len(set(lines)) == 1 and "groups=6" in lines[0]
    q = compile_ctx.compile(tpch.q1_plan(SCHEMA_ONLY), [tabs[1]])
    q.finalize_host(_merge([q1_partial_table_numpy(c, r0) for c, r0 in zip(shards, row0)]))
    assert q.result().text == orc.execute(tpch.q1_plan(whole)).text
    q.close()
    for t in tabs:
        t.close()


def _synthetic_shard(n, groups, shift, seed_rows):
    from resql_amd import datagen
    c = datagen.synthetic_columns(seed_rows, n, groups)
    c["b"] = c["b"] + shift
    return c


def test_where_the_union_is_not_dense_every_shard_takes_the_hash_aggregation(compile_ctx):
    """each shard's group key spans 1024 values (a dense layout on its own), the union spans 2^26: no dense layout exists for the
    table, and ALL shards must say so alike (the multi-GPU handle then merges their group rows by key)"""
    a = _synthetic_shard(3000, 1024, 0, 0)
    b = _synthetic_shard(3000, 1024, 1 << 26, 3000)
    tabs = [compile_ctx.table(tpch.make_table("t", tpch.SYNTH_SCHEMA, c, 3000)) for c in (a, b)]
    tabs[1].set_row0(3000)
    plan = tpch.synthetic_plan(tpch.synthetic_table(0, 1024), 1 << 30)
    for t in tabs:
        q = compile_ctx.compile(plan, [t])
        assert "partial table:" in q.explain
        q.close()
    blobs = [t.stats_blob() for t in tabs]
    for t in tabs:
        t.unify_shard_stats(blobs)
        q = compile_ctx.compile(plan, [t])
        assert "partial table:" not in q.explain
        with pytest.raises(engine.EngineError) as e:
            q.partial_layout()
        assert e.value.status == 3
        q.close()
        t.close()


def test_malformed_statistics_are_refused(compile_ctx):
    shards, row0, _ = shardcases.lineitem_shards()
    t = compile_ctx.table(shardcases.shard_table(shards[0]))
    other = compile_ctx.table(tpch.synthetic_table(100, 8))
    blob = t.stats_blob()
    with pytest.raises(engine.EngineError) as e:
        t.unify_shard_stats([other.stats_blob()])             # another schema
    assert e.value.status == 1
    with pytest.raises(engine.EngineError) as e:
        t.unify_shard_stats([b"\0" * len(blob)])              # not a statistics blob
    assert e.value.status == 1
    u = compile_ctx.table(shardcases.shard_table(shards[1]))
    u.set_row0(row0[1])
    with pytest.raises(engine.EngineError) as e:
        t.unify_shard_stats([u.stats_blob()])                 # its own blob is not among them
    assert e.value.status == 1 and "own" in str(e.value)
    assert t.total_rows == t.n_rows                           # nothing was changed by the refused calls
    for x in (t, other, u):
        x.close()


# ---- two gloo ranks: the exchange bench.py's measured path uses ---------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_path: str):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from resql_amd.dist import PartialMerger, unify_shard_stats
        shards, row0, _ = shardcases.lineitem_shards(drops=("R", "O"))     # rank 0 has no 'R' line, rank 1 no 'O' line
        ctx = engine.Context(device=-1)
        t = ctx.table(shardcases.shard_table(shards[rank]))
        t.set_row0(row0[rank])
        unify_shard_stats(dist, t, world)
        q = ctx.compile(tpch.q1_plan(SCHEMA_ONLY), [t])
        lines = [None] * world
        dist.all_gather_object(lines, shardcases.layout_line(q))
        assert len(set(lines)) == 1 and "groups=6" in lines[0], lines
        n_min, n_max, n_sum = q.partial_layout()
        partial = torch.from_numpy(q1_partial_table_numpy(shards[rank], row0[rank]))
        PartialMerger(dist, partial, n_min, n_max, n_sum, world).merge()
        if rank == 0:
            q.finalize_host(partial.numpy())
            with open(out_path, "w") as f:
                f.write(q.result().text)
        q.close(); t.close(); ctx.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_whose_shards_miss_different_groups(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    _, _, whole = shardcases.lineitem_shards(drops=("R", "O"))
    assert open(out).read() == orc.execute(tpch.q1_plan(whole)).text


def test_unify_never_narrows_below_the_shards_own_rows_and_refuses_inverted_ranges(compile_ctx):
    """a shard's own statistics are folded into the union whatever the blobs say about it (device code trusts the union
    without a range check for engine-owned columns), and a blob with min > max is refused."""
    import struct
    shards, row0, whole = shardcases.lineitem_shards()
    tabs = []
    for c, r0 in zip(shards, row0):
        t = compile_ctx.table(shardcases.shard_table(c))
        t.set_row0(r0)
        tabs.append(t)
    blobs = [bytearray(t.stats_blob()) for t in tabs]
    head, col = 32, 4 + 4 + 8 + 8 + 4 + 4 + 256
    names = [c.name for c in shardcases.shard_table(shards[0]).columns]
    qi = names.index("l_quantity")
    off = head + qi * col + 8
    true_min, true_max = struct.unpack_from("<qq", blobs[0], off)
    # shard 0's blob lies: a range narrower than its rows hold (a stale blob of an earlier, smaller table)
    lying = [bytearray(b) for b in blobs]
    struct.pack_into("<qq", lying[0], off, true_min + 5, true_max - 5)
    tabs[0].unify_shard_stats([bytes(b) for b in lying])
    q = compile_ctx.compile(tpch.q1_plan(SCHEMA_ONLY), [tabs[0]])
    q.close()
    back = tabs[0].stats_blob()          # (the export shows the shard's OWN statistics: untouched by what the blobs claimed)
    assert struct.unpack_from("<qq", back, off) == (true_min, true_max)
    # the union the planner sees covers the own rows: a plan grouped by l_quantity has a cell for every own value
    from resql_amd import plan as P2
    host = shardcases.shard_table(shards[0])
    p = P2.Plan([host])
    node = p.aggregation([p.count(p.star())], [p.attr("l_quantity")], p.scan(host.name))
    node = p.projection([p.attr("l_quantity")], node)
    q = compile_ctx.compile(p.set_root(p.materialize(node)), [tabs[0]])
    line = [l for l in q.explain.splitlines() if l.startswith("partial table:")][0]
    assert f"[{true_min}..{true_max}]" in line, line
    q.close()
    inverted = [bytearray(b) for b in blobs]
    struct.pack_into("<qq", inverted[1], off, 10, 3)
    with pytest.raises(engine.EngineError) as e:
        tabs[1].unify_shard_stats([bytes(b) for b in inverted])
    assert "min > max" in str(e.value)
    for t in tabs:
        t.close()
