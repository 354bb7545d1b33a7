"""rsq_multi_* (include/resql_hip.h): one host process over N shards — the C-ABI form of JitContextFlounder::execute()'s
fan-out-and-join (reference src/JitContextFlounder.h:459-487).  The box has ONE GPU: N-shard handles list device 0
several times and merge with peer copies + the engine's merge kernel (RCCL cannot hold a device twice); the RCCL path
(dlopen of librccl, ncclCommInitAll, the grouped ncclReduce per [min | max | sum] segment) runs with a one-device
communicator.  Results == the oracle on the unsharded table, byte for byte."""
import numpy as np
import pytest

import os
import sys

from resql_amd import datagen, engine, plan as P, tpch
from oracle import orc

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import shardcases  # noqa: E402

pytestmark = pytest.mark.gpu
SF = 0.05


def _schema(cols):
    return tpch.lineitem_table(0.001, cols, n_rows=0)


@pytest.mark.parametrize("n_shards", [1, 2, 3])
@pytest.mark.parametrize("plan_of,cols", [(tpch.q1_plan, tpch.Q1_COLUMNS), (tpch.q6_plan, tpch.Q6_COLUMNS)])
def test_dense_aggregation_over_shards(n_shards, plan_of, cols):
    n = datagen.n_lineitem(SF)
    want = orc.execute(plan_of(tpch.lineitem_table(SF, cols)))
    m = engine.MultiContext([0] * n_shards)
    try:
        assert m.n == n_shards
        assert ("peer copies" in m.merge_name) == (n_shards > 1)
        shards = m.generate(engine.GEN_LINEITEM, n, SF)
        assert sum(t.n_rows for t in shards) == n and [t.n_rows for t in shards] == [m.shard_rows(n, i)[1] for i in range(n_shards)]
        q = m.compile(plan_of(_schema(cols)), [[t] for t in shards])
        for _ in range(3):                                  # steps follow each other without leftovers
            q.execute()
            got = q.result()
            assert got.text == want.text and got.tuples == want.tuples
        rep, per = q.report()
        assert len(per) == n_shards and all(k > 0 for k in per) and rep.kernel_time_ms == max(per)
        assert rep.bytes_read == n * (tpch.Q1_BYTES_PER_ROW if plan_of is tpch.q1_plan else tpch.Q6_BYTES_PER_ROW)
        q.close()
        for t in shards:
            t.close()
    finally:
        m.close()


def test_rccl_path_with_a_one_device_communicator():
    n = datagen.n_lineitem(SF)
    want = orc.execute(tpch.q1_plan(tpch.lineitem_table(SF, tpch.Q1_COLUMNS)))
    m = engine.MultiContext([0], merge=engine.MERGE_RCCL)
    try:
        shards = m.generate(engine.GEN_LINEITEM, n, SF)
        q = m.compile(tpch.q1_plan(_schema(tpch.Q1_COLUMNS)), [[shards[0]]])
        for _ in range(2):
            q.execute()
            assert q.result().text == want.text
        q.close()
        shards[0].close()
    finally:
        m.close()


def test_rccl_refuses_a_device_listed_twice():
    with pytest.raises(engine.EngineError) as e:
        engine.MultiContext([0, 0], merge=engine.MERGE_RCCL)
    assert "same device twice" in str(e.value)


@pytest.mark.parametrize("groups", [1024, 1 << 16])
def test_large_partial_tables_merge(groups):
    """synthetic filter + group-by: the partial tables are LDS-table / HBM-table sized (all three segments in use)"""
    n = 600_000
    host = tpch.synthetic_table(n, groups)
    want = orc.execute(tpch.synthetic_plan(host, 1 << 30))
    m = engine.MultiContext([0, 0])
    try:
        shards = m.generate(engine.GEN_SYNTHETIC, n, 1.0, param=groups)
        q = m.compile(tpch.synthetic_plan(tpch.synthetic_table(0, groups), 1 << 30), [[t] for t in shards])
        q.execute()
        got = q.result()
        assert sorted(got.text.splitlines()) == sorted(want.text.splitlines())
        q.close()
        for t in shards:
            t.close()
    finally:
        m.close()


def _shard_tables(m, shards, row0):
    tabs = []
    for i, (c, r0) in enumerate(zip(shards, row0)):
        t = m.shards[i].table(shardcases.shard_table(c))
        t.set_row0(r0)
        tabs.append(t)
    return tabs


@pytest.mark.parametrize("plan_of,cols", [(tpch.q1_plan, tpch.Q1_COLUMNS), (tpch.q6_plan, tpch.Q6_COLUMNS)])
def test_shards_with_different_value_sets_merge_through_peer_copies(plan_of, cols):
    """a time-clustered table: shard 1 holds no 'R' line, shard 2 no 'O' line.  The reference has one hash table every worker
    reaches (aggregation.h:240-295) and answers whatever the distribution; rsq_multi_query_compile gives every shard the whole
    table's statistics, so all three derive ONE dense layout (round 3 refused this) and the peer-copy merge adds them up"""
    want_cols = sorted(set(tpch.Q1_COLUMNS) | set(cols))
    shards, row0, whole = shardcases.lineitem_shards(n_rows=30_000, columns=want_cols)
    want = orc.execute(plan_of(whole))
    m = engine.MultiContext([0, 0, 0])
    try:
        tabs = _shard_tables(m, shards, row0)
        q = m.compile(plan_of(_schema(cols)), [[t] for t in tabs])
        assert "dense partial tables" in q.merge_name and "peer copies" in q.merge_name
        assert [t.total_rows for t in tabs] == [whole.n_rows] * 3
        for _ in range(2):
            q.execute()
            got = q.result()
            assert got.text == want.text and got.tuples == want.tuples
        q.close()
        for t in tabs:
            t.close()
    finally:
        m.close()


def test_a_shard_planned_as_the_whole_table_through_the_rccl_handle():
    """the RCCL handle (one-device communicator: grouped ncclReduce per segment) over a shard that lacks 'R' and was given the
    statistics of all three shards: its partial table has the whole table's 6 cells, the 'R' cells stay at their identities, and
    the answer is the oracle's on that shard's rows"""
    shards, row0, _ = shardcases.lineitem_shards(n_rows=30_000)
    m = engine.MultiContext([0], merge=engine.MERGE_RCCL)
    ctx = engine.Context(device=0)
    try:
        others = [ctx.table(shardcases.shard_table(c)) for c in shards]
        for t, r0 in zip(others, row0):
            t.set_row0(r0)
        t1 = m.shards[0].table(shardcases.shard_table(shards[1]))
        t1.set_row0(row0[1])
        t1.unify_shard_stats([t.stats_blob() for t in others])
        q = m.compile(tpch.q1_plan(_schema(tpch.Q1_COLUMNS)), [[t1]])
        q.execute()
        got = q.result()
        want = orc.execute(tpch.q1_plan(shardcases.shard_table(shards[1])))
        assert got.text == want.text and 0 < got.n_rows <= 4
        q.close(); t1.close()
        for t in others:
            t.close()
    finally:
        ctx.close()
        m.close()


def test_where_the_union_of_the_shards_is_not_dense_the_general_merge_answers():
    """every shard's group key spans 1024 values (dense on its own), the union spans 2^26: all shards take the hash aggregation
    and the general re-aggregating merge gives the single-table answer (round 3: RSQ_ERR_UNSUPPORTED)"""
    from resql_amd import datagen
    n, groups = 60_000, 1024
    parts = []
    for i, shift in enumerate((0, 1 << 26, 0)):
        c = datagen.synthetic_columns(i * n, n, groups)
        c["b"] = c["b"] + shift
        parts.append(c)
    whole = tpch.make_table("t", tpch.SYNTH_SCHEMA, {k: np.concatenate([c[k] for c in parts]) for k in parts[0]}, 3 * n)
    want = orc.execute(tpch.synthetic_plan(whole, 1 << 30))
    m = engine.MultiContext([0, 0, 0])
    try:
        tabs = []
        for i, c in enumerate(parts):
            t = m.shards[i].table(tpch.make_table("t", tpch.SYNTH_SCHEMA, c, n))
            t.set_row0(i * n)
            tabs.append(t)
        q = m.compile(tpch.synthetic_plan(tpch.synthetic_table(0, groups), 1 << 30), [[t] for t in tabs])
        assert "general merge" in q.merge_name
        q.execute()
        got = q.result()
        assert got.n_rows == want.n_rows == 2 * groups
        assert got.text == want.text
        q.close()
        for t in tabs:
            t.close()
    finally:
        m.close()


def test_device_tail_orders_groups_that_first_occur_in_late_shards():
    """65536 dense groups over 16 shards of 62 500 rows: the root's own rows need 16 bits, the first rows of the merged table 20.
    The radix sort by first row must cover the WHOLE table's row numbers (round-3 advisor: it covered the root shard's), or the
    replay of the reference's hash table sees another insertion order.  No ORDER BY: the text is the emission order, byte for byte"""
    n, groups, n_shards = 1_000_000, 1 << 16, 16
    host = tpch.synthetic_table(n, groups)
    want = orc.execute(tpch.synthetic_plan(host, 1 << 30))
    m = engine.MultiContext([0] * n_shards)
    try:
        shards = m.generate(engine.GEN_SYNTHETIC, n, 1.0, param=groups)
        assert shards[0].n_rows < (1 << 16) < n
        q = m.compile(tpch.synthetic_plan(tpch.synthetic_table(0, groups), 1 << 30), [[t] for t in shards])
        assert "dense partial tables" in q.merge_name
        q.execute()
        got = q.result()
        assert got.n_rows == want.n_rows
        assert got.text == want.text
        q.close()
        for t in shards:
            t.close()
    finally:
        m.close()


def test_general_merge_of_a_few_dozen_groups_sizes_the_replay_from_the_whole_table():
    """40 groups of a computed key over 4 shards: the reference allocates its aggregation table for the WHOLE input
    (aggregation.h:81-92 getSize = child / 512 -> the prime above it); with so few groups no growth evens the sizes out, so a replay
    sized from the root shard's rows (round-3 advisor) would emit the groups in another order"""
    n, groups = 400_000, 40
    host = tpch.synthetic_table(n, groups)

    def plan_of(t):
        p = P.Plan([t])
        key = p.add(p.mul(p.attr("b"), p.constant("7", P.BIGINT)), p.constant("3", P.BIGINT))
        sc, cnt = p.sum(p.attr("c")), p.count(p.star())
        node = p.selection(p.lt(p.attr("a"), p.constant(str(1 << 30), P.BIGINT)), p.scan("t"))
        node = p.aggregation([sc, cnt], [key], node)
        return p.set_root(p.materialize(p.projection([p.as_("k", key), p.as_("s", sc), p.as_("n", cnt)], node)))

    want = orc.execute(plan_of(host))
    m = engine.MultiContext([0, 0, 0, 0])
    try:
        shards = m.generate(engine.GEN_SYNTHETIC, n, 1.0, param=groups)
        q = m.compile(plan_of(tpch.synthetic_table(0, groups)), [[t] for t in shards])
        assert "general merge" in q.merge_name
        q.execute()
        got = q.result()
        assert got.n_rows == want.n_rows == groups
        assert got.text == want.text
        q.close()
        for t in shards:
            t.close()
    finally:
        m.close()


def test_q3_over_key_aligned_shards():
    """joins + many groups: build sides replicated, lineitem sharded on an l_orderkey boundary, every shard runs the whole
    plan (its own top 10) concurrently, the host merges the ordered rows"""
    from resql_amd.dist import shard_rows_on_key
    sf = 0.05
    cu, od, li = tpch.customer_table(sf), tpch.orders_table(sf), tpch.lineitem_table(sf, tpch.Q3_LINEITEM_COLUMNS)
    want = orc.execute(tpch.q3_plan(cu, od, li))
    keys = {c.name: c.data for c in li.columns}["l_orderkey"]
    n = li.n_rows
    m = engine.MultiContext([0, 0, 0])
    try:
        per_shard = []
        for i in range(3):
            row0, cnt = shard_rows_on_key(n, 3, i, lambda r: int(keys[r]))
            ctx = m.shards[i]
            per_shard.append([ctx.table(cu), ctx.table(od), ctx.generate(engine.GEN_LINEITEM, cnt, sf, row0=row0, param=1)])
        q = m.compile(tpch.q3_plan(tpch.customer_table(0.001), tpch.orders_table(0.001),
                                   tpch.lineitem_table(0.001, tpch.Q3_LINEITEM_COLUMNS, n_rows=0)), per_shard)
        q.execute()
        got = q.result()
        assert got.n_rows == want.n_rows == 10
        # ties on (revenue, o_orderdate) may come out in another order than the reference's quicksort leaves them
        key = lambda res: [(res.value(r, 1), res.value(r, 2)) for r in range(res.n_rows)]
        assert key(got) == key(want) and sorted(got.text.splitlines()) == sorted(want.text.splitlines())
        q.close()
        for ts in per_shard:
            for t in ts:
                t.close()
    finally:
        m.close()


# ---- the general merge: any sharding gives the single-GPU answer (round-2 verdict: tile-aligned shards + Q3 used to concatenate
# the partial groups of an order that straddles a cut) -------------------------------------------------------------------------

def _q3_inputs(sf):
    cu, od, li = tpch.customer_table(sf), tpch.orders_table(sf), tpch.lineitem_table(sf, tpch.Q3_LINEITEM_COLUMNS)
    schema = tpch.q3_plan(tpch.customer_table(0.001), tpch.orders_table(0.001), tpch.lineitem_table(0.001, tpch.Q3_LINEITEM_COLUMNS, n_rows=0))
    return cu, od, li, schema


def _close_all(per_shard):
    for ts in per_shard:
        for t in ts:
            t.close()


@pytest.mark.parametrize("n_shards", [2, 5])
def test_q3_over_tile_aligned_shards_takes_the_general_merge(n_shards):
    """rsq_multi_table_generate cuts on 128-row tiles: orders straddle the cuts, so their groups come from two shards and must be
    re-aggregated before ORDER BY ... LIMIT sees them.  Byte-identical to the oracle — tie order included, the tail being the
    single-GPU tail over the merged groups."""
    sf = 0.05
    cu, od, li, schema = _q3_inputs(sf)
    want = orc.execute(tpch.q3_plan(cu, od, li))
    m = engine.MultiContext([0] * n_shards)
    try:
        shards = m.generate(engine.GEN_LINEITEM, li.n_rows, sf, param=1)
        per_shard = [[m.shards[i].table(cu), m.shards[i].table(od), shards[i]] for i in range(n_shards)]
        q = m.compile(schema, per_shard)
        assert "general merge" in q.merge_name and "overlap" in q.merge_name
        for _ in range(2):
            q.execute()
            got = q.result()
            assert got.text == want.text and got.tuples == want.tuples
        q.close()
        _close_all(per_shard)
    finally:
        m.close()


def test_q3_key_aligned_shards_from_the_c_abi(monkeypatch):
    """rsq_multi_table_generate_on_key: no l_orderkey spans two shards -> the statistics prove the groups disjoint -> every shard's
    own top 10 + ordered merge; the general merge, forced, gives the same rows"""
    sf = 0.05
    cu, od, li, schema = _q3_inputs(sf)
    want = orc.execute(tpch.q3_plan(cu, od, li))
    keys = {c.name: c.data for c in li.columns}["l_orderkey"]
    for force in ("0", "1"):
        monkeypatch.setenv("RSQ_MULTI_GENERAL_MERGE", force)
        m = engine.MultiContext([0, 0, 0])
        try:
            shards = m.generate_on_key(engine.GEN_LINEITEM, li.n_rows, sf, "l_orderkey", param=1)
            assert sum(t.n_rows for t in shards) == li.n_rows
            at = 0
            for t in shards[:-1]:                              # every cut sits on a key change, at or behind the tile-aligned cut
                at += t.n_rows
                assert keys[at] != keys[at - 1]
            assert [m.shard_rows(li.n_rows, i)[0] <= sum(t.n_rows for t in shards[:i]) for i in range(3)] == [True] * 3
            per_shard = [[m.shards[i].table(cu), m.shards[i].table(od), shards[i]] for i in range(3)]
            q = m.compile(schema, per_shard)
            assert ("ordered merge" in q.merge_name) == (force == "0"), q.merge_name
            q.execute()
            got = q.result()
            if force == "1":
                assert got.text == want.text
            key = lambda res: [(res.value(r, 1), res.value(r, 2)) for r in range(res.n_rows)]
            assert key(got) == key(want) and sorted(got.text.splitlines()) == sorted(want.text.splitlines())
            q.close()
            _close_all(per_shard)
        finally:
            m.close()


def test_hash_aggregation_without_limit_over_shards_keeps_the_emission_order():
    """a computed group key (generic hash aggregation), every group in every shard: the merged groups leave in the reference's
    hash-table order, which depends on each group's FIRST row over the whole table (shard tables carry row0)"""
    n, groups = 400_000, 5000
    host = tpch.synthetic_table(n, groups)

    def plan_of(t):
        p = P.Plan([t])
        key = p.add(p.mul(p.attr("b"), p.constant("3", P.BIGINT)), p.constant("1", P.BIGINT))
        sc, cnt, lo, hi, av = p.sum(p.attr("c")), p.count(p.star()), p.min(p.attr("d")), p.max(p.attr("c")), p.avg(p.attr("d"))
        node = p.selection(p.lt(p.attr("a"), p.constant(str(1 << 30), P.BIGINT)), p.scan("t"))
        node = p.aggregation([sc, cnt, lo, hi, av], [key], node)
        node = p.projection([p.as_("k", key), p.as_("s", sc), p.as_("n", cnt), p.as_("lo", lo), p.as_("hi", hi), p.as_("av", av)], node)
        return p.set_root(p.materialize(node))

    want = orc.execute(plan_of(host))
    m = engine.MultiContext([0, 0, 0, 0])
    try:
        shards = m.generate(engine.GEN_SYNTHETIC, n, 1.0, param=groups)
        q = m.compile(plan_of(tpch.synthetic_table(0, groups)), [[t] for t in shards])
        assert "general merge" in q.merge_name
        q.execute()
        got = q.result()
        assert got.n_rows == want.n_rows == groups
        assert got.text == want.text
        q.close()
        for t in shards:
            t.close()
    finally:
        m.close()


@pytest.mark.parametrize("order", [False, True])
def test_materialised_rows_over_shards_keep_scan_order(order):
    """no aggregation: the shards' materialised rows back to back are the single-GPU scan order; ORDER BY runs the reference's
    (unstable) quicksort over exactly that order, LIMIT as MaterializeOp / OrderByOp apply it"""
    n = 300_000
    host = tpch.synthetic_table(n, 64)

    def plan_of(t):
        p = P.Plan([t])
        node = p.selection(p.lt(p.attr("a"), p.constant(str(1 << 24), P.BIGINT)), p.scan("t"))
        node = p.projection([p.attr("b"), p.attr("c"), p.as_("e", p.add(p.attr("c"), p.attr("d")))], node)
        if order:
            return p.set_root(p.orderby([p.asc(p.attr("b")), p.desc(p.attr("c"))], node), limit=500)
        return p.set_root(p.materialize(node), limit=700)

    want = orc.execute(plan_of(host))
    m = engine.MultiContext([0, 0, 0])
    try:
        shards = m.generate(engine.GEN_SYNTHETIC, n, 1.0, param=64)
        q = m.compile(plan_of(tpch.synthetic_table(0, 64)), [[t] for t in shards])
        q.execute()
        got = q.result()
        assert got.n_rows == want.n_rows > 100
        assert got.text == want.text
        q.close()
        for t in shards:
            t.close()
    finally:
        m.close()


def test_join_in_front_of_a_dense_aggregation_over_shards():
    """TPC-H Q14's shape: a build pipeline, then probe + ungrouped sums (register accumulators).  The shards cannot be enqueued
    without the host (the build sizes its table), so they run on host threads up to their partial tables; merge and finalize as
    for Q1.  (Round 2 refused this shape: the asynchronous path takes no join pipelines.)"""
    sf = 0.02
    od, li = tpch.orders_table(sf), tpch.lineitem_table(sf, tpch.Q3_LINEITEM_COLUMNS)

    def plan_of(o, l):
        p = P.Plan([o, l])
        rev, cnt = p.sum(p.mul(p.attr("l_extendedprice"), p.sub(p.constant("1", P.BIGINT), p.attr("l_discount")))), p.count(p.star())
        build = p.selection(p.lt(p.attr("o_orderdate"), p.constant("1995-03-15", P.DATE)), p.scan("orders"))
        node = p.hashjoin([p.eq(p.attr("o_orderkey"), p.attr("l_orderkey"))], build, p.scan("lineitem"), single_match=True)
        node = p.aggregation([rev, cnt], [], node)
        return p.set_root(p.materialize(p.projection([p.as_("revenue", rev), p.as_("n", cnt)], node)))

    want = orc.execute(plan_of(od, li))
    m = engine.MultiContext([0, 0, 0])
    try:
        shards = m.generate(engine.GEN_LINEITEM, li.n_rows, sf, param=1)
        per_shard = [[m.shards[i].table(od), shards[i]] for i in range(3)]
        q = m.compile(plan_of(tpch.orders_table(0.001), tpch.lineitem_table(0.001, tpch.Q3_LINEITEM_COLUMNS, n_rows=0)), per_shard)
        assert "dense partial tables" in q.merge_name and "host threads" in q.merge_name
        for _ in range(2):
            q.execute()
            assert q.result().text == want.text
        q.close()
        _close_all(per_shard)
    finally:
        m.close()


def test_closing_the_multi_context_first_closes_its_queries():
    """MultiContext.close() destroys the shard contexts: queries compiled on them go first (a later MultiQuery.close() is a no-op)"""
    m = engine.MultiContext([0, 0])
    shards = m.generate(engine.GEN_LINEITEM, 20_000, 0.01)
    q = m.compile(tpch.q1_plan(_schema(tpch.Q1_COLUMNS)), [[t] for t in shards])
    q.execute()
    m.close()
    assert q.h is None
    q.close()
