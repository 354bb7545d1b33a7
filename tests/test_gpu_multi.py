"""rsq_multi_* (include/resql_hip.h): one host process over N shards — the C-ABI form of JitContextFlounder::execute()'s
fan-out-and-join (reference src/JitContextFlounder.h:459-487).  The box has ONE GPU: N-shard handles list device 0
several times and merge with peer copies + the engine's merge kernel (RCCL cannot hold a device twice); the RCCL path
(dlopen of librccl, ncclCommInitAll, the grouped ncclReduce per [min | max | sum] segment) runs with a one-device
communicator.  Results == the oracle on the unsharded table, byte for byte."""
import numpy as np
import pytest

from resql_amd import datagen, engine, plan as P, tpch
from oracle import orc

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


def test_shards_that_disagree_on_the_layout_are_refused():
    """the dense group layout comes from each shard's column statistics: a shard that misses a group value cannot be merged"""
    li = tpch.lineitem_table(0.01, tpch.Q1_COLUMNS)
    m = engine.MultiContext([0, 0])
    try:
        a = m.shards[0].table(li)
        cols = {c.name: c.data for c in li.columns if c.data is not None}
        keep = cols["l_returnflag"] != ord("R")
        sub = tpch.make_table("lineitem", tpch.LINEITEM_SCHEMA, {k: v[keep] for k, v in cols.items()}, int(keep.sum()))
        b = m.shards[1].table(sub)
        with pytest.raises(engine.EngineError) as e:
            m.compile(tpch.q1_plan(_schema(tpch.Q1_COLUMNS)), [[a], [b]])
        assert e.value.status == 3 and "disagree" in str(e.value)
        a.close(); b.close()
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
