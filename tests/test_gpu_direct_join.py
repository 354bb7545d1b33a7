"""A join whose build side is the bare scan of an engine-owned table in the order of its strictly ascending key column needs no table:
entry number rank(key) is the row's number, and the probes read the payload from the build table's own columns (a string's address is
computed, not loaded) - HashTable::direct, engine_pipelines.cpp sizeJoinTable, codegen_join.cpp consumeMatch.  With dense keys (TPC-H's
c_custkey, p_partkey) nothing at all is built; with gaps in the key range (o_orderkey) only the key bitmap and its index are.
Everything else - a selection in front of the build, keys out of order, duplicate keys, adopted columns, the aggregation's own table -
keeps the forms it had."""
import numpy as np
import pytest

from resql_amd import engine, plan as P
from oracle import orc

pytestmark = pytest.mark.gpu
T = P.TypeInit
NOTE = "the build table's own columns"


def _dim(m, kind, seed=3):
    rng = np.random.default_rng(seed)
    if kind == "dense":
        dk = np.arange(7, 7 + m, dtype=np.int32)
    elif kind == "gaps":
        dk = (7 + np.cumsum(rng.integers(1, 5, m))).astype(np.int32)
    elif kind == "shuffled":
        dk = rng.permutation(np.arange(7, 7 + m, dtype=np.int32))
    else:      # "dups": ascending, one key twice
        dk = np.arange(7, 7 + m, dtype=np.int32)
        dk[m // 2] = dk[m // 2 - 1]
    names = np.array([f"n{int(v) % 977:05d}".encode() for v in range(m)], dtype="S9")
    flag = rng.choice(np.frombuffer(b"ANR", dtype=np.uint8), m)
    dw = rng.integers(-50, 50, m).astype(np.int64)
    dd = rng.integers(8000, 9000, m).astype(np.uint32)
    if kind == "dups":      # (a whole row twice: which of two rows of one key a single-match probe finds is the reference's insertion order)
        for col in (names, flag, dw, dd):
            col[m // 2] = col[m // 2 - 1]
    return P.Table("dim", [P.Column("dk", T.INT(), dk), P.Column("dname", T.CHAR(9), names), P.Column("dflag", T.CHAR(1), flag),
                           P.Column("dw", T.BIGINT(), dw), P.Column("dd", T.DATE(), dd)], m), dk


def _fact(n, dk, seed=5):
    rng = np.random.default_rng(seed)
    k = rng.integers(int(dk.min()) - 3, int(dk.max()) + 4, n).astype(np.int32)
    return P.Table("t", [P.Column("k", T.INT(), k), P.Column("v", T.BIGINT(), rng.integers(0, 1000, n).astype(np.int64))], n)


def _plan(dim, fact, select_build=False, single=True):
    p = P.Plan([dim, fact])
    build = p.scan("dim")
    if select_build:
        build = p.selection(p.ge(p.attr("dw"), p.constant("-20", P.BIGINT)), build)
    j = p.hashjoin([p.eq(p.attr("dk"), p.attr("k"))], build, p.scan("t"), single_match=single)
    s, c = p.sum(p.add(p.attr("v"), p.attr("dw"))), p.count(p.star())
    node = p.aggregation([s, c, p.max(p.attr("dd"))], [p.attr("dflag"), p.attr("dname")], j)
    return p.set_root(p.materialize(p.projection([p.attr("dflag"), p.attr("dname"), p.as_("s", s), p.as_("c", c)], node)))


def _run(gpu_ctx, plan, executions=3):
    tabs = [gpu_ctx.table(t) for t in plan.tables]
    q = gpu_ctx.compile(plan, tabs)
    try:
        want = sorted(orc.execute(plan).text.splitlines())
        for _ in range(executions):
            q.execute()
            assert sorted(q.result().text.splitlines()) == want
        return q.explain
    finally:
        q.close()
        for t in tabs:
            t.close()


@pytest.mark.parametrize("single", [True, False])
@pytest.mark.parametrize("kind,direct", [("dense", "nothing is built"), ("gaps", "only the key bitmap and its index are built"), ("shuffled", None), ("dups", None)])
def test_bare_scan_build_in_key_order_is_read_in_place(gpu_ctx, kind, direct, single):
    dim, dk = _dim(60_000, kind)
    explain = _run(gpu_ctx, _plan(dim, _fact(700_000, dk), single=single))
    assert (NOTE in explain) == (direct is not None), explain
    if direct:
        assert direct in explain


def test_a_selection_in_front_of_the_build_keeps_the_table(gpu_ctx):
    dim, dk = _dim(60_000, "dense")
    assert NOTE not in _run(gpu_ctx, _plan(dim, _fact(400_000, dk), select_build=True))


def test_small_build_tables(gpu_ctx):
    """up to 1 024 build rows a dictionary's index and placement launches cost more than a hash table - but dense keys need neither"""
    dim, dk = _dim(900, "gaps")
    assert NOTE not in _run(gpu_ctx, _plan(dim, _fact(200_000, dk)))
    dim, dk = _dim(900, "dense")
    assert "nothing is built" in _run(gpu_ctx, _plan(dim, _fact(200_000, dk)))
    dim, dk = _dim(25, "dense")
    assert "nothing is built" in _run(gpu_ctx, _plan(dim, _fact(100_000, dk)))


def test_materialized_join_reads_strings_in_place(gpu_ctx):
    """no aggregation: the joined rows themselves, in scan order, with the build side's string delivered from the build table's column"""
    dim, dk = _dim(50_000, "gaps", seed=11)
    fact = _fact(300_000, dk, seed=12)
    p = P.Plan([dim, fact])
    j = p.hashjoin([p.eq(p.attr("dk"), p.attr("k"))], p.scan("dim"), p.selection(p.lt(p.attr("v"), p.constant("40", P.BIGINT)), p.scan("t")), single_match=True)
    plan = p.set_root(p.materialize(p.projection([p.attr("k"), p.attr("dname"), p.attr("dflag"), p.attr("dd"), p.attr("v")], j)))
    tabs = [gpu_ctx.table(t) for t in plan.tables]
    q = gpu_ctx.compile(plan, tabs)
    try:
        want = orc.execute(plan).text
        for _ in range(2):
            q.execute()
            assert q.result().text == want
        assert NOTE in q.explain
    finally:
        q.close()
        for t in tabs:
            t.close()


def test_key_bits_are_built_once_per_table_version(gpu_ctx):
    """keys with gaps: the key bitmap and its index are a function of the build table's key column alone - the second execution of a query
    finds them as the first left them, and so does a FRESH query over the same tables (Context::keyIndexes); appended rows are a new version"""
    dim, dk = _dim(80_000, "gaps", seed=21)
    fact = _fact(500_000, dk, seed=22)
    plan = _plan(dim, fact)
    want = sorted(orc.execute(plan).text.splitlines())
    tabs = [gpu_ctx.table(t) for t in plan.tables]
    try:
        q = gpu_ctx.compile(plan, tabs)
        q.execute()
        first = int(q.report().num_kernels)
        assert sorted(q.result().text.splitlines()) == want
        q.execute()
        again = int(q.report().num_kernels)
        assert sorted(q.result().text.splitlines()) == want
        assert again <= first - 2, (first, again)          # no build pipeline, no index
        q.close()
        q = gpu_ctx.compile(plan, tabs)                    # what a ReSQL host does per statement
        q.execute()
        assert int(q.report().num_kernels) == again and sorted(q.result().text.splitlines()) == want
        # the build table grows: a new version, its own key bits, the right answer (compiled queries are per statement: a fresh one);
        # the statement compiled BEFORE the rows came is refused - its kernels' arguments point at the columns' old place
        stale = q
        tail, tk = _dim(5_000, "gaps", seed=23)
        tail.columns[0].data[:] = tail.columns[0].data + (int(dk[-1]) + 1)      # keys stay strictly ascending across the seam
        tabs[0].append(gpu_ctx.table(tail))
        with pytest.raises(engine.EngineError, match="compile it again"):
            stale.execute()
        stale.close()
        grown_dim = P.Table("dim", [P.Column(a.name, a.type, np.concatenate([a.data, b.data])) for a, b in zip(dim.columns, tail.columns)], 85_000)
        grown = _plan(grown_dim, fact)
        q = gpu_ctx.compile(grown, tabs)
        for _ in range(2):
            q.execute()
            assert sorted(q.result().text.splitlines()) == sorted(orc.execute(grown).text.splitlines())
        q.close()
    finally:
        for t in tabs:
            t.close()
