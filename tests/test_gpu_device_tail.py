"""The tail of a large dense aggregation on the device (resql_amd/csrc/devtail.hip): groups present -> ordered by first row
(radix sort) -> the reference's hashes -> [host: slot order of the reference's table] -> packed tuples gathered on the device.
Same bytes as the host tail and as the oracle: emission order, AVG, MIN / MAX, 4-byte and 1-byte columns, LIMIT on the
materialisation; rsq_config.emission_order = RSQ_EMIT_ANY gives the same rows in another order."""
import numpy as np
import pytest

from resql_amd import engine, plan as P, tpch
from oracle import orc

pytestmark = pytest.mark.gpu
T = P.TypeInit


def _run(ctx, plan, tabs=None):
    own = tabs is None
    tabs = [ctx.table(t) for t in plan.tables] if own else tabs
    q = ctx.compile(plan, tabs)
    try:
        q.execute()
        first = q.result()
        q.execute()
        second = q.result()
        assert second.tuples == first.tuples
        return first, q.report()
    finally:
        q.close()
        if own:
            for t in tabs:
                t.close()


@pytest.mark.parametrize("groups,sel", [(1 << 16, 0.3), (1 << 18, 0.02), (1 << 20, 0.5)])
def test_device_tail_equals_host_tail_equals_oracle(gpu_ctx, monkeypatch, capfd, groups, sel):
    t = tpch.synthetic_table(900_001, groups)
    plan = tpch.synthetic_plan(t, int(sel * (1 << 31)))
    want = orc.execute(plan)
    monkeypatch.setenv("RSQ_TRACE", "1")
    got, _ = _run(gpu_ctx, plan)
    assert "device tail" in capfd.readouterr().err
    assert got.n_rows == want.n_rows and got.text == want.text and got.tuples == want.tuples
    monkeypatch.setenv("RSQ_DEVICE_TAIL", "0")
    host, _ = _run(gpu_ctx, plan)
    assert "device tail" not in capfd.readouterr().err
    assert host.tuples == want.tuples


def _mixed_plan(t, limit=None):
    p = P.Plan([t])
    k1, k2, k3 = p.attr("k1"), p.attr("k2"), p.attr("flag")
    aggs = [p.sum(p.attr("x")), p.min(p.attr("d")), p.max(p.attr("x")), p.avg(p.attr("x")), p.count(p.star()), p.min(p.attr("y")), p.avg(p.attr("y"))]
    node = p.selection(p.gt(p.attr("y"), p.constant("100", P.BIGINT)), p.scan("t"))
    node = p.aggregation(aggs, [k1, k2, k3], node)
    node = p.projection([p.as_("a3", aggs[3]), k2, p.as_("a0", aggs[0]), k3, p.as_("lo", aggs[1]), k1, p.as_("hi", aggs[2]), p.as_("n", aggs[4]),
                         p.as_("ylo", aggs[5]), p.as_("yavg", aggs[6])], node)
    return p.set_root(p.materialize(node), limit=limit)


def _mixed_table(n=400_000, seed=5):
    rng = np.random.default_rng(seed)
    return P.Table("t", [P.Column("k1", T.BIGINT(), rng.integers(-7, 300, n).astype(np.int64)),
                         P.Column("k2", T.DATE(), (19920101 + rng.integers(0, 28, n)).astype(np.uint32)),
                         P.Column("flag", T.CHAR(1), rng.choice(np.frombuffer(b"ANR", dtype=np.uint8), n)),
                         P.Column("d", T.DATE(), (19950101 + rng.integers(0, 28, n)).astype(np.uint32)),
                         P.Column("x", T.DECIMAL(12, 2), rng.integers(-5000, 100000, n).astype(np.int64)),
                         P.Column("y", T.BIGINT(), rng.integers(0, 1000, n).astype(np.int64))], n)


@pytest.mark.parametrize("limit", [None, 1, 1234])
def test_three_keys_of_three_types_min_max_avg_and_limit(gpu_ctx, monkeypatch, capfd, limit):
    """307 x 28 x 3 groups (below the size the device tail starts at by itself: forced), output columns in another order than the
    group row, DATE / CHAR(1) / BIGINT keys (4-, 1- and 8-byte tuple fields), AVG of two accumulators, LIMIT of the MaterializeOp"""
    monkeypatch.setenv("RSQ_DEVICE_TAIL_MIN", "1")
    monkeypatch.setenv("RSQ_TRACE", "1")
    t = _mixed_table()
    plan = _mixed_plan(t, limit)
    want = orc.execute(plan)
    got, _ = _run(gpu_ctx, plan)
    assert "device tail" in capfd.readouterr().err
    assert got.n_rows == want.n_rows
    assert got.text == want.text and got.tuples == want.tuples


def test_computed_projection_or_order_by_keep_the_host_tail(gpu_ctx, monkeypatch, capfd):
    monkeypatch.setenv("RSQ_DEVICE_TAIL_MIN", "1")
    monkeypatch.setenv("RSQ_TRACE", "1")
    t = tpch.synthetic_table(200_000, 1 << 16)
    p = P.Plan([t])
    b = p.attr("b")
    sc, sd, cnt = p.sum(p.attr("c")), p.sum(p.attr("d")), p.count(p.star())
    node = p.aggregation([sc, sd, cnt], [b], p.scan("t"))
    node = p.projection([b, p.as_("plus", p.add(sc, sd)), p.as_("n", cnt)], node)
    plan = p.set_root(p.materialize(node))
    got, _ = _run(gpu_ctx, plan)
    assert "device tail" not in capfd.readouterr().err
    assert got.text == orc.execute(plan).text


def test_emission_order_any_gives_the_same_rows(monkeypatch):
    ctx = engine.Context(device=0, emission_order=engine.EMIT_ANY)
    try:
        t = tpch.synthetic_table(500_000, 1 << 17)
        plan = tpch.synthetic_plan(t, 1 << 30)
        want = orc.execute(plan)
        for dev_tail in ("1", "0"):
            monkeypatch.setenv("RSQ_DEVICE_TAIL", dev_tail)
            got, _ = _run(ctx, plan)
            assert got.n_rows == want.n_rows
            assert sorted(got.text.splitlines()) == sorted(want.text.splitlines())
            b = [got.value(r, 0) for r in range(0, got.n_rows, 997)]
            assert b == sorted(b)            # ... in group-id order: deterministic, not the reference's
    finally:
        ctx.close()


# ---- the same tail over the GROUP ROWS of a hash / join-entry aggregation (engine.cpp runRowsDeviceTail, devtail.hip k_row_*) -------
def _hash_plan(t, order=False, limit=None, strings=False):
    """computed group key (generic hash aggregation), sums / count / min / max / avg; optionally ORDER BY over the result"""
    p = P.Plan([t])
    key = p.add(p.mul(p.attr("b"), p.constant("3", P.BIGINT)), p.constant("1", P.BIGINT))
    sc, cnt, lo, hi, av = p.sum(p.attr("c")), p.count(p.star()), p.min(p.attr("d")), p.max(p.attr("c")), p.avg(p.attr("d"))
    node = p.selection(p.lt(p.attr("a"), p.constant(str(1 << 30), P.BIGINT)), p.scan("t"))
    node = p.aggregation([sc, cnt, lo, hi, av], [key], node)
    node = p.projection([p.as_("k", key), p.as_("s", sc), p.as_("n", cnt), p.as_("lo", lo), p.as_("hi", hi), p.as_("av", av)], node)
    if order:
        return p.set_root(p.orderby([p.desc(p.attr("n")), p.asc(p.attr("lo"))], node), limit=limit)
    return p.set_root(p.materialize(node), limit=limit)


@pytest.mark.parametrize("order,limit", [(False, None), (False, 777), (True, None)])
def test_group_rows_of_a_hash_aggregation_are_finished_on_the_device(gpu_ctx, monkeypatch, capfd, order, limit):
    """100 000 groups of a computed key: no ORDER BY -> the text IS the reference's emission order (aggregation.h:298-343), made by the
    device replay; with ORDER BY the device delivers the tuples in that order and the host runs the reference's quicksort over them
    (ties on both sort keys keep the order that quicksort leaves them in).  Same bytes as the host tail and the oracle."""
    t = tpch.synthetic_table(700_001, 100_000)
    plan = _hash_plan(t, order, limit)
    want = orc.execute(plan)
    monkeypatch.setenv("RSQ_TRACE", "1")
    got, _ = _run(gpu_ctx, plan)
    assert "device tail over" in capfd.readouterr().err
    assert got.n_rows == want.n_rows and got.text == want.text and got.tuples == want.tuples
    monkeypatch.setenv("RSQ_DEVICE_TAIL", "0")
    host, _ = _run(gpu_ctx, plan)
    assert "device tail over" not in capfd.readouterr().err
    assert host.tuples == want.tuples


def test_group_rows_with_string_group_values(gpu_ctx, monkeypatch, capfd):
    """VARCHAR and CHAR(1) group values next to an integer one: hashVarchar / the CHAR(1) rule / the integer rule summed as Values::hash
    does (ValuesJitFlounder.h:65-162), strings written into the tuples NUL-terminated (values.h:136)"""
    # (hashVarchar sums its characters' images, so strings that are permutations of one another collide - qlib/hash.h:131-147 - and the
    # reference's table, which the oracle restates, degrades to long probe chains: a few thousand groups, not hundreds of thousands)
    n = 60_000
    rng = np.random.default_rng(11)
    names = np.array([f"name{(i * 7919) % 4_000:06d}".encode() for i in range(n)], dtype="S12")
    monkeypatch.setenv("RSQ_DEVICE_TAIL_MIN", "1000")
    t = P.Table("t", [P.Column("s", T.VARCHAR(12), names), P.Column("f", T.CHAR(1), rng.integers(65, 68, n).astype(np.uint8)),
                      P.Column("x", T.BIGINT(), rng.integers(0, 1000, n).astype(np.int64))], n)
    p = P.Plan([t])
    sx, cnt = p.sum(p.attr("x")), p.count(p.star())
    node = p.aggregation([sx, cnt], [p.attr("s"), p.attr("f")], p.scan("t"))
    plan = p.set_root(p.materialize(p.projection([p.attr("f"), p.as_("total", sx), p.attr("s"), p.as_("n", cnt)], node)))
    want = orc.execute(plan)
    monkeypatch.setenv("RSQ_TRACE", "1")
    got, _ = _run(gpu_ctx, plan)
    assert "device tail over" in capfd.readouterr().err
    assert got.n_rows == want.n_rows > 4000 and got.text == want.text and got.tuples == want.tuples


def test_q3_without_limit_takes_the_device_tail(gpu_ctx, monkeypatch, capfd):
    """aggregation at the join entry (TPC-H Q3's groups hang off the orders entry), ORDER BY revenue desc, o_orderdate, no LIMIT: the
    device orders, hashes, replays and builds the tuples, the host runs the reference's quicksort - byte for byte the oracle's rows"""
    sf = 0.2
    plan = tpch.q3_plan(tpch.customer_table(sf), tpch.orders_table(sf), tpch.lineitem_table(sf, tpch.Q3_LINEITEM_COLUMNS), limit=None)
    want = orc.execute(plan)
    monkeypatch.setenv("RSQ_TRACE", "1")
    monkeypatch.setenv("RSQ_DEVICE_TAIL_MIN", "1000")
    got, _ = _run(gpu_ctx, plan)
    assert "device tail over" in capfd.readouterr().err
    assert got.n_rows == want.n_rows > 1000 and got.text == want.text
