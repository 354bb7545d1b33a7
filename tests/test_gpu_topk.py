"""GPU parity for ORDER BY ... LIMIT k above join-entry / hash aggregations with many groups: the engine pre-selects the
candidate rows on the device (radix select on the first sort key, aot_kernels.hip selectTopCandidates) and the host tail
sorts only those.  Every case is compared with the CPU oracle byte for byte, with the pre-selection on and off."""
import os

import pytest

from resql_amd import plan as P, tpch
from oracle import orc

pytestmark = pytest.mark.gpu


def run_both_ways(gpu_ctx, plan):
    want = orc.execute(plan)
    got = gpu_ctx.run(plan)
    assert got.text == want.text
    assert got.tuples == want.tuples
    os.environ["RSQ_DEVICE_TOPK"] = "0"
    try:
        off = gpu_ctx.run(plan)
    finally:
        del os.environ["RSQ_DEVICE_TOPK"]
    assert off.text == want.text
    return want


def hash_agg_plan(t, order, limit, shift="1048576", sel=1 << 30):
    """select b*2+1 as k, sum(c - shift) as s, min(d) as lo, count(*) as n from t where a < sel group by b*2+1 order by ... limit"""
    p = P.Plan([t])
    key = p.add(p.mul(p.attr("b"), p.constant("2", P.BIGINT)), p.constant("1", P.BIGINT))
    s = p.sum(p.sub(p.attr("c"), p.constant(shift, P.BIGINT)))
    lo, n = p.min(p.attr("d")), p.count(p.star())
    node = p.selection(p.lt(p.attr("a"), p.constant(str(sel), P.BIGINT)), p.scan(t.name))
    node = p.aggregation([s, lo, n], [key], node)
    node = p.projection([p.as_("k", key), p.as_("s", s), p.as_("lo", lo), p.as_("n", n)], node)
    node = p.orderby(order(p), node)
    return p.set_root(node, limit=limit)


@pytest.mark.parametrize("limit", [1, 10, 100])
def test_sum_descending_and_ascending(gpu_ctx, limit):
    """first key = a SUM accumulator with negative and positive values (c - 2^19 summed over ~20 rows per group)"""
    t = tpch.synthetic_table(200_000, 10_000)
    for order in (lambda p: [p.desc(p.attr("s")), p.attr("k")], lambda p: [p.attr("s"), p.desc(p.attr("k"))]):
        want = run_both_ways(gpu_ctx, hash_agg_plan(t, order, limit, shift="524288"))
        assert want.n_rows == limit


def test_first_key_is_the_group_value(gpu_ctx):
    t = tpch.synthetic_table(100_000, 20_000)
    run_both_ways(gpu_ctx, hash_agg_plan(t, lambda p: [p.desc(p.attr("k"))], 25))
    run_both_ways(gpu_ctx, hash_agg_plan(t, lambda p: [p.attr("k")], 25))


def test_heavy_ties_on_the_first_key_resolved_by_the_second(gpu_ctx):
    """COUNT(*) over uniform groups ties massively; the tie set at the threshold is larger than the candidate buffer or not
    depending on the limit — both ways must agree with the oracle"""
    t = tpch.synthetic_table(120_000, 6_000)
    for limit in (5, 60):
        run_both_ways(gpu_ctx, hash_agg_plan(t, lambda p: [p.desc(p.attr("n")), p.attr("k")], limit))
        run_both_ways(gpu_ctx, hash_agg_plan(t, lambda p: [p.attr("n"), p.desc(p.attr("lo")), p.attr("k")], limit))


def test_ties_on_all_keys_fall_back_to_the_reference_order(gpu_ctx):
    """ORDER BY n only: the leading rows tie on every key, the answer is the reference's quicksort order over its emission
    order, which needs all groups"""
    t = tpch.synthetic_table(120_000, 6_000)
    run_both_ways(gpu_ctx, hash_agg_plan(t, lambda p: [p.desc(p.attr("n"))], 10))
    run_both_ways(gpu_ctx, hash_agg_plan(t, lambda p: [p.attr("lo")], 10))


def test_fewer_groups_than_the_limit_asks_for(gpu_ctx):
    t = tpch.synthetic_table(50_000, 3_000)
    want = run_both_ways(gpu_ctx, hash_agg_plan(t, lambda p: [p.desc(p.attr("s")), p.attr("k")], 2_000))
    assert want.n_rows == 2_000
    want = run_both_ways(gpu_ctx, hash_agg_plan(t, lambda p: [p.desc(p.attr("s")), p.attr("k")], 5_000))
    assert want.n_rows == 3_000


@pytest.mark.parametrize("order", ["date_first", "date_desc", "revenue_asc"])
def test_q3_other_orders(gpu_ctx, order):
    """join-entry aggregation (TPC-H Q3 at SF0.2: ~2 K groups per 0.1 SF) ordered by a DATE group value first (32-bit key,
    ~100 groups per date) or by ascending revenue"""
    sf = 0.3
    orders = {
        "date_first": lambda p: [p.attr("o_orderdate"), p.desc(p.attr("revenue"))],
        "date_desc": lambda p: [p.desc(p.attr("o_orderdate")), p.attr("revenue"), p.attr("l_orderkey")],
        "revenue_asc": lambda p: [p.attr("revenue"), p.attr("l_orderkey")],
    }
    li = tpch.lineitem_table(sf, tpch.Q3_LINEITEM_COLUMNS)
    plan = tpch.q3_plan(tpch.customer_table(sf), tpch.orders_table(sf), li, limit=20, order_by=orders[order])
    run_both_ways(gpu_ctx, plan)


def test_string_group_keys_take_the_candidate_path(gpu_ctx):
    """TPC-H Q10 (group by c_custkey, c_name, ..., five string values; ORDER BY revenue DESC LIMIT 20) at SF 0.2: ~7 K
    groups keyed by 32 words.  No CHAR value ends with a space, the kernel reports that no host merge is needed and the
    candidates are selected on the device; a second table whose CHAR values do end with spaces must take the full path."""
    from resql_amd import tpch_full
    db = tpch_full.database(0.2)
    host = [db[k] for k in sorted(db)]
    tabs = [gpu_ctx.table(t) for t in host]
    try:
        sql = tpch_full.QUERIES["q10"]
        want = orc.execute(gpu_ctx.sql_plan(sql, tabs, host))
        q = gpu_ctx.sql_compile(sql, tabs)
        for _ in range(2):
            q.execute()
            assert q.result().text == want.text
        q.close()
        assert want.n_rows == 20
    finally:
        for t in tabs:
            t.close()
    # CHAR(6) group values that differ only in trailing spaces: one group in the reference, merged on the host
    import numpy as np
    n = 60_000
    r = np.arange(n)
    names = np.array([b"k%03d" % (i % 3000) + (b" " if (i // 3000) % 2 else b"") for i in range(n)], dtype="S6")
    t = P.Table("t", [P.Column("name", P.TypeInit.CHAR(6), names), P.Column("v", P.TypeInit.BIGINT(), (r % 97).astype(np.int64))], n)
    p = P.Plan([t])
    s = p.sum(p.attr("v"))
    node = p.aggregation([s], [p.attr("name")], p.scan("t"))
    node = p.projection([p.attr("name"), p.as_("s", s)], node)
    node = p.orderby([p.desc(p.attr("s")), p.attr("name")], node)
    run_both_ways(gpu_ctx, p.set_root(node, limit=7))


def dense_plan(t, order, limit):
    """select b, sum(c - 2^19) as s, min(d) as lo, count(*) as n from t where a < 2^30 group by b order by ... limit (b is a column with
    statistics: a dense group id, the aggregate table lives in HBM)"""
    p = P.Plan([t])
    b = p.attr("b")
    s = p.sum(p.sub(p.attr("c"), p.constant("524288", P.BIGINT)))
    lo, n = p.min(p.attr("d")), p.count(p.star())
    node = p.selection(p.lt(p.attr("a"), p.constant(str(1 << 30), P.BIGINT)), p.scan(t.name))
    node = p.aggregation([s, lo, n], [b], node)
    node = p.projection([b, p.as_("s", s), p.as_("lo", lo), p.as_("n", n)], node)
    node = p.orderby(order(p), node)
    return p.set_root(node, limit=limit)


def test_large_dense_tables_take_the_candidate_path(gpu_ctx):
    """200 K dense groups (about half of them present): ORDER BY an aggregate ... LIMIT reads back a few candidate rows instead of
    the whole aggregate table; ties on all keys fall back to the table"""
    t = tpch.synthetic_table(300_000, 200_000)
    for order, limit in ((lambda p: [p.desc(p.attr("s")), p.attr("b")], 10), (lambda p: [p.attr("s"), p.desc(p.attr("b"))], 25),
                         (lambda p: [p.desc(p.attr("n")), p.attr("lo"), p.attr("b")], 7), (lambda p: [p.attr("lo")], 5),
                         (lambda p: [p.desc(p.attr("n"))], 5)):
        want = run_both_ways(gpu_ctx, dense_plan(t, order, limit))
        assert want.n_rows == limit


def test_wide_group_rows_select_from_slot_and_key_rows(gpu_ctx):
    """group rows of more than eight words (two CHAR(40) group values) above 6 000 groups: the compaction writes [slot | sort key] rows
    and the selection fetches its candidates from the table (engine.cpp narrowRows).  ORDER BY a sum decides from the candidates;
    ORDER BY count(*) ties every group - the candidates overflow, the execution starts over with full rows; both must agree with the
    oracle, executed twice (the second execution remembers which way it went)."""
    import numpy as np
    n, g = 90_000, 6_000
    r = np.arange(n)
    a = np.array([b"customer-name-%06d-of-the-first-kind" % (i % g) for i in range(n)], dtype="S40")
    b = np.array([b"address-%06d-somewhere-far-away" % ((i % g) * 7 % g) for i in range(n)], dtype="S40")
    v = ((r * 2654435761) % 1000 - 300).astype(np.int64)
    t = P.Table("t", [P.Column("a", P.TypeInit.CHAR(40), a), P.Column("b", P.TypeInit.CHAR(40), b), P.Column("v", P.TypeInit.BIGINT(), v)], n)
    for order in (lambda p, s, c: [p.desc(p.attr("s")), p.attr("a")], lambda p, s, c: [p.desc(p.attr("c")), p.attr("a")], lambda p, s, c: [p.attr("s")]):
        p = P.Plan([t])
        s, c = p.sum(p.attr("v")), p.count(p.star())
        node = p.aggregation([s, c], [p.attr("a"), p.attr("b")], p.scan("t"))
        node = p.projection([p.attr("a"), p.attr("b"), p.as_("s", s), p.as_("c", c)], node)
        node = p.orderby(order(p, s, c), node)
        plan = p.set_root(node, limit=12)
        want = orc.execute(plan)
        tabs = [gpu_ctx.table(t)]
        q = gpu_ctx.compile(plan, tabs)
        try:
            for _ in range(3):
                q.execute()
                got = q.result()
                assert got.text == want.text and got.tuples == want.tuples
        finally:
            q.close(); tabs[0].close()
    run_both_ways(gpu_ctx, plan)
