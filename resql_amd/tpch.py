"""TPC-H schemas (reference tpch/create.sql) and the plans the reference's planner
(src/planner.h:409-497) produces for tpch/queries/q1.sql, q3.sql, q6.sql, hand-built with the
same expression sharing (planner.h:90-99 unifySelectAndGroupby; aggregates inside the select
list are the same nodes the AggregationOp receives, planner.h:430).

The literal texts and their type categories are what the reference's lexer/parser hand to
ExprGen::constant (parser/parseSql.h:85-122, parser/parser.y:152-160): integers -> BIGINT,
numbers with a point -> DECIMAL, multi-character strings -> VARCHAR, `date "..."` -> DATE.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import datagen
from . import plan as P
from .plan import TypeInit as T

LINEITEM_SCHEMA = [
    ("l_orderkey", T.INT()), ("l_partkey", T.INT()), ("l_suppkey", T.INT()), ("l_linenumber", T.INT()),
    ("l_quantity", T.DECIMAL(12, 0)), ("l_extendedprice", T.DECIMAL(12, 2)), ("l_discount", T.DECIMAL(12, 2)),
    ("l_tax", T.DECIMAL(12, 2)), ("l_returnflag", T.CHAR(1)), ("l_linestatus", T.CHAR(1)),
    ("l_shipdate", T.DATE()), ("l_commitdate", T.DATE()), ("l_receiptdate", T.DATE()),
    ("l_shipinstruct", T.CHAR(25)), ("l_shipmode", T.CHAR(10)), ("l_comment", T.VARCHAR(44)),
]
ORDERS_SCHEMA = [
    ("o_orderkey", T.INT()), ("o_custkey", T.INT()), ("o_orderstatus", T.CHAR(1)),
    ("o_totalprice", T.DECIMAL(12, 2)), ("o_orderdate", T.DATE()), ("o_orderpriority", T.CHAR(15)),
    ("o_clerk", T.CHAR(15)), ("o_shippriority", T.INT()), ("o_comment", T.VARCHAR(79)),
]
CUSTOMER_SCHEMA = [
    ("c_custkey", T.INT()), ("c_name", T.VARCHAR(25)), ("c_address", T.VARCHAR(40)), ("c_nationkey", T.INT()),
    ("c_phone", T.CHAR(15)), ("c_acctbal", T.DECIMAL(12, 2)), ("c_mktsegment", T.CHAR(10)),
    ("c_comment", T.VARCHAR(117)),
]

Q1_COLUMNS = ["l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate"]
Q6_COLUMNS = ["l_quantity", "l_extendedprice", "l_discount", "l_shipdate"]
Q3_LINEITEM_COLUMNS = ["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"]

# ALGORITHMIC bytes per row: the column bytes a query must read once
Q1_BYTES_PER_ROW = 8 + 8 + 8 + 8 + 1 + 1 + 4      # 38
Q6_BYTES_PER_ROW = 4 + 8 + 8 + 8                  # 28
SYNTH_BYTES_PER_ROW = 32


def make_table(name: str, schema, data: Dict[str, np.ndarray], n_rows: int) -> P.Table:
    return P.Table(name, [P.Column(n, t, data.get(n)) for n, t in schema], n_rows)


def lineitem_table(sf: float, columns, seed: int = datagen.SEED, n_rows: Optional[int] = None) -> P.Table:
    n = datagen.n_lineitem(sf) if n_rows is None else n_rows
    return make_table("lineitem", LINEITEM_SCHEMA, datagen.lineitem_columns(0, n, sf, seed, set(columns)), n)


def orders_table(sf: float, seed: int = datagen.SEED) -> P.Table:
    n = datagen.n_orders(sf)
    return make_table("orders", ORDERS_SCHEMA, datagen.orders_columns(0, n, sf, seed), n)


def customer_table(sf: float, seed: int = datagen.SEED) -> P.Table:
    n = datagen.n_customer(sf)
    return make_table("customer", CUSTOMER_SCHEMA, datagen.customer_columns(0, n, sf, seed), n)


# ------------------------------------------------------------------------------------------------
def q1_plan(lineitem: P.Table, shipdate: str = "1998-9-02") -> P.Plan:
    """tpch/queries/q1.sql; plan: OrderBy <- Materialize <- Projection <- Aggregation <- Selection <- Scan"""
    p = P.Plan([lineitem])
    rf, ls = p.attr("l_returnflag"), p.attr("l_linestatus")

    def disc_price():
        return p.mul(p.attr("l_extendedprice"), p.sub(p.constant("1", P.BIGINT), p.attr("l_discount")))

    sum_qty = p.sum(p.attr("l_quantity"))
    sum_base = p.sum(p.attr("l_extendedprice"))
    sum_disc = p.sum(disc_price())
    sum_charge = p.sum(p.mul(disc_price(), p.add(p.constant("1", P.BIGINT), p.attr("l_tax"))))
    avg_qty = p.avg(p.attr("l_quantity"))
    avg_price = p.avg(p.attr("l_extendedprice"))
    avg_disc = p.avg(p.attr("l_discount"))
    cnt = p.count(p.star())
    aggs = [sum_qty, sum_base, sum_disc, sum_charge, avg_qty, avg_price, avg_disc, cnt]
    select = [rf, ls,
              p.as_("sum_qty", sum_qty), p.as_("sum_base_price", sum_base), p.as_("sum_disc_price", sum_disc),
              p.as_("sum_charge", sum_charge), p.as_("avg_qty", avg_qty), p.as_("avg_price", avg_price),
              p.as_("avg_disc", avg_disc), p.as_("count_order", cnt)]
    where = p.le(p.attr("l_shipdate"), p.constant(shipdate, P.DATE))
    plan = p.scan("lineitem")
    plan = p.selection(where, plan)
    plan = p.aggregation(aggs, [rf, ls], plan)
    plan = p.projection(select, plan)
    plan = p.orderby([p.attr("l_returnflag"), p.attr("l_linestatus")], plan)
    return p.set_root(plan)


def q6_plan(lineitem: P.Table, date_lo: str = "1994-01-01", date_hi: str = "1995-01-01",
            discount: str = "0.06", quantity: str = "24") -> P.Plan:
    """tpch/queries/q6.sql; BETWEEN is desugared by the parser (parser.y:102-105) and the
    conjuncts are re-chained left-deep by pushDownSelection (planner.h:191-195)."""
    p = P.Plan([lineitem])
    conds = [
        p.ge(p.attr("l_shipdate"), p.constant(date_lo, P.DATE)),
        p.lt(p.attr("l_shipdate"), p.constant(date_hi, P.DATE)),
        p.ge(p.attr("l_discount"), p.sub(p.constant(discount, P.DECIMAL), p.constant("0.01", P.DECIMAL))),
        p.le(p.attr("l_discount"), p.add(p.constant(discount, P.DECIMAL), p.constant("0.01", P.DECIMAL))),
        p.lt(p.attr("l_quantity"), p.constant(quantity, P.BIGINT)),
    ]
    rev = p.sum(p.mul(p.attr("l_extendedprice"), p.attr("l_discount")))
    plan = p.scan("lineitem")
    plan = p.selection(p.conjunction(conds), plan)
    plan = p.aggregation([rev], [], plan)
    plan = p.projection([p.as_("revenue", rev)], plan)
    plan = p.materialize(plan)
    return p.set_root(plan)


def q3_plan(customer: P.Table, orders: P.Table, lineitem: P.Table, segment: str = "BUILDING",
            date: str = "1995-03-15", limit: int = 10, order_by=None) -> P.Plan:
    """tpch/queries/q3.sql; plan:
    OrderBy <- Materialize <- Projection <- Aggregation <-
        HashJoin[ build = HashJoin[ build = sel(customer), probe = sel(orders) ] (multi-match),
                  probe = sel(lineitem) ] (single-match: o_orderkey is unique, planner.h:218-241)"""
    p = P.Plan([customer, orders, lineitem])
    sel_c = p.selection(p.eq(p.attr("c_mktsegment"), p.constant(segment, P.VARCHAR)), p.scan("customer"))
    sel_o = p.selection(p.lt(p.attr("o_orderdate"), p.constant(date, P.DATE)), p.scan("orders"))
    sel_l = p.selection(p.gt(p.attr("l_shipdate"), p.constant(date, P.DATE)), p.scan("lineitem"))
    hj1 = p.hashjoin([p.eq(p.attr("c_custkey"), p.attr("o_custkey"))], sel_c, sel_o, single_match=False)
    hj2 = p.hashjoin([p.eq(p.attr("o_orderkey"), p.attr("l_orderkey"))], hj1, sel_l, single_match=True)
    g_ok, g_od, g_sp = p.attr("l_orderkey"), p.attr("o_orderdate"), p.attr("o_shippriority")
    rev = p.sum(p.mul(p.attr("l_extendedprice"), p.sub(p.constant("1", P.BIGINT), p.attr("l_discount"))))
    plan = p.aggregation([rev], [g_ok, g_od, g_sp], hj2)
    plan = p.projection([g_ok, p.as_("revenue", rev), g_od, g_sp], plan)
    plan = p.orderby(order_by(p) if order_by else [p.desc(p.attr("revenue")), p.attr("o_orderdate")], plan)
    return p.set_root(plan, limit=limit)


def synthetic_plan(table: P.Table, threshold: int) -> P.Plan:
    """BASELINE config 5: select b, sum(c), sum(d), count(*) from t where a < threshold group by b"""
    p = P.Plan([table])
    b = p.attr("b")
    sc, sd, cnt = p.sum(p.attr("c")), p.sum(p.attr("d")), p.count(p.star())
    plan = p.scan(table.name)
    plan = p.selection(p.lt(p.attr("a"), p.constant(str(threshold), P.BIGINT)), plan)
    plan = p.aggregation([sc, sd, cnt], [b], plan)
    plan = p.projection([b, p.as_("sum_c", sc), p.as_("sum_d", sd), p.as_("cnt", cnt)], plan)
    plan = p.materialize(plan)
    return p.set_root(plan)


SYNTH_SCHEMA = [("a", T.BIGINT()), ("b", T.BIGINT()), ("c", T.BIGINT()), ("d", T.BIGINT())]


def synthetic_table(n: int, groups: int, seed: int = datagen.SEED) -> P.Table:
    return make_table("t", SYNTH_SCHEMA, datagen.synthetic_columns(0, n, groups, seed), n)


def standard_plans(sf: float = 0.01):
    """the plans whose pipeline kernels are pre-compiled at build time (BASELINE.json configs)"""
    li = lineitem_table(sf, Q1_COLUMNS + ["l_orderkey"])
    cu, od = customer_table(sf), orders_table(sf)
    plans = [q1_plan(li), q6_plan(li), q3_plan(cu, od, li)]
    for groups in (8, 1024, 1 << 20):
        for thr in (1 << 30,):
            plans.append(synthetic_plan(synthetic_table(4096, groups), thr))
    return plans
