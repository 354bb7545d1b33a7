"""BULK INSERT cases: '.tbl' fixtures (the head of the reference's own tpch/datasets/sf001 files, kept as data under
tests/golden/tbl/) and the plans run over them.  Shared by the golden generator, the CPU tests and the GPU tests."""
import os

from resql_amd import plan as P, tpch

TBL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tbl")
FILES = {"customer": os.path.join(TBL_DIR, "customer.tbl"), "orders": os.path.join(TBL_DIR, "orders.tbl")}
SCHEMAS = {"customer": tpch.CUSTOMER_SCHEMA, "orders": tpch.ORDERS_SCHEMA}


def schema_table(name: str) -> P.Table:
    return P.Table(name, [P.Column(n, t) for n, t in SCHEMAS[name]], 0)


def scan_plan(table: P.Table) -> P.Plan:
    """every column of every row, in file order"""
    p = P.Plan([table])
    return p.set_root(p.materialize(p.scan(table.name)), request_all=True)


def orders_by_status(orders: P.Table) -> P.Plan:
    p = P.Plan([orders])
    st = p.attr("o_orderstatus")
    tot, cnt, lo, hi = p.sum(p.attr("o_totalprice")), p.count(p.star()), p.min(p.attr("o_orderdate")), p.max(p.attr("o_totalprice"))
    avg = p.avg(p.attr("o_totalprice"))
    node = p.selection(p.ge(p.attr("o_orderdate"), p.constant("1993-01-01", P.DATE)), p.scan("orders"))
    node = p.aggregation([tot, cnt, lo, hi, avg], [st], node)
    node = p.projection([st, p.as_("total", tot), p.as_("n", cnt), p.as_("first", lo), p.as_("top", hi), p.as_("mean", avg)], node)
    return p.set_root(p.orderby([p.attr("o_orderstatus")], node))


def building_orders(customer: P.Table, orders: P.Table) -> P.Plan:
    """the customer x orders half of Q3 on real dbgen rows: string filter, join, group by date parts"""
    p = P.Plan([customer, orders])
    sel_c = p.selection(p.eq(p.attr("c_mktsegment"), p.constant("BUILDING", P.VARCHAR)), p.scan("customer"))
    sel_o = p.selection(p.lt(p.attr("o_orderdate"), p.constant("1995-03-15", P.DATE)), p.scan("orders"))
    hj = p.hashjoin([p.eq(p.attr("c_custkey"), p.attr("o_custkey"))], sel_c, sel_o, single_match=True)
    key, tot, cnt = p.attr("o_shippriority"), p.sum(p.attr("o_totalprice")), p.count(p.star())
    bal = p.max(p.attr("c_acctbal"))
    node = p.aggregation([tot, cnt, bal], [p.attr("o_orderstatus"), key], hj)
    node = p.projection([p.attr("o_orderstatus"), key, p.as_("total", tot), p.as_("n", cnt), p.as_("bal", bal)], node)
    return p.set_root(p.materialize(node))


QUERIES = {
    "scan_customer": (("customer",), lambda t: scan_plan(t["customer"])),
    "scan_orders": (("orders",), lambda t: scan_plan(t["orders"])),
    "orders_by_status": (("orders",), lambda t: orders_by_status(t["orders"])),
    "building_orders": (("customer", "orders"), lambda t: building_orders(t["customer"], t["orders"])),
}
