"""The eight-table TPC-H-shaped database for the SQL front-end tests: every column that any of the reference's own
queries (tpch/queries/q1, q3, q5, q6, q10, q12, q14, q19) touches, deterministic and counter-based like datagen.py.

lineitem / orders / customer keep the columns datagen.py already defines (same values, so Q1 / Q3 / Q6 answers do not
change) and gain the ones the other queries need; part, supplier, nation and region are new.  Schemas are those of the
reference's tpch/create.sql.  Value domains follow the TPC-H specification's lists (ship modes, instructions, priorities,
containers, brands, type syllables, nations / regions) so that the queries' constants select something.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np

from . import datagen
from . import plan as P
from . import tpch
from .plan import TypeInit as T

# further random streams (datagen.py uses 1..11 and 21..24)
(S_LSUPP, S_LCOMMIT, S_LINSTR, S_LMODE, S_OPRIO, S_CNATION, S_CBAL, S_CPHONE, S_SNATION, S_SBAL, S_PBRAND, S_PTYPE, S_PSIZE,
 S_PCONT, S_CADDR, S_CCOMM) = range(31, 47)

PART_SCHEMA = [
    ("p_partkey", T.INT()), ("p_name", T.CHAR(55)), ("p_mfgr", T.CHAR(55)), ("p_brand", T.CHAR(10)), ("p_type", T.VARCHAR(25)),
    ("p_size", T.INT()), ("p_container", T.CHAR(10)), ("p_retailprice", T.DECIMAL(12, 2)), ("p_comment", T.VARCHAR(23)),
]
SUPPLIER_SCHEMA = [
    ("s_suppkey", T.INT()), ("s_name", T.CHAR(25)), ("s_address", T.VARCHAR(40)), ("s_nationkey", T.INT()), ("s_phone", T.CHAR(15)),
    ("s_acctbal", T.DECIMAL(12, 2)), ("s_comment", T.VARCHAR(101)),
]
NATION_SCHEMA = [("n_nationkey", T.INT()), ("n_name", T.CHAR(25)), ("n_regionkey", T.INT()), ("n_comment", T.VARCHAR(152))]
REGION_SCHEMA = [("r_regionkey", T.INT()), ("r_name", T.CHAR(25)), ("r_comment", T.VARCHAR(152))]
PARTSUPP_SCHEMA = [("ps_partkey", T.INT()), ("ps_suppkey", T.INT()), ("ps_availqty", T.INT()), ("ps_supplycost", T.DECIMAL(12, 2)),
                   ("ps_comment", T.VARCHAR(199))]

SHIPMODES = [b"REG AIR", b"AIR", b"RAIL", b"SHIP", b"TRUCK", b"MAIL", b"FOB"]
INSTRUCTIONS = [b"DELIVER IN PERSON", b"COLLECT COD", b"NONE", b"TAKE BACK RETURN"]
PRIORITIES = [b"1-URGENT", b"2-HIGH", b"3-MEDIUM", b"4-NOT SPECIFIED", b"5-LOW"]
CONTAINERS = [a + b" " + b for a in (b"SM", b"LG", b"MED", b"JUMBO", b"WRAP") for b in (b"CASE", b"BOX", b"BAG", b"JAR", b"PKG", b"PACK", b"CAN", b"DRUM")]
TYPES = [a + b" " + b + b" " + c for a in (b"STANDARD", b"SMALL", b"MEDIUM", b"LARGE", b"ECONOMY", b"PROMO")
         for b in (b"ANODIZED", b"BURNISHED", b"PLATED", b"POLISHED", b"BRUSHED") for c in (b"TIN", b"NICKEL", b"BRASS", b"STEEL", b"COPPER")]
NATIONS = [(b"ALGERIA", 0), (b"ARGENTINA", 1), (b"BRAZIL", 1), (b"CANADA", 1), (b"EGYPT", 4), (b"ETHIOPIA", 0), (b"FRANCE", 3),
           (b"GERMANY", 3), (b"INDIA", 2), (b"INDONESIA", 2), (b"IRAN", 4), (b"IRAQ", 4), (b"JAPAN", 2), (b"JORDAN", 4), (b"KENYA", 0),
           (b"MOROCCO", 0), (b"MOZAMBIQUE", 0), (b"PERU", 1), (b"CHINA", 2), (b"ROMANIA", 3), (b"SAUDI ARABIA", 4), (b"VIETNAM", 2),
           (b"RUSSIA", 3), (b"UNITED KINGDOM", 3), (b"UNITED STATES", 1)]
REGIONS = [b"AFRICA", b"AMERICA", b"ASIA", b"EUROPE", b"MIDDLE EAST"]
WORDS = [b"slyly", b"final", b"deposits", b"ironic", b"packages", b"carefully", b"bold", b"requests", b"quickly", b"even"]


def n_part(sf: float) -> int:
    return max(1, int(round(200_000 * sf)))


def n_supplier(sf: float) -> int:
    return max(5, int(round(10_000 * sf)))


def _pick(values: List[bytes], idx: np.ndarray, width: int) -> np.ndarray:
    return np.array(values, dtype=f"S{width}")[idx]


def _text(seed: int, stream: int, idx: np.ndarray, width: int, words: int) -> np.ndarray:
    """a few dictionary words per row (deterministic), cut to `width` bytes"""
    parts = [np.array(WORDS, dtype="S12")[datagen.uniform(seed, stream + 100 * k, idx, len(WORDS))] for k in range(words)]
    out = parts[0]
    for p in parts[1:]:
        out = np.char.add(np.char.add(out, b" "), p)
    return out.astype(f"S{width}")


def lineitem(sf: float, seed: int = datagen.SEED) -> P.Table:
    n = datagen.n_lineitem(sf)
    cols = datagen.lineitem_columns(0, n, sf, seed)
    r = np.arange(n, dtype=np.int64)
    block = r // datagen.LINES_PER_BLOCK
    pat = datagen.uniform(seed, datagen.S_PERM, block, 14)
    o = block * datagen.ORDERS_PER_BLOCK + datagen._ORDER_OF[pat, r % datagen.LINES_PER_BLOCK]
    cols["l_suppkey"] = (1 + datagen.uniform(seed, S_LSUPP, r, n_supplier(sf))).astype(np.int32)
    cols["l_commitdate"] = datagen.yyyymmdd(datagen.order_dates(o, seed) + 30 + datagen.uniform(seed, S_LCOMMIT, r, 61))
    cols["l_shipinstruct"] = _pick(INSTRUCTIONS, datagen.uniform(seed, S_LINSTR, r, len(INSTRUCTIONS)), 25)
    cols["l_shipmode"] = _pick(SHIPMODES, datagen.uniform(seed, S_LMODE, r, len(SHIPMODES)), 10)
    return tpch.make_table("lineitem", tpch.LINEITEM_SCHEMA, cols, n)


def orders(sf: float, seed: int = datagen.SEED) -> P.Table:
    n = datagen.n_orders(sf)
    cols = datagen.orders_columns(0, n, sf, seed)
    o = np.arange(n, dtype=np.int64)
    cols["o_orderpriority"] = _pick(PRIORITIES, datagen.uniform(seed, S_OPRIO, o, len(PRIORITIES)), 15)
    return tpch.make_table("orders", tpch.ORDERS_SCHEMA, cols, n)


def customer(sf: float, seed: int = datagen.SEED) -> P.Table:
    n = datagen.n_customer(sf)
    cols = datagen.customer_columns(0, n, sf, seed)
    c = np.arange(n, dtype=np.int64)
    cols["c_name"] = np.array([b"Customer#%09d" % (k + 1) for k in range(n)], dtype="S25")
    cols["c_address"] = _text(seed, S_CADDR, c, 40, 3)
    cols["c_nationkey"] = datagen.uniform(seed, S_CNATION, c, 25).astype(np.int32)
    ph = datagen.uniform(seed, S_CPHONE, c, 10_000_000)
    cols["c_phone"] = np.array([b"%02d-%03d-%03d-%04d" % (10 + int(nk), int(p) // 10000 % 1000, int(p) // 10 % 1000, int(p) % 10000)
                                for nk, p in zip(cols["c_nationkey"], ph)], dtype="S15")
    cols["c_acctbal"] = datagen.uniform(seed, S_CBAL, c, 1_099_999) - 99_999
    cols["c_comment"] = _text(seed, S_CCOMM, c, 117, 6)
    return tpch.make_table("customer", tpch.CUSTOMER_SCHEMA, cols, n)


def part(sf: float, seed: int = datagen.SEED) -> P.Table:
    n = n_part(sf)
    p = np.arange(n, dtype=np.int64)
    pk = p + 1
    m = 1 + datagen.uniform(seed, S_PBRAND, p, 5)
    b = 1 + datagen.uniform(seed, S_PBRAND + 200, p, 5)
    cols = {
        "p_partkey": pk.astype(np.int32),
        "p_brand": np.array([b"Brand#%d%d" % (int(x), int(y)) for x, y in zip(m, b)], dtype="S10"),
        "p_type": _pick(TYPES, datagen.uniform(seed, S_PTYPE, p, len(TYPES)), 25),
        "p_size": (1 + datagen.uniform(seed, S_PSIZE, p, 50)).astype(np.int32),
        "p_container": _pick(CONTAINERS, datagen.uniform(seed, S_PCONT, p, len(CONTAINERS)), 10),
        "p_retailprice": 90000 + ((pk // 10) % 20001) + 100 * (pk % 1000),
    }
    return tpch.make_table("part", PART_SCHEMA, cols, n)


def supplier(sf: float, seed: int = datagen.SEED) -> P.Table:
    n = n_supplier(sf)
    s = np.arange(n, dtype=np.int64)
    cols = {
        "s_suppkey": (s + 1).astype(np.int32),
        "s_name": np.array([b"Supplier#%09d" % (k + 1) for k in range(n)], dtype="S25"),
        "s_nationkey": datagen.uniform(seed, S_SNATION, s, 25).astype(np.int32),
        "s_acctbal": datagen.uniform(seed, S_SBAL, s, 1_099_999) - 99_999,
    }
    return tpch.make_table("supplier", SUPPLIER_SCHEMA, cols, n)


def nation() -> P.Table:
    cols = {
        "n_nationkey": np.arange(25, dtype=np.int32),
        "n_name": np.array([nm for nm, _ in NATIONS], dtype="S25"),
        "n_regionkey": np.array([rk for _, rk in NATIONS], dtype=np.int32),
    }
    return tpch.make_table("nation", NATION_SCHEMA, cols, 25)


def region() -> P.Table:
    cols = {"r_regionkey": np.arange(5, dtype=np.int32), "r_name": np.array(REGIONS, dtype="S25")}
    return tpch.make_table("region", REGION_SCHEMA, cols, 5)


def database(sf: float, seed: int = datagen.SEED, fill_unused: bool = True) -> Dict[str, P.Table]:
    """all tables of tpch/create.sql except partsupp (no query of the reference's set reads it); columns no generator
    fills (comments, clerks, ...) hold zeros / empty strings so that `select *` has something to read — unless
    fill_unused is False (full sizes: l_comment alone would be 2.6 GB of zeros at SF10; such columns then have no data and
    no query of QUERIES touches them)"""
    db = {t.name: t for t in (lineitem(sf, seed), orders(sf, seed), customer(sf, seed), part(sf, seed), supplier(sf, seed),
                              nation(), region())}
    for t in db.values():
        for c in t.columns:
            if c.data is None and fill_unused:
                c.data = np.zeros(t.n_rows, dtype=c.type.np_dtype)
    return db


# the reference's tpch/queries/*.sql, statement text only (the inputs of its test/test_queries.h)
QUERIES = {
    "q1": """select l_returnflag, l_linestatus, sum(l_quantity) as sum_qty, sum(l_extendedprice) as sum_base_price,
        sum(l_extendedprice * (1 - l_discount)) as sum_disc_price, sum(l_extendedprice * (1 - l_discount) * (1 + l_tax)) as sum_charge,
        avg(l_quantity) as avg_qty, avg(l_extendedprice) as avg_price, avg(l_discount) as avg_disc, count(*) as count_order
        from lineitem where l_shipdate <= date '1998-9-02' group by l_returnflag, l_linestatus order by l_returnflag, l_linestatus""",
    "q3": """select l_orderkey, sum(l_extendedprice * (1 - l_discount)) as revenue, o_orderdate, o_shippriority
        from customer, orders, lineitem
        where c_mktsegment = 'BUILDING' and c_custkey = o_custkey and l_orderkey = o_orderkey and o_orderdate < date '1995-03-15'
        and l_shipdate > date '1995-03-15' group by l_orderkey, o_orderdate, o_shippriority order by revenue desc, o_orderdate limit 10""",
    "q5": """select n_name, sum(l_extendedprice * (1 - l_discount)) as revenue
        from customer, orders, lineitem, supplier, nation, region
        where c_custkey = o_custkey and l_orderkey = o_orderkey and l_suppkey = s_suppkey and c_nationkey = s_nationkey
        and s_nationkey = n_nationkey and n_regionkey = r_regionkey and r_name = 'ASIA' and o_orderdate >= date '1994-01-01'
        and o_orderdate < date '1995-01-01' group by n_name order by revenue desc""",
    "q6": """select sum(l_extendedprice * l_discount) as revenue from lineitem
        where l_shipdate >= date '1994-01-01' and l_shipdate < date '1995-01-01' and l_discount between 0.06 - 0.01 and 0.06 + 0.01
        and l_quantity < 24""",
    "q10": """select c_custkey, c_name, sum(l_extendedprice * (1 - l_discount)) as revenue, c_acctbal, n_name, c_address, c_phone, c_comment
        from customer, orders, lineitem, nation
        where c_custkey = o_custkey and l_orderkey = o_orderkey and o_orderdate >= date '1993-10-01' and o_orderdate < date '1994-01-01'
        and l_returnflag = 'R' and c_nationkey = n_nationkey
        group by c_custkey, c_name, c_acctbal, c_phone, n_name, c_address, c_comment order by revenue desc limit 20""",
    "q12": """select l_shipmode,
        sum(case when o_orderpriority = '1-URGENT' or o_orderpriority = '2-HIGH' then 1 else 0 end) as high_line_count,
        sum(case when o_orderpriority <> '1-URGENT' and o_orderpriority <> '2-HIGH' then 1 else 0 end) as low_line_count
        from orders, lineitem
        where o_orderkey = l_orderkey and l_shipmode in ('MAIL', 'SHIP') and l_commitdate < l_receiptdate and l_shipdate < l_commitdate
        and l_receiptdate >= date '1994-01-01' and l_receiptdate < date '1995-01-01' group by l_shipmode order by l_shipmode""",
    "q14": """select 100.00 * sum(case when p_type like 'PROMO%' then l_extendedprice * (1 - l_discount) else 0 end)
        + sum(l_extendedprice * (1 - l_discount)) as promo_revenue
        from lineitem, part where l_partkey = p_partkey and l_shipdate >= date '1995-09-01' and l_shipdate < date '1995-10-01'""",
    "q19": """select l_extendedprice* (1 - l_discount) from lineitem, part
        where p_partkey = l_partkey and l_shipinstruct = 'DELIVER IN PERSON' and l_shipmode in ('AIR', 'AIR REG') and
        ( ( p_brand = 'Brand#12' and p_container in ('SM CASE', 'SM BOX', 'SM PACK', 'SM PKG') and l_quantity >= 1 and l_quantity <= 1 + 10
            and p_size between 1 and 5 )
          or ( p_brand = 'Brand#23' and p_container in ('MED BAG', 'MED BOX', 'MED PKG', 'MED PACK') and l_quantity >= 10
            and l_quantity <= 10 + 10 and p_size between 1 and 10 )
          or ( p_brand = 'Brand#34' and p_container in ('LG CASE', 'LG BOX', 'LG PACK', 'LG PKG') and l_quantity >= 20
            and l_quantity <= 20 + 10 and p_size between 1 and 15 ) )""",
}
