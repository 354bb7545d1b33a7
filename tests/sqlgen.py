"""Random VALID select statements over the eight-table TPC-H database (resql_amd/tpch_full.py): foreign-key join paths,
predicates with literals drawn from the value domains, group-bys on low-cardinality columns with every aggregate kind, order by /
limit.  Used for differential testing of the whole SQL path: reference (its grammar + planner + JIT, fed the token stream) ==
oracle (plan of THIS front end) == engine.  Deterministic per seed."""
import random

# (tables, join conditions)
PATHS = [
    (["lineitem"], []),
    (["orders"], []),
    (["customer"], []),
    (["part"], []),
    (["orders", "lineitem"], ["o_orderkey = l_orderkey"]),
    (["customer", "orders"], ["c_custkey = o_custkey"]),
    (["customer", "orders", "lineitem"], ["c_custkey = o_custkey", "l_orderkey = o_orderkey"]),
    (["lineitem", "part"], ["l_partkey = p_partkey"]),
    (["supplier", "nation"], ["s_nationkey = n_nationkey"]),
    (["nation", "region"], ["n_regionkey = r_regionkey"]),
    (["supplier", "nation", "region"], ["s_nationkey = n_nationkey", "n_regionkey = r_regionkey"]),
    (["customer", "nation"], ["c_nationkey = n_nationkey"]),
    (["lineitem", "supplier"], ["l_suppkey = s_suppkey"]),
]

PREDICATES = {
    "lineitem": [lambda r: f"l_quantity < {r.randrange(2, 50)}", lambda r: f"l_discount between 0.0{r.randrange(0, 5)} and 0.0{r.randrange(5, 10)}",
                 lambda r: f"l_shipdate >= date '199{r.randrange(2, 8)}-0{r.randrange(1, 10)}-01'", lambda r: f"l_returnflag = '{r.choice('RAN')}'",
                 lambda r: f"l_shipmode in ('{r.choice(['MAIL', 'SHIP', 'AIR'])}', '{r.choice(['RAIL', 'FOB', 'TRUCK'])}')",
                 lambda r: "l_commitdate < l_receiptdate", lambda r: f"l_shipinstruct <> '{r.choice(['NONE', 'COLLECT COD'])}'",
                 lambda r: f"l_linenumber <= {r.randrange(1, 7)}", lambda r: f"l_tax > 0.0{r.randrange(0, 8)}"],
    "orders": [lambda r: f"o_orderdate < date '199{r.randrange(3, 9)}-06-15'", lambda r: f"o_orderpriority = '{r.choice(['1-URGENT', '2-HIGH', '5-LOW'])}'",
               lambda r: f"o_orderdate >= date '199{r.randrange(2, 6)}-01-01'", lambda r: f"o_orderpriority like '{r.choice(['%HIGH', '1%', '%-%'])}'"],
    "customer": [lambda r: f"c_mktsegment = '{r.choice(['BUILDING', 'AUTOMOBILE', 'MACHINERY'])}'", lambda r: f"c_acctbal > {r.randrange(-500, 5000)}.00",
                 lambda r: f"c_nationkey < {r.randrange(3, 25)}", lambda r: f"c_acctbal < -{r.randrange(100, 900)}.50"],
    "part": [lambda r: f"p_size between {r.randrange(1, 20)} and {r.randrange(20, 50)}", lambda r: f"p_brand = 'Brand#{r.randrange(1, 6)}{r.randrange(1, 6)}'",
             lambda r: f"p_type like '{r.choice(['PROMO%', '%BRASS', '%ANODIZED%'])}'", lambda r: f"p_container in ('SM CASE', 'LG BOX', 'MED BAG')"],
    "supplier": [lambda r: f"s_acctbal > {r.randrange(0, 8000)}.00", lambda r: f"s_nationkey <> {r.randrange(0, 25)}"],
    "nation": [lambda r: f"n_regionkey = {r.randrange(0, 5)}", lambda r: f"n_name <> '{r.choice(['PERU', 'CHINA', 'FRANCE'])}'"],
    "region": [lambda r: f"r_name = '{r.choice(['ASIA', 'EUROPE', 'AMERICA'])}'", lambda r: "r_regionkey < 4"],
}
GROUPABLE = {
    "lineitem": ["l_returnflag", "l_linestatus", "l_shipmode", "l_shipinstruct", "l_linenumber", "l_quantity", "l_tax"],
    "orders": ["o_orderpriority", "o_shippriority"],
    "customer": ["c_mktsegment", "c_nationkey"],
    "part": ["p_brand", "p_size", "p_container"],
    "supplier": ["s_nationkey"],
    "nation": ["n_name", "n_regionkey"],
    "region": ["r_name"],
}
MEASURES = {
    "lineitem": ["l_quantity", "l_extendedprice", "l_discount", "l_extendedprice * (1 - l_discount)", "l_extendedprice * l_tax"],
    "orders": ["o_shippriority + 1"],
    "customer": ["c_acctbal"],
    "part": ["p_retailprice", "p_size * 2"],
    "supplier": ["s_acctbal"],
    "nation": ["n_nationkey * 1"],
    "region": ["r_regionkey + 10"],
}
PLAIN = {
    "lineitem": ["l_orderkey", "l_linenumber", "l_quantity", "l_shipdate", "l_shipmode"],
    "orders": ["o_orderkey", "o_orderdate", "o_orderpriority"],
    "customer": ["c_custkey", "c_name", "c_acctbal", "c_phone"],
    "part": ["p_partkey", "p_brand", "p_type", "p_size"],
    "supplier": ["s_suppkey", "s_name", "s_acctbal"],
    "nation": ["n_nationkey", "n_name"],
    "region": ["r_regionkey", "r_name"],
}


def statement(seed: int) -> str:
    r = random.Random(seed)
    tables, joins = r.choice(PATHS)
    conds = list(joins)
    for t in tables:
        for _ in range(r.randrange(0, 3)):
            conds.append(r.choice(PREDICATES[t])(r))
    r.shuffle(conds)
    where = (" where " + " and ".join(conds)) if conds else ""
    from_ = ", ".join(r.sample(tables, len(tables)))
    if r.random() < 0.7:
        groups = []
        for t in r.sample(tables, len(tables)):
            if r.random() < 0.6 and len(groups) < 2:
                groups.append(r.choice(GROUPABLE[t]))
        groups = list(dict.fromkeys(groups))
        aggs = []
        for i in range(r.randrange(1, 4)):
            m = r.choice(MEASURES[r.choice(tables)])
            fn = r.choice(["sum", "avg", "min", "max", "count"])
            aggs.append(f"{fn}({m}) as a{i}" if fn != "count" or r.random() < 0.5 else f"count(*) as a{i}")
        sel = ", ".join(groups + aggs)
        s = f"select {sel} from {from_}{where}"
        if groups:
            s += " group by " + ", ".join(groups)
            order = [g + r.choice(["", " desc"]) for g in groups]      # the groups, all of them: a total order
            if r.random() < 0.4:
                order = [f"a0{r.choice(['', ' desc'])}"] + order
            s += " order by " + ", ".join(order)
            if r.random() < 0.4:
                s += f" limit {r.randrange(1, 12)}"
        return s
    cols = []
    for t in tables:
        cols += r.sample(PLAIN[t], r.randrange(1, 3))
    s = f"select {', '.join(cols)} from {from_}{where}"
    return s
