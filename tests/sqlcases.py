"""Shared inputs of the SQL front-end tests: hand-written statements that pin particular rules of the reference's
lexer / grammar / planner, next to the reference's eight TPC-H queries (resql_amd/tpch_full.py QUERIES) and the random
statements of tests/sqlfuzz.py."""

# tokenizer rules of lexer.y that a hand-written tokenizer gets wrong first
TOKEN_CASES = [
    "select a from t where b >= 1 and c <> 2 and d <= 3",
    "select sum(x), summ, counter, int1, integer, dates from t",          # keywords are not prefixes of identifiers
    "select a from t group by a order by a",                              # one space: GROUPBY / ORDERBY
    "select a from t group  by a",                                        # two spaces: identifiers `group`, `by`
    "select 1, 1.5, .5, 5., 1e5, 1.5e-3, 12abc from t",
    "select 'x', 'two words', \"dq\", 'it\\'s', '' from t",
    "select a -- trailing comment\n from t",
    "select a - -1, a--1 from t",                                         # `--1 from t` without a newline is not a comment
    "select a::int, b :: decimal(12,2) from t",
    "select A from t",                                                    # upper case is not in any rule
    "select a from t;",                                                   # neither is ';'
    "select 'unterminated from t",
    "create table t ( a int, b char(3) )",
    "bulk insert t from \"x.tbl\" with ( fieldterminator=\"|\", firstrow=2 )",
    "select a\tfrom\tt\r\n",                                              # tab is white space, carriage return is not
]

# grammar corners: precedence ladder, BETWEEN / IN / CASE rewriting, negated literals
PARSE_CASES = [
    "select a = b < c, a < b = c, a + b * c - d / e from t",
    "select a or b and c or d from t",
    "select a between 1 and 5 and b between c + 1 and d * 2 or e from t",
    "select a between 1 and 2 + 3 between 4 and 5 from t",
    "select a in (1), a in (1, 2, 3), a in ('x', 'yz') from t",
    "select a + b in (1, 2) from t",
    "select case when a then b end, case when a then b else c end, case when a then b when c then d else e end from t",
    "select -1, -2.50, - 3, 4 - -5, a * -1 from t",
    "select a like 'x%' and b like '%', a like 'p' || 'q' from t",
    "select a asc, b desc from t order by a asc, b desc, c",
    "select count(*), count(a), sum(a * b), avg(a) as m, min(a), max(b) from t",
    "select * from t, u where a = b",
    "select *, a from t",
    "select a as b as c from t",
    "select a from t where",
    "select from t",
    "select a from t limit 5 order by a",
    "select a from t limit -1",
    "select date '1994-01-01', date \"1995/02/03\", a :: date from t",
    "select (a), ((a + b)) * (c) from t",
    "select a b from t",
    "select sum(a) over from t",
    "bulk insert t from 'f.tbl'",
    "bulk insert t from 'f.tbl' with ( firstrow=3 )",
    "bulk insert t from f",
    "create table t ( )",
    "create table t ( a int b int )",
    "create table t ( a varchar(10), b decimal(15,2), c date, d bigint, e char(1) )",
]

# planner rules over the eight-table database of tpch_full.database(): push-down, join sides and order, join conditions
# between already joined tables, conditions without attributes, group-by unification, star, limit placement
PLAN_CASES = [
    "select c_name from customer where c_custkey < 10",
    "select * from nation",
    "select * from nation, region where n_regionkey = r_regionkey",
    "select n_name, r_name from region, nation where r_regionkey = n_regionkey and r_name = 'ASIA' limit 3",
    "select o_orderkey from lineitem, orders where o_orderkey = l_orderkey and o_orderdate < date '1992-02-01' and l_quantity < 3",
    "select o_orderkey from orders, lineitem where l_orderkey = o_orderkey and l_quantity < 3",
    "select c_custkey from customer, orders, lineitem where o_custkey = c_custkey and l_orderkey = o_orderkey and l_quantity > 49 and c_acctbal < 0",
    "select s_name from supplier, nation, region where s_nationkey = n_nationkey and n_regionkey = r_regionkey and r_name = 'EUROPE'",
    "select c_name from customer, supplier, nation where c_nationkey = s_nationkey and s_nationkey = n_nationkey and c_nationkey = n_nationkey and n_name = 'PERU'",
    "select l_orderkey from lineitem where l_commitdate < l_receiptdate and l_shipdate < l_commitdate and l_quantity < 2",
    "select l_orderkey from lineitem where l_quantity < 2 and 1 = 1 and l_discount > 0.05",
    "select l_orderkey from lineitem where 1 = 1 and l_quantity < 2",
    "select l_quantity + 1, count(*) from lineitem where l_quantity < 5 group by l_quantity + 1",
    "select l_returnflag, sum(l_quantity) * 2 + count(*) as x from lineitem group by l_returnflag order by l_returnflag desc",
    "select count(*) from lineitem, part where l_partkey = p_partkey and p_size < l_quantity",
    "select n_name from nation where n_nationkey = n_regionkey",
    "select n_name from nation order by n_name limit 4",
    "select n_name from nation limit 0",
    "select p_brand, count(*) as n from part where p_size between 10 and 20 group by p_brand order by n desc, p_brand limit 5",
    "select o_orderpriority, count(*) from orders where o_orderdate >= date '1993-07-01' and o_orderdate < date '1993-10-01' group by o_orderpriority order by o_orderpriority",
    "select c_name, c_acctbal from customer where c_acctbal < -900.00 order by c_acctbal limit 5",      # negated literal
    "select l_quantity :: bigint, (l_extendedprice * 2) :: decimal(19,4) from lineitem where l_linenumber > 6",   # explicit casts
    "select n_nationkey :: bigint + -3 from nation where n_nationkey :: bigint > 20",
    "select l_orderkey from lineitem where l_orderkey < 3",              # INT -> BIGINT cast of keys beyond int16: see oracle g_narrowCasts
    "select n_name, n_name like '%IA', count(*) as n from nation group by n_name order by n_name limit 9",      # LIKE above the aggregation (host tail)
    "select n_name :: int from nation",                                   # a cast the reference has no code for
    "select c_name from customer, nation",                                 # no equality: nested-loops join
    "select x from nosuchtable",
    "select nosuchcolumn from nation",
]
