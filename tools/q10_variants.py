import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resql_amd import engine, tpch_full
sf = 10.0
db = tpch_full.database(sf, fill_unused=False)
names = sorted(db); host = [db[k] for k in names]
ctx = engine.Context(device=0)
tabs = [ctx.table(t) for t in host]
base = """from customer, orders, lineitem, nation
        where c_custkey = o_custkey and l_orderkey = o_orderkey and o_orderdate >= date '1993-10-01' and o_orderdate < date '1994-01-01'
        and l_returnflag = 'R' and c_nationkey = n_nationkey"""
variants = {
 "full q10": tpch_full.QUERIES["q10"],
 "group by c_custkey only": f"select c_custkey, sum(l_extendedprice * (1 - l_discount)) as revenue {base} group by c_custkey order by revenue desc limit 20",
 "c_custkey, c_acctbal": f"select c_custkey, sum(l_extendedprice * (1 - l_discount)) as revenue, c_acctbal {base} group by c_custkey, c_acctbal order by revenue desc limit 20",
 "c_custkey, c_name": f"select c_custkey, c_name, sum(l_extendedprice * (1 - l_discount)) as revenue {base} group by c_custkey, c_name order by revenue desc limit 20",
 "c_custkey, c_comment": f"select c_custkey, c_comment, sum(l_extendedprice * (1 - l_discount)) as revenue {base} group by c_custkey, c_comment order by revenue desc limit 20",
 "ungrouped": f"select sum(l_extendedprice * (1 - l_discount)) as revenue {base}",
}
only = os.environ.get("Q10_ONLY")
for name, sql in variants.items():
    if only and only not in name:
        continue
    q = ctx.sql_compile(sql, tabs)
    best = 1e9
    for _ in range(4):
        q.execute(); best = min(best, q.report().kernel_time_ms)
    print(f"{name}: kernels {best:.3f} ms, {q.report().num_kernels} launches", flush=True)
    q.close()
