import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
os.environ["RSQ_GENERIC"] = "0"
os.environ["RSQ_GENERIC2"] = "0"
from resql_amd import engine, tpch_full
from oracle import orc
ctx = engine.Context(device=0)
db = tpch_full.database(0.01)
host = [db[k] for k in sorted(db)]
tabs = [ctx.table(t) for t in host]
sql = "select c_custkey from customer where c_mktsegment = 'BUILDING'"
want = orc.execute(ctx.sql_plan(sql, tabs, host))
w = set(int(l.strip("|")) for l in want.text.splitlines()[1:])
for rep in range(3):
    cq = ctx.sql_compile(sql, tabs)
    cq.execute()
    got = cq.result()
    g = set(int(l.strip("|")) for l in got.text.splitlines()[1:])
    miss = sorted(w - g)
    print("missing", len(miss), "extra", len(g - w), [(m - 1, (m - 1) >> 7, (m - 1) & 127) for m in miss][:60], flush=True)
    cq.close()
