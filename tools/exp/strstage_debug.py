import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
os.environ["RSQ_GENERIC"] = "0"
os.environ["RSQ_GENERIC2"] = "0"
from resql_amd import engine, tpch_full
from oracle import orc
ctx = engine.Context(device=0)
for sf in (0.001, 0.01, 0.05):
    db = tpch_full.database(sf)
    host = [db[k] for k in sorted(db)]
    tabs = [ctx.table(t) for t in host]
    for sql in ("select count(*) from customer where c_mktsegment = 'BUILDING'",
                "select c_custkey from customer where c_mktsegment = 'BUILDING'",
                "select count(*) from customer, orders where c_mktsegment = 'BUILDING' and c_custkey = o_custkey",
                "select count(*) from lineitem where l_shipmode = 'MAIL'",
                "select count(*) from lineitem where l_shipinstruct = 'DELIVER IN PERSON' and l_shipmode = 'AIR'"):
        try:
            want = orc.execute(ctx.sql_plan(sql, tabs, host))
            cq = ctx.sql_compile(sql, tabs)
            cq.execute()
            got = cq.result()
            same = sorted(got.text.splitlines()) == sorted(want.text.splitlines())
            print(sf, sql[:70], "OK" if same else "DIFF", got.n_rows, want.n_rows, got.text[:80].replace("\n", " ") if not same else "", want.text[:80].replace("\n", " ") if not same else "", flush=True)
            cq.close()
        except Exception as e:
            print(sf, sql[:70], "EXC", e, flush=True)
    for t in tabs: t.close()
