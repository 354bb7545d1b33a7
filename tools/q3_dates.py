import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from resql_amd import datagen, engine, tpch
sf = 10.0
ctx = engine.Context(device=0)
li = ctx.generate(engine.GEN_LINEITEM, datagen.n_lineitem(sf), sf, param=1)
od = ctx.generate(engine.GEN_ORDERS, datagen.n_orders(sf), sf)
cu = ctx.generate(engine.GEN_CUSTOMER, datagen.n_customer(sf), sf)
for date in ("1995-03-15", "1998-12-01", "1992-01-01"):
    plan = tpch.q3_plan(tpch.customer_table(0.001), tpch.orders_table(0.001), tpch.lineitem_table(0.001, tpch.Q3_LINEITEM_COLUMNS, n_rows=0), date=date)
    q = ctx.compile(plan, [cu, od, li])
    for _ in range(3):
        q.execute()
    os.environ["RSQ_TRACE"] = "1"
    print("== date", date, flush=True)
    q.execute()
    del os.environ["RSQ_TRACE"]
    q.close()
