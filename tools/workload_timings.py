"""Q3 / Q6 / synthetic timings on one GPU (development aid; bench.py is the contract benchmark)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resql_amd import datagen, engine, tpch

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = engine.Context(device=0)
nL, nO, nC = datagen.n_lineitem(sf), datagen.n_orders(sf), datagen.n_customer(sf)
li = ctx.generate(engine.GEN_LINEITEM, nL, sf, param=1)
od = ctx.generate(engine.GEN_ORDERS, nO, sf)
cu = ctx.generate(engine.GEN_CUSTOMER, nC, sf)
plan = tpch.q3_plan(tpch.customer_table(0.001), tpch.orders_table(0.001), tpch.lineitem_table(0.001, tpch.Q3_LINEITEM_COLUMNS, n_rows=0))
q = ctx.compile(plan, [cu, od, li])
for i in range(4):
    q.execute(); r = q.report()
    print("q3 sf", sf, "kernel_ms", round(r.kernel_time_ms, 3), "exec_ms", round(r.execution_time_ms, 3), "fin_ms", round(r.finalize_time_ms, 3),
          "GB/s(streaming bytes)", round(r.hbm_gbps, 1), "kernels", r.num_kernels, flush=True)
print(q.result().text)
if len(sys.argv) > 2:
    for groups in (8, 1024, 1 << 20):
        n = 200_000_000
        t = ctx.generate(engine.GEN_SYNTHETIC, n, 1.0, param=groups)
        st = tpch.synthetic_table(16, groups)
        for sel in (0.01, 0.1, 0.5):
            qs = ctx.compile(tpch.synthetic_plan(st, int(sel * (1 << 31))), [t])
            for i in range(3):
                qs.execute()
            r = qs.report()
            print("synthetic groups", groups, "sel", sel, "rows", n, "kernel_ms", round(r.kernel_time_ms, 3), "GB/s", round(r.hbm_gbps, 1),
                  "exec_ms", round(r.execution_time_ms, 3), "fin_ms", round(r.finalize_time_ms, 3), "result rows", qs.result().n_rows, flush=True)
            qs.close()
        t.close()
