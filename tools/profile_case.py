"""One workload, a few executions: the thing to put behind `rocprofv3 --kernel-trace --stats -- python3 tools/profile_case.py ...`.
usage: profile_case.py synthetic ROWS GROUPS SELECTIVITY [REPEAT]   |   profile_case.py q1|q6|q3 SF [REPEAT]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resql_amd import datagen, engine, tpch  # noqa: E402

kind = sys.argv[1]
ctx = engine.Context(device=0)
if kind == "synthetic":
    n, groups, sel = int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
    rep = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    t = ctx.generate(engine.GEN_SYNTHETIC, n, 1.0, param=groups)
    q = ctx.compile(tpch.synthetic_plan(tpch.synthetic_table(16, groups), int(sel * (1 << 31))), [t])
else:
    sf = float(sys.argv[2])
    rep = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    li = ctx.generate(engine.GEN_LINEITEM, datagen.n_lineitem(sf), sf, param=1)
    if kind == "q3":
        od = ctx.generate(engine.GEN_ORDERS, datagen.n_orders(sf), sf)
        cu = ctx.generate(engine.GEN_CUSTOMER, datagen.n_customer(sf), sf)
        plan = tpch.q3_plan(tpch.customer_table(0.001), tpch.orders_table(0.001), tpch.lineitem_table(0.001, tpch.Q3_LINEITEM_COLUMNS, n_rows=0))
        q = ctx.compile(plan, [cu, od, li])
    else:
        cols = tpch.Q1_COLUMNS if kind == "q1" else tpch.Q6_COLUMNS
        plan = (tpch.q1_plan if kind == "q1" else tpch.q6_plan)(tpch.lineitem_table(0.001, cols, n_rows=0))
        q = ctx.compile(plan, [li])
q.await_kernels()
for _ in range(rep):
    q.execute()
    r = q.report()
    print(kind, "kernel_ms", round(r.kernel_time_ms, 3), "exec_ms", round(r.execution_time_ms, 3), "kernels", r.num_kernels, flush=True)
q.close()
ctx.close()
