import sys, time
sys.path.insert(0, '/root/repo')
from resql_amd import engine, tpch, datagen
ctx = engine.Context(device=0)
n = datagen.n_lineitem(1.0)
t = ctx.generate(engine.GEN_LINEITEM, n, 1.0)
q = ctx.compile(tpch.q1_plan(tpch.lineitem_table(0.001, tpch.Q1_COLUMNS, n_rows=0)), [t])
q.await_kernels()
for i in range(4):
    t0 = time.perf_counter(); q.execute(); dt = time.perf_counter() - t0
    r = q.report()
    print(i, "exec ms %.3f kernel ms %.4f" % (dt * 1e3, r.kernel_time_ms), flush=True)
