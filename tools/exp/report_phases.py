import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from resql_amd import engine, tpch_full
ctx = engine.Context(device=0)
db = tpch_full.database(10, fill_unused=False)
host=[db[k] for k in sorted(db)]
tabs=[ctx.table(t) for t in host]
for name in ("q10","q3","q5"):
    cq = ctx.sql_compile(tpch_full.QUERIES[name], tabs)
    cq.await_kernels()
    for i in range(10):
        cq.execute()
    best=None
    for i in range(10):
        cq.execute(); r=cq.report()
        cur=(r.execution_time_ms, r.kernel_time_ms, r.finalize_time_ms)
        best = cur if best is None or cur[0]<best[0] else best
    print(name, "exec %.3f kernel %.3f finalize %.3f" % best, flush=True)
    cq.close()
