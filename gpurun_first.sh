set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python - <<'PY' 2>&1 | tee gpurun_out/first_q1.log
import time, os
from resql_amd import tpch, engine, datagen
ctx = engine.Context(device=0)
print("read bw GB/s", ctx.read_bandwidth(4<<30, 5))
for sf in [1, 10]:
    n = datagen.n_lineitem(sf)
    t = ctx.generate(engine.GEN_LINEITEM, n, sf)
    li = tpch.lineitem_table(0.01, tpch.Q1_COLUMNS)   # schema only
    for name, plan, bpr in [("q1", tpch.q1_plan(li), 38), ("q6", tpch.q6_plan(li), 28)]:
        q = ctx.compile(plan, [t])
        for i in range(5):
            q.execute()
            r = q.report()
            print(name, "sf", sf, "rows", n, "kernel_ms", round(r.kernel_time_ms,4), "exec_ms", round(r.execution_time_ms,3), "GB/s", round(r.hbm_gbps,1), "Mrows/s", round(n/r.kernel_time_ms/1e3,1), "fin_ms", round(r.finalize_time_ms,3), "compile_ms", round(r.compilation_time_ms,1), r.jit_compiles, r.jit_cache_hits)
        print(q.result().text)
        q.close()
    t.close()
PY
