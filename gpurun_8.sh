cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "
from resql_amd import engine
c=engine.Context(0); print('read bw nt GB/s', c.read_bandwidth(8<<30, 10))"
python bench.py --steps 200 --warmup 20 2>&1 | tail -1 > gpurun_out/bench_n1_d.json; cat gpurun_out/bench_n1_d.json
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_q1_d -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_q1_d.log 2>&1
cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch_d -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch_d.log 2>&1
cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq_d -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq_d.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import time
from resql_amd import tpch, engine, datagen
ctx = engine.Context(device=0)
sf = 10
nL, nO, nC = datagen.n_lineitem(sf), datagen.n_orders(sf), datagen.n_customer(sf)
li = ctx.generate(engine.GEN_LINEITEM, nL, sf, param=1)
od = ctx.generate(engine.GEN_ORDERS, nO, sf)
cu = ctx.generate(engine.GEN_CUSTOMER, nC, sf)
s_li = tpch.lineitem_table(0.001, tpch.Q3_LINEITEM_COLUMNS, n_rows=0)
s_od = tpch.orders_table(0.001); s_cu = tpch.customer_table(0.001)
plan = tpch.q3_plan(s_cu, s_od, s_li)
q = ctx.compile(plan, [cu, od, li])
print(q.explain)
for i in range(5):
    q.execute(); r = q.report()
    print("q3 sf10 kernel_ms", round(r.kernel_time_ms,3), "exec_ms", round(r.execution_time_ms,3), "fin_ms", round(r.finalize_time_ms,3), "GB/s", round(r.hbm_gbps,1), "kernels", r.num_kernels)
print(q.result().text)
q6 = ctx.compile(tpch.q6_plan(s_li if False else tpch.lineitem_table(0.001, tpch.Q6_COLUMNS, n_rows=0)), [li])
for i in range(3):
    q6.execute(); r = q6.report(); print("q6 sf10 kernel_ms", round(r.kernel_time_ms,4), "GB/s", round(r.hbm_gbps,1), "exec_ms", round(r.execution_time_ms,3))
for groups in (8, 1024, 1<<20):
    n = 200_000_000
    t = ctx.generate(engine.GEN_SYNTHETIC, n, 1.0, param=groups)
    st = tpch.synthetic_table(16, groups)
    for sel in (0.01, 0.1, 0.5):
        qs = ctx.compile(tpch.synthetic_plan(st, int(sel*(1<<31))), [t])
        for i in range(3): qs.execute()
        r = qs.report(); print("synthetic groups", groups, "sel", sel, "rows", n, "kernel_ms", round(r.kernel_time_ms,3), "GB/s", round(r.hbm_gbps,1), "exec_ms", round(r.execution_time_ms,3), "result rows", qs.result().n_rows)
        qs.close()
    t.close()
PY
