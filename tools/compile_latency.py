"""Compile latency, cold and warm, per plan (the reference: 0.6-3 ms per query, src/JitContextFlounder.h:410-456).
Cold = an empty code-object cache directory; warm = the same plan compiled again (code objects on disk / loaded).
Prints one JSON line per plan: compile ms, first-execution ms, whether the pre-compiled generic pipeline served it, and the
time until the specialised kernel had been built in the background.
usage: python tools/compile_latency.py [sf]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resql_amd import datagen, engine, tpch  # noqa: E402

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
cache = tempfile.mkdtemp(prefix="rsq_cold_")
ctx = engine.Context(device=0, cache_dir=cache)
li = ctx.generate(engine.GEN_LINEITEM, datagen.n_lineitem(sf), sf, param=1)
od = ctx.generate(engine.GEN_ORDERS, datagen.n_orders(sf), sf)
cu = ctx.generate(engine.GEN_CUSTOMER, datagen.n_customer(sf), sf)
plans = {
    "q6": (tpch.q6_plan(tpch.lineitem_table(0.001, tpch.Q6_COLUMNS, n_rows=0)), [li]),
    "q1": (tpch.q1_plan(tpch.lineitem_table(0.001, tpch.Q1_COLUMNS, n_rows=0)), [li]),
    "q3": (tpch.q3_plan(tpch.customer_table(0.001), tpch.orders_table(0.001), tpch.lineitem_table(0.001, tpch.Q3_LINEITEM_COLUMNS, n_rows=0)), [cu, od, li]),
}
for name, (plan, tabs) in plans.items():
    rec = {"plan": name, "sf": sf}
    for phase in ("cold", "warm"):
        t0 = time.perf_counter()
        q = ctx.compile(plan, tabs)
        t1 = time.perf_counter()
        q.execute()
        t2 = time.perf_counter()
        generic = "generic pre-compiled pipeline" in q.explain or "generic pre-compiled interpreter" in q.explain
        rec[phase] = {"compile_ms": round((t1 - t0) * 1e3, 3), "first_execution_ms": round((t2 - t1) * 1e3, 3), "generic_pipeline": generic,
                      "reported_compilation_time_ms": round(q.report().compilation_time_ms, 3)}
        if generic:
            while q.report().jit_compiles == 0 and time.perf_counter() - t0 < 120:
                q.execute()
                time.sleep(0.02)
            rec[phase]["specialised_kernel_ready_after_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
            q.execute()
            rec[phase]["specialised_execution_ms"] = round(q.report().execution_time_ms, 3)
        q.close()
    print(json.dumps(rec), flush=True)
# the reference's statements with joins from SQL text (eight-table database; joins into hash aggregations, CASE / LIKE on strings,
# materialisation): cold through the interpreter for whole pipelines
from resql_amd import tpch_full  # noqa: E402
db = tpch_full.database(sf, fill_unused=False)
tabs8 = [ctx.table(db[k]) for k in sorted(db)]
for name in ("q3", "q5", "q10", "q12", "q14", "q19"):
    sql = tpch_full.QUERIES[name]
    rec = {"plan": name + " (SQL text)", "sf": sf, "compile_helpers": os.environ.get("RSQ_COMPILE_HELPERS", "1") != "0"}
    for phase in ("cold", "warm"):
        t0 = time.perf_counter()
        q = ctx.sql_compile(sql, tabs8)
        t1 = time.perf_counter()
        q.execute()
        t2 = time.perf_counter()
        generic = "generic pre-compiled" in q.explain
        rec[phase] = {"compile_ms": round((t1 - t0) * 1e3, 3), "first_execution_ms": round((t2 - t1) * 1e3, 3), "generic_pipeline": generic,
                      "rows": q.result().n_rows}
        if generic:
            interp = []
            while q.report().jit_compiles == 0 and time.perf_counter() - t0 < 180:
                q.execute()
                r = q.report()
                if r.jit_compiles == 0:
                    interp.append(r.execution_time_ms)
                time.sleep(0.02)
            rec[phase]["specialised_kernel_ready_after_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
            if interp:
                rec[phase]["interpreter_execution_ms"] = round(min(interp), 3)      # steady state on the interpreter, while the kernels were being built
            q.execute()
            best = q.report().execution_time_ms
            for _ in range(3):
                q.execute()
                best = min(best, q.report().execution_time_ms)
            rec[phase]["specialised_execution_ms"] = round(best, 3)
        q.close()
    print(json.dumps(rec), flush=True)
ctx.close()
