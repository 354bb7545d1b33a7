"""A / B of one environment switch the code generator reads (the kernels differ, the process and the resident database are the same): every
statement is compiled once per value, the executions of the variants alternate, and each answer is compared with the first variant's.
usage: python tools/exp/ab_env.py SF NAME=v0,v1[,v2] [--only q3,q10] [--rounds 6] [--per-kernel]"""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from resql_amd import engine, tpch_full  # noqa: E402

sf = float(sys.argv[1])
name, values = sys.argv[2].split("=")
values = values.split(",")
only = sys.argv[sys.argv.index("--only") + 1].split(",") if "--only" in sys.argv else None
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 6
t0 = time.time()
db = tpch_full.database(sf, fill_unused=sf < 1.0)
host = [db[k] for k in sorted(db)]
print(f"# generated SF{sf:g} in {time.time() - t0:.1f} s", flush=True)
ctx = engine.Context(device=0, engine_flags=engine.ENGINE_NO_PLAN_MEMO)
tabs = [ctx.table(t) for t in host]
for qn, sql in tpch_full.QUERIES.items():
    if only and qn not in only:
        continue
    qs = []
    for v in values:
        os.environ[name] = v
        q = ctx.sql_compile(sql, tabs)
        q.await_kernels()
        for _ in range(3):
            q.execute()
        qs.append(q)
    os.environ.pop(name, None)
    want = qs[0].result().text
    dev = [[] for _ in values]
    wall = [[] for _ in values]
    for _ in range(rounds):
        for i, q in enumerate(qs):
            for _ in range(3):
                t = time.perf_counter()
                q.execute()
                wall[i].append((time.perf_counter() - t) * 1e3)
                dev[i].append(q.report().kernel_time_ms)
    out = {"query": qn, "sf": sf, "switch": name}
    for i, v in enumerate(values):
        out[f"{v}"] = {"kernels_ms_median": round(statistics.median(dev[i]), 4), "kernels_ms_min": round(min(dev[i]), 4),
                       "exec_ms_median": round(statistics.median(wall[i]), 4), "exec_ms_min": round(min(wall[i]), 4),
                       "launches": int(qs[i].report().num_kernels), "answer_equal": qs[i].result().text == want}
        if "--per-kernel" in sys.argv:
            out[f"{v}"]["explain"] = [l for l in qs[i].explain.splitlines() if " us" in l or " ms" in l][:24]
    print(json.dumps(out), flush=True)
    for q in qs:
        q.close()
