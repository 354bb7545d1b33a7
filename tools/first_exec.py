"""What a ReSQL host pays per SELECT: compile -> ONE execution -> delete (reference src/execute.h:213-247), against the steady state of a
re-executed query.  For every TPC-H statement (SQL text, resident database, warm code-object cache):
  first_ever  - the very first query of that shape on the context (nothing remembered);
  first[k]    - k-th fresh query of the same statement: compile -> execute once -> destroy (what the plan memo and the arenas are for);
  steady      - executions 2..N of one query.
usage: python tools/first_exec.py [SF] [--only q3,q10] [--fresh 5] [--steady 10] [--driver-alloc] [--no-memo] [--reserve-gb G]
One JSON line per statement (+ one with the context's memory statistics); every answer is compared with the reference's golden where one exists
and with the first answer otherwise."""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resql_amd import engine, tpch_full  # noqa: E402


def arg(name, default, conv=str):
    return conv(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


sf = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 1.0
only = arg("--only", None, lambda s: s.split(","))
n_fresh = arg("--fresh", 5, int)
n_steady = arg("--steady", 10, int)
flags = (engine.ENGINE_DRIVER_ALLOC if "--driver-alloc" in sys.argv else 0) | (engine.ENGINE_NO_PLAN_MEMO if "--no-memo" in sys.argv else 0)
reserve = arg("--reserve-gb", 0.0, float)

t0 = time.time()
db = tpch_full.database(sf, fill_unused=sf < 1.0)
names = sorted(db)
host = [db[k] for k in names]
print(f"# generated SF{sf:g} in {time.time() - t0:.1f} s", flush=True)
ctx = engine.Context(device=0, engine_flags=flags, arena_reserve_bytes=int(reserve * (1 << 30)))
tabs = [ctx.table(t) for t in host]
print("# " + json.dumps({"engine_flags": flags, "memory_after_tables": ctx.memory_stats()}), flush=True)


def once(sql):
    """compile -> execute once -> destroy; returns (compile_ms, exec_ms, launches, text)"""
    t = time.perf_counter()
    q = ctx.sql_compile(sql, tabs)
    q.await_kernels()
    c_ms = (time.perf_counter() - t) * 1e3
    t = time.perf_counter()
    q.execute()
    e_ms = (time.perf_counter() - t) * 1e3
    r = q.report()
    text = q.result().text
    q.close()
    return c_ms, e_ms, int(r.num_kernels), text


for name, sql in tpch_full.QUERIES.items():
    if only and name not in only:
        continue
    gold = os.path.join(ROOT, "tests", "golden", f"ref_full_{name}_sf{sf:g}.tbl")
    want = open(gold, encoding="latin1").read() if os.path.exists(gold) else None
    m0 = ctx.memory_stats()
    c0, e0, k0, text0 = once(sql)
    if want is None:
        want = text0
    ok = text0 == want
    fresh = []
    for _ in range(n_fresh):
        c, e, k, text = once(sql)
        ok = ok and text == want
        fresh.append({"compile_ms": round(c, 3), "exec_ms": round(e, 3), "launches": k})
    m1 = ctx.memory_stats()
    q = ctx.sql_compile(sql, tabs)
    q.await_kernels()
    q.execute()
    steady = []
    for _ in range(n_steady):
        t = time.perf_counter()
        q.execute()
        steady.append((time.perf_counter() - t) * 1e3)
    ok = ok and q.result().text == want
    launches = int(q.report().num_kernels)
    q.close()
    print(json.dumps({"query": name, "sf": sf, "first_ever": {"compile_ms": round(c0, 3), "exec_ms": round(e0, 3), "launches": k0},
                      "first_exec_ms": [f["exec_ms"] for f in fresh], "first_compile_ms": [f["compile_ms"] for f in fresh],
                      "first_launches": [f["launches"] for f in fresh],
                      "steady_exec_ms": round(statistics.median(steady), 3), "steady_min_ms": round(min(steady), 3), "steady_launches": launches,
                      "answers_equal": bool(ok), "checked_against": os.path.relpath(gold, ROOT) if os.path.exists(gold) else "the first answer",
                      "driver_calls_during": {k: m1[k] - m0[k] for k in ("device_slab_allocs", "pinned_slab_allocs", "raw_driver_calls")},
                      "driver_ms_during": round(m1["driver_ms"] - m0["driver_ms"], 3)}), flush=True)
print("# " + json.dumps({"memory_at_end": ctx.memory_stats()}), flush=True)
for t in tabs:
    t.close()
ctx.close()
