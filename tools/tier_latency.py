"""A statement nobody has compiled before (EMPTY code-object cache): when does it answer from the interpreter, when do the quick tier's
kernels (stage 2 called) take over, when the full (inlined) ones - and what does an execution cost on each?  (The reference compiles a
query in 0.6-3 ms, src/JitContextFlounder.h:410-456.)
usage: RSQ_GENERIC=1 python tools/tier_latency.py [SF] [--only q5,q10]      one JSON line per statement"""
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RSQ_GENERIC", "1")
from resql_amd import engine, tpch_full  # noqa: E402

sf = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 1.0
only = sys.argv[sys.argv.index("--only") + 1].split(",") if "--only" in sys.argv else None
db = tpch_full.database(sf, fill_unused=sf < 1.0)
host = [db[k] for k in sorted(db)]
for name, sql in tpch_full.QUERIES.items():
    if only and name not in only:
        continue
    cache = tempfile.mkdtemp(prefix="rsq_tier_")
    ctx = engine.Context(device=0, cache_dir=cache)
    tabs = [ctx.table(t) for t in host]
    gold = os.path.join(ROOT, "tests", "golden", f"ref_full_{name}_sf{sf:g}.tbl")
    want = open(gold, encoding="latin1").read() if os.path.exists(gold) else None
    t0 = time.perf_counter()
    q = ctx.sql_compile(sql, tabs)
    compile_ms = (time.perf_counter() - t0) * 1e3
    reached, exec_ms, ok = {}, {}, True
    deadline = time.time() + 120
    while time.time() < deadline:
        t = time.perf_counter()
        q.execute()
        e = (time.perf_counter() - t) * 1e3
        ex = q.explain
        tier = "full" if ("kernel tier: full" in ex or ("generic pre-compiled" not in ex and "kernel tier" not in ex)) else "quick" if "kernel tier: quick" in ex else "interpreter"
        if tier not in reached:
            reached[tier] = (time.perf_counter() - t0) * 1e3
        exec_ms[tier] = min(exec_ms.get(tier, 1e9), e)
        if want is not None:
            ok = ok and q.result().text == want
        if tier == "full" and (time.perf_counter() - t0) * 1e3 > reached["full"] + 50:      # a few executions on the final kernels, then done
            break
    print(json.dumps({"query": name, "sf": sf, "compile_ms": round(compile_ms, 3),
                      "first_answer_after_ms": round(reached.get("interpreter", reached.get("quick", reached.get("full", 0))), 1),
                      "tier_reached_after_ms": {k: round(v, 1) for k, v in reached.items()},
                      "best_exec_ms_on_tier": {k: round(v, 3) for k, v in exec_ms.items()},
                      "answers_equal_reference": ok if want is not None else None}), flush=True)
    q.close()
    for t in tabs:
        t.close()
    ctx.close()
    shutil.rmtree(cache, ignore_errors=True)
