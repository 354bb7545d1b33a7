"""The reference's eight TPC-H statements as SQL text through the engine (parse -> plan -> HIP pipelines) on a resident
database, with the unmodified reference (its own grammar + planner + asmjit JIT, fed the same token streams) timed on the
host beside it.  usage: python tools/sql_bench.py [SF] [--reference] [--repeat N]
Prints one JSON line per statement: device time, whole execution (best of the repeated executions of ONE query), rows, first_ever_exec_ms (the
first execution of the statement on the context), first_exec_ms (a second, fresh query of the same statement: compile -> execute once, what the
plan memo and the arenas are for) and (with --reference) the reference's `execute:` time."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resql_amd import engine, tpch_full  # noqa: E402

sf = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 1.0
with_ref = "--reference" in sys.argv
repeat = int(sys.argv[sys.argv.index("--repeat") + 1]) if "--repeat" in sys.argv else 5

t0 = time.time()
db = tpch_full.database(sf, fill_unused=sf < 1.0)      # (full sizes: columns no statement reads stay without data)
names = sorted(db)
host = [db[k] for k in names]
print(f"# generated SF{sf:g} in {time.time() - t0:.1f} s: " + ", ".join(f"{t.name} {t.n_rows}" for t in host), flush=True)
ctx = engine.Context(device=0)
tabs = [ctx.table(t) for t in host]
ref_case = None
if with_ref:
    from oracle import orc
    import tempfile
    from resql_amd import plan as P
    tmp = tempfile.mkdtemp(prefix="resql_sqlbench_")
    shell = P.Plan(host); shell.root = 0
    ref_case = orc.write_case(shell, tmp)

only = sys.argv[sys.argv.index("--only") + 1].split(",") if "--only" in sys.argv else None
for name, sql in tpch_full.QUERIES.items():
    if only and name not in only:
        continue
    q = ctx.sql_compile(sql, tabs)
    q.await_kernels()
    q.execute()                                    # the first execution of this statement on the context: nothing remembered
    first_ever = q.report().execution_time_ms
    best_k, best_e = 1e9, 1e9
    for _ in range(repeat):
        q.execute()
        r = q.report()
        if r.kernel_time_ms > 0:
            best_k = min(best_k, r.kernel_time_ms)
        best_e = min(best_e, r.execution_time_ms)
    res = q.result()
    out = {"query": name, "sf": sf, "kernel_ms": round(best_k, 3), "exec_ms": round(best_e, 3), "kernels": r.num_kernels,
           "rows": res.n_rows, "bytes_read": r.bytes_read, "gbps": round(r.bytes_read / (best_k * 1e-3) / 1e9, 1),
           "frac_of_8TBps": round(r.bytes_read / (best_k * 1e-3) / 8e12, 3)}
    gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", f"ref_full_{name}_sf{sf:g}.tbl")
    if os.path.exists(gold):
        out["equals_reference_answer"] = res.text == open(gold, encoding="latin1").read()
    q.close()
    # what a ReSQL host pays per SELECT (compile -> ONE execution -> delete, reference src/execute.h:213-247): a fresh query of the same statement
    q1 = ctx.sql_compile(sql, tabs)
    q1.execute()
    out["first_exec_ms"] = round(q1.report().execution_time_ms, 3)
    out["first_exec_equals_reference_answer"] = (q1.result().text == open(gold, encoding="latin1").read()) if os.path.exists(gold) else None
    out["first_ever_exec_ms"] = round(first_ever, 3)
    q1.close()
    if with_ref:
        import subprocess
        tp = os.path.join(tmp, f"{name}.tokens")
        with open(tp, "w", encoding="latin1") as f:
            f.write(ctx.sql_describe(sql, 0))
        pr = subprocess.run([orc.REF_HARNESS, ref_case, "--sql-tokens", tp, "--repeat", "2", "--quiet"], capture_output=True, text=True)
        ms = [float(l.split()[4]) for l in pr.stderr.splitlines() if l.startswith("#timing")]
        out["reference_exec_ms"] = round(min(ms), 1) if ms else None
        load = [float(l.split()[1]) for l in pr.stderr.splitlines() if l.startswith("#load_ms")]
        out["reference_load_ms"] = round(load[0], 0) if load else None
    print(json.dumps(out), flush=True)
for t in tabs:
    t.close()
ctx.close()
if with_ref:
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
