#!/usr/bin/env python3
"""Random valid statements (tests/sqlgen.py) of seeds beyond the suite's, at SF 0.05, each executed three times (interpreter, specialised
kernels, warm paths) and then twice more as a fresh query executed once (the plan memo's path), against the oracle.  usage: python tools/exp/sql_sweep.py FIRST COUNT"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from resql_amd import engine, tpch_full
from oracle import orc
import sqlgen

first, count = int(sys.argv[1]), int(sys.argv[2])
ctx = engine.Context(device=0)
db = tpch_full.database(0.05)
host = [db[k] for k in sorted(db)]
tabs = [ctx.table(t) for t in host]
ran = skipped = refused = 0
bad = []
t0 = time.time()
for seed in range(first, first + count):
    s = sqlgen.statement(seed)
    try:
        want = orc.execute(ctx.sql_plan(s, tabs, host))
    except orc.OracleError:
        skipped += 1; continue
    except engine.EngineError:
        refused += 1; continue
    try:
        q = ctx.sql_compile(s, tabs)
    except engine.EngineError as e:
        refused += 1; continue
    for rep in range(3):
        if rep == 1: q.await_kernels()
        q.execute()
        if q.result().text != want.text:
            bad.append((seed, rep)); print("MISMATCH", seed, rep, s, flush=True); break
    q.close(); ran += 1
    if not bad or bad[-1][0] != seed:      # ... and what a ReSQL host does: a FRESH query of the same statement, executed once (served by the plan memo)
        for rep in (3, 4):
            q = ctx.sql_compile(s, tabs); q.await_kernels(); q.execute()
            if q.result().text != want.text:
                bad.append((seed, rep)); print("MISMATCH (fresh query)", seed, rep, s, flush=True)
            q.close()
    if ran % 40 == 0: print(f"... {ran} statements, {len(bad)} bad, {time.time() - t0:.0f} s", flush=True)
print(f"done: {ran} statements run, {skipped} the reference dies on, {refused} refused by the engine, {len(bad)} bad: {bad}")
sys.exit(1 if bad else 0)
