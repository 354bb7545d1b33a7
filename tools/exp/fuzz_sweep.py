#!/usr/bin/env python3
"""Seeds beyond the ones the suite pins: fuzz plans and join plans against the oracle, each executed twice (interpreter first, then the
specialised kernels) and once more as a fresh query on the same tables (the plan memo's path).  usage: python tools/exp/fuzz_sweep.py FIRST COUNT [joins]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from resql_amd import engine
from oracle import orc
import fuzzplans
import test_gpu_fuzz_joins as joins

first, count = int(sys.argv[1]), int(sys.argv[2])
kind_joins = len(sys.argv) > 3 and sys.argv[3] == "joins"
ctx = engine.Context(device=0)
bad, ran, skipped, t0 = [], 0, 0, time.time()
for seed in range(first, first + count):
    try:
        if kind_joins:
            plan, what = joins.make(seed); kind = None
        else:
            plan, kind = fuzzplans.make(seed); what = kind
        want = orc.execute(plan)
    except orc.OracleError:
        skipped += 1; continue
    tabs = [ctx.table(t) for t in plan.tables]
    try:
        q = ctx.compile(plan, tabs)
        for rep in range(2):
            if rep == 1: q.await_kernels()
            q.execute()
            got = q.result()
            ok = fuzzplans.same(kind, got.text, want.text) if kind is not None else sorted(got.text.splitlines()) == sorted(want.text.splitlines())
            if not ok:
                bad.append((seed, what, rep)); print("MISMATCH", seed, what, "execution", rep, flush=True); break
        q.close()
        if not bad or bad[-1][0] != seed:      # a FRESH query of the same plan on the same tables, executed once: what the plan memo serves
            q = ctx.compile(plan, tabs); q.await_kernels(); q.execute()
            got = q.result()
            ok = fuzzplans.same(kind, got.text, want.text) if kind is not None else sorted(got.text.splitlines()) == sorted(want.text.splitlines())
            if not ok:
                bad.append((seed, what, "fresh")); print("MISMATCH", seed, what, "fresh query", flush=True)
            q.close()
    except engine.EngineError as e:
        print("ENGINE ERROR", seed, what, str(e)[:160], flush=True); bad.append((seed, what, "error"))
    for t in tabs: t.close()
    ran += 1
    if ran % 50 == 0: print(f"... {ran} plans, {len(bad)} bad, {time.time() - t0:.0f} s", flush=True)
print(f"done: {ran} plans run, {skipped} skipped (oracle refuses), {len(bad)} bad: {bad}")
sys.exit(1 if bad else 0)
