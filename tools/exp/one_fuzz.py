#!/usr/bin/env python3
"""One fuzz plan against the oracle, with the plan's explain text and the differing rows of every execution.
usage: python tools/exp/one_fuzz.py SEED [joins] [EXECUTIONS]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from resql_amd import engine
from oracle import orc
import fuzzplans
import test_gpu_fuzz_joins as joins

seed = int(sys.argv[1])
kind_joins = len(sys.argv) > 2 and sys.argv[2] == "joins"
n = int(sys.argv[-1]) if len(sys.argv) > 2 and sys.argv[-1].isdigit() else 3
plan, kind = joins.make(seed) if kind_joins else fuzzplans.make(seed)
want = orc.execute(plan)
ctx = engine.Context(device=0)
tabs = [ctx.table(t) for t in plan.tables]
q = ctx.compile(plan, tabs)
print(q.explain)
for rep in range(n):
    if rep == 1:
        q.await_kernels()
    q.execute()
    got = q.result()
    g, w = sorted(got.text.splitlines()), sorted(want.text.splitlines())
    print(f"execution {rep}: {got.n_rows} rows (oracle {want.n_rows}), equal as multisets: {g == w}, equal as text: {got.text == want.text}")
    if g != w:
        gs, ws = set(g), set(w)
        print("  only in the engine's answer:", sorted(gs - ws)[:12])
        print("  only in the oracle's answer:", sorted(ws - gs)[:12])
    print("  " + q.explain.splitlines()[-1][:300])
