#!/usr/bin/env python3
"""Host tail of a large dense aggregation in isolation (no GPU): a stand-in partial table of G groups goes through
rsq_query_finalize_host — groups from the table, emission order (sort by first row, the reference's hashes, replay of its hash
table), AVG / projection / materialise — and the phases are timed (RSQ_TRACE=1 prints them).
  python tools/tail_bench.py [log2 groups = 20] [rows = 1250000000] [present fraction = 1.0]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resql_amd import engine, plan as P, tpch  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1_250_000_000
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
G = 1 << lg
T = P.TypeInit
ctx = engine.Context(device=-1)
n = 4096
rng = np.random.default_rng(1)
b = rng.integers(0, G, n).astype(np.int64); b[0] = 0; b[1] = G - 1
t = P.Table("t", [P.Column("a", T.BIGINT(), rng.integers(0, 1 << 31, n).astype(np.int64)), P.Column("b", T.BIGINT(), b),
                  P.Column("c", T.BIGINT(), rng.integers(0, 1 << 20, n).astype(np.int64)),
                  P.Column("d", T.BIGINT(), rng.integers(0, 1 << 20, n).astype(np.int64))], n)
dev = ctx.table(t)
# the reference sizes its table from the scanned relation: give the plan the row count of the real shard
big = P.Table("t", t.columns, n)
q = ctx.compile(tpch.synthetic_plan(big, 1 << 30), [dev])
n_min, n_max, n_sum = q.partial_layout()
assert n_min == G and n_sum == 3 * G, (n_min, n_max, n_sum)
words = np.zeros(n_min + n_max + n_sum, dtype=np.int64)
first = rng.permutation(rows if rows < (1 << 26) else (1 << 26))[:G].astype(np.int64) * max(1, rows >> 26)
present = rng.random(G) < frac
words[:G] = np.where(present, first, np.iinfo(np.int64).max)
words[G:2 * G] = rng.integers(0, 1 << 40, G)
words[2 * G:3 * G] = rng.integers(0, 1 << 40, G)
words[3 * G:] = np.where(present, rng.integers(1, 2000, G), 0)
for i in range(3):
    t0 = time.perf_counter()
    q.finalize_host(words)
    dt = (time.perf_counter() - t0) * 1e3
    print(f"finalize_host: {dt:.2f} ms, {q.result(text=False).n_rows} rows", flush=True)
import hashlib
print("sha256 of the result tuples:", hashlib.sha256(q.result(text=False).tuples).hexdigest())
