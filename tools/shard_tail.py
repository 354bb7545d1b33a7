#!/usr/bin/env python3
"""One 1.25 B-row shard of BASELINE config 5 end to end, several executions of ONE compiled query (cold first, then warm):
kernel ms, tail ms, whole execution ms.  RSQ_TRACE=1 prints the phases.
  python tools/shard_tail.py [log2 groups = 20] [selectivity = 0.1] [rows = 1250000000] [emission = reference|any]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resql_amd import engine, tpch  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sel = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1_250_000_000
emit = engine.EMIT_ANY if len(sys.argv) > 4 and sys.argv[4] == "any" else engine.EMIT_REFERENCE
ctx = engine.Context(device=0, emission_order=emit)
shard = ctx.generate(engine.GEN_SYNTHETIC, rows, 1.0, param=1 << lg)
q = ctx.compile(tpch.synthetic_plan(tpch.synthetic_table(16, 1 << lg), int(sel * (1 << 31))), [shard])
q.await_kernels()
for i in range(5):
    q.execute()
    r = q.report()
    print(f"execution {i}: exec {r.execution_time_ms:.3f} ms, kernels {r.kernel_time_ms:.3f} ms, tail {r.finalize_time_ms:.3f} ms, "
          f"{r.num_kernels} launches, {q.result(text=False).n_rows} rows; exec / kernels = {r.execution_time_ms / r.kernel_time_ms:.2f}", flush=True)
q.close(); shard.close(); ctx.close()
