"""PCIe-inclusive rate for DESIGN.md: upload of the 7 Q1 columns of an SF1 lineitem from pageable host memory through
rsq_table_create (includes the statistics kernels), then one Q1 execution.  Never part of bench.py's `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resql_amd import datagen, engine, tpch
sf = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
li = tpch.lineitem_table(sf, tpch.Q1_COLUMNS)
n = li.n_rows
ctx = engine.Context(device=0)
ctx.table(li).close()                                   # warm-up (first-touch, allocator)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); t = ctx.table(li); dt = time.perf_counter() - t0; best = min(best, dt)
    q = ctx.compile(tpch.q1_plan(li), [t]); q.execute(); q.close(); t.close()
print(f"upload of {n} rows x 38 B = {n * 38 / 1e6:.0f} MB: {best * 1e3:.1f} ms = {n * 38 / best / 1e9:.1f} GB/s = {n / best / 1e9:.2f} G rows/s (PCIe-inclusive)")
ctx.close()
