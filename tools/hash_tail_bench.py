#!/usr/bin/env python3
"""The tail of a hash aggregation with many groups, device against host: select b * 3 + 1, sum(c), count(*), min(d), max(c), avg(d) from t
where a < tau group by b * 3 + 1 over a device-generated synthetic table (a computed key: the generic hash aggregation; no ORDER BY:
the rows leave in the reference's emission order).  Prints per setting the whole execution, the kernels and the tail (finalize) time.
  python tools/hash_tail_bench.py [rows = 200000000] [groups = 1048576]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resql_amd import engine, plan as P, tpch  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20


def plan_of(t):
    p = P.Plan([t])
    key = p.add(p.mul(p.attr("b"), p.constant("3", P.BIGINT)), p.constant("1", P.BIGINT))
    sc, cnt, lo, hi, av = p.sum(p.attr("c")), p.count(p.star()), p.min(p.attr("d")), p.max(p.attr("c")), p.avg(p.attr("d"))
    node = p.selection(p.lt(p.attr("a"), p.constant(str(1 << 30), P.BIGINT)), p.scan("t"))
    node = p.aggregation([sc, cnt, lo, hi, av], [key], node)
    return p.set_root(p.materialize(p.projection([p.as_("k", key), p.as_("s", sc), p.as_("n", cnt), p.as_("lo", lo), p.as_("hi", hi), p.as_("av", av)], node)))


print("# tools/hash_tail_bench.py on one MI355X: the tail of a hash aggregation (computed group key, 6 output columns of 8 bytes, no ORDER BY: rows in\n"
      "# the reference's emission order), device (engine_devtail.cpp runRowsDeviceTail, devtail.hip k_row_*) against the host's worker pool.\n"
      "# \"tail\" includes the copy of the finished tuples to the host (1 M x 48 B = 50 MB over PCIe is ~2 ms of it).\n"
      f"{rows // 1000000} M rows, {groups} groups:", flush=True)
ctx = engine.Context(device=0)
t = ctx.generate(engine.GEN_SYNTHETIC, rows, 1.0, param=groups)
import hashlib
for setting in ("1", "0"):
    os.environ["RSQ_DEVICE_TAIL"] = setting
    q = ctx.compile(plan_of(tpch.synthetic_table(16, groups)), [t])
    q.await_kernels()
    best = None
    for _ in range(4):
        q.execute()
        r = q.report()
        cur = (r.execution_time_ms, r.kernel_time_ms, r.finalize_time_ms)
        best = cur if best is None or cur[0] < best[0] else best
    res = q.result(text=False)
    print(f"RSQ_DEVICE_TAIL={setting}: {res.n_rows} groups, execution {best[0]:.2f} ms, kernels {best[1]:.2f} ms, tail {best[2]:.2f} ms, "
          f"sha256 {hashlib.sha256(res.tuples).hexdigest()[:16]}", flush=True)
    q.close()
