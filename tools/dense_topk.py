"""ORDER BY aggregate DESC LIMIT 10 over a large dense group domain: whole-execution time with the device-side candidate
pre-selection (default) and with RSQ_DEVICE_TOPK=0 (the whole aggregate table is read back and every group materialised on the
host).  usage: python tools/dense_topk.py [ROWS] [GROUPS]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resql_amd import engine, plan as P, tpch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
ctx = engine.Context(device=0)
t = ctx.generate(engine.GEN_SYNTHETIC, n, 1.0, param=groups)
schema = tpch.synthetic_table(16, groups)
for mode in ("1", "0"):
    os.environ["RSQ_DEVICE_TOPK"] = mode
    p = P.Plan([schema])
    b = p.attr("b")
    s, cnt = p.sum(p.attr("c")), p.count(p.star())
    node = p.selection(p.lt(p.attr("a"), p.constant(str(1 << 30), P.BIGINT)), p.scan("t"))
    node = p.aggregation([s, cnt], [b], node)
    node = p.projection([b, p.as_("s", s), p.as_("n", cnt)], node)
    node = p.orderby([p.desc(p.attr("s")), p.attr("b")], node)
    q = ctx.compile(p.set_root(node, limit=10), [t])
    q.await_kernels()
    for _ in range(3):
        q.execute()
    r = q.report()
    print(f"RSQ_DEVICE_TOPK={mode}: kernel_ms {r.kernel_time_ms:.3f} exec_ms {r.execution_time_ms:.3f} finalize_ms {r.finalize_time_ms:.3f}", flush=True)
    first = q.result().text.splitlines()[1]
    print("  first row:", first)
    q.close()
t.close()
ctx.close()
