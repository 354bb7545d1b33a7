"""Timings of the generic (non-BASELINE) device paths at scale: materialisation of a selection, generic hash aggregation
on a computed key, join + materialise.  Development aid.  usage: python tools/generic_paths.py [ROWS]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resql_amd import engine, plan as P, tpch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
ctx = engine.Context(device=0)
schema = tpch.synthetic_table(16, 1 << 20)


def run(label, plan, tables, reps=3):
    q = ctx.compile(plan, tables)
    for _ in range(reps):
        q.execute()
    r = q.report()
    print(f"{label:58s} kernel_ms {r.kernel_time_ms:9.3f} exec_ms {r.execution_time_ms:9.3f} fin_ms {r.finalize_time_ms:8.3f} rows {q.result().n_rows}", flush=True)
    q.close()


for groups in tuple(int(g) for g in os.environ.get('RSQ_TEST_GROUPS', '1024,1048576').split(',')):
    t = ctx.generate(engine.GEN_SYNTHETIC, n, 1.0, param=groups)
    for sel in (0.01, 0.1):
        thr = str(int(sel * (1 << 31)))
        # select a, b, c, d where a < thr  (materialise in scan order)
        p = P.Plan([schema])
        node = p.selection(p.lt(p.attr("a"), p.constant(thr, P.BIGINT)), p.scan("t"))
        run(f"materialize sel={sel} (4 cols of {n} rows)", p.set_root(p.materialize(node), request_all=True), [t], reps=2)
    for sel in (0.1, 0.5):
        thr = str(int(sel * (1 << 31)))
        # group by a computed key: b * 2 + 1  -> generic hash aggregation
        p = P.Plan([schema])
        key = p.add(p.mul(p.attr("b"), p.constant("2", P.BIGINT)), p.constant("1", P.BIGINT))
        sc, cnt = p.sum(p.attr("c")), p.count(p.star())
        node = p.selection(p.lt(p.attr("a"), p.constant(thr, P.BIGINT)), p.scan("t"))
        node = p.aggregation([sc, cnt], [key], node)
        node = p.projection([p.as_("k", key), p.as_("s", sc), p.as_("n", cnt)], node)
        run(f"hash agg computed key groups={groups} sel={sel}", p.set_root(p.materialize(node)), [t])
    t.close()
ctx.close()
