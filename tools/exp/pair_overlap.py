"""Does the tail of one shard's execution overlap the scan of the next when the two run on TWO contexts of the same GPU from two host threads?
(VERDICT r04 "next round" 6 asks for rsq_query_execute_async / _wait; a second context is the engine's existing unit of concurrency: own stream,
own error word, own arenas.)   usage: python tools/exp/pair_overlap.py [selectivity] [rows]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from resql_amd import engine, tpch
sel = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1_250_000_000
groups = 1 << 20
thr = int(sel * (1 << 31))
ctxs = [engine.Context(device=0), engine.Context(device=0)]
schema = tpch.synthetic_table(16, groups)
tabs = [c.generate(engine.GEN_SYNTHETIC, rows, 1.0, row0=i * rows, param=groups) for i, c in enumerate(ctxs)]
qs = [c.compile(tpch.synthetic_plan(schema, thr), [t]) for c, t in zip(ctxs, tabs)]
for q in qs:
    q.await_kernels(); q.execute(); q.execute()
def seq():
    t = time.perf_counter()
    for q in qs: q.execute()
    return (time.perf_counter() - t) * 1e3
def par():
    ths = [threading.Thread(target=q.execute) for q in qs]
    t = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    return (time.perf_counter() - t) * 1e3
def staggered():
    """shard B starts when shard A's kernels are done (A's tail then runs beside B's scan)"""
    done = threading.Event()
    def a():
        qs[0].execute()
    tA = threading.Thread(target=a)
    t = time.perf_counter()
    tA.start()
    time.sleep(max(0.0, qs[0].report().kernel_time_ms * 1e-3 * 0.9))
    qs[1].execute()
    tA.join()
    return (time.perf_counter() - t) * 1e3
s = [seq() for _ in range(4)]; p = [par() for _ in range(4)]; g = [staggered() for _ in range(4)]
r = [q.report() for q in qs]
print(f"sel {sel}: one shard exec {r[0].execution_time_ms:.2f} ms (kernels {r[0].kernel_time_ms:.2f}, tail {r[0].finalize_time_ms:.2f}); two shards one after the other {min(s):.2f} ms, "
      f"from two threads on two contexts {min(p):.2f} ms, staggered {min(g):.2f} ms")
