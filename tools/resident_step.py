"""The resident step (RSQ_PERSISTENT_STEP=1, engine.cpp "the resident step") against the launched one: TPC-H Q1 steps on one GPU.

One process measures one setting (the knob is read once per process): run it twice,
    python tools/resident_step.py --sf 1
    RSQ_PERSISTENT_STEP=1 python tools/resident_step.py --sf 1
Every step's answer is compared with the first step's; the first with the reference's own answer where a golden of that scale
factor is committed (tests/golden/ref_full_q1_sf{1,10}.tbl).  Besides the steady loop: a pause longer than the kernel waits (it
leaves by itself and the next step launches again), another query on the same context between two steps (the kernel is parked),
and a partial execution of the same query.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from resql_amd import engine, tpch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=1.0)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    args = ap.parse_args()
    resident = os.environ.get("RSQ_PERSISTENT_STEP") == "1"
    ctx = engine.Context(device=0)
    li = ctx.generate(engine.GEN_LINEITEM, tpch.datagen.n_lineitem(args.sf), args.sf)
    schema = tpch.lineitem_table(0.001, tpch.Q1_COLUMNS, n_rows=0)
    q = ctx.compile(tpch.q1_plan(schema), [li])
    q.await_kernels()
    q6 = ctx.compile(tpch.q6_plan(tpch.lineitem_table(0.001, tpch.Q6_COLUMNS, n_rows=0)), [li])
    q6.await_kernels()
    q.execute()
    first = q.result().text
    gold = os.path.join(ROOT, "tests", "golden", "ref_full_q1_sf%g.tbl" % args.sf)
    equals_reference = None
    if os.path.exists(gold):
        with open(gold) as f:
            equals_reference = f.read() == first
    ok = True
    for _ in range(args.warmup):
        q.execute()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        q.execute()
    us = (time.perf_counter() - t0) / args.steps * 1e6
    ok &= q.result().text == first
    # every step's answer (a shorter loop: reading the result costs more than the step)
    for _ in range(64):
        q.execute()
        ok &= q.result().text == first
    # the host goes away for longer than the kernel waits
    time.sleep(0.06)
    t1 = time.perf_counter()
    q.execute()
    after_pause_us = (time.perf_counter() - t1) * 1e6
    ok &= q.result().text == first
    # another query of the context in between
    q.execute()
    t2 = time.perf_counter()
    q6.execute()
    other_query_us = (time.perf_counter() - t2) * 1e6
    q6_first = q6.result().text
    q.execute()
    ok &= q.result().text == first
    q6.execute()
    ok &= q6.result().text == q6_first
    # a partial execution of the same query, then a full one
    q.execute()
    q.execute_partial()
    q.finalize()
    ok &= q.result().text == first
    q.execute()
    ok &= q.result().text == first
    line = {"workload": "TPC-H Q1 SF%g, %d steps" % (args.sf, args.steps), "resident_step": resident, "us_per_step": round(us, 2),
            "answers_equal": bool(ok), "answer_sha1": hashlib.sha1(first.encode()).hexdigest(), "equals_reference_answer": equals_reference,
            "first_step_after_60ms_pause_us": round(after_pause_us, 1), "other_query_after_a_step_us": round(other_query_us, 1),
            "device_us_last_step": round(q.report().kernel_time_ms * 1e3, 2)}
    print(json.dumps(line), flush=True)
    q.close(); q6.close(); li.close(); ctx.close()
    return 0 if ok and equals_reference is not False else 1


if __name__ == "__main__":
    sys.exit(main())
