"""Every environment switch the engine still reads selects a kernel form or a host path that also exists as an automatic choice or a
fallback (a look-back that timed out, a table that fell back to its hash form, a host without mapped pinned memory, ...).  A
form nobody runs rots: each switch is flipped here, in a fresh context, over seeded plans of the shapes it touches, and the
answers are compared with the oracle byte for byte (multiset for plans without an order, as the reference's own tests do,
test/test_common.h:152-190).  DESIGN.md lists the switches; `grep -oh '"RSQ_[A-Z0-9_]*"' resql_amd/csrc` must stay within them."""
import os
import re

import pytest

from resql_amd import engine, tpch, tpch_full
from oracle import orc

import fuzzplans
import test_gpu_fuzz_joins as joins

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# switch -> (value, which plan families exercise it)
KNOBS = [
    ("RSQ_POLL", "0", ("tpch", "fuzz")),                 # stream synchronisation instead of polling the sequence number
    ("RSQ_FUSED_STEP", "0", ("tpch", "fuzz")),           # the one-launch step as separate launches (its fallback after an exception)
    ("RSQ_FUSED_SELECT", "0", ("topk",)),                # candidate selection as separate launches (fallback of a timed-out meeting point)
    ("RSQ_SCAN_CHAINED", "0", ("rows", "joins")),        # offset scan in three launches (fallback of a timed-out look-back)
    ("RSQ_RANK_CHAINED", "0", ("joins", "q3")),          # rank index in two launches (fallback of a timed-out look-back)
    ("RSQ_PUBLISH_STATUS", "0", ("joins", "q3", "topk")),  # status words by copies (hosts without a device view of pinned memory)
    ("RSQ_COMPACT", "0", ("joins", "fuzz", "q3")),       # no wave compaction (pipelines that are not selective take this form anyway)
    ("RSQ_JOIN_RANK", "0", ("joins", "q3")),             # hash form of every join table (what a duplicate build key falls back to)
    ("RSQ_JOIN_BITMAP", "0", ("joins", "q3")),           # no key bitmap (what a wide key range falls back to)
    ("RSQ_CHECK_STATS", "1", ("tpch", "fuzz")),          # statistics range checks for engine-owned columns too
    ("RSQ_LATE_LOADS", "0", ("tpch", "fuzz")),
    ("RSQ_LATE_LOADS", "2", ("tpch", "fuzz", "joins")),
    ("RSQ_GENERIC", "0", ("fuzz",)),                     # blocking compile instead of the interpreter in front
    ("RSQ_GENERIC2", "0", ("joins",)),                   # no whole-pipeline interpreter: joins wait for their kernels
    ("RSQ_COMPILE_HELPERS", "0", ("joins",)),            # kernels compiled in process, one after the other
    ("RSQ_DEVICE_TOPK", "0", ("topk", "q3")),
    ("RSQ_GROUP_VALUES_BY_ADDRESS", "0", ("topk",)),     # string group values that depend on the key are copied into the entries (what a chain in its hash form falls back to)
    ("RSQ_DEVICE_TAIL", "0", ("dense_large",)),
    ("RSQ_DEVICE_REPLAY", "0", ("dense_large",)),
    ("RSQ_TAIL_THREADS", "1", ("dense_large", "q3")),    # (read once per process: only checks that the setting is accepted when it is the first)
    ("RSQ_PARTITION", "0", ("dense_large",)),
    ("RSQ_PARTITION", "2", ("dense_large",)),
    ("RSQ_STAGED", "0", ("dense_large",)),
    ("RSQ_AGG_MODE", "5", ("fuzz",)),
    ("RSQ_TRACE", "2", ("tpch",)),                       # the traced paths synchronise between kernels: another order of the same calls
    ("RSQ_DEBUG_TAIL", "1", ("q3",)),                    # device timestamps of the pipelines' workgroups (a kernel argument more)
]


def _check(ctx, plan, kind=None):
    want = orc.execute(plan)
    got = ctx.run(plan)
    if kind is not None:
        assert fuzzplans.same(kind, got.text, want.text)
    else:
        assert sorted(got.text.splitlines()) == sorted(want.text.splitlines()) and got.n_rows == want.n_rows


def _family(ctx, name):
    if name == "tpch":
        li = tpch.lineitem_table(0.02, tpch.Q1_COLUMNS)
        for plan in (tpch.q1_plan(li), tpch.q6_plan(li)):
            want, got = orc.execute(plan), ctx.run(plan)
            assert got.text == want.text
    elif name == "q3":
        sf = 0.05
        plan = tpch.q3_plan(tpch.customer_table(sf), tpch.orders_table(sf), tpch.lineitem_table(sf, tpch.Q3_LINEITEM_COLUMNS))
        want, got = orc.execute(plan), ctx.run(plan)
        key = lambda res: [(res.value(r, 1), res.value(r, 2)) for r in range(res.n_rows)]
        assert key(got) == key(want) and sorted(got.text.splitlines()) == sorted(want.text.splitlines())
    elif name == "fuzz":
        for seed in range(40, 64):
            plan, kind = fuzzplans.make(seed)
            try:
                want = orc.execute(plan)
            except orc.OracleError:
                continue
            got = ctx.run(plan)
            assert fuzzplans.same(kind, got.text, want.text), seed
    elif name == "joins":
        for seed in range(7000, 7016):
            plan, what = joins.make(seed)
            try:
                want = orc.execute(plan)
            except orc.OracleError:
                continue
            got = ctx.run(plan)
            assert sorted(got.text.splitlines()) == sorted(want.text.splitlines()), (seed, what)
    elif name == "rows":
        from resql_amd import plan as P
        t = tpch.synthetic_table(300_000, 64)
        p = P.Plan([t])
        node = p.selection(p.lt(p.attr("a"), p.constant(str(1 << 26), P.BIGINT)), p.scan("t"))
        plan = p.set_root(p.materialize(p.projection([p.attr("b"), p.attr("c")], node)), limit=5000)
        want, got = orc.execute(plan), ctx.run(plan)
        assert got.text == want.text
    elif name == "topk":
        db = tpch_full.database(0.02)
        host = [db[k] for k in sorted(db)]
        tabs = [ctx.table(t) for t in host]
        for q in ("q3", "q10"):
            sql = tpch_full.QUERIES[q]
            want = orc.execute(ctx.sql_plan(sql, tabs, host))
            cq = ctx.sql_compile(sql, tabs)
            cq.execute(); cq.execute()
            got = cq.result()
            cq.close()
            assert sorted(got.text.splitlines()) == sorted(want.text.splitlines()) and got.n_rows == want.n_rows
        for t in tabs:
            t.close()
    elif name == "dense_large":
        t = tpch.synthetic_table(400_000, 1 << 17)
        plan = tpch.synthetic_plan(t, 1 << 30)
        want = orc.execute(plan)
        tabs = [ctx.table(t)]
        cq = ctx.compile(plan, tabs)
        for _ in range(2):
            cq.execute()
            assert cq.result().text == want.text
        cq.close(); tabs[0].close()
    else:
        raise AssertionError(name)


@pytest.mark.parametrize("knob,value,families", KNOBS, ids=[f"{k}={v}" for k, v, _ in KNOBS])
def test_switch_flipped(monkeypatch, tmp_path, knob, value, families):
    monkeypatch.setenv(knob, value)
    ctx = engine.Context(device=0)
    try:
        for fam in families:
            _family(ctx, fam)
    finally:
        ctx.close()
