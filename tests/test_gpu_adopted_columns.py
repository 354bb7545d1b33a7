"""rsq_table_create_device adopts caller-owned device memory.  The statistics gathered at adoption (min / max, byte-value
sets) shape the compiled kernels (dense group ids, key bitmaps); the kernels range-check what they derive from them, so a
caller that refills an adopted column gets RSQ_ERR_RUNTIME — never an out-of-bounds access."""
import numpy as np
import pytest

from resql_amd import engine, plan as P
from oracle import orc

pytestmark = pytest.mark.gpu
T = P.TypeInit


def _plan(tables, with_join):
    p = P.Plan(tables)
    if with_join:
        j = p.hashjoin([p.eq(p.attr("dk"), p.attr("k"))], p.scan("dim"), p.scan("t"), single_match=True)
        s = p.sum(p.attr("v"))
        node = p.aggregation([s], [p.attr("g")], j)
        node = p.projection([p.attr("g"), p.as_("s", s)], node)
    else:
        s, c = p.sum(p.attr("v")), p.count(p.star())
        node = p.aggregation([s, c], [p.attr("k"), p.attr("f")], p.scan("t"))
        node = p.projection([p.attr("k"), p.attr("f"), p.as_("s", s), p.as_("c", c)], node)
    return p.set_root(p.materialize(node))


@pytest.mark.parametrize("with_join", [False, True])
def test_refilled_adopted_column_fails_cleanly(gpu_ctx, with_join):
    import torch
    n = 100_000
    rng = np.random.default_rng(11)
    k = rng.integers(10, 50, n).astype(np.int32)
    f = rng.choice(np.frombuffer(b"AB", dtype=np.uint8), n)
    v = rng.integers(0, 1000, n).astype(np.int64)
    dk = np.arange(10, 50, dtype=np.int32)
    dg = (dk % 7).astype(np.int32)
    dev = {name: torch.from_numpy(a).cuda() for name, a in (("k", k), ("f", f), ("v", v), ("dk", dk), ("dg", dg))}
    t = gpu_ctx.table_from_device("t", n, [("k", T.INT(), dev["k"].data_ptr()), ("f", T.CHAR(1), dev["f"].data_ptr()), ("v", T.BIGINT(), dev["v"].data_ptr())])
    dim = gpu_ctx.table_from_device("dim", len(dk), [("dk", T.INT(), dev["dk"].data_ptr()), ("g", T.INT(), dev["dg"].data_ptr())])
    host_t = P.Table("t", [P.Column("k", T.INT(), k), P.Column("f", T.CHAR(1), f), P.Column("v", T.BIGINT(), v)], n)
    host_d = P.Table("dim", [P.Column("dk", T.INT(), dk), P.Column("g", T.INT(), dg)], len(dk))
    tables, host = ([dim, t], [host_d, host_t]) if with_join else ([t], [host_t])
    q = gpu_ctx.compile(_plan(host, with_join), tables)
    try:
        q.execute()
        want = orc.execute(_plan(host, with_join)).text
        assert sorted(q.result().text.splitlines()) == sorted(want.splitlines())
        # refill with values the statistics never saw: beyond max, below min, a byte outside the set
        saved = dev["dk" if with_join else "k"].clone()
        col = dev["dk"] if with_join else dev["k"]
        col[5] = 1_000_000
        col[6] = -7
        torch.cuda.synchronize()
        with pytest.raises(engine.EngineError) as e:
            q.execute()
        assert e.value.status == 5 and "column statistics" in str(e.value)
        col.copy_(saved)
        if not with_join:
            fs = dev["f"].clone()
            dev["f"][3] = ord("Z")
            torch.cuda.synchronize()
            with pytest.raises(engine.EngineError) as e:
                q.execute()
            assert e.value.status == 5
            dev["f"].copy_(fs)
        torch.cuda.synchronize()
        q.execute()                                  # the original data: the original answer
        assert sorted(q.result().text.splitlines()) == sorted(want.splitlines())
    finally:
        q.close(); t.close(); dim.close()


def test_a_byte_inside_the_range_but_outside_the_value_set_fails_and_refresh_stats_answers(gpu_ctx):
    """VERDICT r04 "what's missing" 3: l_returnflag-like column with the value set {A, N, R}; the host writes 'B' (inside [A, R], not in the
    set).  The dense group ids come from the set: the row must not be counted into a neighbouring group.  The compiled query fails
    with RSQ_ERR_RUNTIME; after rsq_table_refresh_stats a newly compiled query gives the oracle's answer for the changed data."""
    import torch
    n = 50_000
    rng = np.random.default_rng(5)
    f = rng.choice(np.frombuffer(b"ANR", dtype=np.uint8), n)
    s = rng.choice(np.frombuffer(b"FO", dtype=np.uint8), n)
    v = rng.integers(0, 1000, n).astype(np.int64)
    dev = {name: torch.from_numpy(a).cuda() for name, a in (("f", f), ("s", s), ("v", v))}
    t = gpu_ctx.table_from_device("t", n, [("f", T.CHAR(1), dev["f"].data_ptr()), ("s", T.CHAR(1), dev["s"].data_ptr()), ("v", T.BIGINT(), dev["v"].data_ptr())])

    def plan(host_f):
        host = P.Table("t", [P.Column("f", T.CHAR(1), host_f), P.Column("s", T.CHAR(1), s), P.Column("v", T.BIGINT(), v)], n)
        p = P.Plan([host])
        sm, c = p.sum(p.attr("v")), p.count(p.star())
        node = p.aggregation([sm, c], [p.attr("f"), p.attr("s")], p.scan("t"))
        node = p.projection([p.attr("f"), p.attr("s"), p.as_("sm", sm), p.as_("c", c)], node)
        return p.set_root(p.materialize(node))

    q = gpu_ctx.compile(plan(f), [t])
    q2 = None
    try:
        q.execute()
        assert sorted(q.result().text.splitlines()) == sorted(orc.execute(plan(f)).text.splitlines())
        assert "{65,78,82}" in q.explain                  # the dense layout came from the byte-value set
        dev["f"][123] = ord("B")
        dev["f"][4567] = ord("B")
        torch.cuda.synchronize()
        with pytest.raises(engine.EngineError) as e:
            q.execute()
        assert e.value.status == 5 and "column statistics" in str(e.value)
        t.refresh_stats()
        f2 = f.copy(); f2[123] = ord("B"); f2[4567] = ord("B")
        q2 = gpu_ctx.compile(plan(f2), [t])
        assert "{65,66,78,82}" in q2.explain
        q2.execute()
        assert sorted(q2.result().text.splitlines()) == sorted(orc.execute(plan(f2)).text.splitlines())
    finally:
        q.close()
        if q2 is not None:
            q2.close()
        t.close()
