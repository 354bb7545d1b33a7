"""A join probed for ALL matches whose build side has unique keys and contributes nothing but the key (TPC-H Q3's customer side)
is, in the engine's rank form, a key SET: the key bitmap alone, no entries (codegen.cpp consumeBuild, HashTable::setOnly).
Duplicate build keys - found by the sizing pass, or appearing later in adopted columns - put the hash table back."""
import numpy as np
import pytest

from resql_amd import engine, plan as P
from oracle import orc

pytestmark = pytest.mark.gpu
T = P.TypeInit


def _tables(n, m, dup, seed=4):
    rng = np.random.default_rng(seed)
    dk = rng.permutation(np.arange(100, 100 + 4 * m, dtype=np.int32))[:m].copy()
    if dup:
        dk[m // 3] = dk[m // 2]
    k = rng.integers(90, 110 + 4 * m, n).astype(np.int32)
    dim = P.Table("dim", [P.Column("dk", T.INT(), dk)], m)
    fact = P.Table("t", [P.Column("k", T.INT(), k), P.Column("g", T.INT(), rng.integers(0, 7, n).astype(np.int32)),
                         P.Column("v", T.BIGINT(), rng.integers(0, 1000, n).astype(np.int64))], n)
    return dim, fact


def _plan(dim, fact):
    p = P.Plan([dim, fact])
    j = p.hashjoin([p.eq(p.attr("dk"), p.attr("k"))], p.scan("dim"), p.scan("t"), single_match=False)
    s, c = p.sum(p.attr("v")), p.count(p.star())
    node = p.aggregation([s, c], [p.attr("g")], j)
    return p.set_root(p.materialize(p.projection([p.attr("g"), p.as_("s", s), p.as_("c", c)], node)))


@pytest.mark.parametrize("dup", [False, True])
def test_key_set_or_hash_table(gpu_ctx, monkeypatch, capfd, dup):
    monkeypatch.setenv("RSQ_TRACE", "1")
    dim, fact = _tables(400_000, 50_000, dup)
    plan = _plan(dim, fact)
    tabs = [gpu_ctx.table(t) for t in plan.tables]
    q = gpu_ctx.compile(plan, tabs)
    try:
        assert "nothing but the bitmap when the build keys prove unique" in q.explain
        q.execute()
        first = q.result().text
        q.execute()
        assert q.result().text == first
        err = capfd.readouterr().err
        assert ("ht0: hash table" in err) == dup and ("ht0: bitmap-rank dictionary" in err) == (not dup)
        assert sorted(first.splitlines()) == sorted(orc.execute(plan).text.splitlines())
    finally:
        q.close()
        for t in tabs:
            t.close()


def test_duplicates_appearing_later_put_the_hash_table_back(gpu_ctx):
    import torch
    dim, fact = _tables(300_000, 20_000, False, seed=8)
    dk = torch.from_numpy(dim.columns[0].data).cuda()
    ddim = gpu_ctx.table_from_device("dim", dim.n_rows, [("dk", T.INT(), dk.data_ptr())])
    dfact = gpu_ctx.table(fact)
    q = gpu_ctx.compile(_plan(dim, fact), [ddim, dfact])
    try:
        q.execute()
        assert sorted(q.result().text.splitlines()) == sorted(orc.execute(_plan(dim, fact)).text.splitlines())
        dim.columns[0].data[7] = dim.columns[0].data[11]            # two build rows with one key (inside the column's recorded range)
        dk.copy_(torch.from_numpy(dim.columns[0].data))
        torch.cuda.synchronize()
        for _ in range(2):
            q.execute()
            assert sorted(q.result().text.splitlines()) == sorted(orc.execute(_plan(dim, fact)).text.splitlines())
    finally:
        q.close(); ddim.close(); dfact.close()


def test_component_bitmap_of_a_two_key_join_is_tested_in_front_of_the_compaction(gpu_ctx):
    """A join on TWO keys whose build side holds few values of its first key component (TPC-H Q5's supplier side: the suppliers of one
    region): the build sets one bit per component value, the probe pipeline - whose value for that component is a column of its own
    scan - tests the bit in stage 1, with the word fetched beside the tile, so rows that cannot match never enter the compaction
    queue.  A necessary condition only: the answer is the oracle's, also for probe values outside the build side's range and on a
    second execution (the bits are never cleared: the build side's columns are immutable)."""
    import numpy as np
    from resql_amd import plan as P
    T = P.TypeInit
    rng = np.random.default_rng(5)
    nb, npr = 4_000, 400_000
    dk = rng.choice(np.arange(1000, 21_000), nb, replace=False).astype(np.int32)            # 4 000 of 20 000 possible first components
    d = P.Table("d", [P.Column("dk", T.INT(), dk), P.Column("dn", T.INT(), (dk % 25).astype(np.int32)), P.Column("dv", T.BIGINT(), rng.integers(0, 100, nb).astype(np.int64))], nb)
    fk = rng.integers(0, 30_000, npr).astype(np.int32)                                     # ... also below and above the build side's range
    f = P.Table("f", [P.Column("fk", T.INT(), fk), P.Column("fn", T.INT(), rng.integers(0, 25, npr).astype(np.int32)),
                      P.Column("fx", T.BIGINT(), rng.integers(0, 1000, npr).astype(np.int64)), P.Column("fsel", T.BIGINT(), rng.integers(0, 100, npr).astype(np.int64))], npr)
    p = P.Plan([d, f])
    probe = p.selection(p.lt(p.attr("fsel"), p.constant("60", P.BIGINT)), p.scan("f"))
    node = p.hashjoin([p.eq(p.attr("dk"), p.attr("fk")), p.eq(p.attr("dn"), p.attr("fn"))], p.scan("d"), probe, single_match=False)
    s, c = p.sum(p.add(p.attr("fx"), p.attr("dv"))), p.count(p.star())
    node = p.aggregation([s, c], [p.attr("dn")], node)
    plan = p.set_root(p.materialize(p.projection([p.attr("dn"), p.as_("s", s), p.as_("c", c)], node)))
    want = orc.execute(plan)
    tabs = [gpu_ctx.table(d), gpu_ctx.table(f)]
    q = gpu_ctx.compile(plan, tabs)
    assert "component bitmap of ht0 tested in front of the probe" in q.explain
    assert "pf_ht0_c" in q.source and "ht0_c_bm" in q.source
    q.await_kernels()
    for _ in range(3):
        q.execute()
        got = q.result()
        assert got.n_rows == want.n_rows > 0 and sorted(got.text.splitlines()) == sorted(want.text.splitlines())
    q.close()
    for t in tabs:
        t.close()
