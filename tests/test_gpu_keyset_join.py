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
