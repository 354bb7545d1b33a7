"""String join keys of different kinds and declared lengths.  The reference hashes each side with its own type's function
(hashChar pads with spaces to the DECLARED length, hashVarchar sums the characters: qlib/hash.h:116-147), ht_get returns only
entries with an equal hash (:427-477), and the keys are compared the probe side's way (hashjoin.h:142/191).  So CHAR(n) against
VARCHAR matches a value that is exactly n characters long, CHAR(a) against CHAR(b) never matches, VARCHAR against VARCHAR
matches equal strings.  Engine == oracle (which restates exactly that) for every combination."""
import itertools

import numpy as np
import pytest

from resql_amd import plan as P
from oracle import orc

pytestmark = pytest.mark.gpu
T = P.TypeInit


def _strs(vals, width):
    a = np.zeros((len(vals), width), dtype=np.uint8)
    for i, v in enumerate(vals):
        b = v.encode()[:width]
        a[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
    return a


def _values(rng, n, width):
    out = []
    for _ in range(n):
        core = "".join(rng.choice(list("ab"), size=int(rng.integers(0, width + 1))))
        out.append((core + " " * int(rng.integers(0, 3)))[:width])
    return out


KINDS = [("CHAR", 4), ("VARCHAR", 4), ("CHAR", 6), ("VARCHAR", 6), ("CHAR", 11), ("VARCHAR", 11)]


@pytest.mark.parametrize("build,probe", [(b, p) for b, p in itertools.product(KINDS, KINDS) if b != p])
def test_join_on_strings_of_different_kinds(gpu_ctx, build, probe):
    rng = np.random.default_rng(hash((build, probe)) % (1 << 31))
    bt, pt = getattr(T, build[0])(build[1]), getattr(T, probe[0])(probe[1])
    bvals = sorted(set(_values(rng, 60, build[1])))
    pvals = _values(rng, 400, probe[1])
    dim = P.Table("dim", [P.Column("dk", bt, _strs(bvals, build[1])), P.Column("dv", T.BIGINT(), np.arange(len(bvals), dtype=np.int64))], len(bvals))
    fact = P.Table("t", [P.Column("k", pt, _strs(pvals, probe[1])), P.Column("v", T.BIGINT(), np.arange(len(pvals), dtype=np.int64))], len(pvals))
    p = P.Plan([dim, fact])
    j = p.hashjoin([p.eq(p.attr("dk"), p.attr("k"))], p.scan("dim"), p.scan("t"), single_match=False)
    plan = p.set_root(p.materialize(p.projection([p.attr("dv"), p.attr("v")], j)))
    want = orc.execute(plan)
    got = gpu_ctx.run(plan)
    assert sorted(got.text.splitlines()) == sorted(want.text.splitlines())
    if build[0] == "VARCHAR" and probe[0] == "VARCHAR":
        assert want.n_rows > 20          # equal strings match whatever the declared lengths
    if build[0] == "CHAR" and probe[0] == "CHAR":
        assert want.n_rows == 0          # the pad spaces never hash alike


# ---- string constants as keys (refused until round 2) -----------------------------------------------------------------------

def _same(gpu_ctx, plan):
    want, got = orc.execute(plan), gpu_ctx.run(plan)
    assert sorted(got.text.splitlines()) == sorted(want.text.splitlines())
    return want


def _fact_and_dim():
    rng = np.random.default_rng(1)
    n = 5000
    t = P.Table("t", [P.Column("k", T.VARCHAR(4), _strs([["ab", "abcd", "x", "ab "][i % 4] for i in range(n)], 4)),
                      P.Column("g", T.INT(), rng.integers(0, 5, n).astype(np.int32)),
                      P.Column("v", T.BIGINT(), rng.integers(0, 100, n).astype(np.int64))], n)
    dim = P.Table("dim", [P.Column("dk", T.VARCHAR(4), _strs(["ab", "zz", "abcd"], 4)), P.Column("dv", T.BIGINT(), np.arange(3, dtype=np.int64))], 3)
    return t, dim


@pytest.mark.parametrize("kind", [P.VARCHAR, P.CHAR])
def test_string_constant_as_group_key(gpu_ctx, kind):
    t, _ = _fact_and_dim()
    for with_column in (True, False):
        p = P.Plan([t])
        s, c = p.sum(p.attr("v")), p.constant("abc", kind)
        keys = [c, p.attr("g")] if with_column else [c]
        want = _same(gpu_ctx, p.set_root(p.materialize(p.projection(keys + [p.as_("s", s)], p.aggregation([s], keys, p.scan("t"))))))
        assert want.n_rows == (5 if with_column else 1)


@pytest.mark.parametrize("kind", [P.VARCHAR, P.CHAR])
@pytest.mark.parametrize("side", ["probe", "build"])
def test_string_constant_as_join_key(gpu_ctx, kind, side):
    """build side: every build row carries the same key, so every matching probe row has three matches - with a hash
    aggregation (behind a wave compaction) above the join, which used to keep only the last match of a row"""
    t, dim = _fact_and_dim()
    p = P.Plan([dim, t])
    c = p.constant("ab", kind)
    eq = p.eq(p.attr("dk"), c) if side == "probe" else p.eq(c, p.attr("k"))
    j = p.hashjoin([eq], p.scan("dim"), p.scan("t"), single_match=False)
    s, cnt = p.sum(p.attr("v")), p.count(p.star())
    plan = p.set_root(p.materialize(p.projection([p.attr("dv"), p.as_("s", s), p.as_("c", cnt)], p.aggregation([s, cnt], [p.attr("dv")], j))))
    want = _same(gpu_ctx, plan)
    assert want.n_rows == (1 if side == "probe" else 3)


def test_every_match_of_a_row_reaches_an_aggregation_behind_the_join(gpu_ctx):
    """duplicate build keys (string keys: no key bitmap, the table walk stays in the row function) and a selective hash
    aggregation above: three matches per probe row, all three counted"""
    rng = np.random.default_rng(3)
    n, m = 40_000, 300
    names = [f"k{i:03d}" for i in range(100)]
    dim = P.Table("dim", [P.Column("dk", T.VARCHAR(6), _strs([names[i % 100] for i in range(m)], 6)),
                          P.Column("dv", T.BIGINT(), np.arange(m, dtype=np.int64))], m)
    fact = P.Table("t", [P.Column("k", T.VARCHAR(6), _strs([names[int(i)] for i in rng.integers(0, 120, n) % 110 % 100], 6)),
                         P.Column("v", T.BIGINT(), rng.integers(0, 50, n).astype(np.int64))], n)
    p = P.Plan([dim, fact])
    j = p.hashjoin([p.eq(p.attr("dk"), p.attr("k"))], p.scan("dim"), p.scan("t"), single_match=False)
    s, cnt = p.sum(p.attr("v")), p.count(p.star())
    key = p.add(p.attr("dv"), p.constant("1", P.BIGINT))                 # a computed key: generic hash aggregation
    plan = p.set_root(p.materialize(p.projection([p.as_("key", key), p.as_("s", s), p.as_("c", cnt)], p.aggregation([s, cnt], [key], j))))
    want = _same(gpu_ctx, plan)
    assert want.n_rows == m
