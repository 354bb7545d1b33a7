"""Differential test of join shapes the seeded plan generator (tests/fuzzplans.py, pinned against the reference) does not reach:
keys without a key bitmap (wide integer ranges, computed keys, strings of several kinds), duplicate build keys under joins probed
for all matches, and what sits above the join (a selection, a dense or a hash aggregation, a plain materialisation).
Engine == oracle as multisets of rows."""
import random

import numpy as np
import pytest

from resql_amd import engine, plan as P
from oracle import orc

pytestmark = pytest.mark.gpu
T = P.TypeInit


def _strs(vals, width):
    a = np.zeros((len(vals), width), dtype=np.uint8)
    for i, v in enumerate(vals):
        b = v.encode()[:width]
        a[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
    return a


def make(seed, with_spec=False):
    """-> (plan, description); with_spec=True: (plan, description, spec) where `spec` holds every decision taken, for an
    independent evaluation of the same query (tests/test_join_shapes_pandas.py)"""
    r = random.Random(seed)
    spec = {}
    rng = np.random.default_rng(seed)
    n, m = r.choice([300, 5000, 40000]), r.choice([7, 200, 3000])
    key_kind = r.choice(["narrow", "wide", "computed", "varchar", "char", "mixed"])
    unique = r.random() < 0.4
    domain = max(4, int(m * r.choice([0.3, 1.0, 4.0])))
    if unique:
        base = rng.permutation(max(domain, m))[:m]
    else:
        base = rng.integers(0, domain, m)
    fbase = rng.integers(0, max(domain, m) + 3, n)
    scale = (1 << 33) + 12345 if key_kind == "wide" else 1
    names = lambda a: [f"s{int(x):05d}" + (" " if int(x) % 5 == 0 else "") for x in a]
    dcols = [P.Column("dk", T.BIGINT(), (base * scale).astype(np.int64)), P.Column("dg", T.INT(), rng.integers(0, 9, m).astype(np.int32)),
             P.Column("dx", T.BIGINT(), rng.integers(0, 1000, m).astype(np.int64))]
    fcols = [P.Column("fk", T.BIGINT(), (fbase * scale).astype(np.int64)), P.Column("fg", T.INT(), rng.integers(0, 6, n).astype(np.int32)),
             P.Column("fx", T.BIGINT(), rng.integers(0, 1000, n).astype(np.int64))]
    if key_kind in ("varchar", "char", "mixed"):
        dt = T.VARCHAR(8) if key_kind != "char" else T.CHAR(8)
        ft = T.VARCHAR(10) if key_kind == "varchar" else T.CHAR(8) if key_kind == "char" else T.CHAR(7)
        dcols.append(P.Column("ds", dt, _strs(names(base), 8)))
        fcols.append(P.Column("fs", ft, _strs(names(fbase), 10 if key_kind == "varchar" else 8 if key_kind == "char" else 7)))
    dim, fact = P.Table("d", dcols, m), P.Table("f", fcols, n)
    p = P.Plan([dim, fact])
    if key_kind in ("narrow", "wide"):
        eq = p.eq(p.attr("dk"), p.attr("fk"))
    elif key_kind == "computed":
        eq = p.eq(p.add(p.attr("dk"), p.constant("1", P.BIGINT)), p.add(p.attr("fk"), p.constant("1", P.BIGINT)))
    else:
        eq = p.eq(p.attr("ds"), p.attr("fs"))
    left = p.scan("d")
    if r.random() < 0.3:
        spec["dx_below"] = r.choice([100, 500, 900])
        left = p.selection(p.lt(p.attr("dx"), p.constant(str(spec["dx_below"]), P.BIGINT)), left)
    right = p.scan("f")
    if r.random() < 0.4:
        spec["fx_from"] = r.choice([50, 500, 950])
        right = p.selection(p.ge(p.attr("fx"), p.constant(str(spec["fx_from"]), P.BIGINT)), right)
    node = p.hashjoin([eq], left, right, single_match=unique and r.random() < 0.5)
    if r.random() < 0.3:
        spec["sum_below"] = r.choice([400, 1000, 1600])
        node = p.selection(p.lt(p.add(p.attr("dx"), p.attr("fx")), p.constant(str(spec["sum_below"]), P.BIGINT)), node)
    top = r.choice(["dense", "hash", "rows", "global"])
    spec["key_kind"], spec["top"] = key_kind, top
    s, c, mx = p.sum(p.attr("fx")), p.count(p.star()), p.max(p.attr("dx"))
    if top == "dense":
        keys = [p.attr("dg"), p.attr("fg")] if r.random() < 0.5 else [p.attr("fg")]
        spec["dense_keys"] = len(keys)
        node = p.projection(keys + [p.as_("s", s), p.as_("c", c), p.as_("mx", mx)], p.aggregation([s, c, mx], keys, node))
    elif top == "hash":
        key = p.add(p.attr("dx"), p.attr("fg"))
        node = p.projection([p.as_("k", key), p.as_("s", s), p.as_("c", c)], p.aggregation([s, c], [key], node))
    elif top == "global":
        node = p.projection([p.as_("s", s), p.as_("c", c), p.as_("mx", mx)], p.aggregation([s, c, mx], [], node))
    else:
        node = p.projection([p.attr("dx"), p.attr("fx"), p.attr("fg")], node)
    plan, what = p.set_root(p.materialize(node)), f"{key_kind}/{'unique' if unique else 'dups'}/{top}"
    return (plan, what, spec) if with_spec else (plan, what)


@pytest.mark.parametrize("block", range(0, 120, 20))
def test_join_shapes_engine_matches_oracle(gpu_ctx, block):
    failures = []
    for seed in range(block, block + 20):
        plan, what = make(7000 + seed)
        try:
            want = orc.execute(plan)
        except orc.OracleError:
            with pytest.raises(engine.EngineError):
                gpu_ctx.run(plan)
            continue
        try:
            got = gpu_ctx.run(plan)
        except engine.EngineError as e:
            failures.append((seed, what, "engine refused: " + str(e)[:80]))
            continue
        if sorted(got.text.splitlines()) != sorted(want.text.splitlines()):
            failures.append((seed, what, f"{got.n_rows} rows vs {want.n_rows}"))
    assert not failures, failures
