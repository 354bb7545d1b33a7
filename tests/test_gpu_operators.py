"""GPU parity for the reference's own operator tests (test/test_operators.h) and for the golden outputs of the
unmodified reference (tests/golden/ref_*.tbl), through the C ABI."""
from collections import Counter

import numpy as np
import pytest

from resql_amd import engine, plan as P, tpch
from oracle import orc

import goldens
import refcases

pytestmark = pytest.mark.gpu

# plan shapes this engine version does not lower: they must be refused loudly (RSQ_ERR_UNSUPPORTED), never routed
# to a CPU path.  (Empty since the device-side materialisation and the generic hash aggregation exist.)
NOT_YET = set()


@pytest.mark.parametrize("case", sorted(refcases.CASES))
def test_reference_operator_cases(gpu_ctx, case):
    plan = refcases.CASES[case]()
    if case in NOT_YET:
        with pytest.raises(engine.EngineError) as e:
            gpu_ctx.run(plan)
        assert e.value.status == 3
        return
    got = gpu_ctx.run(plan)
    refcases.check_against_literals(case, got)          # the reference's expected table
    want = orc.execute(plan)
    if case == "hashjoin":                               # matches of one probe row come in hash-table order: multiset
        assert Counter(got.rows()) == Counter(want.rows())
    else:
        assert got.text == want.text                     # byte for byte, incl. emission / scan order


@pytest.mark.parametrize("name", goldens.NAMES)
def test_matches_reference_golden(gpu_ctx, name):
    got = gpu_ctx.run(goldens.golden_plan(name))
    assert got.text == goldens.golden_text(name)


def test_selection_materialises_in_scan_order_with_strings(gpu_ctx):
    """selection over a table with a CHAR(10) column, output in scan order (materialize.h appends sequentially)"""
    cu = tpch.customer_table(0.05)
    p = P.Plan([cu])
    cond = p.eq(p.attr("c_mktsegment"), p.constant("MACHINERY", P.VARCHAR))
    p.set_root(p.materialize(p.selection(cond, p.scan("customer"))), request_all=True)
    got, want = gpu_ctx.run(p), orc.execute(p)
    assert got.n_rows == want.n_rows > 1000
    assert got.tuples == want.tuples


@pytest.mark.parametrize("limit", [0, 1, 7, 1000])
def test_limit_on_materialize(gpu_ctx, limit):
    t = tpch.synthetic_table(50_000, 64)
    p = P.Plan([t])
    sel = p.selection(p.lt(p.attr("a"), p.constant(str(1 << 29), P.BIGINT)), p.scan("t"))
    p.set_root(p.materialize(sel), limit=limit, request_all=True)
    got, want = gpu_ctx.run(p), orc.execute(p)
    assert got.n_rows == want.n_rows == max(limit, 1)
    assert got.tuples == want.tuples


def test_order_by_over_selection(gpu_ctx):
    t = tpch.synthetic_table(20_000, 50)
    p = P.Plan([t])
    sel = p.selection(p.lt(p.attr("a"), p.constant(str(1 << 30), P.BIGINT)), p.scan("t"))
    p.set_root(p.orderby([p.attr("b"), p.desc(p.attr("c"))], sel), limit=500, request_all=True)
    got, want = gpu_ctx.run(p), orc.execute(p)
    assert got.text == want.text                          # same quicksort on the same input order => same ties


def test_join_output_multiset(gpu_ctx):
    sf = 0.02
    cu, od = tpch.customer_table(sf), tpch.orders_table(sf)
    p = P.Plan([cu, od])
    sel_c = p.selection(p.eq(p.attr("c_mktsegment"), p.constant("BUILDING", P.VARCHAR)), p.scan("customer"))
    hj = p.hashjoin([p.eq(p.attr("c_custkey"), p.attr("o_custkey"))], sel_c, p.scan("orders"))
    proj = p.projection([p.attr("o_orderkey"), p.attr("c_custkey"), p.attr("o_orderdate")], hj)
    p.set_root(p.materialize(proj))
    got, want = gpu_ctx.run(p), orc.execute(p)
    assert got.n_rows == want.n_rows > 1000
    assert got.tuples == want.tuples          # c_custkey is unique: one match per probe row, so even the order agrees


@pytest.mark.parametrize("n,groups", [(20_000, 300), (400_000, 70_000)])
def test_hash_aggregation_on_computed_keys(gpu_ctx, n, groups):
    """group by a computed expression (test_operators.h:760-831 at scale): generic hash aggregation, incl. table
    re-sizing when the reference's estimate is too small"""
    t = tpch.synthetic_table(n, groups)
    p = P.Plan([t])
    key = p.add(p.attr("b"), p.mul(p.attr("c"), p.constant("0", P.BIGINT)))
    agg = p.aggregation([p.sum(p.attr("d")), p.count(p.star()), p.min(p.attr("c")), p.max(p.attr("c"))],
                        [key], p.scan("t"))
    p.set_root(p.materialize(agg), request_all=True)
    got, want = gpu_ctx.run(p), orc.execute(p)
    assert got.n_rows == want.n_rows
    assert got.text == want.text


def test_string_join_keys_of_different_declared_lengths(gpu_ctx):
    """VARCHAR(a) = VARCHAR(b): equal strings match (hashVarchar stops at the NUL); CHAR(a) = CHAR(b), a != b: never a match
    in the reference (hashChar pads with spaces to the declared length, qlib/hash.h:131-147) — engine == oracle == reference"""
    import numpy as np
    from resql_amd import plan as P
    for T in (P.TypeInit.VARCHAR, P.TypeInit.CHAR):
        a = P.Table("a", [P.Column("ak", T(6), np.array([b"x1", b"x2 ", b"longer", b"q"], dtype="S6")),
                          P.Column("av", P.TypeInit.BIGINT(), np.arange(4, dtype=np.int64))], 4)
        b = P.Table("b", [P.Column("bk", T(12), np.array([b"x1  ", b"x2", b"longer", b"longerstill", b"x2 ", b"q"], dtype="S12")),
                          P.Column("bv", P.TypeInit.BIGINT(), np.arange(6, dtype=np.int64) * 10)], 6)
        p = P.Plan([a, b])
        j = p.hashjoin([p.eq(p.attr("ak"), p.attr("bk"))], p.scan("a"), p.scan("b"))
        plan = p.set_root(p.materialize(p.projection([p.attr("av"), p.attr("bv")], j)))
        want = orc.execute(plan)
        if orc.have_reference():
            assert orc.run_reference(plan)[0] == want.text
        got = gpu_ctx.run(plan)
        assert got.text == want.text
        assert (want.n_rows > 0) == (T is P.TypeInit.VARCHAR)
