"""Late loads (codegen.cpp finishPipeline): behind a selective leading selection the other columns are loaded only by lanes
that hold a passing row.  Chosen from the column statistics; same bytes as the plain form and the oracle."""
import numpy as np
import pytest

from resql_amd import plan as P, tpch
from oracle import orc
import fuzzplans

T = P.TypeInit


def test_chosen_from_the_statistics(compile_ctx):
    li = tpch.lineitem_table(0.01, tpch.Q6_COLUMNS)
    q6 = compile_ctx.compile(tpch.q6_plan(li), [compile_ctx.table(li)])
    assert "late loads" in q6.explain and "lead_pred(" in q6.source
    li1 = tpch.lineitem_table(0.01, tpch.Q1_COLUMNS)
    q1 = compile_ctx.compile(tpch.q1_plan(li1), [compile_ctx.table(li1)])
    assert "late loads" not in q1.explain and "lead_pred(" not in q1.source        # 98 % of the rows pass l_shipdate <= 1998-09-02
    t = tpch.synthetic_table(50_000, 8)
    for sel, late in ((0.01, True), (0.1, True), (0.5, False)):
        q = compile_ctx.compile(tpch.synthetic_plan(t, int(sel * (1 << 31))), [compile_ctx.table(t)])
        assert ("late loads" in q.explain) == late
        if late:
            assert "~%.1f %%" % (sel * 100) in q.explain


def test_estimate_understands_both_operand_orders_and_connectives(compile_ctx):
    n = 10_000
    rng = np.random.default_rng(2)
    t = P.Table("t", [P.Column("x", T.BIGINT(), rng.integers(0, 1000, n).astype(np.int64)),
                      P.Column("y", T.BIGINT(), rng.integers(0, 100, n).astype(np.int64)),
                      P.Column("v", T.BIGINT(), rng.integers(0, 9, n).astype(np.int64))], n)
    def explain(pred_of):
        p = P.Plan([t])
        s = p.sum(p.attr("v"))
        node = p.aggregation([s], [], p.selection(pred_of(p), p.scan("t")))
        return compile_ctx.compile(p.set_root(p.materialize(p.projection([p.as_("s", s)], node))), [compile_ctx.table(t)]).explain
    c = lambda p, v, ty=P.BIGINT: p.constant(str(v), ty)
    assert "~5.0 %" in explain(lambda p: p.lt(p.attr("x"), c(p, 50)))
    assert "~5.0 %" in explain(lambda p: p.gt(c(p, 50), p.attr("x")))                       # constant on the left
    assert "~0.5 %" in explain(lambda p: p.and_(p.lt(p.attr("x"), c(p, 50)), p.lt(p.attr("y"), c(p, 10))))
    assert "late loads" not in explain(lambda p: p.or_(p.lt(p.attr("x"), c(p, 50)), p.ge(p.attr("y"), c(p, 10))))
    assert "late loads" not in explain(lambda p: p.lt(p.attr("x"), p.attr("v")))              # nothing the statistics can say


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["1", "2", "0"])
def test_late_loads_give_the_oracles_bytes(gpu_ctx, monkeypatch, mode):
    monkeypatch.setenv("RSQ_LATE_LOADS", mode)
    li = tpch.lineitem_table(0.05, tpch.Q6_COLUMNS)
    cases = [tpch.q6_plan(li)]
    for groups, sel in ((8, 0.01), (1024, 0.05), (1 << 17, 0.02), (1 << 17, 0.6)):
        cases.append(tpch.synthetic_plan(tpch.synthetic_table(300_001, groups), int(sel * (1 << 31))))
    for plan in cases:
        tabs = [gpu_ctx.table(t) for t in plan.tables]
        q = gpu_ctx.compile(plan, tabs)
        try:
            if mode == "0":
                assert "late loads" not in q.explain
            if mode == "2":
                assert "late loads" in q.explain
            q.execute()
            assert q.result().text == orc.execute(plan).text
        finally:
            q.close()
            for t in tabs:
                t.close()


@pytest.mark.gpu
def test_fuzz_plans_with_late_loads_forced(gpu_ctx, monkeypatch):
    from resql_amd import engine
    monkeypatch.setenv("RSQ_LATE_LOADS", "2")
    late = 0
    for seed in range(200, 280):
        plan, kind = fuzzplans.make(seed)
        try:
            want = orc.execute(plan)
        except orc.OracleError:
            continue
        tabs = [gpu_ctx.table(t) for t in plan.tables]
        try:
            q = gpu_ctx.compile(plan, tabs)
        except engine.EngineError as e:
            for t in tabs:
                t.close()
            if e.status == 3:
                continue
            raise
        try:
            late += "late loads" in q.explain
            q.execute()
            got = q.result()
            assert fuzzplans.same(kind, got.text, want.text), f"seed {seed} ({kind})"
        finally:
            q.close()
            for t in tabs:
                t.close()
    assert late >= 5
