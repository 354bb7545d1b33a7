"""LIKE: the oracle against the reference's answers (tests/golden/like_reference.json), the engine's typing / refusal
rules on CPU, and (marked gpu) the device implementation against both."""
import json
import os

import pytest

from resql_amd import engine, plan as P
from oracle import orc

import likecases

with open(os.path.join(os.path.dirname(__file__), "golden", "like_reference.json")) as f:
    GOLDEN = json.load(f)


def test_golden_matches_the_case_file():
    assert GOLDEN["strings"] == likecases.STRINGS and GOLDEN["patterns"] == likecases.PATTERNS


@pytest.mark.parametrize("kind", ["char", "varchar"])
def test_oracle_like_matches_reference(kind):
    assert orc.execute(likecases.plan(kind)).text == GOLDEN["flags_" + kind]
    for pat, want in GOLDEN["select_" + kind].items():
        assert orc.execute(likecases.select_plan(kind, pat)).text == want


def test_reference_quirks_are_kept():
    """prefix and suffix may overlap in the string: 'ab' LIKE 'abab' holds in the reference"""
    rows = {int(l.split("|")[0]): l.split("|")[1:-1] for l in GOLDEN["flags_varchar"].splitlines()[1:]}
    s, p = likecases.STRINGS.index("ab"), likecases.PATTERNS.index("abab")
    assert rows[s][p] == "1"
    assert rows[likecases.STRINGS.index("abc")][likecases.PATTERNS.index("%abc%")] == "1"
    assert rows[likecases.STRINGS.index("abc")][likecases.PATTERNS.index("a_b")] == "0"


def _char1_plan():
    t = P.Table("s", [P.Column("c", P.TypeInit.CHAR(1), None), P.Column("v", P.TypeInit.VARCHAR(4), None)], 0)
    p = P.Plan([t])
    sel = p.selection(p.like(p.attr("c"), p.constant("a%", P.VARCHAR)), p.scan("s"))
    return p.set_root(p.materialize(sel), request_all=True)


def test_like_on_char1_is_refused_by_oracle_and_engine(compile_ctx):
    with pytest.raises(orc.OracleError):
        orc.execute(_char1_plan())
    plan = _char1_plan()
    tabs = [compile_ctx.table(t) for t in plan.tables]
    with pytest.raises(engine.EngineError) as e:
        compile_ctx.compile(plan, tabs)
    assert e.value.status == 2


def test_like_needs_string_operands(compile_ctx):
    t = likecases.table("char")
    p = P.Plan([t])
    p.set_root(p.materialize(p.selection(p.like(p.attr("id"), p.constant("1%", P.VARCHAR)), p.scan("s"))), request_all=True)
    with pytest.raises(orc.OracleError):
        orc.execute(p)
    with pytest.raises(engine.EngineError):
        compile_ctx.compile(p, [compile_ctx.table(t)])


def test_like_pipelines_compile_for_gfx950(compile_ctx):
    for kind in ("char", "varchar"):
        plan = likecases.plan(kind, likecases.PATTERNS[:6])
        q = compile_ctx.compile(plan, [compile_ctx.table(t) for t in plan.tables])
        assert "rsq::like(" in q.source
        q.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["char", "varchar"])
def test_gpu_like_matches_reference_and_oracle(gpu_ctx, kind):
    got = gpu_ctx.run(likecases.plan(kind)).text
    assert got == GOLDEN["flags_" + kind]
    assert got == orc.execute(likecases.plan(kind)).text
    for pat, want in GOLDEN["select_" + kind].items():
        assert gpu_ctx.run(likecases.select_plan(kind, pat)).text == want
