"""SQL front end (resql_amd/csrc/sqlfront.cpp) against the unmodified reference's: its Lemon grammar and its planner, driven
by token streams (tests/golden/sql_reference.json, made by tests/golden/make_sql_golden.py; live where the reference harness
exists).  CPU only: parsing and planning need no GPU; results are checked through the oracle here and through the engine in
tests/test_gpu_sql.py."""
import json
import os

import pytest

import sqlcases
import sqlfuzz
from resql_amd import engine, tpch_full
from oracle import orc

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "sql_reference.json")) as f:
    GOLD = json.load(f)


def my_parse(ctx, sql: str) -> str:
    try:
        return ctx.sql_describe(sql, 1)
    except engine.EngineError as e:
        if "FLOAT constants" in str(e):
            return "FLOAT\n"
        assert "Syntax error." in str(e), str(e)
        return "SYNTAX ERROR\n"


def same_parse(mine: str, ref: str) -> bool:
    # the reference leaves some broken statements through error_msg() -> exit (no answer at all): any refusal matches
    if ref == "REFUSED\n":
        return mine in ("SYNTAX ERROR\n", "FLOAT\n")
    if mine == "FLOAT\n":       # parsed by the reference (as a FLOAT constant it has no code generation for), refused here
        return "FLOAT" in ref
    return mine == ref


@pytest.fixture(scope="module")
def database(compile_ctx):
    db = tpch_full.database(GOLD["sf"])
    host = [db[k] for k in GOLD["tables"]]
    tabs = [compile_ctx.table(t) for t in host]
    yield host, tabs
    for t in tabs:
        t.close()


def test_tokenizer_rules(compile_ctx):
    """lexer.y by hand: longest match, earlier rule on a tie, multi-word keywords with exactly one space, the three number
    forms, both quote styles with escapes, comments need their newline, unknown characters end the statement"""
    def toks(s):
        return [l.split(" ", 1)[0] for l in compile_ctx.sql_describe(s, 0).splitlines()]
    assert toks("select summ, sum(x)") == ["SELECT_TK", "IDENTIFIER", "COMMA", "SUM_TK", "LPAREN", "IDENTIFIER", "RPAREN"]
    assert toks("group by") == ["GROUPBY"] and toks("group  by") == ["IDENTIFIER", "IDENTIFIER"]
    assert toks("1 1.5 .5 5. 1e5 1.e5 .5e-3") == ["INTEGER_CONSTANT", "DECIMAL_CONSTANT", "DECIMAL_CONSTANT", "DECIMAL_CONSTANT",
                                                 "FLOAT_CONSTANT", "FLOAT_CONSTANT", "FLOAT_CONSTANT"]
    assert toks("12abc") == ["INTEGER_CONSTANT", "IDENTIFIER"]
    assert toks("a<>b<=c<d") == ["IDENTIFIER", "NEQ_TK", "IDENTIFIER", "LE_TK", "IDENTIFIER", "LT_TK", "IDENTIFIER"]
    assert toks("a::int") == ["IDENTIFIER", "TYPECAST_TK", "INT_TK"] and toks("integer") == ["IDENTIFIER"]
    assert toks("a -- c\nb") == ["IDENTIFIER", "IDENTIFIER"] and toks("a --b") == ["IDENTIFIER", "MINUS_TK", "MINUS_TK", "IDENTIFIER"]
    assert toks("'it\\'s' \"x\"") == ["STRING_CONSTANT", "STRING_CONSTANT"]
    assert toks("a;")[-1] == "ERROR" and toks("A")[-1] == "ERROR" and toks("'open")[-1] == "ERROR" and toks("a\rb")[-1] == "ERROR"


def test_parses_match_the_reference_grammar(compile_ctx):
    """hand-written corner cases, the eight TPC-H statements and 400 random statements (a third of them damaged): same token
    stream as when the golden was made, same Query as the reference's Lemon parser builds from it — or refused by both"""
    assert len(GOLD["parse"]) > 400
    bad = []
    for g in GOLD["parse"]:
        if compile_ctx.sql_describe(g["sql"], 0) != g["tokens"]:
            bad.append(("tokens", g["sql"]))
        elif not same_parse(my_parse(compile_ctx, g["sql"]), g["dump"]):
            bad.append(("parse", g["sql"], my_parse(compile_ctx, g["sql"]), g["dump"]))
    assert not bad, bad[:3]
    kinds = {g["dump"].split("\n", 1)[0].split(" ")[0] for g in GOLD["parse"]}
    assert {"SELECT", "CREATE_TABLE", "BULK_INSERT", "SYNTAX"} <= kinds


@pytest.mark.skipif(not orc.have_reference(), reason="needs the compiled reference (build container)")
def test_parses_match_the_reference_grammar_live(compile_ctx):
    bad = []
    for seed in range(1000, 1250):
        s = sqlfuzz.statement(seed)
        try:
            ref = orc.reference_parse(compile_ctx.sql_describe(s, 0))
        except orc.OracleError:
            ref = ""
        if not same_parse(my_parse(compile_ctx, s), ref if ref.strip() else "REFUSED\n"):
            bad.append((seed, s))
    assert not bad, bad[:3]


def test_plans_match_the_reference_planner(compile_ctx, database):
    """buildQuery (planner.h:409-497) on the eight TPC-H statements and the planner cases: same operator tree, same
    expressions, same build sides, join order and single-match flags; a statement the reference refuses, or plans with a
    nested-loops join, is refused here"""
    host, tabs = database
    assert len(GOLD["plans"]) >= 30
    for g in GOLD["plans"]:
        ref = g["dump"]
        if ref.startswith("REFUSED") or "NESTEDLOOPSJOIN" in ref:
            with pytest.raises(engine.EngineError):
                compile_ctx.sql_plan_text(g["sql"], tabs)
            continue
        assert compile_ctx.sql_plan_text(g["sql"], tabs) == ref, g["sql"]


def test_planner_error_messages(compile_ctx, database):
    host, tabs = database
    with pytest.raises(engine.EngineError, match="Table nosuchtable does not exist."):
        compile_ctx.sql_plan_text("select x from nosuchtable", tabs)
    with pytest.raises(engine.EngineError, match="Syntax error."):
        compile_ctx.sql_plan_text("select from nation", tabs)
    with pytest.raises(engine.EngineError, match="nested-loops"):
        compile_ctx.sql_plan_text("select c_name from customer, nation", tabs)
    with pytest.raises(engine.EngineError, match="not a select"):
        compile_ctx.sql_plan_text("create table t ( a int )", tabs)


RESULT_NAMES = sorted(GOLD["results"])


@pytest.mark.parametrize("name", RESULT_NAMES)
def test_results_through_the_oracle_match_the_reference(compile_ctx, database, name):
    """the reference executed the statement end to end (its parser, planner and JIT); the oracle executes the plan THIS
    front end made from the same text: byte-identical relations, or refused by both"""
    host, tabs = database
    g = GOLD["results"][name]
    if "refused" in g or "NESTEDLOOPSJOIN" in next(p["dump"] for p in GOLD["plans"] if p["sql"] == g["sql"]):
        with pytest.raises((engine.EngineError, orc.OracleError)):
            orc.execute(compile_ctx.sql_plan(g["sql"], tabs, host))
        return
    res = orc.execute(compile_ctx.sql_plan(g["sql"], tabs, host))
    if g.get("reference_undefined"):
        # the reference read past its hash table here (qlib/hash.h:441-451) and emitted a group twice; the oracle counts it
        assert res.ref_oob_probes > 0 or res.ref_narrow_casts > 0
        assert res.text == g["text"]         # the golden keeps the oracle's (intended) answer for these
        return
    assert res.text == g["text"]


def test_the_eight_queries_compile_for_gfx950(compile_ctx, database):
    host, tabs = database
    for name, sql in tpch_full.QUERIES.items():
        q = compile_ctx.sql_compile(sql, tabs)
        assert "pipeline 0" in q.explain, name
        q.close()


def test_database_statement_loop_without_gpu(compile_ctx):
    """CREATE TABLE through the C ABI's statement loop (rsq_db_*; execute.h:508-545, 263-281); running a SELECT needs a device"""
    db = engine.Database(compile_ctx)
    try:
        assert db.execute("create table nation ( n_nationkey int, n_name char(25), n_regionkey int, n_comment varchar(152) )") is None
        with pytest.raises(engine.EngineError, match="Table nation already exists."):
            db.execute("create table nation ( x int )")
        with pytest.raises(engine.EngineError, match="Table region does not exist."):
            db.execute("bulk insert region from 'r.tbl'")
        with pytest.raises(engine.EngineError, match="Syntax error."):
            db.execute("create table t ( )")
        with pytest.raises(engine.EngineError, match="no device|compile-only"):
            db.execute("select n_name from nation")          # parsed, planned and compiled; executing needs a GPU
    finally:
        db.close()


def test_reference_int16_cast_switch_reproduces_the_jit(compile_ctx, database, monkeypatch):
    """the reference's asmjit back end sign-extends INT -> BIGINT casts from 16 bits; with
    RSQ_REFERENCE_INT16_CAST=1 the oracle (and the engine, tests/test_gpu_sql.py) give the JIT's own answer"""
    host, tabs = database
    cases = [g for g in GOLD["results"].values() if "reference_text" in g]
    assert cases
    monkeypatch.setenv("RSQ_REFERENCE_INT16_CAST", "1")
    for g in cases:
        res = orc.execute(compile_ctx.sql_plan(g["sql"], tabs, host))
        assert res.text == g["reference_text"] and res.text != g["text"]


def _sqlgen_gold():
    with open(os.path.join(HERE, "golden", "sqlgen_reference.json")) as f:
        return json.load(f)


def test_random_valid_statements_match_the_reference(compile_ctx, database):
    """160 random valid statements (tests/sqlgen.py: foreign-key join paths, predicates, group-bys with every aggregate,
    order by / limit): the oracle's answer on the plan THIS front end makes == the digest of the reference's own answer"""
    import hashlib
    import sqlgen
    host, tabs = database
    gold = _sqlgen_gold()
    assert gold["sf"] == GOLD["sf"]
    compared = 0
    for seed, g in gold["seeds"].items():
        s = sqlgen.statement(int(seed))
        if "refused" in g:
            with pytest.raises((engine.EngineError, orc.OracleError)):
                orc.execute(compile_ctx.sql_plan(s, tabs, host))
            continue
        res = orc.execute(compile_ctx.sql_plan(s, tabs, host))
        if "undefined" in g:
            assert res.ref_oob_probes or res.ref_narrow_casts, s
            continue
        assert hashlib.sha1(res.text.encode("latin1")).hexdigest() == g["sha1"], s
        compared += 1
    assert compared >= 140
