"""GPU parity for SQL text: statement -> the engine's front end (tokens, grammar, planner of the reference) -> HIP pipelines,
against (a) the answers of the UNMODIFIED reference executing the same statements end to end (tests/golden/sql_reference.json,
SF 0.01) and (b) the CPU oracle on a larger database."""
import json
import os

import numpy as np
import pytest

from resql_amd import engine, tpch_full
from oracle import orc

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "sql_reference.json")) as f:
    GOLD = json.load(f)


@pytest.fixture(scope="module")
def small_db(gpu_ctx):
    db = tpch_full.database(GOLD["sf"])
    host = [db[k] for k in GOLD["tables"]]
    tabs = [gpu_ctx.table(t) for t in host]
    yield host, tabs
    for t in tabs:
        t.close()


@pytest.mark.parametrize("name", sorted(GOLD["results"]))
def test_sql_results_match_the_reference(gpu_ctx, small_db, name):
    host, tabs = small_db
    g = GOLD["results"][name]
    plan_dump = next(p["dump"] for p in GOLD["plans"] if p["sql"] == g["sql"])
    if "refused" in g or "NESTEDLOOPSJOIN" in plan_dump:
        with pytest.raises(engine.EngineError):
            q = gpu_ctx.sql_compile(g["sql"], tabs)
            q.execute()
        return
    q = gpu_ctx.sql_compile(g["sql"], tabs)
    try:
        q.execute()
        got = q.result()
    finally:
        q.close()
    # "reference_undefined" cases (a group emitted twice after an out-of-bounds probe, 16-bit INT -> BIGINT casts of the
    # asmjit back end; see test_sql_frontend.py) keep the oracle's answer in the golden
    assert got.text == g["text"]


@pytest.mark.parametrize("name", sorted(tpch_full.QUERIES))
def test_the_eight_queries_on_a_larger_database(gpu_ctx, name):
    """SF 0.05 (300 K lineitem rows): join tables, group counts and string group keys well beyond a wave / a workgroup"""
    db = tpch_full.database(0.05)
    host = [db[k] for k in sorted(db)]
    tabs = [gpu_ctx.table(t) for t in host]
    try:
        sql = tpch_full.QUERIES[name]
        want = orc.execute(gpu_ctx.sql_plan(sql, tabs, host))
        q = gpu_ctx.sql_compile(sql, tabs)
        q.execute()
        got = q.result()
        q.close()
        assert got.text == want.text
        assert got.tuples == want.tuples
        if name == "q19":
            assert got.n_rows > 0
    finally:
        for t in tabs:
            t.close()


def test_statement_loop_create_load_select(gpu_ctx, tmp_path):
    """executeStatement's three statement kinds (execute.h:508-545) through engine.Database: CREATE TABLE, BULK INSERT of a
    '|' separated file, SELECT — the flow of the reference's tpch/create.sql + load_*.sql + queries"""
    nat = tpch_full.nation()
    reg = tpch_full.region()
    with open(tmp_path / "nation.tbl", "w") as f:
        for i in range(nat.n_rows):
            f.write(f"{nat.col('n_nationkey').data[i]}|{nat.col('n_name').data[i].decode()}|{nat.col('n_regionkey').data[i]}|about {i}|\n")
    with open(tmp_path / "region.tbl", "w") as f:
        for i in range(reg.n_rows):
            f.write(f"{reg.col('r_regionkey').data[i]}|{reg.col('r_name').data[i].decode()}|none|\n")
    db = engine.Database(gpu_ctx)
    try:
        db.execute_script("""
            create table nation ( n_nationkey int, n_name char(25), n_regionkey int, n_comment varchar(152) );
            create table region ( r_regionkey int, r_name char(25), r_comment varchar(152) );""")
        assert db.execute("select n_name from nation where n_nationkey < 3").n_rows == 0        # a created, still empty relation
        db.execute_script(f"""
            bulk insert nation from "{tmp_path}/nation.tbl" with ( fieldterminator="|" );
            bulk insert region from "{tmp_path}/region.tbl" with ( fieldterminator="|" );
        """)
        res = db.execute("select r_name, count(*) as n from nation, region where n_regionkey = r_regionkey group by r_name order by r_name")
        assert res.text.splitlines()[1:] == [f"{r.decode():<25}|5|" for r in sorted(tpch_full.REGIONS)]
        res = db.execute("select n_name, n_comment from nation where n_nationkey in (3, 7) order by n_name desc")
        assert res.text.splitlines()[1:] == ["GERMANY                  |about 7|", "CANADA                   |about 3|"]
        with pytest.raises(engine.EngineError, match="Syntax error."):
            db.execute("select n_name from nation where")
        with pytest.raises(engine.EngineError, match="single-character field terminators"):          # execute.h:340-342
            db.execute(f'bulk insert region from "{tmp_path}/region.tbl" with ( fieldterminator="||" )')
        # a second BULK INSERT appends, as the reference's does (AppendIterator on the existing relation, execute.h:348-350)
        before = db.execute("select count(*) as n from region").text.splitlines()[1]
        db.execute(f'bulk insert region from "{tmp_path}/region.tbl" with ( fieldterminator="|" )')
        after = db.execute("select count(*) as n from region").text.splitlines()[1]
        assert int(after.rstrip("|")) == 2 * int(before.rstrip("|"))
        assert db.report().num_kernels >= 1
        # a generated table handed over to the database
        li = gpu_ctx.generate(engine.GEN_LINEITEM, 60_000, 0.01, param=1)
        db.add_table(li)
        res = db.execute("select count(*) as n from lineitem where l_quantity < 26")
        assert 25_000 < int(res.text.splitlines()[1].split("|")[0]) < 35_000
    finally:
        db.close()


def test_reference_int16_cast_switch_reproduces_the_jit(small_db):
    """rsq_config.compat_flags = RSQ_COMPAT_JIT_INT16_CAST: INT -> BIGINT casts extend the low 16 bits, as the reference's asmjit
    back end does — a context created with the bit returns the JIT's own answer for `l_orderkey < 3`
    (keys 65537.. included), on the specialised kernels and on the interpreter; the default context returns what the source says"""
    host, _ = small_db
    cases = [g for g in GOLD["results"].values() if "reference_text" in g]
    assert cases
    for interpreter in ("0", "1"):
        os.environ["RSQ_FORCE_GENERIC"] = interpreter
        try:
            ctx = engine.Context(device=0, compat_flags=engine.COMPAT_JIT_INT16_CAST)
            tabs = [ctx.table(t) for t in host]
            for g in cases:
                q = ctx.sql_compile(g["sql"], tabs)
                q.execute()
                got = q.result()
                q.close()
                assert got.text == g["reference_text"] and got.text != g["text"]
            ctx.close()
        finally:
            os.environ.pop("RSQ_FORCE_GENERIC", None)


def test_config_is_validated():
    """struct_size says how much of rsq_config the host's header knew; unknown settings are refused instead of guessed"""
    import ctypes as C
    L = engine.lib()
    h = C.c_void_p()
    cfg = engine.rsq_config.make(0)
    cfg.struct_size = 0                                            # an uninitialised struct
    assert L.rsq_ctx_create(C.byref(cfg), C.byref(h)) == 1 and b"struct_size" in L.rsq_last_error(None)
    cfg = engine.rsq_config.make(0, emission_order=7)
    assert L.rsq_ctx_create(C.byref(cfg), C.byref(h)) == 1 and b"emission_order" in L.rsq_last_error(None)
    cfg = engine.rsq_config.make(0, compat_flags=1 << 9)
    assert L.rsq_ctx_create(C.byref(cfg), C.byref(h)) == 1 and b"compat_flags" in L.rsq_last_error(None)
    # a host built against the header before emission_order / compat_flags existed: garbage behind its struct is never read
    cfg = engine.rsq_config.make(0, emission_order=7, compat_flags=1 << 9)
    cfg.struct_size = engine.rsq_config.emission_order.offset
    assert L.rsq_ctx_create(C.byref(cfg), C.byref(h)) == 0
    L.rsq_ctx_destroy(h)


def test_random_valid_statements(gpu_ctx, small_db):
    """the first 60 statements of tests/sqlgen.py (the oracle's answers for all 160 are pinned on the reference by
    tests/test_sql_frontend.py): engine == oracle byte for byte, a statement the reference dies on is refused by both"""
    import sqlgen
    host, tabs = small_db
    with open(os.path.join(HERE, "golden", "sqlgen_reference.json")) as f:
        gold = json.load(f)["seeds"]
    ran = 0
    for seed in range(45):
        s = sqlgen.statement(seed)
        if "refused" in gold[str(seed)]:
            with pytest.raises(engine.EngineError):
                q = gpu_ctx.sql_compile(s, tabs)
                q.execute()
            continue
        want = orc.execute(gpu_ctx.sql_plan(s, tabs, host))
        q = gpu_ctx.sql_compile(s, tabs)
        try:
            q.execute()
            got = q.result()
        finally:
            q.close()
        assert got.text == want.text, s
        assert got.tuples == want.tuples, s
        ran += 1
    assert ran >= 43


def test_random_valid_statements_larger_and_repeated(gpu_ctx):
    """25 further statements at SF 0.1 (600 K lineitem rows), each executed three times: the second and third execution
    may take the late-load form of a pipeline (chosen from the first one's row counts) and reuse table capacities, entry
    counts and the LDS front table paths — every execution must give the oracle's answer"""
    import sqlgen
    db = tpch_full.database(0.1)
    host = [db[k] for k in sorted(db)]
    tabs = [gpu_ctx.table(t) for t in host]
    try:
        ran = 0
        for seed in range(200, 225):
            s = sqlgen.statement(seed)
            try:
                want = orc.execute(gpu_ctx.sql_plan(s, tabs, host))
            except orc.OracleError:
                continue                      # a statement the reference dies on (see test_sql_frontend.py)
            q = gpu_ctx.sql_compile(s, tabs)
            try:
                for _ in range(3):
                    q.execute()
                    assert q.result().text == want.text, s
            finally:
                q.close()
            ran += 1
        assert ran >= 20
    finally:
        for t in tabs:
            t.close()


def test_plain_c_host_end_to_end(tmp_path):
    """integration/examples/sql_host.c on the GPU: CREATE TABLE, BULK INSERT and SELECT through the C ABI from a C program"""
    import subprocess
    root = os.path.dirname(HERE)
    exe = str(tmp_path / "sql_host")
    subprocess.check_call(["gcc", "-std=c11", "-I" + os.path.join(root, "include"), os.path.join(root, "integration", "examples", "sql_host.c"),
                           "-L" + os.path.join(root, "resql_amd"), "-lresql_hip", "-Wl,-rpath," + os.path.join(root, "resql_amd"), "-o", exe])
    nat = tpch_full.nation()
    with open(tmp_path / "nation.tbl", "w") as f:
        for i in range(nat.n_rows):
            f.write(f"{nat.col('n_nationkey').data[i]}|{nat.col('n_name').data[i].decode()}|{nat.col('n_regionkey').data[i]}|c{i}|\n")
    pr = subprocess.run([exe, "create table nation ( n_nationkey int, n_name char(25), n_regionkey int, n_comment varchar(152) )",
                         f'bulk insert nation from "{tmp_path}/nation.tbl" with ( fieldterminator="|" )',
                         "select n_name, n_comment from nation where n_regionkey = 2 order by n_name desc limit 3",
                         "showperf", "showperf=true", "tables",
                         "select count(*) as n from nation"],
                        capture_output=True, text=True)
    assert pr.returncode == 0, pr.stdout + pr.stderr
    out = pr.stdout.splitlines()
    assert out[:3] == ["create table ok", "bulk insert ok", "3 row(s)"]
    assert out[3:6] == ["VIETNAM                  |c21|", "JAPAN                    |c12|", "INDONESIA                |c9|"]
    # control statements through the same C entry point (processControl, execute.h:454-474)
    assert out[6] == "false"
    assert out[7].startswith("┌") and any("nation" in l and " 25 " in l for l in out[8:14])
    rest = out[out.index(next(l for l in out if l.endswith("1 tables"))) + 1:]
    assert [l.split()[0] for l in rest if l.startswith(("compile:", "execute:", "device:"))] == ["compile:", "execute:", "device:"]
    assert rest[-2:] == ["1 row(s)", "25|"]


def test_control_variables_shape_what_a_select_returns(gpu_ctx, tmp_path, monkeypatch):
    """processControl's variables (execute.h:454-474) and what executeSelectPlan / printQueryResult do with them
    (execute.h:213-247, 173-200; showReport JitContextFlounder.h:132-150): showplan, showperf, showasm, showfln, tofile"""
    monkeypatch.chdir(tmp_path)                                   # tofile writes "qres.tbl" into the working directory
    nat = tpch_full.nation()
    db = engine.Database(gpu_ctx)
    try:
        db.add_table(gpu_ctx.table(nat))
        sql = "select n_regionkey, count(*) as c from nation group by n_regionkey order by n_regionkey"
        plain = db.execute(sql)
        assert db.last_kind == "SELECT" and db.message == "\n" and not os.path.exists("qres.tbl")
        for stmt in ("showperf=true", "showplan = true", "tofile=true", "threads=8"):
            assert db.execute(stmt) is None and db.last_kind == "CONTROL"
        res = db.execute(sql)
        assert res.text == plain.text
        msg = db.message.splitlines()
        assert any("AGGREGATION" in l.upper() for l in msg) and any("SCAN" in l.upper() for l in msg)        # the operator tree
        perf = [l for l in msg if l.startswith(("compile: ", "execute: ", "device:  "))]
        assert [l.split()[0] for l in perf] == ["compile:", "execute:", "device:"] and all(l.split()[2] == "ms" or l.split()[2] == "ms," for l in perf)
        assert float(perf[1].split()[1]) > 0
        with open("qres.tbl") as f:                                # serializeRelation's format (dbdata.h:688-701)
            assert f.read() == "".join(l + "\n" for l in plain.text.splitlines() if not l.startswith("#"))
        db.execute("showplan=false"); db.execute("showperf=false"); db.execute("tofile=false")
        db.execute("showasm=true")
        db.execute(sql)
        assert "__global__" in db.message and "rsq_device.h" in db.message                                   # the generated HIP source
        db.execute("showasm=false"); db.execute("showfln=true")
        db.execute(sql)
        assert "pipeline 0:" in db.message and "__global__" not in db.message                                # the pipeline description
        db.execute("threads")
        assert db.message == "8\n"
    finally:
        db.close()
