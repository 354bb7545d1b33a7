"""The reference's control statements (processControl, /root/reference/src/execute.h:407-474) in the engine's statement loop
(rsq_db_execute): variable syntax, error texts, `tables`, and what they do to later statements.  CPU part: a compile-only
context (control statements, CREATE TABLE and BULK INSERT need no GPU).  SELECT effects are in tests/test_gpu_sql.py."""
import os

import pytest

from resql_amd import engine


@pytest.fixture()
def db():
    ctx = engine.Context(device=-1)
    d = engine.Database(ctx)
    yield d
    d.close()
    ctx.close()


def test_bool_variables_set_and_print(db):
    for var in ("showplan", "tofile", "showperf", "showasm", "showfln", "optimize", "emitmc"):
        assert db.execute(var) is None and db.last_kind == "CONTROL" and db.message == "false\n"
        db.execute(f"{var}=true")
        assert db.message == ""
        db.execute(var)
        assert db.message == "true\n"
        db.execute(f" {var} = false ")                       # blanks are dropped before matching (execute.h:413)
        db.execute(var)
        assert db.message == "false\n"


def test_threads_variable(db):
    db.execute("threads")
    assert db.message == "1\n"                               # JitConfig::numThreads default (JitContextFlounder.h:96)
    db.execute("threads=16")
    db.execute("threads")
    assert db.message == "16\n"
    db.execute("threads = 4")
    db.execute("threads")
    assert db.message == "4\n"
    with pytest.raises(engine.EngineError):                  # the reference dies on std::invalid_argument here
        db.execute("threads=many")


def test_error_texts_of_set_bool_var(db):
    with pytest.raises(engine.EngineError) as e:
        db.execute("showperf=yes")
    assert "Expected true or false" in str(e.value)
    with pytest.raises(engine.EngineError) as e:
        db.execute("showperf:true")
    assert "Expected varname=value" in str(e.value)
    # rfind, not a prefix test (execute.h:414): the name may stand anywhere in the line, and a line that merely CONTAINS a
    # variable name is a (malformed) control statement — also a select over a column called tofile
    with pytest.raises(engine.EngineError) as e:
        db.execute("select tofile from t")
    assert "Expected varname=value" in str(e.value)


def test_tables_listing(db, tmp_path):
    db.execute("tables")
    first = db.message.splitlines()
    assert first[1].split("│")[1:4] == [" Table name ", " Number of attributes ", " Number of tuples "]
    assert first[-1].strip() == "0 tables"
    db.execute("create table nation ( n_nationkey int, n_name char(25), n_regionkey int )")
    assert db.message == "Created table nation\n"
    db.execute("create table r ( a int )")
    p = tmp_path / "n.tbl"
    p.write_text("0|ALGERIA|0|\n1|ARGENTINA|1|\n2|BRAZIL|1|\n")
    db.execute(f'bulk insert nation from "{p}" with ( fieldterminator="|" )')
    assert db.message == "Inserted 3 tuples\n" and db.last_kind == "BULK_INSERT"
    db.execute("tables")
    lines = db.message.splitlines()
    assert lines[0].startswith("┌") and lines[2].startswith("├") and lines[-2].startswith("└")
    rows = [[c.strip() for c in l.split("│")[1:4]] for l in lines[3:-2]]
    assert rows == [["nation", "3", "3"], ["r", "1", "0"]]
    widths = {len(l) for l in lines[:-1]}
    assert len(widths) == 1                                   # a rectangle
    assert lines[-1].endswith("2 tables") and len(lines[-1]) == len(lines[0]) - 1    # setw(sum of widths + columns), dbdata.h:620-624


def test_tables_must_be_the_whole_line(db):
    with pytest.raises(engine.EngineError) as e:
        db.execute("tables ")
    assert "Syntax error." in str(e.value)


def test_bulk_insert_appends_and_checks_the_terminator(db, tmp_path):
    db.execute("create table t ( a int, b decimal(6,2) )")
    p = tmp_path / "t.tbl"
    p.write_text("1|1.50|\n2|2.25|\n")
    db.execute(f'bulk insert t from "{p}" with ( fieldterminator="|" )')
    db.execute(f'bulk insert t from "{p}" with ( fieldterminator="|" )')       # the reference appends (execute.h:348-350)
    db.execute("tables")
    assert [c.strip() for c in db.message.splitlines()[3].split("│")[1:4]] == ["t", "2", "4"]
    with pytest.raises(engine.EngineError) as e:
        db.execute(f'bulk insert t from "{p}" with ( fieldterminator="||" )')
    assert "Bulk insert only supports single-character field terminators." in str(e.value)
