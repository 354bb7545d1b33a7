"""Known-answer TPC-H: the reference's own SF 0.01 answers (test/reference/q{1,3,5,6,10,12,19}.tbl, which its
test/test_queries.h:5-110 compares IN ORDER with what its engine returns) replayed against the oracle (CPU) and the HIP
engine (GPU).

The reference's snapshot has no lineitem.tbl, so the input is regenerated: tests/tpch_dbgen.py restates the TPC's data
generator, and is pinned here against the seven data files the reference does ship — every generated field of orders
(incl. o_totalprice and o_orderstatus, which are functions of the order's lineitems), customer, part, supplier, partsupp,
nation and region equals the reference's file (compared directly where /root/reference exists, through committed SHA-256
digests of those files everywhere).  The tables then take the reference's own way in: '.tbl' text -> BULK INSERT
(rsq_table_load_tbl) -> the reference's query texts through the SQL front end.
Q10 prints c_comment, which dbgen takes from its 300 MB text pool (not restated): that one column is left out of the
comparison.  Q14 has no committed answer in the reference."""
import hashlib
import json
import os

import numpy as np
import pytest

from resql_amd import engine, plan as P, tpch, tpch_full
from oracle import orc
import tpch_dbgen as G

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "tpch_sf001")
REF_DATA = "/root/reference/tpch/datasets/sf001"
SCHEMAS = {"lineitem": tpch.LINEITEM_SCHEMA, "orders": tpch.ORDERS_SCHEMA, "customer": tpch.CUSTOMER_SCHEMA, "part": tpch_full.PART_SCHEMA,
           "supplier": tpch_full.SUPPLIER_SCHEMA, "partsupp": tpch_full.PARTSUPP_SCHEMA, "nation": tpch_full.NATION_SCHEMA,
           "region": tpch_full.REGION_SCHEMA}
ANSWERED = ["q1", "q3", "q5", "q6", "q10", "q12", "q19"]


@pytest.fixture(scope="module")
def generated():
    return G.Tables(0.01)


@pytest.fixture(scope="module")
def tbl_dir(generated, tmp_path_factory):
    d = tmp_path_factory.mktemp("tpch_sf001")
    generated.write(str(d))
    return str(d)


def _fields(line, fields):
    f = line.split("|")
    return "|".join(f[i] for i in fields)


def test_generator_reproduces_the_reference_data_files(generated):
    with open(os.path.join(GOLD, "reference_data_digests.json")) as f:
        digests = json.load(f)["tables"]
    assert len(generated.lineitem) == 60175 and len(generated.orders) == 15000          # dbgen's SF 0.01 sizes
    for name, d in digests.items():
        h, n = hashlib.sha256(), 0
        for line in generated.tbl_lines(name):
            h.update((_fields(line, d["fields"]) + "\n").encode())
            n += 1
        assert n == d["rows"] and h.hexdigest() == d["sha256"], f"{name}: generated fields differ from the reference's {name}.tbl"
        if os.path.isdir(REF_DATA):                # in the build container: field by field, so that a difference names its row
            with open(os.path.join(REF_DATA, name + ".tbl")) as f:
                for i, (mine, theirs) in enumerate(zip(generated.tbl_lines(name), f.read().splitlines())):
                    assert _fields(mine, d["fields"]) == _fields(theirs, d["fields"]), f"{name} row {i}"


def _golden(q):
    with open(os.path.join(GOLD, q + ".tbl")) as f:
        return [l.split("|")[:-1] for l in f.read().splitlines()]


def _check(q, text):
    got = [l.split("|")[:-1] for l in text.splitlines() if not l.startswith("#")]
    want = _golden(q)
    if q == "q10":                                 # c_comment comes from dbgen's text pool
        got, want = [r[:-1] for r in got], [r[:-1] for r in want]
    assert len(got) == len(want), f"{q}: {len(got)} rows, the reference has {len(want)}"
    for i, (a, b) in enumerate(zip(got, want)):    # in order, as checkRelations(..., true) compares (test_common.h:125-148)
        assert [x.rstrip() for x in a] == [x.rstrip() for x in b], f"{q} row {i}: {a} != {b}"


def _create_statement(name):
    cols = ", ".join(f"{c} {str(t).lower()}" for c, t in SCHEMAS[name])
    return f"create table {name} ( {cols} )"


def _load_all(db, tbl_dir):
    for name in G.TABLE_NAMES:
        db.execute(_create_statement(name))
        db.execute(f'bulk insert {name} from "{tbl_dir}/{name}.tbl" with ( fieldterminator="|" )')
        assert db.message.startswith("Inserted ")


@pytest.fixture(scope="module")
def host_database(tbl_dir):
    """the generated files through the engine's BULK INSERT into a compile-only context (host copies), read back as host tables"""
    ctx = engine.Context(device=-1)
    host, dev = [], []
    for name in sorted(G.TABLE_NAMES):
        schema = P.Table(name, [P.Column(c, t) for c, t in SCHEMAS[name]], 0)
        dt = ctx.load_tbl(schema, os.path.join(tbl_dir, name + ".tbl"), "|")
        cols = [P.Column(c, t, dt.read_column(c, t.np_dtype)) for c, t in SCHEMAS[name]]
        host.append(P.Table(name, cols, dt.n_rows))
        dev.append(dt)
    yield ctx, host, dev
    for d in dev:
        d.close()
    ctx.close()


@pytest.mark.parametrize("q", ANSWERED)
def test_oracle_gives_the_reference_answers(host_database, q):
    ctx, host, dev = host_database
    plan = ctx.sql_plan(tpch_full.QUERIES[q], dev, host)
    _check(q, orc.execute(plan).text)


@pytest.mark.gpu
def test_engine_gives_the_reference_answers(gpu_ctx, tbl_dir):
    """create table + bulk insert + the reference's query texts, all through the statement loop on the GPU"""
    db = engine.Database(gpu_ctx)
    try:
        _load_all(db, tbl_dir)
        db.execute("tables")
        assert "lineitem" in db.message and " 60175 " in db.message
        for q in ANSWERED:
            res = db.execute(tpch_full.QUERIES[q])
            _check(q, res.text)
            res2 = db.execute(tpch_full.QUERIES[q])            # and again (table capacities, late-load forms, rank dictionaries reused)
            assert res2.text == res.text
    finally:
        db.close()


@pytest.mark.skipif(not orc.have_reference(), reason="needs the compiled reference (oracle/_ref, build container only)")
@pytest.mark.parametrize("q", ["q1", "q3", "q12"])
def test_the_reference_itself_reproduces_its_answers_on_the_generated_data(host_database, q):
    """closing the loop: ReSQL's own grammar, planner and asmjit JIT (oracle/_ref/ref_harness), fed the regenerated tables,
    return what its repository records as known answers"""
    ctx, host, dev = host_database
    _check(q, orc.run_reference_sql(host, ctx.sql_describe(tpch_full.QUERIES[q], 0)))
