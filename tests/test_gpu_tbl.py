"""BULK INSERT on the GPU: '.tbl' files -> device columns -> queries through the C ABI; answers equal what the unmodified
reference gives for the same files (tests/golden/tbl_reference.json) and what the oracle computes."""
import json
import os

import pytest

from oracle import orc

import tblcases
from test_tbl_ingest import host_table

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(__file__), "golden", "tbl_reference.json")) as f:
    GOLDEN = json.load(f)


@pytest.mark.parametrize("name", ["orders_by_status", "building_orders"])
def test_queries_over_bulk_inserted_tables(gpu_ctx, name):
    needed, make = tblcases.QUERIES[name]
    schemas = {t: tblcases.schema_table(t) for t in needed}
    dev = {t: gpu_ctx.load_tbl(schemas[t], tblcases.FILES[t]) for t in needed}
    plan = make(schemas)
    q = gpu_ctx.compile(plan, [dev[t.name] for t in plan.tables])
    q.execute()
    got = q.result().text
    q.close()
    assert got == GOLDEN[name]["text"]
    host = {t: host_table(dev[t], schemas[t]) for t in needed}
    assert got == orc.execute(make(host)).text
    for d in dev.values():
        d.close()


def test_scan_of_bulk_inserted_table_keeps_strings_and_order(gpu_ctx):
    import hashlib
    schema = tblcases.schema_table("customer")
    dev = gpu_ctx.load_tbl(schema, tblcases.FILES["customer"])
    q = gpu_ctx.compile(tblcases.scan_plan(schema), [dev])
    q.execute()
    text = q.result().text
    q.close(); dev.close()
    assert hashlib.sha256(text.encode("latin1")).hexdigest() == GOLDEN["scan_customer"]["sha256"]
