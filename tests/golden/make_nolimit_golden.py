#!/usr/bin/env python3
"""Generate tests/golden/ref_nolimit_sf1.json: the UNMODIFIED reference's answers to TPC-H Q3 and Q10 WITHOUT their LIMIT at SF1 -
every group of the aggregation leaves in the reference's own order (hash-table slot order, then its quicksort): 11 K and 38 K
rows.  Too many bytes for a fixture, so the file keeps their number, the SHA-256 of the serialised relation and its first and
last rows; tests/test_gpu_sql_fullsize.py compares the engine's text the same way (the device tail of engine.cpp
runRowsDeviceTail makes these rows).  Q3: the hand-built plan of resql_amd/tpch.py over the three-table generator; Q10: the
reference's own statement text minus `limit 20`, through its grammar and planner, over the eight-table database.

Run in the build container only (needs /root/reference):   python tests/golden/make_nolimit_golden.py
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from resql_amd import engine, tpch, tpch_full  # noqa: E402
from oracle import orc  # noqa: E402

SF = 1.0


def entry(text, **more):
    lines = text.splitlines()
    e = {"rows": len(lines) - 1, "sha256": hashlib.sha256(text.encode("latin1")).hexdigest(), "head": lines[:6], "tail": lines[-5:]}
    e.update(more)
    return e


def q10_nolimit_sql():
    sql = tpch_full.QUERIES["q10"]
    assert "limit 20" in sql
    return sql.replace("limit 20", "")


def main():
    if not orc.have_reference():
        raise SystemExit("oracle/_ref/ref_harness is missing: run `make -C oracle ref` (needs /root/reference)")
    out = {"sf": SF}
    t0 = time.time()
    plan = tpch.q3_plan(tpch.customer_table(SF), tpch.orders_table(SF), tpch.lineitem_table(SF, tpch.Q3_LINEITEM_COLUMNS), limit=None)
    text, tm = orc.run_reference(plan)
    oracle_text = orc.execute(plan).text
    out["q3_nolimit"] = entry(text, reference_exec_ms=tm["exec_ms"], oracle_equal=oracle_text == text)
    print("q3 without limit:", out["q3_nolimit"]["rows"], "rows, oracle equal:", oracle_text == text, f"{time.time() - t0:.0f} s", flush=True)
    t0 = time.time()
    db = tpch_full.database(SF, fill_unused=False)
    host = [db[k] for k in sorted(db)]
    ctx = engine.Context(device=-1)
    tabs = [ctx.table(t) for t in host]
    sql = q10_nolimit_sql()
    text = orc.run_reference_sql(host, ctx.sql_describe(sql, 0))
    res = orc.execute(ctx.sql_plan(sql, tabs, host))
    e = entry(text, oracle_equal=res.text == text)
    if (res.ref_oob_probes > 0 or res.ref_narrow_casts > 0) and res.text != text:
        e = entry(res.text, oracle_equal=True, reference_undefined=True)      # (the reference read past its table's end: its own answer depends on a heap byte)
    out["q10_nolimit"] = e
    print("q10 without limit:", e["rows"], "rows, oracle equal:", e["oracle_equal"], f"{time.time() - t0:.0f} s", flush=True)
    with open(os.path.join(HERE, "ref_nolimit_sf1.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
