"""Rebuild the plans of the committed golden files (tests/golden/ref_*.tbl, written by
tests/golden/make_golden.py from the unmodified reference) from their recorded parameters."""
import json
import os

from resql_amd import tpch

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "ref_index.json")) as f:
    INDEX = json.load(f)


def golden_text(name: str) -> str:
    with open(os.path.join(HERE, "golden", f"ref_{name}.tbl")) as f:
        return f.read()


def golden_plan(name: str):
    p = INDEX[name]
    if p["plan"] == "synthetic":
        return tpch.synthetic_plan(tpch.synthetic_table(p["n"], p["groups"]), p["threshold"])
    sf = p["sf"]
    if p["plan"] == "q1":
        li = tpch.lineitem_table(sf, tpch.Q1_COLUMNS)
        return tpch.q1_plan(li, shipdate=p.get("shipdate", "1998-9-02"))
    if p["plan"] == "q6":
        li = tpch.lineitem_table(sf, tpch.Q6_COLUMNS)
        return tpch.q6_plan(li, p.get("date_lo", "1994-01-01"), p.get("date_hi", "1995-01-01"), p.get("discount", "0.06"),
                            p.get("quantity", "24"))
    if p["plan"] == "q3":
        li = tpch.lineitem_table(sf, tpch.Q3_LINEITEM_COLUMNS)
        return tpch.q3_plan(tpch.customer_table(sf), tpch.orders_table(sf), li, limit=p.get("limit"))
    raise KeyError(p["plan"])


NAMES = sorted(INDEX)
