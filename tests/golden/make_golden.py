#!/usr/bin/env python3
"""Generate tests/golden/ref_*.tbl: outputs of the UNMODIFIED reference (oracle/_ref/ref_harness, built from
/root/reference by `make -C oracle ref`) on seeded inputs.

Inputs are never stored: they are regenerated bit-identically from (seed, scale factor) by
resql_amd/datagen.py.  Each golden file holds the plan name, its parameters and the reference's serialised
result (`#schema` line + serializeRelation output), so the tests can replay the same plan through the oracle
(CPU) and through the HIP engine (GPU) and compare byte for byte.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from resql_amd import tpch  # noqa: E402
from oracle import orc  # noqa: E402


def cases():
    for sf in (0.01, 0.05):
        li = tpch.lineitem_table(sf, tpch.Q1_COLUMNS + ["l_orderkey"])
        cu, od = tpch.customer_table(sf), tpch.orders_table(sf)
        yield f"q1_sf{sf}", {"plan": "q1", "sf": sf}, tpch.q1_plan(li)
        yield f"q6_sf{sf}", {"plan": "q6", "sf": sf}, tpch.q6_plan(li)
        yield f"q3_sf{sf}", {"plan": "q3", "sf": sf, "limit": 10}, tpch.q3_plan(cu, od, li)
        yield f"q3_nolimit_sf{sf}", {"plan": "q3", "sf": sf, "limit": None}, tpch.q3_plan(cu, od, li, limit=None)
    # parameter variants (other substitution values of the TPC-H templates)
    li = tpch.lineitem_table(0.02, tpch.Q1_COLUMNS + ["l_orderkey"])
    yield "q1_sf0.02_1998-08-01", {"plan": "q1", "sf": 0.02, "shipdate": "1998-08-01"}, tpch.q1_plan(li, shipdate="1998-08-01")
    yield "q6_sf0.02_1996", {"plan": "q6", "sf": 0.02, "date_lo": "1996-01-01", "date_hi": "1997-01-01", "discount": "0.03",
                             "quantity": "25"}, tpch.q6_plan(li, "1996-01-01", "1997-01-01", "0.03", "25")
    for groups, sel in ((8, 0.5), (1024, 0.1), (4096, 0.5)):
        t = tpch.synthetic_table(100_000, groups)
        yield f"synth_g{groups}_s{sel}", {"plan": "synthetic", "n": 100_000, "groups": groups, "threshold": int(sel * (1 << 31))}, \
            tpch.synthetic_plan(t, int(sel * (1 << 31)))


def main():
    if not orc.have_reference():
        raise SystemExit("oracle/_ref/ref_harness is missing: run `make -C oracle ref` (needs /root/reference)")
    index = {}
    for name, params, plan in cases():
        text, tm = orc.run_reference(plan)
        with open(os.path.join(HERE, f"ref_{name}.tbl"), "w") as f:
            f.write(text)
        index[name] = params
        print(name, len(text.splitlines()) - 1, "rows", tm["exec_ms"])
    with open(os.path.join(HERE, "ref_index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
