#!/usr/bin/env python3
"""Generate tests/golden/ref_full_*.tbl: answers of the UNMODIFIED reference (oracle/_ref/ref_harness, built from
/root/reference by `make -C oracle ref`) at BASELINE.json's full sizes — TPC-H Q1 / Q6 / Q3 at SF1 and SF10 over the
repo's deterministic synthetic tables (resql_amd/datagen.py; SF10: 59 999 996 lineitem rows, 15 M orders, 1.5 M customers).

The inputs are never stored: the GPU tests regenerate them on the device (bit-identical generator, proven at small sizes
by tests/test_gpu_tpch.py) and compare the engine's answer with these few rows byte for byte
(tests/test_gpu_fullsize.py), the way the reference's own test/test_queries.h:5-60 compares against test/reference/q*.tbl.

Run in the build container only (needs /root/reference, ~25 GB of memory and a few minutes at SF10):
    python tests/golden/make_fullsize_golden.py [sf ...]
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from resql_amd import tpch  # noqa: E402
from oracle import orc  # noqa: E402


def main():
    if not orc.have_reference():
        raise SystemExit("oracle/_ref/ref_harness is missing: run `make -C oracle ref` (needs /root/reference)")
    sfs = [float(a) for a in sys.argv[1:]] or [1.0, 10.0]
    index_path = os.path.join(HERE, "ref_full_index.json")
    index = json.load(open(index_path)) if os.path.exists(index_path) else {}
    for sf in sfs:
        t0 = time.time()
        li = tpch.lineitem_table(sf, tpch.Q1_COLUMNS + ["l_orderkey"])
        cu, od = tpch.customer_table(sf), tpch.orders_table(sf)
        print(f"SF{sf:g}: generated {li.n_rows} lineitem rows in {time.time() - t0:.0f} s", flush=True)
        for name, plan in ((f"q1_sf{sf:g}", tpch.q1_plan(li)), (f"q6_sf{sf:g}", tpch.q6_plan(li)), (f"q3_sf{sf:g}", tpch.q3_plan(cu, od, li))):
            t1 = time.time()
            text, tm = orc.run_reference(plan)
            with open(os.path.join(HERE, f"ref_full_{name}.tbl"), "w") as f:
                f.write(text)
            index[name] = {"plan": name.split("_")[0], "sf": sf, "lineitem_rows": li.n_rows, "orders_rows": od.n_rows, "customer_rows": cu.n_rows,
                           "reference_exec_ms": tm["exec_ms"]}
            print(name, len(text.splitlines()) - 1, "rows; reference execute:", tm["exec_ms"], f"ms; wall {time.time() - t1:.0f} s", flush=True)
        del li, cu, od
    with open(index_path, "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
