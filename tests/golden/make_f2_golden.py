#!/usr/bin/env python3
"""Generate tests/golden/ref_full_{q5,q10,q12,q14,q19}_sf{1,10}.tbl: answers of the UNMODIFIED reference — its own grammar
(src/parser/parser.y via its vendored lemon), its planner (src/planner.h) and its asmjit JIT, fed the statements' token
streams — on the eight-table database of resql_amd/tpch_full.py at SF1 and SF10 (SF10: 59 999 996 lineitem rows, 15 M orders,
1.5 M customers, 2 M parts, 100 K suppliers).  The statements are the reference's own tpch/queries/q{5,10,12,14,19}.sql
(test/test_queries.h:5-110 runs them at SF0.01).

The inputs are never stored: tests/test_gpu_sql_fullsize.py regenerates them (same numpy generator), uploads them and
compares the engine's answers byte for byte.  The CPU oracle runs the same plans here and counts probes of the reference's
hash table that read past its end (qlib/hash.h:441-451): such answers are recorded as "reference undefined".

Run in the build container only (needs /root/reference, ~30 GB of memory and ~15 minutes at SF10):
    python tests/golden/make_f2_golden.py [sf ...]
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from resql_amd import engine, tpch_full  # noqa: E402
from oracle import orc  # noqa: E402

NAMES = ("q5", "q10", "q12", "q14", "q19")


def main():
    if not orc.have_reference():
        raise SystemExit("oracle/_ref/ref_harness is missing: run `make -C oracle ref` (needs /root/reference)")
    sfs = [float(a) for a in sys.argv[1:]] or [1.0, 10.0]
    index_path = os.path.join(HERE, "ref_full_index.json")
    index = json.load(open(index_path)) if os.path.exists(index_path) else {}
    ctx = engine.Context(device=-1)
    for sf in sfs:
        t0 = time.time()
        db = tpch_full.database(sf, fill_unused=False)
        host = [db[k] for k in sorted(db)]
        print(f"SF{sf:g}: generated the database in {time.time() - t0:.0f} s: " + ", ".join(f"{t.name} {t.n_rows}" for t in host), flush=True)
        tabs = [ctx.table(t) for t in host]
        for name in NAMES:
            sql = tpch_full.QUERIES[name]
            t1 = time.time()
            text = orc.run_reference_sql(host, ctx.sql_describe(sql, 0))
            tag = f"{name}_sf{sf:g}"
            entry = {"plan": name, "sf": sf, "rows": {t.name: t.n_rows for t in host}, "result_rows": len(text.splitlines()),
                     "reference_wall_s": round(time.time() - t1, 1)}
            if os.environ.get("RSQ_GOLDEN_SKIP_ORACLE") != "1":
                t2 = time.time()
                res = orc.execute(ctx.sql_plan(sql, tabs, host))
                entry["oracle_equal"] = res.text == text
                entry["oracle_wall_s"] = round(time.time() - t2, 1)
                if (res.ref_oob_probes > 0 or res.ref_narrow_casts > 0) and res.text != text:
                    entry["reference_undefined"] = True
                    text = res.text          # the golden keeps the oracle's answer (the reference's depends on a heap byte)
            with open(os.path.join(HERE, f"ref_full_{tag}.tbl"), "w", encoding="latin1") as f:
                f.write(text)
            index[tag] = entry
            print(tag, entry, flush=True)
        for t in tabs:
            t.close()
        del db, host, tabs
    with open(index_path, "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
