#!/usr/bin/env python3
"""tests/golden/sqlgen_reference.json: per seed of tests/sqlgen.py, the digest of the UNMODIFIED reference's answer (its Lemon
grammar + planner + asmjit JIT, fed the engine's token stream) over the SF 0.01 database of resql_amd/tpch_full.py — or why there
is none ("refused": the reference refuses the statement; "undefined": its answer is an artefact of a known defect).
Run in the build container only:  python tests/golden/make_sqlgen_golden.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

from resql_amd import engine, tpch_full  # noqa: E402
from oracle import orc  # noqa: E402
import sqlgen  # noqa: E402

SEEDS = 160
ctx = engine.Context(device=-1)
db = tpch_full.database(0.01)
host = [db[k] for k in sorted(db)]
tabs = [ctx.table(t) for t in host]
out = {"sf": 0.01, "seeds": {}}
for seed in range(SEEDS):
    s = sqlgen.statement(seed)
    try:
        ref = orc.run_reference_sql(host, ctx.sql_describe(s, 0))
    except orc.OracleError as e:
        out["seeds"][str(seed)] = {"refused": str(e).strip().splitlines()[-1][:160]}
        continue
    entry = {"sha1": hashlib.sha1(ref.encode("latin1")).hexdigest(), "rows": len(ref.splitlines()) - 1}
    try:
        res = orc.execute(ctx.sql_plan(s, tabs, host))
        if res.text != ref and (res.ref_oob_probes or res.ref_narrow_casts):
            entry = {"undefined": "oob probe" if res.ref_oob_probes else "int16 cast", "rows": entry["rows"]}
    except Exception:
        pass
    out["seeds"][str(seed)] = entry
with open(os.path.join(HERE, "sqlgen_reference.json"), "w") as f:
    json.dump(out, f, indent=0)
kinds = [("sha1" in v, "refused" in v, "undefined" in v) for v in out["seeds"].values()]
print("answers", sum(k[0] for k in kinds), "refused", sum(k[1] for k in kinds), "undefined", sum(k[2] for k in kinds))
