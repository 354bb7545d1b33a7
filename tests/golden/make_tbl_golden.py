"""Generate tests/golden/tbl_reference.json: what the UNMODIFIED reference answers after loading tests/golden/tbl/*.tbl
with its own field parser (oracle/ref/ref_harness.cpp `tbl` directive = the loop of executeBulkInsert, reference
src/execute.h:357-385, over ExprGen::constant + ValueMoves::toAddress).  Run in the build container:
    python tests/golden/make_tbl_golden.py
Scans are stored as sha256 of the serialised relation (they are as large as the input), queries as text."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import tblcases  # noqa: E402
from oracle import orc  # noqa: E402


def run_reference_on_tbl(name):
    tables_needed, make = tblcases.QUERIES[name]
    tabs = {t: tblcases.schema_table(t) for t in tables_needed}
    plan = make(tabs)
    with tempfile.TemporaryDirectory(prefix="resql_tbl_") as d:
        path = os.path.join(d, "plan.case")
        with open(path, "w") as f:
            f.write(plan.to_text(tbl_files={t: tblcases.FILES[t] for t in tables_needed}))
        pr = subprocess.run([orc.REF_HARNESS, path], capture_output=True, text=True, errors="replace")
        if pr.returncode != 0 or "#timing" not in pr.stderr:
            raise SystemExit(f"{name}: reference failed: {pr.stderr[-800:]}")
        return pr.stdout


def main():
    out = {}
    for name in tblcases.QUERIES:
        text = run_reference_on_tbl(name)
        if name.startswith("scan_"):
            out[name] = {"sha256": hashlib.sha256(text.encode("latin1")).hexdigest(), "rows": len(text.splitlines()) - 1,
                         "head": text.splitlines()[:3]}
        else:
            out[name] = {"text": text}
    with open(os.path.join(HERE, "tbl_reference.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, v in out.items():
        print(k, v.get("rows", ""), (v.get("text") or "")[:300])


if __name__ == "__main__":
    main()
