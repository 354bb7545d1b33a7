"""Records the REFERENCE's answers (oracle/_ref/ref_harness, the unmodified reference compiled by oracle/Makefile) for the string
join-key cases of tests/stringjoincases.py into tests/golden/string_join_reference.json (sha256 of the canonical text + row count;
the texts of the small cases in full).  Run in the build container:  python tests/golden/make_string_join_golden.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import orc          # noqa: E402
import stringjoincases as sj     # noqa: E402

out = {}
skipped = 0
for name in sorted(sj.all_cases()):
    # data on which the reference loses matches behind its table's last slot (stringjoincases.kinds_plan) is skipped: the next salt
    for salt in range(0, 20):
        plan = sj.all_cases({name: salt})[name]()
        text, _ = orc.run_reference(plan)
        if sj.canonical(text) == sj.canonical(orc.execute(plan).text):
            break
        skipped += 1
        if not name.startswith("kinds_"):
            raise SystemExit(f"{name}: the reference and the oracle disagree and the case has no data to vary")
    else:
        raise SystemExit(f"{name}: no salt on which the reference returns every match")
    canon = "\n".join(sj.canonical(text))
    entry = {"rows": len(canon.splitlines()) - 1, "sha256": hashlib.sha256(canon.encode("latin1")).hexdigest(), "salt": salt}
    if len(canon) < 600:
        entry["text"] = canon
    out[name] = entry
with open(os.path.join(HERE, "string_join_reference.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(len(out), "cases;", sum(e["rows"] for e in out.values()), "rows in all;", skipped, "data sets skipped (reference lost matches at its table's end)")
