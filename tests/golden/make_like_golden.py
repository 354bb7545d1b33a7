"""tests/golden/like_reference.json: the unmodified reference's answers for tests/likecases.py (string table x patterns).
Run in the build container:  python tests/golden/make_like_golden.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import likecases  # noqa: E402
from oracle import orc  # noqa: E402

out = {"strings": likecases.STRINGS, "patterns": likecases.PATTERNS}
for kind in ("char", "varchar"):
    out["flags_" + kind] = orc.run_reference(likecases.plan(kind))[0]
    out["select_" + kind] = {pat: orc.run_reference(likecases.select_plan(kind, pat))[0] for pat in ("%BRASS", "PROMO%", "%special%requests%", "a_b")}
with open(os.path.join(HERE, "like_reference.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print("ok", len(out["flags_char"]))
