"""Generate tests/golden/fuzz_reference.json: for every fuzz seed, what the UNMODIFIED reference (oracle/_ref/ref_harness,
built from /root/reference by oracle/Makefile) answers.  Run in the build container:  python tests/golden/make_fuzz_golden.py

Per seed: {"kind": exact|multiset|count, "refused": bool, "digest": sha256 of the canonical text, "rows": n,
           "ref_undefined": true when the oracle saw the reference read past its hash table (qlib/hash.h:441-451) —
           the reference's answer is then heap-dependent and no digest is stored}.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import fuzzplans  # noqa: E402
from oracle import orc  # noqa: E402

N_SEEDS = 400


def main():
    out = {}
    for seed in range(N_SEEDS):
        plan, kind = fuzzplans.make(seed)
        entry = {"kind": kind}
        try:
            text, _ = orc.run_reference(plan)
            entry["refused"] = False
        except orc.OracleError as e:
            entry["refused"] = True
            entry["message"] = str(e).splitlines()[-1][-120:]
            out[str(seed)] = entry
            continue
        try:
            oob = orc.execute(plan).ref_oob_probes
        except orc.OracleError:
            oob = 0
        if oob:
            entry["ref_undefined"] = True
        else:
            entry["digest"] = fuzzplans.digest(kind, text)
            entry["rows"] = len(text.splitlines()) - 1
        out[str(seed)] = entry
    with open(os.path.join(HERE, "fuzz_reference.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("seeds", len(out), "refused", sum(e["refused"] for e in out.values()),
          "undefined", sum(1 for e in out.values() if e.get("ref_undefined")))


if __name__ == "__main__":
    main()
