"""Generate tests/golden/sweep_finds.json: the fuzz seeds BEYOND the suite's range on which a wide sweep on the GPU (tools/exp/fuzz_sweep.py,
tools/fuzz_sweep.py) once found the engine wrong, with what the UNMODIFIED reference (oracle/_ref/ref_harness) answers for them.
Run in the build container:  python tests/golden/make_sweep_finds_golden.py
Per seed: {"kind", "digest" (sha256 of the canonical text), "rows", "what" (the defect the seed found)}."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import fuzzplans  # noqa: E402
from oracle import orc  # noqa: E402

FINDS = {
    30102: "round 5: two build rows of ONE join key carry the CHAR(6) values 'MAIL' and 'MAIL  '; grouped by the probe key and that value "
           "(carried, compared in full because the build keys repeat) they came out as two groups - nobody told the host to merge them",
}


def main():
    out = {}
    for seed, what in sorted(FINDS.items()):
        plan, kind = fuzzplans.make(seed)
        text, _ = orc.run_reference(plan)
        assert orc.execute(plan).ref_oob_probes == 0
        out[str(seed)] = {"kind": kind, "digest": fuzzplans.digest(kind, text), "rows": len(text.splitlines()) - 1, "what": what}
    with open(os.path.join(HERE, "sweep_finds.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(out)


if __name__ == "__main__":
    main()
