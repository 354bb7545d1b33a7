"""String join keys of different kinds and declared lengths, string constants as keys.  The reference hashes each side with its
own type's function (hashChar pads with spaces to the DECLARED length, hashVarchar sums the characters: qlib/hash.h:116-147),
ht_get returns only entries with an equal hash (:427-477), and the keys are compared the probe side's way (hashjoin.h:142/191).
So CHAR(n) against VARCHAR matches a value that is exactly n characters long, CHAR(a) against CHAR(b) never matches, VARCHAR
against VARCHAR matches equal strings.  Engine == the reference's recorded answers (tests/golden/string_join_reference.json) ==
oracle, for every case of tests/stringjoincases.py."""
import hashlib
import json
import os

import pytest

from oracle import orc
import stringjoincases as sj

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(__file__), "golden", "string_join_reference.json")) as f:
    GOLDEN = json.load(f)
CASES = sj.all_cases({k: v.get("salt", 0) for k, v in GOLDEN.items()})


@pytest.mark.parametrize("name", sorted(CASES))
def test_engine_equals_reference_and_oracle(gpu_ctx, name):
    plan = CASES[name]()
    got = sj.canonical(gpu_ctx.run(plan).text)
    assert got == sj.canonical(orc.execute(plan).text)
    canon = "\n".join(got)
    assert (len(got) - 1, hashlib.sha256(canon.encode("latin1")).hexdigest()) == (GOLDEN[name]["rows"], GOLDEN[name]["sha256"])
