"""The oracle against the REFERENCE's own answers (tests/golden/string_join_reference.json, recorded by
tests/golden/make_string_join_golden.py from the unmodified reference) for string join keys of different kinds and lengths,
string constants as keys, and a join with several matches per row under a hash aggregation."""
import hashlib
import json
import os

import pytest

from oracle import orc
import stringjoincases as sj

with open(os.path.join(os.path.dirname(__file__), "golden", "string_join_reference.json")) as f:
    GOLDEN = json.load(f)
CASES = sj.all_cases({k: v.get("salt", 0) for k, v in GOLDEN.items()})


def digest(text):
    canon = "\n".join(sj.canonical(text))
    return len(canon.splitlines()) - 1, hashlib.sha256(canon.encode("latin1")).hexdigest()


def test_every_case_has_a_reference_answer():
    assert sorted(GOLDEN) == sorted(CASES) and len(CASES) == 39


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_equals_the_reference(name):
    rows, sha = digest(orc.execute(CASES[name]()).text)
    assert (rows, sha) == (GOLDEN[name]["rows"], GOLDEN[name]["sha256"])


def test_the_rules_the_answers_follow():
    """CHAR(a) against CHAR(b) never matches; VARCHAR against VARCHAR matches equal strings whatever the declared lengths; a
    constant build key gives every matching probe row all three build rows"""
    for b, p in sj.KIND_PAIRS:
        rows = GOLDEN["kinds_" + sj.pair_name(b, p)]["rows"]
        if b[0] == "CHAR" and p[0] == "CHAR":
            assert rows == 0
        if b[0] == "VARCHAR" and p[0] == "VARCHAR":
            assert rows > 20
    assert GOLDEN["const_join_varchar_build"]["rows"] == 3 and GOLDEN["const_join_varchar_probe"]["rows"] == 1
    assert GOLDEN["multi_match_agg"]["rows"] == 300


@pytest.mark.skipif(not orc.have_reference(), reason="oracle/_ref/ref_harness not built (needs /root/reference)")
@pytest.mark.parametrize("name", ["kinds_char4_varchar6", "kinds_varchar11_char4", "const_join_char_build", "multi_match_agg"])
def test_live_reference_still_gives_the_recorded_answers(name):
    text, _ = orc.run_reference(CASES[name]())
    assert digest(text) == (GOLDEN[name]["rows"], GOLDEN[name]["sha256"])
