"""The oracle on the seeded random plans of tests/fuzzplans.py, pinned two ways:
  * against tests/golden/fuzz_reference.json — digests of what the unmodified reference answered for every seed
    (written by tests/golden/make_fuzz_golden.py in the build container); runs anywhere;
  * against the compiled reference, live, where oracle/_ref/ref_harness exists.
Runs on CPU."""
import json
import os

import pytest

from oracle import orc

import fuzzplans

with open(os.path.join(os.path.dirname(__file__), "golden", "fuzz_reference.json")) as f:
    GOLDEN = json.load(f)


@pytest.mark.parametrize("block", range(0, len(GOLDEN), 50))
def test_fuzz_oracle_matches_reference_digests(block):
    bad = []
    for seed in range(block, min(block + 50, len(GOLDEN))):
        g = GOLDEN[str(seed)]
        plan, kind = fuzzplans.make(seed)
        assert kind == g["kind"], f"seed {seed}: the generator drifted from the committed goldens; re-run make_fuzz_golden.py"
        try:
            res = orc.execute(plan)
        except orc.OracleError as e:
            if not g["refused"]:
                bad.append((seed, "oracle refused", str(e)))
            continue
        if g["refused"]:
            bad.append((seed, "reference refused, oracle did not", g.get("message")))
        elif g.get("ref_undefined"):
            assert res.ref_oob_probes > 0          # the oracle knows the reference reads past its table here
        elif fuzzplans.digest(kind, res.text) != g["digest"]:
            bad.append((seed, kind, "digest differs"))
    assert not bad, bad


def test_oracle_matches_reference_on_the_seeds_sweeps_found():
    """seeds beyond the suite's range on which a wide GPU sweep once found the engine wrong (tests/golden/sweep_finds.json, written by
    make_sweep_finds_golden.py from the unmodified reference); tests/test_gpu_fuzz.py runs the engine on them"""
    with open(os.path.join(os.path.dirname(__file__), "golden", "sweep_finds.json")) as f:
        finds = json.load(f)
    assert finds
    for seed, g in finds.items():
        plan, kind = fuzzplans.make(int(seed))
        res = orc.execute(plan)
        assert kind == g["kind"] and res.n_rows == g["rows"] and fuzzplans.digest(kind, res.text) == g["digest"], seed


def test_fuzz_covers_the_shapes():
    kinds = [g["kind"] for g in GOLDEN.values() if not g["refused"]]
    assert kinds.count("exact") > 200 and kinds.count("multiset") > 40
    assert 5 < sum(g["refused"] for g in GOLDEN.values()) < 60


@pytest.mark.skipif(not orc.have_reference(), reason="oracle/_ref/ref_harness not built (needs /root/reference)")
def test_fuzz_oracle_matches_live_reference():
    bad = []
    for seed in range(400, 480):                   # seeds beyond the committed goldens
        plan, kind = fuzzplans.make(seed)
        try:
            want, _ = orc.run_reference(plan)
        except orc.OracleError:
            with pytest.raises(orc.OracleError):
                orc.execute(plan)
            continue
        res = orc.execute(plan)
        if not res.ref_oob_probes and not fuzzplans.same(kind, res.text, want):
            bad.append(seed)
    assert not bad, bad
