"""The newest round's files under profiles/ carry the commit they were collected at (tools/stamp_profiles.py, the last step of
tools/collect_profiles.sh), all of them the same one, and that commit is in this history."""
import glob
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import stamp_profiles  # noqa: E402


def test_stamp_round_trip(tmp_path):
    for name, text in (("r99_a.json", '{"x": 1}'), ("r99_b.jsonl", '{"y": 2}\n{"y": 3}\n'), ("r99_c.csv", "a,b\n1,2\n"), ("r99_d.json", "[1, 2]")):
        (tmp_path / name).write_text(text)
    for sha in ("abc", "def"):                                   # a second stamp replaces the first
        for p in glob.glob(str(tmp_path / "r99_*")):
            stamp_profiles.stamp(p, sha)
        assert {stamp_profiles.read_stamp(p) for p in glob.glob(str(tmp_path / "r99_*"))} == {sha}
    assert (tmp_path / "r99_c.csv").read_text() == "a,b\n1,2\n# head def\n"
    assert (tmp_path / "r99_b.jsonl").read_text().count("head_sha") == 1


def test_the_newest_round_is_stamped_with_one_commit_of_this_history():
    files = [p for p in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*")) if os.path.isfile(p)]
    newest = max(int(re.match(r"r(\d+)_", os.path.basename(p)).group(1)) for p in files)
    if newest < 4:
        return                                                   # (stamping began in round 4)
    cur = [p for p in files if os.path.basename(p).startswith("r%02d_" % newest)]
    stamps = {os.path.basename(p): stamp_profiles.read_stamp(p) for p in cur}
    assert all(stamps.values()), sorted(k for k, v in stamps.items() if not v)
    assert len(set(stamps.values())) == 1, stamps
    sha = next(iter(stamps.values()))
    if os.path.isdir(os.path.join(ROOT, ".git")):
        r = subprocess.run(["git", "-C", ROOT, "merge-base", "--is-ancestor", sha, "HEAD"], capture_output=True)
        assert r.returncode == 0, f"profiles of round {newest} are stamped with {sha}, which is not an ancestor of HEAD"


def _newest_round():
    files = [p for p in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*")) if os.path.isfile(p)]
    return max(int(re.match(r"r(\d+)_", os.path.basename(p)).group(1)) for p in files)


def _plausible_q1_pmc(pmc: dict):
    """a --pmc pass over the headline kernel measured that kernel: as many dispatches as the command launches, all alike, and not
    fewer bytes than a full scan must read (round 4's file averaged 262 SF1 + SF10 dispatches: 0.217 x; five collections, nobody looked)"""
    assert "invalid" not in pmc, pmc["invalid"]
    ratio = pmc["hbm_read_bytes_per_launch_corrected"] / pmc["algorithmic_bytes_per_launch"]
    assert 0.98 <= ratio <= 1.5, ratio
    if "expected_launches" in pmc:
        assert pmc["launches"] == pmc["expected_launches"]


def test_the_plausibility_check_refuses_round_4s_polluted_pass():
    import json
    with open(os.path.join(ROOT, "profiles", "r04_q1_sf10_pmc.json")) as f:
        bad = json.load(f)
    with pytest.raises(AssertionError):
        _plausible_q1_pmc(bad)
    with open(os.path.join(ROOT, "profiles", "r03_q1_sf10_pmc.json")) as f:
        _plausible_q1_pmc(json.load(f))
    # bench.py quotes the newest committed pass that passes the same check, and says which one it refused
    sys.path.insert(0, ROOT)
    import bench
    traffic, source = bench.committed_traffic(38 * 59999996)
    assert traffic is not None and 0.98 <= traffic / (38 * 59999996) <= 1.5
    assert "r04_q1_sf10_pmc.json" not in source.split("; refused:")[0]


def test_the_newest_rounds_counter_files_are_plausible_and_its_kernel_average_fits_the_step():
    """from round 5 on: every committed *_pmc.json of the newest round passes the plausibility bound of its kind, and the rocprofv3
    average of the headline kernel is not longer than the whole step of the same collection's bench line"""
    import csv
    import json
    newest = _newest_round()
    if newest < 5:
        return
    tag = "r%02d_" % newest
    prof = os.path.join(ROOT, "profiles")
    q1 = os.path.join(prof, tag + "q1_sf10_pmc.json")
    assert os.path.exists(q1), "the headline kernel's counter pass is missing"
    with open(q1) as f:
        _plausible_q1_pmc(json.load(f))
    for p in glob.glob(os.path.join(prof, tag + "*_pmc.json")):
        with open(p) as f:
            d = json.load(f)
        for case in d.get("cases", {}).values():                   # late loads: fetched bytes never exceed what the columns hold by much
            assert 0.0 < case["traffic_frac_of_8TBps"] <= 1.0, (p, case)
            assert case["hbm_read_bytes_corrected"] <= 1.5 * case["algorithmic_bytes"], (p, case)
        for sel in d.get("selectivity", {}).values():              # staged partitioning: the input once (less what late loads skip at 10 %) + records twice
            assert 0.8 <= sel["over_algorithmic"] <= 3.0, (p, sel["over_algorithmic"])
    # (the line of the PROFILED process itself: the same launches, the same minute of the same box)
    bench_line = os.path.join(prof, tag + "q1_sf10_kernel_stats_run.json")
    stats = os.path.join(prof, tag + "q1_sf10_kernel_stats.csv")
    if os.path.exists(bench_line) and os.path.exists(stats):
        with open(bench_line) as f:
            line = json.load(f)                                    # (one JSON object; tools/stamp_profiles.py re-writes it indented)
        with open(stats) as f:
            rows = [r for r in csv.DictReader(l for l in f if not l.startswith("#")) if r["Name"].startswith("rsq_p0_lineitem_aggregate")]
        assert rows, "the headline kernel is not in the kernel statistics"
        avg_ms = float(rows[0]["AverageNs"]) / 1e6
        assert avg_ms <= line["ms_per_step"] * 1.02, (avg_ms, line["ms_per_step"])
        assert abs(avg_ms - line["roofline"]["kernel_ms"]) <= 0.05 * line["roofline"]["kernel_ms"], (avg_ms, line["roofline"]["kernel_ms"])
