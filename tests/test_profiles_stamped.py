"""The newest round's files under profiles/ carry the commit they were collected at (tools/stamp_profiles.py, the last step of
tools/collect_profiles.sh), all of them the same one, and that commit is in this history."""
import glob
import os
import re
import subprocess
import sys

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
