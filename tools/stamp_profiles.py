#!/usr/bin/env python3
"""Records the commit a round's profile files were collected at INSIDE every one of them.

usage: python tools/stamp_profiles.py DIR rNN SHA
  *.json  -> key "head_sha" (objects) ; *.jsonl -> a last line {"head_sha": ...} ; anything else (csv, txt, log) -> a last line "# head <sha>".
A file that already carries a stamp gets the new one in its place.  tools/collect_profiles.sh runs this as its last step (HEAD_SHA in the
environment: the GPU box has no .git); tests/test_profiles_stamped.py checks that the stamps of the newest round agree with each other."""
import glob
import json
import os
import sys


def stamp(path: str, sha: str) -> None:
    if path.endswith(".json"):
        try:
            d = json.load(open(path))
        except Exception:
            d = None
        if isinstance(d, dict):
            d["head_sha"] = sha
            json.dump(d, open(path, "w"), indent=1)
            return
    lines = open(path, errors="replace").read().splitlines()
    lines = [l for l in lines if not l.startswith("# head ") and not l.startswith('{"head_sha"')]
    lines.append(json.dumps({"head_sha": sha}) if path.endswith((".jsonl", ".json")) else "# head " + sha)
    open(path, "w").write("\n".join(lines) + "\n")


def read_stamp(path: str):
    if path.endswith(".json"):
        try:
            d = json.load(open(path))
            if isinstance(d, dict):
                return d.get("head_sha")
        except Exception:
            pass
    for l in reversed(open(path, errors="replace").read().splitlines()):
        if l.startswith("# head "):
            return l[len("# head "):].strip()
        if l.startswith('{"head_sha"'):
            return json.loads(l)["head_sha"]
    return None


if __name__ == "__main__":
    d, rnd, sha = sys.argv[1], sys.argv[2], sys.argv[3]
    n = 0
    for p in sorted(glob.glob(os.path.join(d, rnd + "_*"))):
        if os.path.isfile(p):
            stamp(p, sha); n += 1
    print(f"   stamped {n} file(s) of {rnd} with {sha}")
