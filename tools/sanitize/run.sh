#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the SQL front end (tokens, grammar, planner) on the CPU: the host-only
# sources (sqlfront.cpp, expr.cpp) are compiled with g++ into two small drivers and fed thousands of statements (the grammar
# fuzzer, the valid-statement generator, the hand-written cases, byte-level damage).  GPU sanitizers are not available on the
# pool; the device code is covered by the parity tests instead.     usage: bash tools/sanitize/run.sh
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=${TMPDIR:-/tmp}/rsq_sanitize; mkdir -p $W
python3 - "$ROOT" "$W" <<'PY'
import random, sys
root, w = sys.argv[1], sys.argv[2]
sys.path.insert(0, root); sys.path.insert(0, root + "/tests")
import sqlfuzz, sqlgen, sqlcases
from resql_amd import tpch_full
out = [sqlfuzz.statement(s) for s in range(4000)] + [sqlgen.statement(s) for s in range(1000)]
out += sqlcases.TOKEN_CASES + sqlcases.PARSE_CASES + sqlcases.PLAN_CASES
r = random.Random(7)
for s in range(2000):
    t = list(sqlfuzz.statement(s))
    for _ in range(r.randrange(1, 4)):
        t[r.randrange(len(t))] = r.choice("()'\"*,.-:<>=%_ \tabz019")
    out.append("".join(t))
open(w + "/stmts.txt", "w").write("\n".join(x.replace("\n", "\x01") for x in out) + "\n")
db = tpch_full.database(0.001)
with open(w + "/schema.txt", "w") as f:
    for k in sorted(db):
        t = db[k]
        f.write(f"table {t.name} {t.n_rows * 1000} {len(t.columns)}\n")
        for c in t.columns:
            f.write(f"{c.name} {c.type.tag} {c.type.precision} {c.type.scale} {c.type.len}\n")
plans = [sqlgen.statement(s) for s in range(3000)] + list(tpch_full.QUERIES.values()) + sqlcases.PLAN_CASES
open(w + "/plans.txt", "w").write("\n".join(" ".join(x.split()) for x in plans) + "\n")
PY
FLAGS="-std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$ROOT/include -I$ROOT/resql_amd/csrc"
g++ $FLAGS $ROOT/tools/sanitize/parse_driver.cpp $ROOT/resql_amd/csrc/sqlfront.cpp $ROOT/resql_amd/csrc/expr.cpp -o $W/parse_driver
g++ $FLAGS $ROOT/tools/sanitize/plan_driver.cpp $ROOT/resql_amd/csrc/sqlfront.cpp $ROOT/resql_amd/csrc/expr.cpp -o $W/plan_driver
$W/parse_driver $W/stmts.txt
ASAN_OPTIONS=detect_leaks=0 $W/plan_driver $W/schema.txt $W/plans.txt     # (the tables are never freed: ~Table lives in the HIP part of the library)
# '.tbl' ingest: well-formed files and damaged ones (missing / extra fields, bad numbers, no trailing newline, empty lines)
g++ $FLAGS -pthread $ROOT/tools/sanitize/tbl_driver.cpp $ROOT/resql_amd/csrc/tbl.cpp $ROOT/resql_amd/csrc/expr.cpp -o $W/tbl_driver
python3 - "$W" <<'PY'
import random, sys
w = sys.argv[1]
r = random.Random(11)
rows = [f"{i}|name {i}|{r.randrange(-99999, 999999) / 100:.2f}|19{r.randrange(92, 99)}-{r.randrange(1, 13):02d}-{r.randrange(1, 29):02d}|{'ab'[i % 2]}|" for i in range(20000)]
open(w + "/good.tbl", "w").write("\n".join(rows) + "\n")
open(w + "/nonl.tbl", "w").write("\n".join(rows[:100]))
bad = list(rows[:3000])
for k in range(300):
    i = r.randrange(len(bad)); m = r.randrange(5)
    f = bad[i].split("|")
    if m == 0: del f[r.randrange(len(f) - 1)]
    elif m == 1: f.insert(r.randrange(len(f)), "extra")
    elif m == 2: f[0] = "12x" + f[0]
    elif m == 3: f[2] = "1.2.3"
    else: f = [""]
    bad[i] = "|".join(f)
    open(w + f"/bad{k % 6}.tbl", "w").write("\n".join(bad[max(0, i - 50):i + 50]) + "\n")
PY
for f in good nonl bad0 bad1 bad2 bad3 bad4 bad5; do $W/tbl_driver $W/$f.tbl '|' INT VARCHAR:12 DECIMAL:12:2 DATE CHAR:1 | tr '\n' ' '; echo; done
