"""Run fuzz seeds one per subprocess with a timeout (development aid: finds the seed behind a hang or a crash).
usage: python tools/fuzz_seeds.py FIRST LAST [TIMEOUT_S]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import fuzzplans
from resql_amd import engine
from oracle import orc
seed = int(sys.argv[1])
plan, kind = fuzzplans.make(seed)
try:
    want = orc.execute(plan)
except orc.OracleError as e:
    want = None
ctx = engine.Context(device=0)
try:
    got = ctx.run(plan)
except engine.EngineError as e:
    print("refused-both" if want is None else "ENGINE-REFUSED " + str(e)); sys.exit(0)
if want is None:
    print("ORACLE-REFUSED-ONLY")
else:
    print("ok" if fuzzplans.same(kind, got.text, want.text) else "DIFF %%d vs %%d rows" %% (got.n_rows, want.n_rows))
""" % (ROOT, ROOT)

first, last = int(sys.argv[1]), int(sys.argv[2])
tmo = float(sys.argv[3]) if len(sys.argv) > 3 else 60
for seed in range(first, last):
    try:
        pr = subprocess.run([sys.executable, "-c", CHILD, str(seed)], capture_output=True, text=True, timeout=tmo,
                            env=dict(os.environ, RSQ_TRACE="1"))
        out = pr.stdout.strip().splitlines()
        status = out[-1] if out else f"rc={pr.returncode} " + pr.stderr[-300:]
    except subprocess.TimeoutExpired as e:
        status = "TIMEOUT; stderr tail: " + ((e.stderr or b"").decode(errors="replace")[-600:])
    if status != "ok" and status != "refused-both":
        print(seed, status, flush=True)
print("done", first, last, flush=True)
