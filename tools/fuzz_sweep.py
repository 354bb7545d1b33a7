"""One-off wide sweep of fuzz seeds on the GPU against the oracle, in process (development aid; the committed suite runs
seeds 0..399, pinned by reference digests).  usage: python tools/fuzz_sweep.py FIRST LAST"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzzplans  # noqa: E402
from resql_amd import engine  # noqa: E402
from oracle import orc  # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
ctx = engine.Context(device=0)
bad, refused, t0 = [], 0, time.time()
for seed in range(first, last):
    plan, kind = fuzzplans.make(seed)
    try:
        want = orc.execute(plan)
    except orc.OracleError:
        want = None
    try:
        got = ctx.run(plan)
    except engine.EngineError as e:
        if want is not None:
            bad.append((seed, "engine refused", str(e)[:200]))
        else:
            refused += 1
        continue
    if want is None:
        bad.append((seed, "oracle refused, engine did not", ""))
    elif not fuzzplans.same(kind, got.text, want.text):
        bad.append((seed, kind, f"{got.n_rows} vs {want.n_rows} rows"))
    if (seed - first) % 100 == 99:
        print(f"... {seed + 1 - first} seeds, {len(bad)} bad, {time.time() - t0:.0f} s", flush=True)
print("seeds", last - first, "refused by both", refused, "BAD", len(bad))
for b in bad:
    print(b)
ctx.close()
