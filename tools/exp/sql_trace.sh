# per-dispatch kernel trace of SQL statements at SF10, in launch order with durations: bash tools/exp/sql_trace.sh q12,q3 [OUTDIR]
# (the last execution of every statement is printed: name, duration in us)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=${2:-$R/gpurun_out/r05x}
mkdir -p $OUT
rm -rf /tmp/trace_sql
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_sql -- python3 $R/tools/sql_bench.py 10 --repeat 4 --only ${1:-q12} > /tmp/sqlt.log 2>&1
f=$(find /tmp/trace_sql -name '*kernel_trace.csv' | head -1)
python3 - "$f" > $OUT/sql_trace_${1:-q12}.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# split into executions at gaps > 200 us, print the last execution of each distinct kernel-name sequence
execs, cur, last_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and s - last_end > 200000 and cur:
        execs.append(cur); cur = []
    cur.append((r["Kernel_Name"][:70], (e - s) / 1e3, (s - last_end) / 1e3 if last_end else 0.0)); last_end = e
if cur: execs.append(cur)
seen = {}
for ex in execs:
    seen[tuple(k for k, _, _ in ex)] = ex
for key, ex in seen.items():
    if len(ex) > 40 or any("k_gen_" in k or "k_minmax" in k or "k_byteset" in k for k in key): continue
    print(f"--- execution of {len(ex)} launches, {sum(d for _, d, _ in ex):.1f} us in kernels, {sum(g for _, _, g in ex[1:]):.1f} us between them")
    for k, d, g in ex: print(f"  {d:8.1f} us  (+{g:5.1f})  {k}")
PY
grep '^{' /tmp/sqlt.log | cut -c1-160
