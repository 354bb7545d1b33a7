#!/bin/bash
# per-dispatch durations of one statement at SF10 (the split form's two kernels carry one name: the trace tells them apart by order)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
q=${1:-q5}
rm -rf /tmp/prof_split
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_split -- python3 $R/tools/sql_bench.py 10 --repeat 4 --only $q > /tmp/split.log 2>&1
f=$(find /tmp/prof_split -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-40:]
prev_end = None
for r in tail:
    gap = (int(r["Start_Timestamp"]) - prev_end) / 1e3 if prev_end is not None else 0.0
    prev_end = int(r["End_Timestamp"])
    g = r.get('Grid_Size') or r.get('Grid_Size_X') or '?'
    w = r.get('Workgroup_Size') or r.get('Workgroup_Size_X') or '?'
    print(f"{r['Kernel_Name'][:50]:50s} grid {g:>8s} wg {w:>4s} vgpr {r.get('VGPR_Count','?'):>4s} lds {r.get('LDS_Block_Size','?'):>6s} us {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}  idle before {gap:7.1f}")
PY
grep '^{' /tmp/split.log | cut -c1-160
