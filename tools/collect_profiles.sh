#!/bin/bash
# Collect the per-round profile artefacts on the GPU box (run through gpurun from the repo root):
#   bench line, rocprofv3 kernel statistics of the same command, and a separate --pmc FETCH_SIZE pass.
# rocprofv3 is run from /tmp with TMPDIR=/tmp (it hangs when started inside the repo snapshot), every step under a hard timeout.
# usage: bash tools/collect_profiles.sh rNN
set -u
R=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 python3 $ROOT/bench.py > $OUT/${R}_bench_n1.log 2>&1
grep '^{' $OUT/${R}_bench_n1.log | tail -1 > $OUT/${R}_bench_n1.json
rm -rf /tmp/prof_ks /tmp/prof_pmc
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -- python3 $ROOT/bench.py --no-cpu-baseline > /tmp/ks.log 2>&1
cp "$(find /tmp/prof_ks -name '*kernel_stats.csv' | head -1)" $OUT/${R}_q1_sf10_kernel_stats.csv
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_pmc -- python3 $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 1 > /tmp/pmc.log 2>&1
python3 - "$(find /tmp/prof_pmc -name '*counter_collection.csv' | head -1)" $OUT/${R}_q1_sf10_pmc.json <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("rsq_pipeline") and r["Counter_Name"] == "FETCH_SIZE"]
per_dispatch = {}
for r in rows:
    per_dispatch[r["Dispatch_Id"]] = per_dispatch.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
n = len(per_dispatch)
kb = sum(per_dispatch.values()) / max(n, 1)
json.dump({"kernel": "rsq_pipeline (TPC-H Q1 SF10: scan 7 columns + filter + 6-group aggregation)", "launches": n,
           "FETCH_SIZE_KB_per_launch": kb,
           "note": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md, HBM section) -> x2; collected in its own --pmc pass",
           "hbm_read_bytes_per_launch_corrected": kb * 1024 * 2,
           "algorithmic_bytes_per_launch": 38 * 59999996}, open(sys.argv[2], "w"), indent=1)
print("pmc launches", n, "corrected bytes", kb * 1024 * 2)
PY
# TPC-H Q3 at SF10 (BASELINE config 3): kernel statistics of a few executions, and the eight SQL statements at SF1
rm -rf /tmp/prof_q3
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q3 -- python3 $ROOT/tools/profile_case.py q3 10 12 > $OUT/${R}_q3_sf10_runs.log 2>&1
cp "$(find /tmp/prof_q3 -name '*kernel_stats.csv' | head -1)" $OUT/${R}_q3_sf10_kernel_stats.csv
timeout -s KILL 300 python3 $ROOT/tools/sql_bench.py 1 --reference > $OUT/${R}_sql_sf1.log 2>&1
grep '^{' $OUT/${R}_sql_sf1.log > $OUT/${R}_sql_sf1.jsonl
head -3 $OUT/${R}_q1_sf10_kernel_stats.csv | cut -c1-150
head -4 $OUT/${R}_q3_sf10_kernel_stats.csv | cut -c1-150
cut -c1-400 $OUT/${R}_bench_n1.json
