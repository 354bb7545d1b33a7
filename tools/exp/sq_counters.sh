#!/bin/bash
# Instruction and cycle counters of the join statements' kernels (is stage 1 issue-bound or memory-bound?).  Separate --pmc passes.
# usage: bash tools/exp/sq_counters.sh OUTDIR q10,q5,q12
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; ONLY=${2:-q10}
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT" ;; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU"; do
    i=$((i + 1))
    rm -rf /tmp/sqc_$i
    timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/sqc_$i -- python3 "$ROOT/tools/sql_bench.py" 10 --repeat 3 --only $ONLY > /tmp/sqc_$i.log 2>&1 || echo "pass $i failed: $(tail -n 3 /tmp/sqc_$i.log)"
done
python3 - "$OUT/sq_counters.txt" <<'PY'
import csv, glob, sys
acc = {}
for d in sorted(glob.glob("/tmp/sqc_*")):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        per = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if not (k.startswith("rsq_p")): continue
            per.setdefault((k, r["Counter_Name"]), {}).setdefault(r["Dispatch_Id"], 0.0)
            per[(k, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for (k, c), dd in per.items():
            vals = sorted(dd.values())
            acc.setdefault(k, {})[c] = vals[len(vals) // 2]          # median dispatch
with open(sys.argv[1], "w") as o:
    for k in sorted(acc):
        o.write(k + "\n")
        for c, v in sorted(acc[k].items()): o.write(f"    {c:28s} {v:16.0f}\n")
print(open(sys.argv[1]).read())
PY
