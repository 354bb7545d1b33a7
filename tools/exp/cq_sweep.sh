#!/bin/bash
# compaction-pipeline loop variants at SF10: the reference's TPC-H statements, whole execution and device ms
cd "${GRAFT_REPO_ROOT:-.}"
for v in "RSQ_CQ_PIPELINE=0" "RSQ_CQ_PIPELINE=1" "RSQ_CQ_PIPELINE=1 RSQ_CQ_PIPELINE_REGS=64" "RSQ_CQ_PIPELINE=1 RSQ_COMPACT_UNROLL=2" "RSQ_CQ_PIPELINE=1 RSQ_COMPACT_UNROLL=6"; do
    echo "== $v"
    env $v timeout -k 10 400 python3 tools/sql_bench.py 10 --repeat 6 2>&1 | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['query'], 'kernel_ms', d['kernel_ms'], 'exec_ms', d['exec_ms'], 'kernels', d['kernels'], 'ok', d.get('equals_reference_answer'))"
done
