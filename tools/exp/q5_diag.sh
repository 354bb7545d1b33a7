#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
one() { echo "== $1 | $2"; env $1 timeout -k 10 200 python3 tools/sql_bench.py 10 --repeat 6 --only $2 2>&1 | grep -E '^\{|rsq tail|rsq trace\]     rsq_p' | cut -c1-420 | tail -${3:-3}; }
one "RSQ_DEBUG_TAIL=1" q5 12
one "RSQ_TRACE=1" q5 14
one "RSQ_HASH_LDS=0" q5
one "RSQ_COMPACT_GRID=2" q5
one "RSQ_COMPACT_GRID=4" q5
one "RSQ_COMPACT_GRID=8" q5
one "RSQ_LAZY_COLUMNS=0" q5
one "RSQ_BITMAP_PREFETCH=0" q5
one "RSQ_JOIN_RANK=0" q5
one "RSQ_DEBUG_TAIL=1" q10 12
one "RSQ_COMPACT_GRID=4" q10
one "RSQ_COMPACT_GRID=8" q10
one "RSQ_GROUP_FD=0" q10
