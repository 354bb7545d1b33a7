#!/bin/bash
# late-load form variants on one GPU: kernel ms of the synthetic 1.25 B-row shard (G = 8, 1 % and 10 %) and TPC-H Q6 at SF10
cd "${GRAFT_REPO_ROOT:-.}"
run() {  # run "ENV..." workload...
    local envs="$1"; shift
    local out
    out=$(env $envs RSQ_KERNEL_CACHE_TAG=$RANDOM timeout -k 10 180 python3 tools/profile_case.py "$@" 2>&1 | grep kernel_ms | awk '{print $3}' | sort -n | head -1)
    echo "$envs | $* | best kernel_ms $out"
}
for w in "synthetic 1250000000 8 0.01 4" "synthetic 1250000000 8 0.1 4" "q6 10 4"; do
    run "RSQ_LATE_PIPELINE=0 RSQ_UNROLL=2" $w
    run "RSQ_LATE_PIPELINE=1 RSQ_UNROLL=2" $w
    run "RSQ_LATE_PIPELINE=1 RSQ_UNROLL=3" $w
    run "RSQ_LATE_PIPELINE=1 RSQ_UNROLL=4" $w
    run "RSQ_LATE_PIPELINE=1 RSQ_UNROLL=2 RSQ_BLOCK=256" $w
    run "RSQ_LATE_PIPELINE=1 RSQ_UNROLL=3 RSQ_BLOCK=256" $w
    run "RSQ_LATE_PIPELINE=1 RSQ_UNROLL=4 RSQ_BLOCK=256" $w
    run "RSQ_LATE_PIPELINE=0 RSQ_UNROLL=2 RSQ_BLOCK=256" $w
done
