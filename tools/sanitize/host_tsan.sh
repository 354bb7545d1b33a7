#!/bin/bash
# ThreadSanitizer over the engine's host concurrency: the persistent worker pool (hostpar.cpp), the cluster-parallel replay of
# the reference's hash table (hostref.cpp), the parallel group decoding / merging of the tail (tail.cpp) and the shard threads
# of rsq_multi_* (multi.cpp; on a machine with a GPU).  The library is built again with -fsanitize=thread (host code only: GPU
# sanitizers are not available on the pool) into /tmp, swapped in for the CPU tests that reach those paths through the C ABI -
# emission order (sequential vs cluster-parallel vs the library's own threads), finalisation from merged partial tables, the
# shard statistics, the two-rank gloo merges - and swapped back.  RSQ_TAIL_THREADS=8 makes the pool real on small machines.
# usage: bash tools/sanitize/host_tsan.sh [log file]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
LOG=${1:-${TMPDIR:-/tmp}/rsq_host_tsan.log}
cp $ROOT/resql_amd/libresql_hip.so ${TMPDIR:-/tmp}/libresql_hip.good.so
trap 'cp ${TMPDIR:-/tmp}/libresql_hip.good.so $ROOT/resql_amd/libresql_hip.so' EXIT
B=${TMPDIR:-/tmp}/rsq_host_thread; mkdir -p $B
cd $ROOT/resql_amd/csrc
for f in expr.cpp hostref.cpp hostpar.cpp runtime.cpp codegen.cpp codegen_join.cpp codegen_agg.cpp codegen_loop.cpp tail.cpp engine.cpp engine_pipelines.cpp engine_devtail.cpp generic.cpp generic2.cpp tbl.cpp sqlfront.cpp api.cpp multi.cpp aot_kernels.hip devtail.hip generic_kernels.hip; do
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -fsanitize=thread -fno-gpu-sanitize -fno-omit-frame-pointer --offload-arch=gfx950 -I../../include -I. -x hip -c $f -o $B/$f.o &
    while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.2; done
done
wait
/opt/rocm/bin/hipcc -shared -fPIC -fsanitize=thread -fno-gpu-sanitize --offload-arch=gfx950 -o $B/libresql_hip.so $B/*.o -lhiprtc -ldl -lpthread
RT=$(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.tsan-x86_64.so' | head -n 1)
cp $B/libresql_hip.so $ROOT/resql_amd/libresql_hip.so
cd $ROOT
echo "== -fsanitize=thread ($RT)" | tee $LOG
# (second_deadlock_stack for readable reports; halt_on_error so that a race fails the run; the Python interpreter itself is not
# instrumented, its own threads do not appear)
LD_PRELOAD=$RT RSQ_TAIL_THREADS=8 TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1 report_signal_unsafe=0" python3 -m pytest \
    tests/test_emission_order.py tests/test_engine_host.py tests/test_shard_stats.py \
    -q -m "not gpu" -p no:cacheprovider -k "not plain_c_host and not two_ranks" 2>&1 | tee -a $LOG
# (left out: the plain-C host test would need the sanitizer runtime at link time; the two-rank gloo tests spend their threads inside
# torch's ProcessGroupGloo, which is not instrumented and reports races of its own - the engine calls they make, finalisation from a
# merged table, run single-threaded there and multi-threaded in the tests above)
