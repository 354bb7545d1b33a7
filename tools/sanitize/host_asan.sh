#!/bin/bash
# The engine's whole host side (typing, pipeline extraction, code generation, SQL front end, '.tbl' ingest, control statements,
# hiprtc compile path of a context without a device) under AddressSanitizer and then UndefinedBehaviorSanitizer: the library is
# built again with -fsanitize=... -fno-gpu-sanitize (host code only; GPU sanitizers are not available on the pool) into /tmp,
# swapped in for the duration of the CPU tests that drive it through the C ABI, and swapped back.
# usage: bash tools/sanitize/host_asan.sh
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cp $ROOT/resql_amd/libresql_hip.so ${TMPDIR:-/tmp}/libresql_hip.good.so
trap 'cp ${TMPDIR:-/tmp}/libresql_hip.good.so $ROOT/resql_amd/libresql_hip.so' EXIT
for SAN in address undefined; do
    B=${TMPDIR:-/tmp}/rsq_host_$SAN; mkdir -p $B
    EXTRA=""; [ $SAN = undefined ] && EXTRA="-fno-sanitize=vptr"
    cd $ROOT/resql_amd/csrc
    for f in expr.cpp hostref.cpp hostpar.cpp runtime.cpp codegen.cpp codegen_join.cpp codegen_agg.cpp codegen_loop.cpp tail.cpp engine.cpp engine_pipelines.cpp engine_devtail.cpp generic.cpp generic2.cpp tbl.cpp sqlfront.cpp api.cpp multi.cpp aot_kernels.hip devtail.hip generic_kernels.hip; do
        /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -fsanitize=$SAN $EXTRA -fno-gpu-sanitize -fno-omit-frame-pointer --offload-arch=gfx950 -I../../include -I. -x hip -c $f -o $B/$f.o
    done
    /opt/rocm/bin/hipcc -shared -fPIC -fsanitize=$SAN $EXTRA -fno-gpu-sanitize --offload-arch=gfx950 -o $B/libresql_hip.so $B/*.o -lhiprtc -ldl -lpthread
    if [ $SAN = address ]; then RT=$(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -n 1)
    else RT=$(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.ubsan_standalone-x86_64.so' | head -n 1); fi
    cp $B/libresql_hip.so $ROOT/resql_amd/libresql_hip.so
    cd $ROOT
    echo "== -fsanitize=$SAN"
    # (the test that links a plain-C host against the library is left out: it would need the sanitizer runtime at link time)
    LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python3 -m pytest tests/test_engine_host.py \
        tests/test_late_loads.py tests/test_like.py tests/test_sql_frontend.py tests/test_control_statements.py tests/test_tbl_ingest.py -q -m "not gpu" \
        -k "not plain_c_host and not links" -p no:cacheprovider
    cp ${TMPDIR:-/tmp}/libresql_hip.good.so $ROOT/resql_amd/libresql_hip.so
done
