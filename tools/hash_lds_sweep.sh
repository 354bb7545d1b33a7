# LDS front table of the generic hash aggregation (RSQ_HASH_LDS_SLOTS, RSQ_HASH_LDS=0) against group counts; 100 M rows
run() { echo "== $*"; env "$@" RSQ_TEST_GROUPS=64,1024,16384 timeout -k 10 200 python tools/generic_paths.py 100000000 2>&1 | grep "hash agg" | grep "sel=0.5" | cut -c1-92; }
run RSQ_HASH_LDS=1
run RSQ_HASH_LDS=0
