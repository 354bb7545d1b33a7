# per-kernel statistics of some of the SQL statements at SF10: bash tools/sql_prof.sh q5,q10
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_sql
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sql -- python3 $R/tools/sql_bench.py 10 --repeat 5 --only ${1:-q5,q10} > /tmp/sqlp.log 2>&1
f=$(find /tmp/prof_sql -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/sql_kernel_stats_${1:-q5,q10}.csv
grep '^{' /tmp/sqlp.log | cut -c1-200
