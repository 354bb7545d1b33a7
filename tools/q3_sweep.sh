cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_q3
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q3 -- python3 $R/tools/profile_case.py q3 10 12 > /tmp/q3prof.log 2>&1
cp "$(find /tmp/prof_q3 -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/q3_kernel_stats.csv
cp "$(find /tmp/prof_q3 -name '*kernel_trace.csv' | head -1)" $R/gpurun_out/q3_kernel_trace.csv
cut -c1-200 $R/gpurun_out/q3_kernel_stats.csv
for v in "RSQ_MAXGRID=512" "RSQ_MAXGRID=1024" "RSQ_MAXGRID=4096" "RSQ_UNROLL=1" "RSQ_UNROLL=2" "RSQ_UNROLL=4" "RSQ_BLOCK=512" "RSQ_BLOCK=1024"; do
  echo "== $v"
  env $v RSQ_TRACE=1 timeout -k 10 100 python3 $R/tools/profile_case.py q3 10 3 2>&1 | grep -E "^\[rsq trace\] [0-9.]+ ms" | tail -3 | cut -c1-60
done
