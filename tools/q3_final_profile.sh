cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/profiles
rm -rf /tmp/prof_q3
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q3 -- python3 $R/tools/profile_case.py q3 10 12 > /tmp/q3_runs.log 2>&1
grep '^q3 ' /tmp/q3_runs.log > $R/gpurun_out/profiles/r03_q3_sf10_runs.txt
f=$(find /tmp/prof_q3 -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/profiles/r03_q3_sf10_kernel_stats.csv
RSQ_DEBUG_TAIL=1 timeout -k 10 200 python3 $R/tools/profile_case.py q3 10 8 2>&1 | grep "rsq tail" | tail -6 > $R/gpurun_out/profiles/r03_q3_sf10_workgroup_timestamps.txt
RSQ_EXEC_TRACE=1 timeout -k 10 200 python3 $R/tools/profile_case.py q3 10 24 2>&1 | grep "rsq exec" | tail -2 >> $R/gpurun_out/profiles/r03_q3_sf10_workgroup_timestamps.txt
cut -c1-200 $R/gpurun_out/profiles/r03_q3_sf10_workgroup_timestamps.txt
