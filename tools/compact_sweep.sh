cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pt in 16 32 64; do
  rm -rf /tmp/prof_c
  RSQ_COMPACT_PT=$pt timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -- python3 $R/tools/profile_case.py q3 10 10 > /tmp/c.log 2>&1
  echo "PT=$pt $(grep k_compact_entries $(find /tmp/prof_c -name '*kernel_stats.csv' | head -1) | sed 's/.*)",//' | cut -c1-60)  $(grep '^q3 kernel_ms' /tmp/c.log | tail -1)"
done
