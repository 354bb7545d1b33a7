R=$GRAFT_REPO_ROOT
for v in "RSQ_BITMAP_PREFETCH=1" "RSQ_BITMAP_PREFETCH=2" "RSQ_BITMAP_PREFETCH=0" "RSQ_QCAP=128"; do
  echo "== $v"
  env $v timeout -k 10 100 python3 $R/tools/profile_case.py q3 10 6 2>&1 | tail -n 2
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_q3
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q3 -- python3 $R/tools/profile_case.py q3 10 12 > /tmp/q3prof.log 2>&1
cut -c1-160 "$(find /tmp/prof_q3 -name '*kernel_stats.csv' | head -1)" | grep -v "k_gen\|k_minmax\|k_byteset"
