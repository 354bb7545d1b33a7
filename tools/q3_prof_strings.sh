cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 1 0; do
rm -rf /tmp/prof_q3
RSQ_STRING_PREFETCH=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q3 -- python3 $R/tools/profile_case.py q3 10 12 > /tmp/q3prof.log 2>&1
echo "== RSQ_STRING_PREFETCH=$v"
f=$(find /tmp/prof_q3 -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/q3_kernel_stats_sp$v.csv
grep -v "k_gen\|k_minmax\|k_byteset\|rocclr" "$f" | cut -d, -f1-4 | cut -c1-110
done
