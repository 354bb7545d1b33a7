R=$GRAFT_REPO_ROOT
for v in "X=1" "RSQ_NT=0" "RSQ_NT=0 RSQ_UNROLL=2" "RSQ_NT=0 RSQ_MAXGRID=1024" "X=2"; do
  echo "== $v"
  env $v timeout -k 10 200 python3 $R/bench.py --sf 1 --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1000,1), 'us/step  kernel', round(d['roofline']['kernel_ms']*1000,1), 'us')"
done
