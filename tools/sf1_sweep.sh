R=$GRAFT_REPO_ROOT
for v in "X=1" "RSQ_UNROLL=2" "RSQ_UNROLL=3" "RSQ_MAXGRID=1024" "RSQ_MAXGRID=2048" "RSQ_MAXGRID=1024 RSQ_UNROLL=2" "RSQ_BLOCK=256" "RSQ_BLOCK=1024"; do
  echo "== $v"
  env $v timeout -k 10 200 python3 $R/bench.py --sf 1 --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1000,1), 'us/step  kernel', round(d['roofline']['kernel_ms']*1000,1), 'us')"
done
