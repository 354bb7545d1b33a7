set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8
python bench.py --steps 50 --warmup 5 2>&1 | tee gpurun_out/bench_n1.json | tail -3
for U in 1 2 3 6 8; do RSQ_UNROLL=$U python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('U=$U', 'value', round(d['value']/1e9,2), 'Grows/s  kernel_ms', round(d['roofline']['kernel_ms'],4), 'GB/s', round(d['roofline']['achieved'],1), 'ms/step', round(d['ms_per_step'],4))"; done
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_q1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_q1.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/prof_q1 -name "*stats*" | head; tail -2 gpurun_out/prof_q1.log
