cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 100 --warmup 10 2>&1 | tail -1 > gpurun_out/bench_n1_b.json; cat gpurun_out/bench_n1_b.json
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_q1_u2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_q1_u2.log 2>&1
cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log 2>&1
cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R gpurun_out | head -40
