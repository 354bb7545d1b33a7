"""The closest thing to N > 1 a one-GPU box allows (VERDICT r04 "next round" 5): `bench.py --gpus 2 --share-gpu` starts two FRESH rank
processes that both drive device 0 — each its own engine context, its own row-range shard of lineitem, the real kernels, the shards'
statistics unified across the processes, the asynchronous partial execution, the partial tables staged through host memory and merged
over gloo, rank 0 finalising — and compares the answer with the unmodified reference's (tests/golden/ref_full_*_sf1.tbl).  The same for
TPC-H Q3's key-aligned sharding (replicated build sides, every rank's top 10 merged by the sort keys).
The reference's fan-out is JitContextFlounder::execute (reference src/JitContextFlounder.h:459-487): N workers on one compiled plan."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=420, env=env)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, pr.stdout[-1000:]
    return json.loads(lines[0])


def test_two_rank_processes_share_the_gpu_q1():
    line = _bench("--gpus", "2", "--share-gpu", "--sf", "1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extras")
    c = line["config"]
    assert line["parity_checked"] is True and "ref_full_q1_sf1.tbl" in line["parity_source"]
    assert c["ranks"] == 2 and c["world_size"] == 2 and c["share_gpu"] is True and c["rccl_ranks_seen"] == 2 and c["self_launched"] is True
    assert c["rows_per_gpu"] * 2 <= c["rows"] < c["rows_per_gpu"] * 2 + 256            # two row-range shards on tile boundaries
    assert len(c["phases"]["kernel_ms_per_rank"]) == 2 and all(v > 0 for v in c["phases"]["kernel_ms_per_rank"])
    assert "gloo" in c["backend"] and line["n_gpus"] == 1


def test_two_rank_processes_share_the_gpu_q3_key_aligned():
    line = _bench("--gpus", "2", "--share-gpu", "--workload", "q3", "--sf", "1", "--steps", "3", "--warmup", "1")
    c = line["config"]
    assert line["parity_checked"] is True and "ref_full_q3_sf1.tbl" in line["parity_source"]
    assert c["ranks"] == 2 and c["share_gpu"] is True and len(c["lineitem_rows_per_rank"]) == 2
    assert sum(c["lineitem_rows_per_rank"]) == 5999980                                  # every lineitem row lives on exactly one rank
    assert all(v > 0 for v in c["phases"]["kernel_ms_per_rank"])
