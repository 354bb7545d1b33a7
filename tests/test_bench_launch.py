"""bench.py's own way into N > 1 ranks, on CPU: `python bench.py --gpus 2` (no launcher) must start two fresh rank
processes itself, rendezvous them, run the sharding / layout check / merge / finalize plumbing and relay rank 0's JSON
line; the same under torch.distributed.run (how the driver launches N > 1); a failing rank must fail the whole call.
`--backend gloo --no-gpu` is bench.py's dry mode: same code path up to the device work (see its docstring)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _line(stdout: str) -> dict:
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def _check(rec: dict, world: int, self_launched: bool):
    assert rec["dry_run"] is True and rec["n_gpus"] == world
    cfg = rec["config"]
    assert cfg["world_size"] == world and cfg["backend"] == "gloo" and cfg["self_launched"] is self_launched
    assert cfg["merged_ok"] is True and cfg["result_groups"] == 6
    assert cfg["phases"]["kernel_ms_per_rank"] == [float(r + 1) for r in range(world)]      # per-rank figures arrive in rank order
    shards = cfg["shards"]
    assert shards[0][0] == 0 and sum(n for _, n in shards) == cfg["rows"] == 59_999_996
    for (a0, an), (b0, _) in zip(shards, shards[1:]):
        assert a0 + an == b0 and b0 % 128 == 0


def test_bench_starts_its_own_ranks():
    pr = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--no-gpu", "--steps", "3", "--warmup", "1"],
                        env=_env(), capture_output=True, text=True, timeout=240)
    assert pr.returncode == 0, pr.stderr[-2000:]
    _check(_line(pr.stdout), 2, True)


def test_ranks_whose_shards_hold_different_group_values_agree_on_the_layout():
    """rank 0's stand-in shard has no 'R' line, rank 1's no 'O' line, rank 2's no 'A' line: planned from their own statistics the
    ranks would derive three different dense layouts (round 3 ended the run with "shards disagree"); the statistics are unified
    across the ranks before compiling, so all plan the 6-group table of the whole"""
    env = _env()
    env["RSQ_BENCH_DRY_DROP"] = "0:R,1:O,2:A"
    pr = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--backend", "gloo", "--no-gpu", "--steps", "2", "--warmup", "1"],
                        env=env, capture_output=True, text=True, timeout=240)
    assert pr.returncode == 0, pr.stderr[-2000:]
    rec = _line(pr.stdout)
    _check(rec, 3, True)
    cfg = rec["config"]
    assert "groups=6" in cfg["layout"] and "l_returnflag{65,78,82} x l_linestatus{70,79}" in cfg["layout"]
    assert cfg["shard_group_values"] == [[[65, 78], [70, 79]], [[65, 78, 82], [70]], [[78, 82], [70, 79]]]
    assert 0 < cfg["shard_rows_total"] < 6144                       # the summed row count of the (thinned) shards


def test_bench_under_torch_distributed_run():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    pr = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH,
                         "--gpus", "2", "--backend", "gloo", "--no-gpu", "--steps", "2", "--warmup", "1"],
                        env=_env(), capture_output=True, text=True, timeout=240)
    assert pr.returncode == 0, pr.stderr[-2000:]
    _check(_line(pr.stdout), 2, False)


def test_a_failing_rank_fails_the_call_quickly():
    """rank 1 dies before the rendezvous; rank 0 would sit in init_process_group until its timeout (120 s) — the launcher watches
    all ranks, ends rank 0 and returns rank 1's exit code"""
    import time
    env = _env()
    env["RSQ_BENCH_DRY_FAIL_RANK"] = "1"
    t0 = time.time()
    pr = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--no-gpu", "--steps", "1", "--warmup", "0"],
                        env=env, capture_output=True, text=True, timeout=240)
    took = time.time() - t0
    assert pr.returncode == 3, (pr.returncode, pr.stderr[-1000:])
    assert took < 30, f"the launcher took {took:.0f} s to notice the dead rank"
    assert "rank 1 exited with code 3" in pr.stderr
    assert not [l for l in pr.stdout.splitlines() if l.startswith("{")]


def test_a_failing_rank_zero_ends_the_others():
    import time
    env = _env()
    env["RSQ_BENCH_DRY_FAIL_RANK"] = "0"
    t0 = time.time()
    pr = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--backend", "gloo", "--no-gpu", "--steps", "1", "--warmup", "0"],
                        env=env, capture_output=True, text=True, timeout=240)
    assert pr.returncode == 3 and time.time() - t0 < 30
    assert not [l for l in pr.stdout.splitlines() if l.startswith("{")]


def test_world_size_mismatch_is_refused():
    env = _env()
    env.update({"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    pr = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--backend", "gloo", "--no-gpu"], env=env, capture_output=True, text=True, timeout=120)
    assert pr.returncode != 0 and "WORLD_SIZE=2" in pr.stderr
