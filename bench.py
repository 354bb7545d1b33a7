#!/usr/bin/env python3
"""bench.py — TPC-H Q1 rows/s at SF10 on MI355X through the engine's C ABI.

  python bench.py --gpus N --steps K --warmup W

N > 1 needs one process per GPU.  Two ways in, same code path afterwards:
  * under a launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`): RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* come from the environment;
  * plain `python bench.py --gpus N`: this process — BEFORE it imports torch or touches a GPU — starts N fresh rank
    processes of itself with those variables set (rendezvous on 127.0.0.1, a free port), relays rank 0's JSON line and
    exits non-zero if any rank does; it watches ALL of them and ends the others as soon as one has failed.  (Mirrors the
    reference's execute(): one call fans out to all workers and joins them, reference src/JitContextFlounder.h:459-487.)

One *step* = one execution of the compiled Q1 plan over the whole lineitem table that is already resident in HBM
(device-generated, deterministic; H2D is not part of any timed region): scan 7 columns -> filter -> 6-group aggregation
kernel -> group-by merge (one RCCL all-gather of the 42-word partial tables + one fused merge kernel when N > 1) -> host
finalisation (AVG, projection, ORDER BY) to ReSQL's result relation.  `value` = lineitem rows of the whole job / wall
time of the K steps (max over ranks).  After the timed region the answer is compared with the UNMODIFIED reference's answer
on the same rows (tests/golden/ref_full_q1_sf{1,10}.tbl): "parity_checked", and a non-zero exit code on a mismatch.

N > 1 shards the SF10 table by row range across the ranks ("morsel-sharded", BASELINE.json config 4): total work is
fixed, so scaling is "strong".

`--path capi` runs the same step through rsq_multi_* (include/resql_hip.h): ONE host process over the N GPUs, RCCL reduce
issued by the library — the way a C++ ReSQL host would call it.  Under a launcher only rank 0 works in that mode.

Extra objects on the JSON line:
  roofline     — the scan+aggregate kernel: ALGORITHMIC bytes (38 B/row x rows per launch) / its average
                 duration measured with HIP events on the engine's stream, against the 8 TB/s HBM3E peak of
                 /opt/skills/guides/MI355X_MICROARCH.md.  `traffic` is NOT measured by this run: it is the FETCH_SIZE
                 figure of the committed rocprofv3 --pmc pass named in `traffic_source`, quoted only when this run has the
                 same bytes per launch.
  cpu_baseline — the UNMODIFIED reference (oracle/_ref/ref_harness: ReSQL's asmjit path, threads=1; ReSQL's aggregation
                 pipelines are single-threaded by construction) timed on this box's host cores on the SAME
                 rows the value is quoted on (SF10 lineitem read back from the device table), rank 0, N = 1 only.
  config.phases (N > 1, or --dist-path) — where a step's time goes, measured in a second, UNTIMED loop of the same steps:
                 kernel_ms per rank (the engine's HIP events), collective_ms (an event pair around the merge on the step's
                 stream), finalize_ms (rank 0's host tail): at 8 GPUs the SF10 kernel is ~45 us per GPU, so the step is
                 launch- and collective-latency shaped, and the line says so.
  extras.weak_scaling — BASELINE config 5, the scaling showcase: every rank holds ONE 1.25 B-row shard
                 (40 GB) of the 10 B-row synthetic table, `a < tau` at 10 % selectivity, group by b into G = 8 (register
                 accumulators, 3 merged words) and G = 2^20 (HBM table, 32 MB all-reduced) — rows/s of the whole job and
                 the fraction of N x 8 TB/s in algorithmic bytes (32 B/row).  Untimed for `value`; skipped with --no-extras.

  extras.q1_sf1 / q6_sf10 / q3_sf10 (N = 1) — the other single-GPU BASELINE configurations (2: Q1 at SF1; 3: Q3 at SF10) and Q6 at
                 SF10, whole executions through the C ABI after the timed region: exec_ms / ms_per_step, kernel_ms (HIP events),
                 launches, fraction of 8 TB/s in algorithmic bytes (for Q3: streaming bytes, an upper bound; for Q6 also the
                 FETCH_SIZE-based traffic_frac of the committed PMC pass), each answer compared with the reference's recorded one.

`--gpus N --share-gpu` is the rehearsal of N > 1 a ONE-GPU box allows: N fresh rank processes that all drive device 0 (own engine context,
own row-range shard, the real kernels, statistics unified across the processes, asynchronous partial execution), the partial tables staged
through host memory and merged over gloo (RCCL cannot hold one device twice; the line says so: `config.backend`, `n_gpus` = 1, `config.ranks`
= N), rank 0 finalises and the answer is compared with the reference's.  `--workload q3` runs TPC-H Q3 the multi-GPU way instead (replicated
build sides, lineitem cut on an l_orderkey boundary, every rank's top 10 merged by the sort keys: resql_amd/dist.py).

extras.q1_sf10_one_shot / first_exec_ms / first_ever_exec_ms - what a ReSQL host pays per SELECT, compile -> ONE execution -> delete (reference
src/execute.h:213-247): the first execution of a FRESH query of the same plan on the same context (plan memo + arenas), and the very first
execution of the plan on the context, beside the steady state of the re-executed query (exec_ms / ms_per_step).

`--backend gloo --no-gpu` is a dry mode for machines without a GPU (the CPU test of the launch path): every rank takes a
compile-only engine context, a deterministic stand-in partial table goes through the same sharding, layout check, merge
and finalisation calls, and the line carries "dry_run": true instead of a measurement.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec); ~6.3 TB/s achievable
PMC_PROFILES = ("profiles/r05_q1_sf10_pmc.json", "profiles/r04_q1_sf10_pmc.json", "profiles/r03_q1_sf10_pmc.json", "profiles/r02_q1_sf10_pmc.json",
                "profiles/r01_q1_sf10_pmc.json")
TRAFFIC_PLAUSIBLE = (0.98, 1.5)      # counter bytes / algorithmic bytes of a full scan: below 0.98 the pass did not measure this kernel
INIT_TIMEOUT_S = 120        # rendezvous of the ranks (a rank that never arrives must not hold the others for minutes)


def _mark(what: str):
    """progress on stderr with RSQ_BENCH_TRACE=1 (which phase a run that printed no line reached)"""
    if os.environ.get("RSQ_BENCH_TRACE"):
        print(f"[bench {time.strftime('%H:%M:%S')}] {what}", file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--sf", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-sf", type=float, default=None,
                    help="scale factor of the reference's sample (default: --sf, i.e. the rows `value` is quoted on)")
    ap.add_argument("--dist-path", action="store_true",
                    help="take the multi-rank step (async partial + merge + finalize) even with one rank: lets a 1-GPU box "
                         "exercise the exact code the N > 1 runs use")
    ap.add_argument("--path", choices=["dist", "capi"], default="dist",
                    help="dist: one process per GPU, torch.distributed (RCCL) merge; capi: ONE process over all GPUs through rsq_multi_*")
    ap.add_argument("--capi-devices", type=str, default=None,
                    help="--path capi: comma-separated device ordinals (default 0..gpus-1; a device may repeat: shards sharing a GPU)")
    ap.add_argument("--no-extras", action="store_true", help="skip extras.weak_scaling (BASELINE config 5 shards)")
    ap.add_argument("--weak-rows", type=int, default=1_250_000_000, help="rows per GPU of the weak-scaling extra")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the group-by merge (nccl = RCCL over xGMI; gloo only with --no-gpu)")
    ap.add_argument("--no-gpu", action="store_true", help="dry mode, see the module docstring")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal of N > 1 on a ONE-GPU box: the N rank processes all drive device 0 (own context, own row-range shard, real "
                         "kernels), the partial tables are staged through host memory and merged over gloo (RCCL cannot hold one device twice)")
    ap.add_argument("--workload", choices=["q1", "q3"], default="q1",
                    help="q3 (with --share-gpu or under a launcher): TPC-H Q3 with replicated build sides, lineitem sharded on an l_orderkey "
                         "boundary, every rank's own top 10 merged by the sort keys (resql_amd/dist.py merge_ordered_results)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------
# self-launch: no torch, no HIP in this function or before it
# ------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv, poll_s: float = 0.05, grace_s: float = 5.0) -> int:
    """start n rank processes of this script, watch ALL of them, relay rank 0's stdout; returns the exit code.
    The first rank that exits non-zero ends the job: the others (which would sit in the rendezvous or in a collective
    until its timeout) are terminated, then killed — fresh children only, nothing here has touched a GPU."""
    port = _free_port()
    procs = []
    for rank in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RSQ_BENCH_SELF_LAUNCHED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if rank == 0 else sys.stderr, text=(rank == 0) or None))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)     # rank 0's pipe must be drained while we poll
    reader.start()
    codes = [None] * n
    failed = None
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
                if codes[r] not in (None, 0) and failed is None:
                    failed = r
        if failed is not None:
            break
        time.sleep(poll_s)
    if failed is not None:
        print(f"bench.py: rank {failed} exited with code {codes[failed]}; ending the other ranks", file=sys.stderr)
        for r, p in enumerate(procs):
            if codes[r] is None:
                p.terminate()
        deadline = time.time() + grace_s
        for r, p in enumerate(procs):
            if codes[r] is None:
                try:
                    codes[r] = p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    codes[r] = p.wait()
    reader.join(timeout=10)
    if failed is None and out0 and out0[0]:
        sys.stdout.write(out0[0])
        sys.stdout.flush()
    if failed is not None:
        return codes[failed] if codes[failed] and codes[failed] > 0 else 1
    return 0


# ------------------------------------------------------------------------------------------------------------------
def cpu_baseline(sample_sf: float, device_table=None, repeat: int = 3):
    """time the reference itself (or, if its binary is absent, the C restatement) on a bounded sample.
    device_table: the resident lineitem table of the run — its columns are read back so that the reference scans exactly
    the rows the GPU number is quoted on (the numpy generator makes the same bits, but takes minutes at SF10)."""
    from resql_amd import tpch, datagen
    from oracle import orc
    import numpy as np
    n = datagen.n_lineitem(sample_sf)
    if device_table is not None and device_table.n_rows == n:
        dt = {"l_quantity": np.int64, "l_extendedprice": np.int64, "l_discount": np.int64, "l_tax": np.int64,
              "l_returnflag": np.uint8, "l_linestatus": np.uint8, "l_shipdate": np.uint32}
        data = {c: device_table.read_column(c, dt[c]) for c in tpch.Q1_COLUMNS}
        li = tpch.make_table("lineitem", tpch.LINEITEM_SCHEMA, data, n)
        source = "read back from the device table"
    else:
        li = tpch.lineitem_table(sample_sf, tpch.Q1_COLUMNS)
        source = "numpy generator (same bits as the device generator)"
    plan = tpch.q1_plan(li)
    cores = os.cpu_count() or 1
    if orc.have_reference():
        _, tm = orc.run_reference(plan, threads=1, repeat=repeat, quiet=True)
        ms = min(tm["exec_ms"])
        # threads=N for the record: ReSQL's aggregation pipelines run under a SingleThreadGuard, so N threads do not help
        nthr = max(2, min(16, cores))
        try:
            _, tmn = orc.run_reference(plan, threads=nthr, repeat=2, quiet=True)
            many = f"; threads={nthr}: {min(tmn['exec_ms']):.1f} ms"
        except Exception:
            many = ""
        return {"value": n / (ms * 1e-3), "unit": "rows/s", "cores": 1, "kind": "reference",
                "sample": f"TPC-H Q1 over {n} synthetic lineitem rows (SF{sample_sf:g}, {source}), ReSQL asmjit path "
                          f"threads=1, best of {repeat} `execute:` times ({ms:.1f} ms){many}; host has {cores} cores but "
                          f"ReSQL runs aggregation pipelines on one thread (SingleThreadGuard)"}
    t0 = time.time()
    orc.execute(plan)
    dt_s = time.time() - t0
    return {"value": n / dt_s, "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": f"TPC-H Q1 over {n} synthetic lineitem rows (SF{sample_sf:g}, {source}), oracle/resql_oracle.c, 1 thread"}


def committed_traffic(bytes_per_launch: int):
    """HBM bytes per launch of the Q1 kernel from the rocprofv3 --pmc pass committed under profiles/ (FETCH_SIZE,
    collected in its own run and corrected as MI355X_MICROARCH.md prescribes for gfx950) — only for the configuration
    it was measured on, and only if the figure can be one: a full scan cannot fetch less than its algorithmic bytes
    (round 4's file averaged SF1 and SF10 dispatches of the same kernel name and said 0.22 x).  Returns (traffic, source);
    a file that is refused is named in the source with the reason, and the next older one is tried."""
    refused = []
    for rel in PMC_PROFILES:
        try:
            with open(os.path.join(ROOT, rel)) as f:
                pmc = json.load(f)
        except Exception:
            continue
        if pmc.get("algorithmic_bytes_per_launch") != bytes_per_launch:
            continue
        if pmc.get("invalid"):
            refused.append(f"{rel}: {pmc['invalid']}")
            continue
        got = pmc.get("hbm_read_bytes_per_launch_corrected")
        ratio = (got / bytes_per_launch) if got else 0.0
        if not (TRAFFIC_PLAUSIBLE[0] <= ratio <= TRAFFIC_PLAUSIBLE[1]):
            refused.append(f"{rel}: {ratio:.3f} x the algorithmic bytes over {pmc.get('launches')} launches is not a measurement of this kernel")
            continue
        src = f"{rel} (committed rocprofv3 --pmc FETCH_SIZE pass over {pmc.get('launches')} launches; not measured by this run)"
        if refused:
            src += "; refused: " + "; ".join(refused)
        return got, src
    return None, ("refused: " + "; ".join(refused)) if refused else None


def golden_answer(sf: float):
    """the UNMODIFIED reference's Q1 answer on the rows of this scale factor (tests/golden/make_fullsize_golden.py), or None"""
    path = os.path.join(ROOT, "tests", "golden", f"ref_full_q1_sf{sf:g}.tbl")
    try:
        with open(path) as f:
            return f.read(), os.path.relpath(path, ROOT)
    except OSError:
        return None, None


def check_parity(result_text: str, sf: float, out: dict) -> int:
    """compare the timed plan's answer with the reference's; fills out['parity_checked' / 'parity_source'], returns the exit code"""
    want, src = golden_answer(sf)
    if want is None:
        out["parity_checked"] = None
        out["parity_source"] = f"no committed reference answer for SF{sf:g} (tests/golden/ref_full_q1_sf{{1,10}}.tbl)"
        return 0
    ok = result_text == want
    out["parity_checked"] = bool(ok)
    out["parity_source"] = f"{src}: the unmodified reference's answer on the same rows, compared byte for byte after the timed region"
    if not ok:
        print(f"bench.py: the answer of the timed plan differs from {src}:\n{result_text}\nvs\n{want}", file=sys.stderr)
        return 4
    return 0


# ------------------------------------------------------------------------------------------------------------------
def dry_run(args, world: int, rank: int) -> int:
    """--no-gpu: launch path, process group, sharding, layout check, merge and finalisation without a device"""
    import torch
    import torch.distributed as dist
    from resql_amd import datagen, engine, tpch
    from resql_amd.dist import PartialMerger, shard_rows
    if os.environ.get("RSQ_BENCH_DRY_FAIL_RANK") == str(rank):     # test hook: a rank that dies before the rendezvous
        print(f"rank {rank}: failing on request", file=sys.stderr)
        return 3
    if world > 1 or args.dist_path:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import datetime
        dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=INIT_TIMEOUT_S))
        grouped = True
    else:
        grouped = False
    n_total = datagen.n_lineitem(args.sf)
    row0, n_rows = shard_rows(n_total, world, rank)
    ctx = engine.Context(device=-1)
    schema_only = tpch.lineitem_table(0.001, tpch.Q1_COLUMNS, n_rows=0)
    # a small stand-in shard with REAL rows (rows [s0, s0 + sn) of a 6144-row table), so that every rank has its own column
    # statistics; RSQ_BENCH_DRY_DROP="rank:char,..." removes the rows with that l_returnflag / l_linestatus value from a rank's
    # shard (test hook: shards whose statistics differ).  The statistics are unified across the ranks before compiling — the
    # step the measured path takes too — so every rank derives the same 6-group layout.
    from resql_amd.dist import unify_shard_stats
    s0, sn = shard_rows(6144, world, rank)
    cols = datagen.lineitem_columns(s0, sn, 0.001, columns=set(tpch.Q1_COLUMNS))
    for item in filter(None, os.environ.get("RSQ_BENCH_DRY_DROP", "").split(",")):
        r, ch = item.split(":")
        if int(r) == rank:
            keep = (cols["l_returnflag"] != ord(ch)) & (cols["l_linestatus"] != ord(ch))
            cols = {k: v[keep] for k, v in cols.items()}
    shard = ctx.table(tpch.make_table("lineitem", tpch.LINEITEM_SCHEMA, cols, len(cols["l_quantity"])))
    shard.set_row0(s0)
    own_layout_values = sorted({int(v) for v in cols["l_returnflag"]}), sorted({int(v) for v in cols["l_linestatus"]})
    unify_shard_stats(dist if grouped else None, shard, world)
    q = ctx.compile(tpch.q1_plan(schema_only), [shard])
    n_min, n_max, n_sum = q.partial_layout()
    words = n_min + n_max + n_sum
    # stand-in partial table: first-row trackers = this shard's first rows, sums = (rank + 1) * (word index + 1)
    partial = torch.empty(words, dtype=torch.int64)
    partial[:n_min] = row0 + torch.arange(n_min)
    partial[n_min:n_min + n_max] = rank
    partial[n_min + n_max:] = (rank + 1) * (torch.arange(n_sum) + 1)
    layout = [l for l in q.explain.splitlines() if l.startswith("partial table:")]
    every_layout, every_values = [layout], [own_layout_values]
    if grouped and world > 1:
        every_layout, every_values = [None] * world, [None] * world
        dist.all_gather_object(every_layout, layout)
        dist.all_gather_object(every_values, own_layout_values)
        if any(e != every_layout[0] for e in every_layout):      # cannot happen after unify_shard_stats: a defect, not a property of the data
            raise SystemExit(f"rank {rank}: internal error: ranks planned from the same statistics disagree on the layout: {every_layout}")
    merger = PartialMerger(dist if grouped else None, partial, n_min, n_max, n_sum, world)
    for _ in range(args.warmup + args.steps):
        mine = partial.clone()
        merger.partial = mine
        merger.merge()
    # the per-rank phase figures travel the way the measured path sends them (one all-gather of a small tensor)
    per_rank = gather_per_rank(dist if grouped else None, world, float(rank + 1), device=None)
    if grouped:
        dist.barrier()
    if rank == 0:
        q.finalize_host(mine.numpy())
        res = q.result()
        tri = world * (world + 1) // 2
        expect_sum = [tri * (i + 1) for i in range(n_sum)]
        ok = mine[:n_min].tolist() == list(range(n_min)) and mine[n_min + n_max:].tolist() == expect_sum
        ok = ok and per_rank == [float(r + 1) for r in range(world)]
        print(json.dumps({"metric": "TPC-H Q1 rows/s at SF10", "dry_run": True, "value": None, "unit": "rows/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "config": {"workload": "launch / merge / finalize plumbing only (no GPU)", "rows": n_total,
                                     "world_size": dist.get_world_size() if grouped else 1,
                                     "backend": dist.get_backend() if grouped else "none",
                                     "self_launched": os.environ.get("RSQ_BENCH_SELF_LAUNCHED") == "1",
                                     "shards": [list(shard_rows(n_total, world, r)) for r in range(world)],
                                     "merge": merger.strategy, "merged_ok": bool(ok), "result_groups": res.n_rows,
                                     "layout": layout[0] if layout else None, "shard_rows_total": shard.total_rows,
                                     "shard_group_values": every_values,
                                     "phases": {"kernel_ms_per_rank": per_rank}}}),
              flush=True)
        if not ok:
            return 1
    q.close()
    shard.close()
    ctx.close()
    if grouped:
        dist.destroy_process_group()
    return 0


def gather_per_rank(dist, world: int, value: float, device):
    """[value of rank 0, ..., value of rank world-1] on every rank (one all-gather of one double per rank)"""
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist is None or world == 1:
        return [float(t.item())]
    every = torch.empty(world, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(every, t)
    return [float(v) for v in every.cpu().tolist()]


# ------------------------------------------------------------------------------------------------------------------
# --path capi: ONE host process over N GPUs through rsq_multi_* (the C ABI a C++ ReSQL host binds)
# ------------------------------------------------------------------------------------------------------------------
def run_capi(args) -> int:
    from resql_amd import datagen, engine, tpch
    devices = [int(d) for d in args.capi_devices.split(",")] if args.capi_devices else list(range(args.gpus))
    n_dev = len(devices)
    n_total = datagen.n_lineitem(args.sf)
    m = engine.MultiContext(devices)
    try:
        shards = m.generate(engine.GEN_LINEITEM, n_total, args.sf)
        schema_only = tpch.lineitem_table(0.001, tpch.Q1_COLUMNS, n_rows=0)
        q = m.compile(tpch.q1_plan(schema_only), [[t] for t in shards])
        for _ in range(max(1, args.warmup)):
            q.execute()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            q.execute()
        elapsed = time.perf_counter() - t0
        # per-shard kernel times: a second, untimed loop (a report per step inside the timed loop would time its own bookkeeping)
        probe = max(1, min(args.steps, 20))
        per = [0.0] * n_dev
        fin = 0.0
        coll = 0.0
        for _ in range(probe):
            q.execute()
            rep, k = q.report()
            per = [a + b for a, b in zip(per, k)]
            fin += rep.finalize_time_ms
            coll += q.collective_ms
        per = [v / probe for v in per]
        result = q.result()
        rows_per = [t.n_rows for t in shards]
        bytes_per_launch = tpch.Q1_BYTES_PER_ROW * max(rows_per)
        slowest = max(per)
        achieved = bytes_per_launch / (slowest * 1e-3) / 1e9
        out = {
            "metric": "TPC-H Q1 rows/s at SF10", "value": n_total * args.steps / elapsed, "unit": "rows/s",
            "n_gpus": len(set(devices)), "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": f"TPC-H Q1 over lineitem SF{args.sf:g} ({n_total} rows, 7 columns, 38 B/row) resident in HBM, "
                                   f"row-range sharded over {n_dev} shard(s) on devices {devices}",
                       "rows": n_total, "rows_per_shard": rows_per, "result_groups": result.n_rows, "path": "capi",
                       "parallelism": f"one host process, rsq_multi_* over {n_dev} shard(s): {m.merge_name}; {q.merge_name}",
                       "phases": {"kernel_ms_per_rank": per, "collective_ms": coll / probe, "finalize_ms": fin / probe,
                                  "step_minus_slowest_kernel_ms": elapsed / args.steps * 1e3 - slowest}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": None, "traffic_source": None,
                         "kernel": "scan+filter+dense aggregation pipeline (the slowest shard's launches)", "kernel_ms": slowest,
                         "bytes_per_launch": bytes_per_launch},
        }
        rc = check_parity(result.text, args.sf, out)
        print(json.dumps(out), flush=True)
        q.close()
        for t in shards:
            t.close()
        return rc
    finally:
        m.close()


# ------------------------------------------------------------------------------------------------------------------
# extras.weak_scaling: BASELINE config 5, one 1.25 B-row shard per GPU
# ------------------------------------------------------------------------------------------------------------------
def weak_scaling_extra(args, ctx, dist, world: int, rank: int, device, steps: int = 3):
    """every rank: one shard of `--weak-rows` rows of the synthetic 4 x int64 table (rows [rank * rows, (rank + 1) * rows) of the
    whole), `a < tau` at 10 %, group by b, 2 sums + count: partial aggregation -> merge (PartialMerger: all-gather for the
    3-word G = 8 table, one all-reduce per segment for the 2^20-group table) -> rank 0 finalises.  Returns a list of dicts
    (rank 0) or None."""
    import torch
    from resql_amd import engine, tpch
    from resql_amd.dist import PartialMerger
    rows = int(args.weak_rows)
    out = []
    threshold = int(0.10 * (1 << 31))
    for groups in (8, 1 << 20):
        shard = ctx.generate(engine.GEN_SYNTHETIC, rows, 1.0, row0=rank * rows, param=groups)
        q = ctx.compile(tpch.synthetic_plan(tpch.synthetic_table(16, groups), threshold), [shard])
        q.await_kernels()
        n_min, n_max, n_sum = q.partial_layout()
        partial = torch.zeros(n_min + n_max + n_sum, dtype=torch.int64, device=device)
        q.bind_partial(partial.data_ptr(), partial.numel() * 8)
        merger = PartialMerger(dist, partial, n_min, n_max, n_sum, world, always_collective=args.dist_path, query=q) if dist is not None else None

        def step():
            if merger is None:
                q.execute()
                return
            q.execute_partial_async()
            merger.merge()
            if rank == 0:
                q.finalize()

        def fence():
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        step()                                           # first execution: sizes regions / picks the form
        fence()
        q.kernel_time_stats(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        if merger is not None and rank != 0:
            q.finalize()
        ksum, kn = q.kernel_time_stats()
        kms = gather_per_rank(dist, world, ksum / max(1, kn), device)
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if rank == 0:
            res = q.result(text=False)
            total_rows = rows * world
            ms = dt / steps * 1e3
            out.append({"groups": groups, "selectivity": 0.10, "rows_per_gpu": rows, "rows": total_rows, "ms_per_step": ms,
                        "rows_per_s": total_rows / (ms * 1e-3), "result_groups": res.n_rows,
                        "algorithmic_gbps": total_rows * 32 / (ms * 1e-3) / 1e9,
                        # ALGORITHMIC bytes (32 B/row) over the whole step against N x 8 TB/s.  At 10 % selectivity the late-load form skips most
                        # lines of c and d, so this can exceed what the bytes really fetched allow: it is a throughput figure, not a roofline
                        # fraction - traffic_frac (FETCH_SIZE of the committed --pmc pass over the kernel time) is, where a pass exists
                        "algorithmic_frac": total_rows * 32 / (ms * 1e-3) / 1e9 / (HBM_PEAK_GBPS * world),
                        "traffic_frac": None,
                        "traffic_note": "late loads apply (a < tau at 10 %): see profiles/*_late_loads_pmc.json for the fetched-bytes fraction of the 1 % case; "
                                        "no --pmc pass is committed for this selectivity",
                        "kernel_ms_per_rank": kms, "merge": merger.strategy if merger is not None else "single GPU",
                        "partial_table_words": n_min + n_max + n_sum})
        q.close()
        shard.close()
        del partial
    return out if rank == 0 else None


# ------------------------------------------------------------------------------------------------------------------
# extras.q1_sf1 / q6_sf10 / q3_sf10: the other single-GPU BASELINE configurations, untimed for `value`
# ------------------------------------------------------------------------------------------------------------------
LATE_LOAD_PROFILES = ("profiles/r05_late_loads_pmc.json", "profiles/r04_late_loads_pmc.json", "profiles/r03_late_loads_pmc.json")


def _golden_text(name: str):
    try:
        with open(os.path.join(ROOT, "tests", "golden", f"ref_full_{name}.tbl")) as f:
            return f.read()
    except OSError:
        return None


def _committed_late_load_traffic(case: str, rows: int):
    """FETCH_SIZE-based fraction of a late-load plan from the committed rocprofv3 --pmc pass (same rows only), or (None, None)"""
    for rel in LATE_LOAD_PROFILES:
        try:
            with open(os.path.join(ROOT, rel)) as f:
                c = json.load(f)["cases"][case]
            if c["rows"] == rows:
                return c["traffic_frac_of_8TBps"], f"{rel} (committed --pmc FETCH_SIZE pass x 2 / kernel time / 8 TB/s; not measured by this run)"
        except Exception:
            continue
    return None, None


def _measure_plan(q, steps: int, golden: str, bytes_alg: int, fresh=None):
    """whole executions of one compiled query: exec_ms (host wall time of rsq_query_execute, launches to result relation),
    kernel_ms (HIP events on the engine's stream), launches, fraction of the 8 TB/s peak in ALGORITHMIC bytes, parity.
    first_ever_exec_ms: the first execution of the plan on the context; first_exec_ms (with `fresh`, a function that compiles the same
    plan anew): compile -> ONE execution -> delete, what a ReSQL host pays per SELECT (reference src/execute.h:213-247)"""
    q.await_kernels()
    q.execute()
    first_ever = q.report().execution_time_ms
    for _ in range(2):
        q.execute()
    q.kernel_time_stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        q.execute()
    ms = (time.perf_counter() - t0) / steps * 1e3
    ksum, kn = q.kernel_time_stats()
    rep = q.report()
    kernel_ms = ksum / max(1, kn) if kn else rep.kernel_time_ms
    want = _golden_text(golden)
    first_exec, first_ok = None, None
    if fresh is not None:
        q1 = fresh()
        q1.execute()
        first_exec = q1.report().execution_time_ms
        first_ok = (q1.result().text == want) if want is not None else None
        q1.close()
    return {"exec_ms": ms, "first_exec_ms": first_exec, "first_ever_exec_ms": first_ever, "first_exec_parity_checked": first_ok,
            "kernel_ms": kernel_ms, "launches": int(rep.num_kernels), "finalize_ms": rep.finalize_time_ms,
            "algorithmic_bytes": bytes_alg, "algorithmic_frac": bytes_alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if kernel_ms > 0 else None,
            "parity_checked": (q.result().text == want) if want is not None else None,
            "parity_source": f"tests/golden/ref_full_{golden}.tbl (the unmodified reference's answer on the same rows)"}


def config_extras(ctx, sf10_table, sf: float):
    """BASELINE configs 2 (Q1 SF1) and 3 (Q3 SF10) and Q6 at SF10 on this GPU; each result is compared with the reference's
    recorded answer.  Q1 at SF1 is a 228 MB scan (~30 us at peak): launch-latency shaped, ms_per_step says so."""
    from resql_amd import datagen, engine, tpch
    out = {}
    schema_q1 = tpch.lineitem_table(0.001, tpch.Q1_COLUMNS, n_rows=0)
    # ---- config 2: Q1 at SF1 ----
    n1 = datagen.n_lineitem(1.0)
    t1 = ctx.generate(engine.GEN_LINEITEM, n1, 1.0)
    q = ctx.compile(tpch.q1_plan(schema_q1), [t1])
    m = _measure_plan(q, 200, "q1_sf1", tpch.Q1_BYTES_PER_ROW * n1, fresh=lambda: ctx.compile(tpch.q1_plan(schema_q1), [t1]))
    out["q1_sf1"] = {"workload": f"TPC-H Q1 over lineitem SF1 ({n1} rows) on one GPU (BASELINE config 2)", "ms_per_step": m["exec_ms"],
                     "first_exec_ms": m["first_exec_ms"], "first_ever_exec_ms": m["first_ever_exec_ms"],
                     "rows_per_s": n1 / (m["exec_ms"] * 1e-3), "kernel_ms": m["kernel_ms"], "frac": m["algorithmic_frac"], "launches": m["launches"],
                     "parity_checked": m["parity_checked"], "parity_source": m["parity_source"]}
    q.close(); t1.close()
    # ---- Q6 at SF10 (late loads behind the selective date filter: fewer bytes fetched than the columns hold) ----
    if sf == 10.0:
        n = sf10_table.n_rows
        # the headline statement the way a ReSQL host issues it: compile -> ONE execution -> delete (the timed loop above re-executes one query)
        t0 = time.perf_counter()
        q = ctx.compile(tpch.q1_plan(schema_q1), [sf10_table])
        t1 = time.perf_counter()
        q.execute()
        out["q1_sf10_one_shot"] = {"compile_ms": (t1 - t0) * 1e3, "first_exec_ms": q.report().execution_time_ms,
                                   "parity_checked": (q.result().text == _golden_text("q1_sf10")) if _golden_text("q1_sf10") is not None else None,
                                   "note": "a fresh query of the timed plan on the same context: what the plan memo and the arenas leave of the first-execution cost"}
        q.close()
        q = ctx.compile(tpch.q6_plan(schema_q1), [sf10_table])
        m = _measure_plan(q, 50, "q6_sf10", tpch.Q6_BYTES_PER_ROW * n, fresh=lambda: ctx.compile(tpch.q6_plan(schema_q1), [sf10_table]))
        tf, src = _committed_late_load_traffic("q6", n)
        m.update({"workload": f"TPC-H Q6 over lineitem SF10 ({n} rows): scan + 5 comparisons + ungrouped sum", "traffic_frac": tf, "traffic_source": src,
                  "note": "algorithmic_frac can exceed what the bytes actually fetched allow (late loads skip cache lines): traffic_frac is the roofline figure"})
        out["q6_sf10"] = m
        q.close()
        # ---- config 3: Q3 at SF10 ----
        nL, nO, nC = datagen.n_lineitem(10.0), datagen.n_orders(10.0), datagen.n_customer(10.0)
        li = ctx.generate(engine.GEN_LINEITEM, nL, 10.0, param=1)
        od = ctx.generate(engine.GEN_ORDERS, nO, 10.0)
        cu = ctx.generate(engine.GEN_CUSTOMER, nC, 10.0)
        plan = tpch.q3_plan(tpch.customer_table(0.001), tpch.orders_table(0.001), tpch.lineitem_table(0.001, tpch.Q3_LINEITEM_COLUMNS, n_rows=0))
        q = ctx.compile(plan, [cu, od, li])
        streaming = (4 + 10) * nC + (4 + 4 + 4 + 4) * nO + (4 + 4 + 8 + 8) * nL
        m = _measure_plan(q, 30, "q3_sf10", streaming, fresh=lambda: ctx.compile(plan, [cu, od, li]))
        m.update({"workload": f"TPC-H Q3 at SF10 (customer {nC}, orders {nO}, lineitem {nL}): 2 joins + aggregation at the join entry + ORDER BY ... LIMIT 10 "
                              f"(BASELINE config 3)",
                  "note": "algorithmic_frac is against the STREAMING bytes of the three scans: an upper-bound figure, the probes are random accesses (SURVEY.md §8d)"})
        out["q3_sf10"] = m
        q.close()
        for t in (li, od, cu):
            t.close()
    return out


# ------------------------------------------------------------------------------------------------------------------
# --workload q3: joins + many groups across ranks (resql_amd/dist.py): replicated build sides, lineitem cut on an
# l_orderkey boundary, the whole plan per rank, ONE all-gather of every rank's 10 rows merged by the sort keys
# ------------------------------------------------------------------------------------------------------------------
def run_q3_sharded(args, dist, world: int, rank: int, local_rank: int, device) -> int:
    import torch
    from resql_amd import datagen, engine, tpch
    from resql_amd.dist import merge_ordered_results, shard_rows_on_key
    share = bool(args.share_gpu)
    sf = args.sf
    nL, nO, nC = datagen.n_lineitem(sf), datagen.n_orders(sf), datagen.n_customer(sf)
    # the cut needs the clustering key around the equal-shard boundaries only: a window of rows there is generated on the host
    per = (nL // (128 * world)) * 128
    def key_at_factory():
        cache = {}
        def key_at(i):
            base = (i // 4096) * 4096
            if base not in cache:
                cache[base] = datagen.lineitem_columns(base, min(4096, nL - base), sf, columns={"l_orderkey"})["l_orderkey"]
            return int(cache[base][i - base])
        return key_at
    row0, n_rows = shard_rows_on_key(nL, world, rank, key_at_factory())
    ctx = engine.Context(device=local_rank)
    li = ctx.generate(engine.GEN_LINEITEM, n_rows, sf, row0=row0, param=1)
    od = ctx.generate(engine.GEN_ORDERS, nO, sf)
    cu = ctx.generate(engine.GEN_CUSTOMER, nC, sf)
    plan = tpch.q3_plan(tpch.customer_table(0.001), tpch.orders_table(0.001), tpch.lineitem_table(0.001, tpch.Q3_LINEITEM_COLUMNS, n_rows=0))
    q = ctx.compile(plan, [cu, od, li])
    q.await_kernels()
    cdev = None if (share or dist is None) else device
    order = [("revenue", False), ("o_orderdate", True)]

    def step():
        q.execute()
        return merge_ordered_results(dist if world > 1 else None, q.result(text=False), order, 10, world, cdev)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        merged = step()
    fence()
    q.kernel_time_stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        merged = step()
    fence()
    elapsed = time.perf_counter() - t0
    ksum, kn = q.kernel_time_stats()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cpu") if share else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    per_kernel = gather_per_rank(dist, world, ksum / max(1, kn), None if share else device)
    rows_per = gather_per_rank(dist, world, float(n_rows), None if share else device)
    rc = 0
    if rank == 0:
        want = _golden_text(f"q3_sf{sf:g}")
        out = {"metric": "TPC-H Q3 rows/s", "value": (nL + nO + nC) * args.steps / elapsed, "unit": "rows/s (lineitem + orders + customer rows of the whole job)",
               "n_gpus": 1 if share else world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
               "config": {"workload": f"TPC-H Q3 at SF{sf:g}: customer {nC} and orders {nO} replicated on every rank, lineitem {nL} rows cut at l_orderkey "
                                      f"boundaries into {world} shard(s); every rank runs the whole plan, one all-gather of 10 rows per rank, merged by (revenue desc, o_orderdate)",
                          "ranks": world, "share_gpu": share, "lineitem_rows_per_rank": [int(v) for v in rows_per],
                          "backend": "none" if dist is None else ("gloo over host memory (--share-gpu)" if share else dist.get_backend() + " (RCCL)"),
                          "self_launched": os.environ.get("RSQ_BENCH_SELF_LAUNCHED") == "1",
                          "phases": {"kernel_ms_per_rank": per_kernel}},
               "parity_checked": (merged.text == want) if want is not None else None,
               "parity_source": f"tests/golden/ref_full_q3_sf{sf:g}.tbl (the unmodified reference's answer on the unsharded tables)" if want is not None else None}
        if want is not None and merged.text != want:
            print(f"bench.py: the merged Q3 answer differs from the reference's:\n{merged.text}\nvs\n{want}", file=sys.stderr)
            rc = 4
        print(json.dumps(out), flush=True)
    q.close()
    for t in (li, od, cu):
        t.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    return rc


# ------------------------------------------------------------------------------------------------------------------
def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.backend == "gloo" and not args.no_gpu:
        raise SystemExit("--backend gloo exists for the --no-gpu dry mode only: the measured path merges over RCCL")
    if args.path == "capi":
        if int(os.environ.get("RANK", "0")) != 0:
            return 0                                      # under a launcher: ONE process drives all GPUs, the other ranks have nothing to do
        return run_capi(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, argv)              # nothing GPU-related has been imported yet

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and not (args.gpus == 1 and world == 1):
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.no_gpu:
        return dry_run(args, world, rank)

    import torch
    from resql_amd import datagen, engine, tpch

    share = bool(args.share_gpu)
    if share:
        local_rank = 0                                    # every rank process drives the one GPU
    dist = None
    if world > 1 or args.dist_path:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        if share:      # RCCL refuses two ranks on one device: the exchange goes through host memory and gloo, and the line says so
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=INIT_TIMEOUT_S))
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=INIT_TIMEOUT_S))
    else:
        torch.cuda.set_device(0)
    device = torch.device("cuda", local_rank if world > 1 else 0)
    if args.workload == "q3":
        return run_q3_sharded(args, dist, world, rank, local_rank, device)

    # ---- data: this rank's row range of the SF table, generated in HBM ----
    from resql_amd.dist import PartialMerger, shard_rows
    n_total = datagen.n_lineitem(args.sf)
    row0, n_rows = shard_rows(n_total, world, rank)   # shard boundaries on 128-row tiles
    ctx = engine.Context(device=local_rank if world > 1 else 0)
    table = ctx.generate(engine.GEN_LINEITEM, n_rows, args.sf, row0=row0)
    # every rank plans its shard as the WHOLE table (union of the shards' column statistics, summed row count): one dense group
    # layout on all ranks whatever their rows hold — the reference has one hash table all workers reach (aggregation.h:240-295)
    from resql_amd.dist import unify_shard_stats
    unify_shard_stats(dist if world > 1 else None, table, world, None if share else device)
    schema_only = tpch.lineitem_table(0.001, tpch.Q1_COLUMNS, n_rows=0)
    q = ctx.compile(tpch.q1_plan(schema_only), [table])
    q.await_kernels()          # the measured path is the specialised kernel, never the generic pipeline a cold cache starts on
    n_min, n_max, n_sum = q.partial_layout()
    words = n_min + n_max + n_sum
    partial = torch.zeros(words, dtype=torch.int64, device=device)
    q.bind_partial(partial.data_ptr(), partial.numel() * 8)

    multi = dist is not None
    merger = None
    ranks_seen = 1
    cdev = torch.device("cpu") if share else device      # where the collectives' tensors live (gloo: host memory)
    if multi:
        # how many ranks the collectives of this run really span: one word of ones, all-reduced over RCCL
        ones = torch.ones(1, dtype=torch.int64, device=cdev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
    if multi and world > 1:
        # (after unify_shard_stats every rank derives the same layout; a difference would be a defect, not a property of the data)
        layout = [l for l in q.explain.splitlines() if l.startswith("partial table:")]
        every = [None] * world
        dist.all_gather_object(every, layout)
        if any(e != every[0] for e in every):
            raise SystemExit(f"rank {rank}: internal error: ranks planned from the same statistics disagree on the layout: {every}")
    if multi:
        # one stream for the whole step: scan+aggregate kernel -> merge collective (RCCL over xGMI) -> read-back, with a
        # single host synchronisation (inside finalize) per step on rank 0 and none on the other ranks
        torch.cuda.set_stream(torch.cuda.Stream(device))       # not the null stream: it serialises against every blocking stream
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        if share:
            staged = torch.empty(words, dtype=torch.int64).pin_memory()      # host staging of the partial table (gloo moves host memory)
            merger = PartialMerger(dist, staged, n_min, n_max, n_sum, world, always_collective=args.dist_path, query=None)
        else:
            merger = PartialMerger(dist, partial, n_min, n_max, n_sum, world, always_collective=args.dist_path, query=q)

    def step():
        if not multi:
            q.execute()
            return
        q.execute_partial_async()                     # enqueue: identity image -> kernel(s)
        if share:
            staged.copy_(partial, non_blocking=True)  # D2H behind the kernel, then the host-side exchange
            torch.cuda.current_stream().synchronize()
            merger.merge()
            if rank == 0:
                partial.copy_(staged, non_blocking=True)
                q.finalize()
            return
        merger.merge()                                # the one exchange step of the path
        if rank == 0:
            q.finalize()                              # D2H of the merged table, sync, AVG / projection / ORDER BY

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    _mark("compiled; warm-up")
    for _ in range(args.warmup):
        step()
    fence()
    _mark("timed region")
    q.kernel_time_stats(reset=True)                   # the engine sums the device time of every launch (HIP events); read once below
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if multi and rank != 0:
        q.finalize()                                  # the other ranks check their device error word once, untimed
    ksum, kn = q.kernel_time_stats()
    avg_kernel_ms = ksum / max(1, kn)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    _mark(f"timed region done: {elapsed / args.steps * 1e3:.4f} ms per step")
    # ---- where a multi-rank step's time goes: a second, UNTIMED loop of the same steps with an event pair around the merge ----
    phases = None
    if multi and share:
        phases = {"kernel_ms_per_rank": gather_per_rank(dist, world, avg_kernel_ms, None),
                  "note": "two rank processes share ONE GPU: their kernels run side by side or one after the other as the hardware queues "
                          "decide, and the exchange is a D2H copy, a gloo all-gather in host memory and an H2D copy - a rehearsal of the code "
                          "path, not a measurement of xGMI"}
    if multi and not share:
        probe = max(1, min(args.steps, 20))
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(probe)]
        fin_ms = 0.0
        step_ms = 0.0
        for i in range(probe):
            fence()
            ts = time.perf_counter()
            q.execute_partial_async()
            ev[i][0].record()
            merger.merge()
            ev[i][1].record()
            if rank == 0:
                q.finalize()
                fin_ms += q.report().finalize_time_ms
            torch.cuda.synchronize()
            step_ms += (time.perf_counter() - ts) * 1e3
        if rank != 0:
            q.finalize()
        coll = sum(a.elapsed_time(b) for a, b in ev) / probe
        per_kernel = gather_per_rank(dist, world, avg_kernel_ms, device)
        per_coll = gather_per_rank(dist, world, coll, device)
        phases = {"kernel_ms_per_rank": per_kernel, "collective_ms_per_rank": per_coll, "collective_ms": max(per_coll),
                  "finalize_ms": fin_ms / probe, "isolated_step_ms_rank0": step_ms / probe,
                  "note": "untimed second loop of the same steps, every step fenced (barrier + synchronize): collective_ms is the "
                          "event pair around the merge on the step's stream and includes waiting for the slowest rank's kernel"}

    _mark("phases done")
    # ---- the weak-scaling extra runs on every rank (its collectives need all of them), after the headline measurement ----
    weak = None
    weak_error = None
    if not args.no_extras and not share:
        try:
            # (--dist-path takes the multi-rank step here too, with a process group of one: partial execution, merge collective, finalize)
            weak = weak_scaling_extra(args, ctx, dist if (dist is not None and (world > 1 or args.dist_path)) else None, world, rank, device)
        except Exception as e:  # an extra never costs the bench line
            weak_error = f"{type(e).__name__}: {e}"

    configs = None
    configs_error = None
    if not args.no_extras and world == 1 and rank == 0:
        try:
            configs = config_extras(ctx, table, args.sf)
        except Exception as e:
            configs_error = f"{type(e).__name__}: {e}"
    _mark(f"extras done ({weak_error}, {configs_error})")
    rc = 0
    if rank == 0:
        result = q.result()
        bytes_per_launch = tpch.Q1_BYTES_PER_ROW * n_rows
        traffic, traffic_source = committed_traffic(bytes_per_launch) if world == 1 else (None, None)
        achieved = bytes_per_launch / (avg_kernel_ms * 1e-3) / 1e9
        # the read-only streaming roofline of THIS box, measured after the timed region with the access form the scan
        # uses (16 B per lane, non-temporal, 2 workgroups per CU): SURVEY.md §8d asks for the fraction against it too
        try:
            measured = ctx.read_bandwidth(4 << 30, 5)
        except Exception:
            measured = None
        config = {"workload": f"TPC-H Q1 over lineitem SF{args.sf:g} ({n_total} rows, 7 columns, 38 B/row) "
                              f"resident in HBM, row-range sharded over {world} GPU(s)",
                  "rows": n_total, "rows_per_gpu": n_rows, "result_groups": result.n_rows, "path": "dist",
                  "world_size": dist.get_world_size() if dist is not None else 1,
                  "rccl_ranks_seen": ranks_seen,
                  "backend": ("gloo over host memory (--share-gpu: the rank processes share GPU 0, RCCL cannot hold a device twice)" if share else
                              (dist.get_backend() + " (RCCL)")) if dist is not None else "none (single process, single GPU)",
                  "share_gpu": share, "ranks": world,
                  "self_launched": os.environ.get("RSQ_BENCH_SELF_LAUNCHED") == "1",
                  "parallelism": (f"row-range shards x{world}, group-by merge " + ("staged through host memory over gloo" if share else "over RCCL") + f": {merger.strategy}")
                  if multi else "single GPU"}
        if phases is not None:
            config["phases"] = phases
        if multi:
            # ("what's weak" 8: say what bounds the strong-scaling case in the line that first measures it)
            config["scaling_note"] = ("strong scaling of a 0.34 ms step: every rank scans rows / N (kernel_ms_per_rank), then ONE exchange of the "
                                      "partial aggregate tables (collective_ms, a few hundred bytes per rank: latency, not bandwidth) and the merge + "
                                      "finalize on rank 0 (finalize_ms); at N = 8 the kernel is ~45 us, so the exchange's latency bounds the speed-up")
        out = {
            "metric": "TPC-H Q1 rows/s at SF10",
            "value": n_total * args.steps / elapsed,
            "unit": "rows/s",
            "n_gpus": 1 if share else world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic",
            "config": config,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "scan+filter+dense aggregation pipeline (rank 0's launches)", "kernel_ms": avg_kernel_ms,
                         "bytes_per_launch": bytes_per_launch,
                         "measured_read_roofline": measured,
                         "frac_of_measured_read_roofline": (achieved / measured) if measured else None},
        }
        rc = check_parity(result.text, args.sf, out)
        if configs is not None or configs_error is not None:
            out.setdefault("extras", {}).update(configs if configs is not None else {"configs_error": configs_error})
            bad = [k for k, v in (configs or {}).items() if v.get("parity_checked") is False]
            if bad:
                print(f"bench.py: extras {bad} differ from the reference's recorded answers", file=sys.stderr)
                rc = rc or 4
        if weak is not None or weak_error is not None:
            out.setdefault("extras", {}).update({"weak_scaling": weak if weak is not None else {"error": weak_error},
                             "weak_scaling_note": "BASELINE config 5: one synthetic 4 x int64 shard per GPU (rows_per_gpu), a < tau at 10 %, "
                                                  "group by b, 2 sums + count; whole step incl. merge and rank 0's host tail; untimed for `value`"})
        if world == 1 and not args.no_cpu_baseline:
            want_sf = args.cpu_baseline_sf if args.cpu_baseline_sf is not None else args.sf
            try:
                out["cpu_baseline"] = cpu_baseline(want_sf, table if want_sf == args.sf else None)
            except Exception as e:  # the baseline is a reported extra; never lose the bench line over it
                try:
                    out["cpu_baseline"] = cpu_baseline(1.0)
                    out["cpu_baseline"]["sample"] += f" [SF{want_sf:g} sample failed: {e}]"
                except Exception as e2:
                    out["cpu_baseline"] = {"value": None, "unit": "rows/s", "cores": 0, "kind": "reference",
                                           "sample": f"failed: {e2}"}
        print(json.dumps(out), flush=True)
        _mark("line printed")

    q.close()
    table.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
