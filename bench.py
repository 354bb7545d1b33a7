#!/usr/bin/env python3
"""bench.py — TPC-H Q1 rows/s at SF10 on MI355X through the engine's C ABI.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run,
                                                          one rank per GPU, RCCL)

One *step* = one execution of the compiled Q1 plan over the whole lineitem table that is already
resident in HBM (device-generated, deterministic; H2D is not part of any timed region):
scan 7 columns -> filter -> 6-group aggregation kernel -> group-by merge (RCCL all-reduce of the
partial aggregate table when N > 1) -> host finalisation (AVG, projection, ORDER BY) to ReSQL's
result relation.  `value` = lineitem rows of the whole job / wall time of the K steps (max over
ranks).

N > 1 shards the SF10 table by row range across the ranks ("morsel-sharded", BASELINE.json config 4):
total work is fixed, so scaling is "strong".

Extra objects on the JSON line:
  roofline     — the scan+aggregate kernel: ALGORITHMIC bytes (38 B/row x rows per launch, SURVEY.md
                 §8d) / its average duration measured with HIP events on the engine's stream, against
                 the 8 TB/s HBM3E peak of /opt/skills/guides/MI355X_MICROARCH.md.
  cpu_baseline — the UNMODIFIED reference (oracle/_ref/ref_harness: ReSQL's asmjit path, threads=1;
                 ReSQL's aggregation pipelines are single-threaded by construction) timed
                 on this box's host cores on a bounded sample (SF1 Q1), rank 0, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec); ~6.3 TB/s achievable


def cpu_baseline(sample_sf: float = 1.0, repeat: int = 3):
    """time the reference itself (or, if its binary is absent, the C restatement) on a bounded sample"""
    from resql_amd import tpch, datagen
    from oracle import orc
    n = datagen.n_lineitem(sample_sf)
    li = tpch.lineitem_table(sample_sf, tpch.Q1_COLUMNS)
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
                "sample": f"TPC-H Q1 over {n} synthetic lineitem rows (SF{sample_sf:g}), ReSQL asmjit path threads=1, "
                          f"best of {repeat} `execute:` times ({ms:.1f} ms){many}; host has {cores} cores but ReSQL runs "
                          f"aggregation pipelines on one thread (SingleThreadGuard)"}
    t0 = time.time()
    orc.execute(plan)
    dt = time.time() - t0
    return {"value": n / dt, "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": f"TPC-H Q1 over {n} synthetic lineitem rows (SF{sample_sf:g}), oracle/resql_oracle.c, 1 thread"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--sf", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-path", action="store_true",
                    help="take the multi-rank step (async partial + merge + finalize) even with one rank: lets a 1-GPU box "
                         "exercise the exact code the N > 1 runs use")
    args = ap.parse_args()

    import torch
    from resql_amd import datagen, engine, tpch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    dist = None
    if world > 1 or args.dist_path:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    device = torch.device("cuda", local_rank if world > 1 else 0)

    # ---- data: this rank's row range of the SF table, generated in HBM ----
    from resql_amd.dist import PartialMerger, shard_rows
    n_total = datagen.n_lineitem(args.sf)
    row0, n_rows = shard_rows(n_total, world, rank)   # shard boundaries on 128-row tiles
    ctx = engine.Context(device=local_rank if world > 1 else 0)
    table = ctx.generate(engine.GEN_LINEITEM, n_rows, args.sf, row0=row0)
    schema_only = tpch.lineitem_table(0.001, tpch.Q1_COLUMNS, n_rows=0)
    q = ctx.compile(tpch.q1_plan(schema_only), [table])
    n_min, n_max, n_sum = q.partial_layout()
    words = n_min + n_max + n_sum
    partial = torch.zeros(words, dtype=torch.int64, device=device)
    q.bind_partial(partial.data_ptr(), partial.numel() * 8)

    multi = dist is not None
    merger = None
    if multi and world > 1:
        # the partial tables are only mergeable if every rank derived the same dense group layout from its shard's
        # column statistics (same byte-value sets / ranges): compare the layout line of `explain` across the ranks
        layout = [l for l in q.explain.splitlines() if l.startswith("partial table:")]
        every = [None] * world
        dist.all_gather_object(every, layout)
        if any(e != every[0] for e in every):
            raise SystemExit(f"rank {rank}: shards disagree on the partial aggregate table layout: {every}")
    if multi:
        # one stream for the whole step: scan+aggregate kernel -> merge collective (RCCL over xGMI) -> read-back, with a
        # single host synchronisation (inside finalize) per step on rank 0 and none on the other ranks
        torch.cuda.set_stream(torch.cuda.Stream(device))       # not the null stream: it serialises against every blocking stream
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        merger = PartialMerger(dist, partial, n_min, n_max, n_sum, world, always_collective=args.dist_path)

    def step():
        if not multi:
            q.execute()
            return
        q.execute_partial_async()                     # enqueue: identity image -> kernel(s)
        merger.merge()                                # the one exchange step of the path
        if rank == 0:
            q.finalize()                              # D2H of the merged table, sync, AVG / projection / ORDER BY

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(q.report().kernel_time_ms)
    fence()
    elapsed = time.perf_counter() - t0
    if multi and rank != 0:
        q.finalize()                                  # the other ranks check their device error word once, untimed
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        result = q.result()
        # HBM traffic of this kernel from the PMC pass committed under profiles/ (FETCH_SIZE, collected in its own
        # rocprofv3 --pmc run and corrected x2 as MI355X_MICROARCH.md prescribes for gfx950); only quoted for the
        # configuration it was measured on
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_q1_sf10_pmc.json")) as f:
                pmc = json.load(f)
            if world == 1 and pmc.get("algorithmic_bytes_per_launch") == tpch.Q1_BYTES_PER_ROW * n_rows:
                traffic = pmc["hbm_read_bytes_per_launch_corrected"]
        except Exception:
            traffic = None
        avg_kernel_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = tpch.Q1_BYTES_PER_ROW * n_rows / (avg_kernel_ms * 1e-3) / 1e9
        # the read-only streaming roofline of THIS box, measured after the timed region with the access form the scan
        # uses (16 B per lane, non-temporal, 2 workgroups per CU): SURVEY.md §8d asks for the fraction against it too
        try:
            measured = ctx.read_bandwidth(4 << 30, 5)
        except Exception:
            measured = None
        out = {
            "metric": "TPC-H Q1 rows/s at SF10",
            "value": n_total * args.steps / elapsed,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic",
            "config": {"workload": f"TPC-H Q1 over lineitem SF{args.sf:g} ({n_total} rows, 7 columns, 38 B/row) "
                                   f"resident in HBM, row-range sharded over {world} GPU(s)",
                       "rows": n_total, "result_groups": result.n_rows,
                       "parallelism": f"row-range shards x{world}, group-by merge over RCCL: {merger.strategy}"
                       if multi else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "rsq_pipeline (scan+filter+dense aggregation)", "kernel_ms": avg_kernel_ms,
                         "bytes_per_launch": tpch.Q1_BYTES_PER_ROW * n_rows,
                         "measured_read_roofline": measured,
                         "frac_of_measured_read_roofline": (achieved / measured) if measured else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline is a reported extra; never lose the bench line over it
                out["cpu_baseline"] = {"value": None, "unit": "rows/s", "cores": 0, "kind": "reference",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)

    q.close()
    table.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
