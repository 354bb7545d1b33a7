"""BASELINE.json config 5 on ONE GPU: 10 B rows x 4 int64 columns, `select b, sum(c), sum(d), count(*) from t
where a < tau group by b`, as 8 sequential device-generated shards of 1.25 B rows (40 GB each; 320 GB does not fit the
288 GB of one MI355X).  Generation is untimed, the scan+aggregate of every shard is timed (HIP events around the
engine's launches), the partial group tables are merged exactly as the multi-GPU path merges them (min | max | sum
segments, resql_amd/dist.py) and finalised once.

Checks (printed in the JSON):
  * prefix parity  — the first 10 M rows through the same plan equal the CPU oracle byte for byte;
  * checksum       — sum over groups of (cnt, sum_c, sum_d) equals an UNGROUPED aggregate over the same shards, which
                     runs through a different kernel (register accumulators instead of the LDS / HBM group table).

usage: python tools/synthetic_10b.py [--rows 10000000000] [--shards 8] [--out profiles/r01_synthetic_10b.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from resql_amd import engine, plan as P, tpch  # noqa: E402
from resql_amd.dist import shard_rows  # noqa: E402


def ungrouped_plan(table, threshold):
    p = P.Plan([table])
    sc, sd, cnt = p.sum(p.attr("c")), p.sum(p.attr("d")), p.count(p.star())
    node = p.selection(p.lt(p.attr("a"), p.constant(str(threshold), P.BIGINT)), p.scan(table.name))
    node = p.aggregation([sc, sd, cnt], [], node)
    node = p.materialize(p.projection([p.as_("sum_c", sc), p.as_("sum_d", sd), p.as_("cnt", cnt)], node))
    return p.set_root(node)


def merge_into(total, part, n_min, n_max, n_sum):
    if n_min:
        torch.minimum(total[:n_min], part[:n_min], out=total[:n_min])
    if n_max:
        torch.maximum(total[n_min:n_min + n_max], part[n_min:n_min + n_max], out=total[n_min:n_min + n_max])
    total[n_min + n_max:] += part[n_min + n_max:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000_000)
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--prefix", type=int, default=10_000_000)
    ap.add_argument("--out", default="")
    ap.add_argument("--groups", default="8,1024,1048576", help="group counts to run (comma separated)")
    ap.add_argument("--per-shard", action="store_true", help="print every shard's partial-execution kernel time")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ctx = engine.Context(device=0)
    schema_only = tpch.synthetic_table(16, 8)
    results = []
    roof = ctx.read_bandwidth(8 << 30, 5)
    for groups in [int(g) for g in args.groups.split(",")]:
        # ---- prefix parity against the oracle ----
        from oracle import orc
        host = tpch.synthetic_table(args.prefix, groups)
        pt = ctx.generate(engine.GEN_SYNTHETIC, args.prefix, 1.0, param=groups)
        prefix_ok = True
        for sel in (0.01, 0.5):
            thr = int(sel * (1 << 31))
            q = ctx.compile(tpch.synthetic_plan(schema_only, thr), [pt])
            q.execute()
            prefix_ok &= q.result().text == orc.execute(tpch.synthetic_plan(host, thr)).text
            q.close()
        pt.close(); del host

        for sel in (0.01, 0.1, 0.5):
            thr = int(sel * (1 << 31))
            total = None; utotal = None
            kernel_ms = 0.0; ukernel_ms = 0.0; rows_done = 0
            exec_ms = 0.0; exec_kernel_ms = 0.0; tail_ms = 0.0; shard_groups = 0
            layout = None
            t_wall = time.perf_counter()
            for s in range(args.shards):
                row0, n = shard_rows(args.rows, args.shards, s)
                t = ctx.generate(engine.GEN_SYNTHETIC, n, 1.0, row0=row0, param=groups)     # untimed
                q = ctx.compile(tpch.synthetic_plan(schema_only, thr), [t])
                q.await_kernels()
                n_min, n_max, n_sum = q.partial_layout()
                if layout is None:
                    layout = (n_min, n_max, n_sum)
                assert layout == (n_min, n_max, n_sum), "shards must agree on the partial table layout"
                part = torch.zeros(n_min + n_max + n_sum, dtype=torch.int64, device=dev)
                q.bind_partial(part.data_ptr(), part.numel() * 8)
                q.execute_partial()
                kernel_ms += q.report().kernel_time_ms
                if args.per_shard:
                    print(f"# groups {groups} sel {sel} shard {s}: partial execution kernels {q.report().kernel_time_ms:.3f} ms, {q.report().num_kernels} launches", flush=True)
                if total is None:
                    total = part.clone()
                else:
                    merge_into(total, part, n_min, n_max, n_sum)
                # the same shard END TO END: one full execution — kernels, tail, this shard's result relation in host memory
                torch.cuda.synchronize()          # (torch's clone / merge of `part` above run on torch's stream: done before the engine rewrites it)
                q.execute()
                rep = q.report()
                exec_ms += rep.execution_time_ms; exec_kernel_ms += rep.kernel_time_ms; tail_ms += rep.finalize_time_ms
                shard_groups += q.result(text=False).n_rows
                # the checksum query: no group key, register accumulators
                uq = ctx.compile(ungrouped_plan(schema_only, thr), [t])
                uq.await_kernels()
                a, b, c = uq.partial_layout()
                upart = torch.zeros(a + b + c, dtype=torch.int64, device=dev)
                uq.bind_partial(upart.data_ptr(), upart.numel() * 8)
                uq.execute_partial()
                ukernel_ms += uq.report().kernel_time_ms
                if utotal is None:
                    utotal = upart.clone()
                else:
                    merge_into(utotal, upart, a, b, c)
                rows_done += n
                if s < args.shards - 1:
                    q.close(); uq.close(); t.close()
            # finalise the merged tables through the last shard's queries
            part.copy_(total); torch.cuda.synchronize(); q.finalize()
            upart.copy_(utotal); torch.cuda.synchronize(); uq.finalize()
            res, ures = q.result(), uq.result()
            q.close(); uq.close(); t.close()
            names = res.names
            sums = [sum(res.value(r, names.index(k)) for r in range(res.n_rows)) for k in ("sum_c", "sum_d", "cnt")]
            usums = [ures.value(0, ures.names.index(k)) for k in ("sum_c", "sum_d", "cnt")] if ures.n_rows else [0, 0, 0]
            gbps = 32.0 * rows_done / (kernel_ms * 1e-3) / 1e9
            e2e = 32.0 * rows_done / (exec_ms * 1e-3) / 1e9
            line = {"workload": "synthetic 4xint64 filter + group-by", "rows": rows_done, "shards": args.shards,
                    "groups": groups, "selectivity": sel, "result_groups": res.n_rows,
                    "kernel_ms_total": round(kernel_ms, 3), "rows_per_s": rows_done / (kernel_ms * 1e-3),
                    "achieved_GBps": round(gbps, 1), "algorithmic_frac_of_8TBps": round(gbps / 8000.0, 3),
                    "frac_of_measured_read_roofline": round(gbps / roof, 3),
                    # end to end: every shard as a full execution (kernels + tail -> that shard's result relation on the host)
                    "exec_ms_total": round(exec_ms, 3), "exec_kernel_ms_total": round(exec_kernel_ms, 3), "tail_ms_total": round(tail_ms, 3),
                    "exec_over_kernel": round(exec_ms / max(exec_kernel_ms, 1e-9), 2), "shard_result_rows_total": shard_groups,
                    "end_to_end_rows_per_s": rows_done / (exec_ms * 1e-3), "end_to_end_algorithmic_frac_of_8TBps": round(e2e / 8000.0, 3),
                    "note": ("late loads skip most cache lines of c, d at this selectivity: algorithmic bytes / time exceeds the peak and is NOT a "
                             "roofline fraction; the FETCH_SIZE-based figure is in profiles/r03_synth_late_loads_pmc.json" if gbps > 8000.0 else ""),
                    "checksum_ok": sums == usums, "checksum": sums, "ungrouped_kernel_ms_total": round(ukernel_ms, 3),
                    "prefix_parity_ok": bool(prefix_ok),
                    "wall_s_incl_generation": round(time.perf_counter() - t_wall, 2)}
            results.append(line)
            print(json.dumps(line), flush=True)
    out = {"measured_read_roofline_GBps": round(roof, 1), "lines": results}
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            json.dump(out, f, indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
