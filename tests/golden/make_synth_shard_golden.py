#!/usr/bin/env python3
"""Checksums of BASELINE.json config 5's shards (10 B rows x 4 int64 as 8 shards of 1.25 B rows): for shard 0 and
shard 7, group counts G in {8, 2^20} and filter selectivities, the expected answer of
    select b, sum(c), sum(d), count(*) from t where a < tau group by b
computed by streaming the numpy generator (resql_amd/datagen.py synthetic_columns, the same bits the device generator
makes) through np.bincount in 8 M-row chunks on all host cores.  Stored per case: number of groups, totals of the three
aggregates, and the SHA-256 of the sorted serialised result ("b|sum_c|sum_d|cnt|" lines, the reference's
serializeRelation format).  tests/test_gpu_fullsize.py runs the same shard on the device and compares.

Independent of the engine and of the oracle (plain numpy); takes ~10 minutes per case on 8 cores:
    python tests/golden/make_synth_shard_golden.py
"""
import hashlib
import json
import os
import sys
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from resql_amd import datagen  # noqa: E402
from resql_amd.dist import shard_rows  # noqa: E402

TOTAL_ROWS, SHARDS = 10_000_000_000, 8
CHUNK = 8 << 20
CASES = [(0, 8, 0.5), (0, 1 << 20, 0.5), (0, 1 << 20, 0.01), (7, 8, 0.5), (7, 1 << 20, 0.1),
         (0, 1024, 0.1), (7, 1024, 0.5)]   # (shard, groups, selectivity); G = 1024 is the workgroup-LDS table (round 3)


def _chunk(args):
    row0, n, groups_list, thresholds = args
    r = np.arange(row0, row0 + n, dtype=np.int64)
    a = datagen.uniform(datagen.SEED, datagen.S_A, r, 1 << 31)
    c = datagen.uniform(datagen.SEED, datagen.S_C, r, 1 << 20)
    d = datagen.uniform(datagen.SEED, datagen.S_D, r, 1 << 20)
    out = {}
    for g in groups_list:
        b = datagen.uniform(datagen.SEED, datagen.S_B, r, g)
        for thr in thresholds[g]:
            m = a < thr
            bm = b[m]
            # float64 weights are exact here: per-chunk sums stay below 2^53 (8 M rows x 2^20)
            out[(g, thr)] = (np.bincount(bm, minlength=g).astype(np.int64),
                             np.bincount(bm, weights=c[m], minlength=g).astype(np.int64),
                             np.bincount(bm, weights=d[m], minlength=g).astype(np.int64))
    return out


def main():
    path = os.path.join(HERE, "synth_shard_checksums.json")
    result = json.load(open(path)) if os.path.exists(path) else {}
    by_shard = {}
    for shard, g, sel in CASES:
        if f"shard{shard}_g{g}_thr{int(sel * (1 << 31))}" in result and not os.environ.get("GOLDEN_REDO"):
            continue                                  # (already recorded: a shard pass takes ~10 minutes)
        by_shard.setdefault(shard, {}).setdefault(g, []).append(int(sel * (1 << 31)))
    for shard, per_g in by_shard.items():
        row0, n = shard_rows(TOTAL_ROWS, SHARDS, shard)
        jobs = [(row0 + o, min(CHUNK, n - o), sorted(per_g), per_g) for o in range(0, n, CHUNK)]
        acc = {}
        with Pool(int(os.environ.get("GOLDEN_PROCS", "6"))) as pool:
            for i, part in enumerate(pool.imap_unordered(_chunk, jobs)):
                for k, (cnt, sc, sd) in part.items():
                    if k not in acc:
                        acc[k] = [cnt, sc, sd]
                    else:
                        acc[k][0] += cnt; acc[k][1] += sc; acc[k][2] += sd
                if i % 16 == 0:
                    print(f"shard {shard}: {i + 1}/{len(jobs)} chunks", flush=True)
        for (g, thr), (cnt, sc, sd) in acc.items():
            present = np.nonzero(cnt)[0]
            lines = sorted(f"{b}|{sc[b]}|{sd[b]}|{cnt[b]}|" for b in present)
            digest = hashlib.sha256(("\n".join(lines) + "\n").encode()).hexdigest()
            result[f"shard{shard}_g{g}_thr{thr}"] = {"shard": shard, "row0": row0, "rows": n, "groups": g, "threshold": thr,
                                                      "result_groups": int(present.size), "cnt": int(cnt.sum()), "sum_c": int(sc.sum()),
                                                      "sum_d": int(sd.sum()), "sha256_sorted_lines": digest}
            print(f"shard{shard}_g{g}_thr{thr}", result[f"shard{shard}_g{g}_thr{thr}"], flush=True)
        with open(path, "w") as f:
            json.dump(result, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
