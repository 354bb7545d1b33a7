"""Row-range shards whose column statistics differ (a time-clustered table: one shard has no 'R' line, another no 'O' line).
The reference has ONE relation and ONE hash table every worker reaches (reference src/operators/aggregation.h:240-295,
src/JitContextFlounder.h:459-487), so any distribution of the rows gives the same answer; the engine plans dense group ids from
column statistics, so its shards must plan from the statistics of the WHOLE table (rsq_table_unify_shard_stats)."""
import numpy as np

from resql_amd import datagen, tpch

SF = 0.001


def lineitem_shards(n_rows: int = 6000, drops=("", "R", "O"), columns=tpch.Q1_COLUMNS):
    """the rows of shard i = rows [i * n / k, (i + 1) * n / k) of the generated table WITHOUT those whose l_returnflag or l_linestatus is
    drops[i].  Returns ([columns of shard i], [first row number of shard i], the concatenated table as a P.Table)."""
    cols = datagen.lineitem_columns(0, n_rows, SF, columns=set(columns))
    k = len(drops)
    shards = []
    for i, drop in enumerate(drops):
        lo, hi = i * n_rows // k, (i + 1) * n_rows // k
        c = {name: v[lo:hi] for name, v in cols.items()}
        keep = np.ones(hi - lo, dtype=bool)
        for ch in drop:
            keep &= (c["l_returnflag"] != ord(ch)) & (c["l_linestatus"] != ord(ch))
        shards.append({name: np.ascontiguousarray(v[keep]) for name, v in c.items()})
    row0 = [0]
    for c in shards[:-1]:
        row0.append(row0[-1] + len(c["l_quantity"]))
    whole = {name: np.concatenate([c[name] for c in shards]) for name in shards[0]}
    return shards, row0, tpch.make_table("lineitem", tpch.LINEITEM_SCHEMA, whole, len(whole["l_quantity"]))


def shard_table(columns: dict):
    return tpch.make_table("lineitem", tpch.LINEITEM_SCHEMA, columns, len(columns["l_quantity"]))


def layout_line(query) -> str:
    return [l for l in query.explain.splitlines() if l.startswith("partial table:")][0]
