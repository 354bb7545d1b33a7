"""Executions after the first run on what the previous one found - a hash aggregation's group count and table size, a materialisation's
row total - without a host round trip in the middle, and check it against the final status words (engine.cpp hashWarm,
engine_pipelines.cpp materializePipeline).  Here the data of ADOPTED device columns changes between executions (inside the recorded
column ranges): more groups than the table was sized for, more groups than rows were provided for, fewer groups, another row total,
an empty result - every execution must give the oracle's answer for the data as it is then."""
import numpy as np
import pytest

from resql_amd import plan as P
from oracle import orc

pytestmark = pytest.mark.gpu
T = P.TypeInit


def _agg_plan(t):
    """select k * 3 + 1, sum(v), count(*) from t where f < 90 group by k * 3 + 1   (a computed key: the generic hash aggregation)"""
    p = P.Plan([t])
    key = p.add(p.mul(p.attr("k"), p.constant("3", P.BIGINT)), p.constant("1", P.BIGINT))
    s, c = p.sum(p.attr("v")), p.count(p.star())
    node = p.selection(p.lt(p.attr("f"), p.constant("90", P.BIGINT)), p.scan("t"))
    node = p.aggregation([s, c], [key], node)
    return p.set_root(p.materialize(p.projection([p.as_("kk", key), p.as_("sv", s), p.as_("cn", c)], node)))


def _mat_plan(t):
    p = P.Plan([t])
    node = p.selection(p.lt(p.attr("f"), p.constant("3", P.BIGINT)), p.scan("t"))
    return p.set_root(p.materialize(p.projection([p.attr("k"), p.attr("v")], node)))


def _host_table(k, v, f):
    return P.Table("t", [P.Column("k", T.BIGINT(), k), P.Column("v", T.BIGINT(), v), P.Column("f", T.BIGINT(), f)], len(k))


def _adopt(gpu_ctx, torch, arrays):
    dev = [torch.from_numpy(a).cuda() for a in arrays]
    tab = gpu_ctx.table_from_device("t", len(arrays[0]), [(n, T.BIGINT(), d.data_ptr()) for n, d in zip("kvf", dev)])
    return dev, tab


def test_hash_aggregation_whose_groups_change_between_executions(gpu_ctx):
    import torch
    n = 400_000
    rng = np.random.default_rng(11)
    v = rng.integers(0, 1000, n).astype(np.int64)
    f = rng.integers(0, 100, n).astype(np.int64)
    # the recorded range of k is [0, 2^20): the first data uses 50 values of it
    k0 = rng.integers(0, 50, n).astype(np.int64); k0[0], k0[1] = 0, (1 << 20) - 1
    dev, tab = _adopt(gpu_ctx, torch, [k0.copy(), v, f])
    q = gpu_ctx.compile(_agg_plan(_host_table(k0, v, f)), [tab])
    try:
        q.await_kernels()
        steps = [k0,                                                          # 50 groups (+2), twice: cold, then warm
                 k0,
                 rng.integers(0, 200_000, n).astype(np.int64),               # ~170 K groups: far beyond the table sized for ~50
                 rng.integers(0, 200_000, n).astype(np.int64),               # as many again, warm
                 rng.integers(0, 260_000, n).astype(np.int64),               # more groups than rows were provided for, the table still fits
                 rng.integers(0, 7, n).astype(np.int64),                     # back to a handful (they travel with the status words)
                 rng.integers(0, 7, n).astype(np.int64)]
        for i, k in enumerate(steps):
            dev[0].copy_(torch.from_numpy(k)); torch.cuda.synchronize()
            want = orc.execute(_agg_plan(_host_table(k, v, f)))
            q.execute()
            got = q.result()
            assert got.n_rows == want.n_rows and sorted(got.text.splitlines()) == sorted(want.text.splitlines()), i
    finally:
        q.close(); tab.close()


def test_materialisation_whose_row_total_changes_between_executions(gpu_ctx):
    import torch
    n = 300_000
    rng = np.random.default_rng(12)
    k = np.arange(n, dtype=np.int64)
    v = rng.integers(0, 1000, n).astype(np.int64)
    f0 = rng.integers(0, 100, n).astype(np.int64); f0[0], f0[1] = 0, 99
    dev, tab = _adopt(gpu_ctx, torch, [k, v, f0.copy()])
    q = gpu_ctx.compile(_mat_plan(_host_table(k, v, f0)), [tab])
    try:
        q.await_kernels()
        steps = [f0, f0,                                                     # ~3 %: cold, then warm (the result lives in host-mapped memory)
                 rng.integers(0, 50, n).astype(np.int64),                    # twice the rows: another total than remembered
                 rng.integers(0, 50, n).astype(np.int64),
                 rng.integers(0, 4, n).astype(np.int64),                     # 75 % of the rows: beyond the small result's 256 KB - device columns
                 np.full(n, 50, dtype=np.int64),                             # nothing
                 rng.integers(0, 100, n).astype(np.int64)]
        for i, f in enumerate(steps):
            dev[2].copy_(torch.from_numpy(f)); torch.cuda.synchronize()
            want = orc.execute(_mat_plan(_host_table(k, v, f)))
            q.execute()
            got = q.result()
            assert got.n_rows == want.n_rows and got.text == want.text, i
    finally:
        q.close(); tab.close()
