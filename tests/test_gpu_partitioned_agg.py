"""Aggregation into a large dense group table (more cells than an LDS table holds): the engine either merges with HBM
atomics or partitions the passing rows (count -> scatter records -> per-partition LDS aggregation), chosen per
execution.  Both forms must give the oracle's bytes, including the emission order (first-row tracker)."""
import numpy as np
import pytest

from resql_amd import plan as P, tpch
from oracle import orc

pytestmark = pytest.mark.gpu
T = P.TypeInit


def run_forced(gpu_ctx, monkeypatch, plan, mode):
    monkeypatch.setenv("RSQ_PARTITION", str(mode))
    tabs = [gpu_ctx.table(t) for t in plan.tables]
    q = gpu_ctx.compile(plan, tabs)
    try:
        q.execute()
        first = q.result().text
        q.execute()                                   # buffers are reused by the second run
        assert q.result().text == first
        return first, q.explain
    finally:
        q.close()
        for t in tabs:
            t.close()


@pytest.mark.parametrize("groups", [1 << 13, 1 << 16, 1 << 20])
@pytest.mark.parametrize("selectivity", [0.0, 0.02, 0.5, 1.0])
def test_partitioned_equals_atomics_equals_oracle(gpu_ctx, monkeypatch, groups, selectivity):
    t = tpch.synthetic_table(300_001, groups)
    plan = tpch.synthetic_plan(t, int(selectivity * (1 << 31)))
    want = orc.execute(plan).text
    got, explain = run_forced(gpu_ctx, monkeypatch, plan, 2)
    assert "partitions" in explain
    assert got == want
    assert run_forced(gpu_ctx, monkeypatch, plan, 0)[0] == want


def test_partitioned_with_min_max_avg_and_two_keys(gpu_ctx, monkeypatch):
    rng = np.random.default_rng(5)
    n = 200_000
    t = P.Table("t", [P.Column("k1", T.BIGINT(), rng.integers(0, 300, n).astype(np.int64)),
                      P.Column("k2", T.INT(), rng.integers(-20, 80, n).astype(np.int32)),
                      P.Column("d", T.DATE(), (19920101 + rng.integers(0, 28, n)).astype(np.uint32)),
                      P.Column("x", T.DECIMAL(12, 2), rng.integers(-5000, 100000, n).astype(np.int64)),
                      P.Column("y", T.BIGINT(), rng.integers(0, 1000, n).astype(np.int64))], n)
    p = P.Plan([t])
    k1, k2 = p.attr("k1"), p.attr("k2")
    aggs = [p.sum(p.attr("x")), p.min(p.attr("d")), p.max(p.attr("x")), p.avg(p.attr("x")), p.count(p.star()),
            p.sum(p.mul(p.attr("x"), p.attr("y"))), p.min(p.attr("y"))]
    node = p.selection(p.gt(p.attr("y"), p.constant("100", P.BIGINT)), p.scan("t"))
    node = p.aggregation(aggs, [k1, k2], node)
    node = p.projection([k1, k2] + [p.as_(f"a{i}", a) for i, a in enumerate(aggs)], node)
    plan = p.set_root(p.materialize(node))
    want = orc.execute(plan).text
    got, explain = run_forced(gpu_ctx, monkeypatch, plan, 2)
    assert "partitions" in explain and got == want


def test_automatic_choice_on_a_larger_table(gpu_ctx, monkeypatch, capfd):
    """above 4 M rows the engine decides from the expected selectivity (column statistics, else a sampled pass); both outcomes must agree with the oracle"""
    monkeypatch.delenv("RSQ_PARTITION", raising=False)
    monkeypatch.setenv("RSQ_TRACE", "1")                # (the trace says which form ran)
    from resql_amd import engine
    n, groups = 6_000_000, 1 << 18
    host = tpch.synthetic_table(n, groups)
    dev = gpu_ctx.generate(engine.GEN_SYNTHETIC, n, 1.0, param=groups)
    for sel in (0.005, 0.6):
        plan = tpch.synthetic_plan(host, int(sel * (1 << 31)))
        q = gpu_ctx.compile(plan, [dev])
        q.await_kernels()                          # (an unseen shape starts on the generic pipeline; this test is about the specialised forms)
        capfd.readouterr()
        q.execute()
        err = capfd.readouterr().err
        got = q.result().text
        q.close()
        assert got == orc.execute(plan).text
        assert ("staged partitioning" in err) == (sel > 0.1), err      # partitioned: scatter + aggregate (regions from the column statistics); atomics: one kernel
    dev.close()


def test_generic_hash_aggregation_with_many_groups(gpu_ctx):
    """a computed group key (no dense id) with about as many groups as the first table has slots: the table must grow
    before linear probing degenerates (this shape took 90 ns per row — seconds — until the probe length was bounded)"""
    n, groups = 2_000_000, 1 << 19
    t = tpch.synthetic_table(n, groups)
    p = P.Plan([t])
    key = p.add(p.mul(p.attr("b"), p.constant("2", P.BIGINT)), p.constant("1", P.BIGINT))
    sc, cnt, lo = p.sum(p.attr("c")), p.count(p.star()), p.min(p.attr("d"))
    node = p.aggregation([sc, cnt, lo], [key], p.scan("t"))
    node = p.projection([p.as_("k", key), p.as_("s", sc), p.as_("n", cnt), p.as_("lo", lo)], node)
    plan = p.set_root(p.materialize(node))
    tabs = [gpu_ctx.table(t)]
    q = gpu_ctx.compile(plan, tabs)
    assert "hash aggregation" in q.explain
    q.execute()                       # finds the table too small, grows it
    q.execute()                       # steady state
    assert q.report().kernel_time_ms < 100
    got = q.result()
    q.close(); tabs[0].close()
    want = orc.execute(plan)
    assert got.n_rows == want.n_rows > 400_000
    assert got.text == want.text


# ---- form 3: packed records staged through LDS rings (rsq_device.h "staged partitioning") ---------------------------------

def _synthetic_like(n, b, seed=3, c_max=1 << 20):
    rng = np.random.default_rng(seed)
    cols = [P.Column("a", T.BIGINT(), rng.integers(0, 1 << 31, n).astype(np.int64)), P.Column("b", T.BIGINT(), b.astype(np.int64)),
            P.Column("c", T.BIGINT(), rng.integers(0, c_max, n).astype(np.int64)), P.Column("d", T.BIGINT(), rng.integers(0, 1 << 20, n).astype(np.int64))]
    return P.Table("t", cols, n)


@pytest.mark.parametrize("staged", ["1", "0"])
def test_both_partitioned_forms_stay_under_test(gpu_ctx, monkeypatch, staged):
    monkeypatch.setenv("RSQ_STAGED", staged)
    t = tpch.synthetic_table(700_003, 1 << 20)
    plan = tpch.synthetic_plan(t, int(0.4 * (1 << 31)))
    got, explain = run_forced(gpu_ctx, monkeypatch, plan, 2)
    assert ("staged through LDS rings" in explain) == (staged == "1")
    assert "8-byte records" in explain or staged == "0"
    assert got == orc.execute(plan).text


@pytest.mark.parametrize("n", [77, 4096 * 3 + 77, 1024 * 4 * 2, 128 * 16 * 2 * 5 + 1])
def test_staged_rounds_with_ragged_ends(gpu_ctx, monkeypatch, n):
    """fewer rows than one round, a partial last round, rows behind the last whole tile"""
    rng = np.random.default_rng(n)
    t = _synthetic_like(n, rng.integers(0, 1 << 17, n))
    plan = tpch.synthetic_plan(t, 1 << 31)
    got, explain = run_forced(gpu_ctx, monkeypatch, plan, 2)
    assert "staged through LDS rings" in explain
    assert got == orc.execute(plan).text


def test_staged_regions_run_full_on_clustered_keys_and_are_counted(gpu_ctx, monkeypatch, capfd):
    """rows sorted by the group key: a round's records all go to one or two partitions (they wait for ring slots), and a
    workgroup's share of a partition is nothing like the average the sampled regions assume"""
    n = 3_000_000
    rng = np.random.default_rng(9)
    t = _synthetic_like(n, np.sort(rng.integers(0, 1 << 18, n)))
    plan = tpch.synthetic_plan(t, int(0.9 * (1 << 31)))
    monkeypatch.setenv("RSQ_TRACE", "1")
    got, explain = run_forced(gpu_ctx, monkeypatch, plan, 2)
    err = capfd.readouterr().err
    assert "staged through LDS rings" in explain
    assert "a region ran full" in err and "counted" in err
    assert got == orc.execute(plan).text


def test_staged_with_one_hot_group(gpu_ctx, monkeypatch):
    n = 1_500_000
    rng = np.random.default_rng(10)
    b = rng.integers(0, 1 << 17, n)
    b[rng.random(n) < 0.97] = 4242
    t = _synthetic_like(n, b)
    plan = tpch.synthetic_plan(t, 1 << 31)
    got, explain = run_forced(gpu_ctx, monkeypatch, plan, 2)
    assert "staged through LDS rings" in explain
    assert got == orc.execute(plan).text


def test_staged_with_absent_groups_and_wide_inputs(gpu_ctx, monkeypatch):
    """not every group of the domain occurs (no watermark: the tracker works to the end); inputs too wide for one word"""
    n = 400_000
    rng = np.random.default_rng(12)
    b = rng.integers(0, 1 << 16, n) * 4            # three quarters of the domain never occur
    t = _synthetic_like(n, b, c_max=1 << 40)
    plan = tpch.synthetic_plan(t, int(0.7 * (1 << 31)))
    got, explain = run_forced(gpu_ctx, monkeypatch, plan, 2)
    assert "16-byte records" in explain
    assert got == orc.execute(plan).text


def test_staged_reuses_regions_and_notices_changed_data(gpu_ctx, monkeypatch, capfd):
    import torch
    from resql_amd import engine
    monkeypatch.setenv("RSQ_PARTITION", "2")
    monkeypatch.setenv("RSQ_TRACE", "1")
    n = 2_000_000
    rng = np.random.default_rng(14)
    host = _synthetic_like(n, rng.integers(0, 1 << 17, n))
    dev = {c.name: torch.from_numpy(c.data).cuda() for c in host.columns}
    t = gpu_ctx.table_from_device("t", n, [(c.name, T.BIGINT(), dev[c.name].data_ptr()) for c in host.columns])
    plan = tpch.synthetic_plan(host, int(0.1 * (1 << 31)))
    q = gpu_ctx.compile(plan, [t])
    try:
        q.execute(); first = q.result().text
        assert first == orc.execute(plan).text
        q.execute()
        assert q.result().text == first
        assert "as in the last execution" in capfd.readouterr().err
        # the same rows, but now nearly all of them pass the filter: the remembered regions are far too small
        dev["a"].fill_(1)
        torch.cuda.synchronize()
        host.columns[0].data[:] = 1
        q.execute()
        assert q.result().text == orc.execute(plan).text
        assert "a region ran full" in capfd.readouterr().err
        # an accumulator input outside the range its column had when the table was adopted: refused, no silent truncation
        dev["c"][17] = 1 << 40
        torch.cuda.synchronize()
        with pytest.raises(engine.EngineError) as e:
            q.execute()
        assert "column statistics" in str(e.value)
    finally:
        q.close(); t.close()


def test_statistics_mislead_and_the_sampled_pass_takes_over(gpu_ctx, monkeypatch, capfd):
    """group keys crowded into a few partitions: the even shares laid out from the column statistics run full, the sampled
    pass (per-partition estimates) then sizes the regions"""
    monkeypatch.delenv("RSQ_PARTITION", raising=False)
    monkeypatch.setenv("RSQ_TRACE", "1")
    n = 5_000_000
    rng = np.random.default_rng(21)
    b = np.where(rng.random(n) < 0.9, rng.integers(0, 1 << 13, n), rng.integers(0, 1 << 18, n))       # 90 % of the rows in 2 of 64 partitions
    t = _synthetic_like(n, b)
    plan = tpch.synthetic_plan(t, int(0.5 * (1 << 31)))
    tabs = [gpu_ctx.table(t)]
    q = gpu_ctx.compile(plan, tabs)
    try:
        q.execute()
        err = capfd.readouterr().err
        assert "from the column statistics" in err and "a region ran full" in err and "(sampled)" in err
        assert q.result().text == orc.execute(plan).text
    finally:
        q.close(); tabs[0].close()


def test_overflowed_staged_attempt_followed_by_the_atomics_form(gpu_ctx, monkeypatch, capfd):
    """The column statistics overstate the pass rate (values of `a` sit at the ends of its range: 'uniform' says 50 %, 2 % pass)
    and the passing rows' keys crowd into ONE partition: the first, tentative staged attempt overflows its even-share regions
    after rsq_staged_agg has already stored partial sums, the sampled pass then picks the HBM-atomics form — which adds onto the
    table.  The table must be back at its identities in between (round-2 advisor finding)."""
    monkeypatch.delenv("RSQ_PARTITION", raising=False)
    monkeypatch.setenv("RSQ_TRACE", "1")
    n = 6_000_000
    rng = np.random.default_rng(21)
    passing = rng.random(n) < 0.02
    a = np.where(passing, 0, (1 << 31) - 1).astype(np.int64)
    b = np.where(passing, rng.integers(0, 1500, n), rng.integers(0, 1 << 20, n)).astype(np.int64)
    b[0] = (1 << 20) - 1                           # (keeps the key range at 2^20 whatever the draw)
    cols = [P.Column("a", T.BIGINT(), a), P.Column("b", T.BIGINT(), b),
            P.Column("c", T.BIGINT(), rng.integers(0, 1 << 20, n).astype(np.int64)), P.Column("d", T.BIGINT(), rng.integers(0, 1 << 20, n).astype(np.int64))]
    t = P.Table("t", cols, n)
    plan = tpch.synthetic_plan(t, 1 << 30)
    want = orc.execute(plan).text
    tabs = [gpu_ctx.table(t)]
    q = gpu_ctx.compile(plan, tabs)
    try:
        q.await_kernels()
        q.execute()
        err = capfd.readouterr().err
        assert "from the column statistics" in err and "a region ran full" in err      # the tentative attempt ran and overflowed
        assert "rows pass; atomics" in err                                               # ... and the sample decided afterwards
        assert q.result().text == want
        q.execute()
        assert q.result().text == want
    finally:
        q.close(); tabs[0].close()
