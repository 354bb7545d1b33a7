"""Differential parity: seeded random plans (tests/fuzzplans.py) through the C ABI on the GPU vs the oracle.

The oracle itself is pinned on the same seeds against the unmodified reference in tests/test_oracle.py
(test_fuzz_oracle_matches_reference), so a green run here is engine == oracle == reference on these shapes."""
import pytest

from resql_amd import engine
from oracle import orc

import fuzzplans

pytestmark = pytest.mark.gpu

SEEDS = list(range(320))


@pytest.mark.parametrize("block", range(0, len(SEEDS), 20))
def test_fuzz_engine_matches_oracle(gpu_ctx, block):
    failures = []
    for seed in SEEDS[block:block + 20]:
        plan, kind = fuzzplans.make(seed)
        try:
            want = orc.execute(plan)
        except orc.OracleError as e:
            # the reference refuses this plan (type rule / missing emitter): the engine must refuse it too
            with pytest.raises(engine.EngineError):
                gpu_ctx.run(plan)
            continue
        try:
            got = gpu_ctx.run(plan)
        except engine.EngineError as e:
            failures.append((seed, "engine refused", str(e)))
            continue
        if not fuzzplans.same(kind, got.text, want.text):
            failures.append((seed, kind, f"{got.n_rows} rows vs {want.n_rows}"))
    assert not failures, failures


def test_seeds_wide_sweeps_found(gpu_ctx):
    """tests/golden/sweep_finds.json: seeds beyond 0..319 on which a wide sweep once found the engine wrong, pinned by the reference's digests;
    every plan executed three times (first execution, warm paths)"""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "sweep_finds.json")) as f:
        finds = json.load(f)
    for seed, g in finds.items():
        plan, kind = fuzzplans.make(int(seed))
        tabs = [gpu_ctx.table(t) for t in plan.tables]
        q = gpu_ctx.compile(plan, tabs)
        for _ in range(3):
            q.execute()
            got = q.result()
            assert got.n_rows == g["rows"] and fuzzplans.digest(kind, got.text) == g["digest"], (seed, g["what"])
        q.close()
        for t in tabs:
            t.close()


@pytest.mark.parametrize("groups", [1, 3, 1024])
def test_hash_aggregation_under_contention(gpu_ctx, monkeypatch, groups):
    """Generic insert-or-find aggregation when every lane of a wave wants the same few slots at once (the case where a
    naive 'spin until published' loop dead-locks a wave): forced with RSQ_AGG_MODE=5 on a key that would otherwise get
    a dense id."""
    from resql_amd import tpch
    monkeypatch.setenv("RSQ_AGG_MODE", "5")
    t = tpch.synthetic_table(300_000, groups)
    plan = tpch.synthetic_plan(t, 1 << 30)
    tabs = [gpu_ctx.table(t)]
    q = gpu_ctx.compile(plan, tabs)
    assert "hash aggregation" in q.explain
    q.execute()
    got = q.result()
    q.close(); tabs[0].close()
    want = orc.execute(plan)
    assert got.n_rows == want.n_rows == groups
    assert got.text == want.text
