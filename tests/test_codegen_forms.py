"""Which form of a pipeline the code generator picks, checked without a GPU (describe + generate + hiprtc for gfx950):
the staged partitioning's record layouts, the key-set / rank-dictionary joins, the flush of register accumulators."""
import numpy as np

from resql_amd import plan as P, tpch

import pytest

T = P.TypeInit


@pytest.fixture(scope="module")
def compile_ctx(tmp_path_factory):
    """a compile-only context of this module's own (the session's shared one counts cache hits that other tests assert on)"""
    from resql_amd import engine
    ctx = engine.Context(device=-1, cache_dir=str(tmp_path_factory.mktemp("kcache_forms")))
    yield ctx
    ctx.close()


def _explain_and_source(ctx, plan):
    q = ctx.compile(plan, [ctx.table(t) for t in plan.tables])
    return q.explain, q.source


def test_staged_partitioning_record_layouts(compile_ctx):
    t = tpch.synthetic_table(300_000, 1 << 20)
    ex, src = _explain_and_source(compile_ctx, tpch.synthetic_plan(t, 1 << 30))
    assert "partitioned as packed 8-byte records staged through LDS rings" in ex       # 12 + 20 + 20 bits
    assert "rsq::stage_commit<1, 256, " in src and "rsq::stage_track(" in src
    # inputs too wide for one word: two words
    rng = np.random.default_rng(1)
    n = 100_000
    wide = P.Table("t", [P.Column("a", T.BIGINT(), rng.integers(0, 1 << 31, n).astype(np.int64)),
                         P.Column("b", T.BIGINT(), rng.integers(0, 1 << 18, n).astype(np.int64)),
                         P.Column("c", T.BIGINT(), rng.integers(0, 1 << 40, n).astype(np.int64)),
                         P.Column("d", T.BIGINT(), rng.integers(0, 1 << 20, n).astype(np.int64))], n)
    ex, _ = _explain_and_source(compile_ctx, tpch.synthetic_plan(wide, 1 << 30))
    assert "packed 16-byte records" in ex
    # a table the LDS aggregation handles needs no partitioning at all
    ex, _ = _explain_and_source(compile_ctx, tpch.synthetic_plan(tpch.synthetic_table(50_000, 1024), 1 << 30))
    assert "staged through LDS rings" not in ex


def test_rank_dictionary_and_key_set_joins(compile_ctx):
    cu, od, li = tpch.customer_table(0.01), tpch.orders_table(0.01), tpch.lineitem_table(0.01, tpch.Q3_LINEITEM_COLUMNS)
    ex, src = _explain_and_source(compile_ctx, tpch.q3_plan(cu, od, li))
    assert "ht0 (1 key(s), 0 payload word(s)" in ex and "nothing but the bitmap when the build keys prove unique" in ex      # customer side: a key set
    assert "ht1 (1 key(s), 2 payload word(s)" in ex and "a bitmap-rank dictionary instead when the build keys prove unique" in ex
    assert "rsq::rank_of(a.ht1_bm" in src and "a.ht0_rank" in src
    assert "for (i64 t = wave * tstep; t < ntiles; t += nwaves * tstep * 4)" in src          # four tiles in flight behind the wave compaction


def test_register_flush_goes_lane_by_lane(compile_ctx):
    li = tpch.lineitem_table(0.01, tpch.Q1_COLUMNS)
    _, src = _explain_and_source(compile_ctx, tpch.q1_plan(li))
    assert "s_lane[" in src and "rsq::wave_reduce_to_lane63<" in src and "rsq::wave_to_lds<" not in src
    assert "tile_ctr" not in src and "for (i64 t = wave * tstep; t < ntiles;" in src            # tiles dealt up front (the dynamic hand-out measured slower and is gone)
    assert "a.fin_out" in src                                                                  # the step in one launch
