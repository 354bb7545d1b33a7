"""The SQL planner shares ONE expression node between identical expressions, as the reference's does (which dies on such statements:
tests/sqlgen.py seed 203, "the reference dies on" in test_gpu_sql.py).  Two equal aggregates over an expression of the group column
cut the shared node's operand chain; the engine must refuse the statement with a type error, not walk a null operand."""
import pytest

from resql_amd import engine, tpch_full


def test_two_equal_aggregates_over_the_group_column_are_refused_cleanly(tmp_path):
    ctx = engine.Context(device=-1, cache_dir=str(tmp_path))
    db = tpch_full.database(0.001)
    tabs = [ctx.table(db[k]) for k in sorted(db)]
    try:
        for sql in ("select o_shippriority, sum(o_shippriority + 1) as a0, sum(o_shippriority + 1) as a1 from orders group by o_shippriority",
                    "select o_shippriority, avg(o_shippriority + 1) as a0, avg(o_shippriority + 1) as a1, count(*) as a2 from orders group by o_shippriority order by o_shippriority desc"):
            with pytest.raises(engine.EngineError) as e:
                ctx.sql_compile(sql, tabs)
            assert "malformed expression" in str(e.value)
        # the same aggregates over another column's expression are fine
        ctx.sql_compile("select o_orderstatus, sum(o_shippriority + 1) as a0, sum(o_shippriority + 1) as a1 from orders group by o_orderstatus", tabs).close()
    finally:
        for t in tabs:
            t.close()
        ctx.close()
