"""GPU parity: TPC-H Q1 / Q6 through the C ABI vs the CPU oracle, byte for byte."""
import pytest

from resql_amd import tpch
from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sf", [0.01, 0.2])
def test_q1_matches_oracle(gpu_ctx, sf):
    li = tpch.lineitem_table(sf, tpch.Q1_COLUMNS)
    plan = tpch.q1_plan(li)
    got = gpu_ctx.run(plan)
    want = orc.execute(plan)
    assert got.text == want.text
    assert got.tuples == want.tuples


@pytest.mark.parametrize("sf", [0.01, 0.2])
def test_q6_matches_oracle(gpu_ctx, sf):
    li = tpch.lineitem_table(sf, tpch.Q6_COLUMNS)
    plan = tpch.q6_plan(li)
    got = gpu_ctx.run(plan)
    want = orc.execute(plan)
    assert got.text == want.text
    assert got.tuples == want.tuples


def test_q1_ragged_row_counts(gpu_ctx):
    """row counts that are not a multiple of the 128-row wave tile, incl. tiny and empty inputs"""
    for n in [0, 1, 2, 127, 128, 129, 1000, 4097]:
        li = tpch.lineitem_table(0.01, tpch.Q1_COLUMNS, n_rows=n)
        plan = tpch.q1_plan(li)
        got = gpu_ctx.run(plan)
        want = orc.execute(plan)
        assert got.text == want.text, n
