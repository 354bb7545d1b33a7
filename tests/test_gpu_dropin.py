"""The drop-in, exercised with the reference's REAL classes: oracle/_ref/ref_harness builds ReSQL's own operator
tree (RelOperator / Expr objects, compiled from /root/reference in the build container) and runs it twice — through
ReSQL's Flounder/asmjit JIT and through integration/resql_hip_binding.h -> C ABI -> HIP engine.  Both results are
serialised by ReSQL's own serializeRelation and must be identical."""
import pytest

from resql_amd import tpch
from oracle import orc

import refcases

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not orc.have_reference(), reason="oracle/_ref/ref_harness was not shipped")]


def both(plan):
    ref, _ = orc.run_reference(plan)
    hip, tm = orc.run_reference(plan, engine="hip")
    return ref, hip


@pytest.mark.parametrize("sf", [0.01, 0.1])
def test_tpch_through_resql_objects(sf):
    li = tpch.lineitem_table(sf, tpch.Q1_COLUMNS + ["l_orderkey"])
    cu, od = tpch.customer_table(sf), tpch.orders_table(sf)
    for plan in (tpch.q1_plan(li), tpch.q6_plan(li), tpch.q3_plan(cu, od, li)):
        ref, hip = both(plan)
        assert hip == ref


@pytest.mark.parametrize("case", ["aggregation", "aggregation2", "aggregation3", "aggregation4"])
def test_reference_operator_cases_through_resql_objects(case):
    ref, hip = both(refcases.CASES[case]())
    assert hip == ref
