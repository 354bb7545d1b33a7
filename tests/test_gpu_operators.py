"""GPU parity for the reference's own operator tests (test/test_operators.h) and for the golden outputs of the
unmodified reference (tests/golden/ref_*.tbl), through the C ABI."""
import pytest

from resql_amd import engine
from oracle import orc

import goldens
import refcases

pytestmark = pytest.mark.gpu

# plan shapes this engine version does not lower yet: they must be refused loudly (RSQ_ERR_UNSUPPORTED), never
# routed to a CPU path
NOT_YET = {
    "selection_decimal", "selection_decimal2", "selection_date", "selection_combined", "hashjoin", "orderby",
    "aggregation5",
}


@pytest.mark.parametrize("case", sorted(refcases.CASES))
def test_reference_operator_cases(gpu_ctx, case):
    plan = refcases.CASES[case]()
    if case in NOT_YET:
        with pytest.raises(engine.EngineError) as e:
            gpu_ctx.run(plan)
        assert e.value.status == 3
        return
    got = gpu_ctx.run(plan)
    refcases.check_against_literals(case, got)          # the reference's expected table
    assert got.text == orc.execute(plan).text            # and byte-for-byte the oracle (incl. emission order)


@pytest.mark.parametrize("name", goldens.NAMES)
def test_matches_reference_golden(gpu_ctx, name):
    got = gpu_ctx.run(goldens.golden_plan(name))
    assert got.text == goldens.golden_text(name)
