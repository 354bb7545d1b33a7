"""Plans for the reference's own operator tests (test/test_operators.h), built with the same
operator/expression constructors, over the literal tables stored in tests/golden/reference_literals.json.
Used by the oracle tests (CPU) and by the GPU parity tests, so both read like the reference's tests."""
from __future__ import annotations

import json
import os
from collections import Counter
from typing import List, Tuple

from resql_amd import plan as P

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_literals.json")) as f:
    LIT = json.load(f)


def _schema(spec) -> List[Tuple[str, P.SqlType]]:
    return [(n, P.parse_type(t.split())) for n, t in spec]


def _table(name, schema_spec, rows) -> P.Table:
    return P.table_from_strings(name, _schema(schema_spec), rows)


def expected_rows(case: str, types: List[P.SqlType]):
    """expected cells parsed like the reference's relationFromStrings (test_common.h:38-62)"""
    out = []
    for row in LIT[case]["expected"]:
        vals = []
        for cell, t in zip(row, types):
            v = P.column_from_strings(t, [cell])[0]
            vals.append(v.item() if hasattr(v, "item") else v)
        out.append(tuple(vals))
    return out


def selection_decimal() -> P.Plan:       # test_operators.h:23-73
    c = LIT["selection_decimal"]
    p = P.Plan([_table("rel", c["schema"], c["rows"])])
    cond = p.or_(p.lt(p.attr("quantity"), p.constant("10.0", P.DECIMAL)),
                 p.gt(p.attr("quantity"), p.constant("1000.0", P.DECIMAL)))
    return p.set_root(p.materialize(p.selection(cond, p.scan("rel"))), request_all=True)


def selection_decimal2() -> P.Plan:      # test_operators.h:76-121
    c = LIT["selection_decimal2"]
    p = P.Plan([_table("rel", c["schema"], c["rows"])])
    return p.set_root(p.materialize(p.selection(p.lt(p.attr("rateA"), p.attr("rateB")), p.scan("rel"))), request_all=True)


def selection_date() -> P.Plan:          # test_operators.h:124-179
    c = LIT["selection_date"]
    p = P.Plan([_table("rel", c["schema"], c["rows"])])
    cond = p.and_(p.ge(p.attr("date"), p.constant("1966/06/15", P.DATE)),
                  p.le(p.attr("date"), p.constant("1988/10/25", P.DATE)))
    return p.set_root(p.materialize(p.selection(cond, p.scan("rel"))), request_all=True)


def selection_combined() -> P.Plan:      # test_operators.h:183-232
    c = LIT["selection_combined"]
    p = P.Plan([_table("rel", c["schema"], c["rows"])])
    cond = p.and_(p.lt(p.attr("ratio"), p.constant("0.222", P.DECIMAL)),
                  p.or_(p.lt(p.attr("date"), p.constant("2000/01/01", P.DATE)),
                        p.lt(p.attr("quantity"), p.constant("120", P.BIGINT))))
    return p.set_root(p.materialize(p.selection(cond, p.scan("rel"))), request_all=True)


def hashjoin() -> P.Plan:                # test_operators.h:443-464
    c = LIT["hashjoin"]
    p = P.Plan([_table("R", c["schema_R"], c["rows_R"]), _table("S", c["schema_S"], c["rows_S"])])
    hj = p.hashjoin([p.eq(p.attr("attributeA"), p.attr("attributeC"))], p.scan("R"), p.scan("S"))
    return p.set_root(p.materialize(hj), request_all=True)


def aggregation() -> P.Plan:             # test_operators.h:511-565
    c = LIT["aggregation"]
    p = P.Plan([_table("rel", c["schema"], c["rows"])])
    agg = p.aggregation([p.sum(p.attr("attributeB"))], [p.attr("attributeA")], p.scan("rel"))
    return p.set_root(p.materialize(agg), request_all=True)


def aggregation2() -> P.Plan:            # test_operators.h:568-637
    c = LIT["aggregation2"]
    p = P.Plan([_table("rel", c["schema"], c["rows"])])
    agg = p.aggregation([p.sum(p.attr("attributeC")), p.count(p.attr("attributeC"))],
                        [p.attr("attributeA"), p.attr("attributeB")], p.scan("rel"))
    return p.set_root(p.materialize(agg), request_all=True)


def aggregation3() -> P.Plan:            # test_operators.h:640-696
    c = LIT["aggregation2"]
    p = P.Plan([_table("rel", c["schema"], c["rows"])])
    agg = p.aggregation([p.sum(p.attr("attributeC")), p.count(p.attr("attributeC"))], [], p.scan("rel"))
    return p.set_root(p.materialize(agg), request_all=True)


def aggregation4() -> P.Plan:            # test_operators.h:699-757
    c = LIT["aggregation2"]
    p = P.Plan([_table("rel", c["schema"], c["rows"])])
    agg = p.aggregation([], [p.attr("attributeA"), p.attr("attributeB")], p.scan("rel"))
    return p.set_root(p.materialize(agg), request_all=True)


def aggregation5() -> P.Plan:            # test_operators.h:760-831
    c = LIT["aggregation"]
    p = P.Plan([_table("rel", c["schema"], c["rows"])])
    agg = p.aggregation([p.sum(p.add(p.attr("attributeA"), p.attr("attributeB")))],
                        [p.add(p.attr("attributeA"), p.attr("attributeB"))], p.scan("rel"))
    return p.set_root(p.materialize(agg), request_all=True)


def orderby() -> P.Plan:                 # test_operators.h:834-888
    c = LIT["orderby"]
    p = P.Plan([_table("rel", c["schema"], c["rows"])])
    return p.set_root(p.orderby([p.attr("attributeA")], p.scan("rel")), request_all=True)


CASES = {
    "selection_decimal": selection_decimal, "selection_decimal2": selection_decimal2,
    "selection_date": selection_date, "selection_combined": selection_combined,
    "hashjoin": hashjoin,
    "aggregation": aggregation, "aggregation2": aggregation2, "aggregation3": aggregation3,
    "aggregation4": aggregation4, "aggregation5": aggregation5,
    "orderby": orderby,
}


def check_against_literals(case: str, result: P.Result):
    """compare like the reference's checkRelations (test_common.h:196-222): as a multiset unless in_order"""
    want = expected_rows(case, result.types)
    got = [tuple(v if not isinstance(v, bytes) else v for v in row) for row in result.rows()]
    if LIT[case]["in_order"]:
        assert got == want, (case, got, want)
    else:
        assert Counter(got) == Counter(want), (case, got, want)
