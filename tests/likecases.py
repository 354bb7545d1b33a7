"""LIKE (reference src/qlib/scalar.h:49-118 stringLikeCheck) over a table of awkward strings x a list of awkward patterns.
One plan per string column: every row comes back with one 0/1 flag per pattern (CASE WHEN col LIKE pat THEN 1 ELSE 0)."""
import numpy as np

from resql_amd import plan as P

T = P.TypeInit

STRINGS = ["", "a", "ab", "abc", "abab", "ababab", "aXb", "a%b", "a_b", "abcabc", "xabc", "abcx", "xabcx", "MAIL", "AIR",
           "REG AIR", "AIR REG", "AIRMAIL", "special requests", "specialrequests", "a special request s", "requests special",
           "BRASS", "LARGE BRASS", "BRASSY", "aaa", "aaaa", "aaaaa", "baaab", "b", "ba", "bab", "abba", "abbba", "%", "_", "%%",
           "a b", "a  b", "ab ", " ab", "forest green", "green forest", "greenforestgreen", "PROMO BURNISHED", "PROMO"]
PATTERNS = ["", "%", "%%", "_", "__", "a", "a%", "%a", "%a%", "a%b", "a_b", "ab", "abab", "%abc", "abc%", "%abc%", "a%b%c",
            "%special%requests%", "%BRASS", "PROMO%", "forest%", "%green%", "a%a", "a%a%a", "_b%", "%b_", "%_", "_%", "b%b", "%ab%ab%"]


def table(kind: str) -> P.Table:
    n = len(STRINGS)
    typ = T.CHAR(20) if kind == "char" else T.VARCHAR(20)
    data = np.array([s.encode() for s in STRINGS], dtype=np.dtype(("S", 20)))
    return P.Table("s", [P.Column("id", T.BIGINT(), np.arange(n, dtype=np.int64)), P.Column("txt", typ, data)], n)


def plan(kind: str, patterns=None) -> P.Plan:
    t = table(kind)
    p = P.Plan([t])
    one, zero = p.constant("1", P.BIGINT), p.constant("0", P.BIGINT)
    outs = [p.attr("id")]
    for i, pat in enumerate(PATTERNS if patterns is None else patterns):
        flag = p.case(p.when_then(p.like(p.attr("txt"), p.constant(pat, P.VARCHAR)), one), zero)
        outs.append(p.as_(f"p{i}", flag))
    return p.set_root(p.materialize(p.projection(outs, p.scan("s"))))


def select_plan(kind: str, pattern: str) -> P.Plan:
    """LIKE as a selection predicate (the shape TPC-H uses: ... where p_type like '%BRASS')"""
    t = table(kind)
    p = P.Plan([t])
    sel = p.selection(p.like(p.attr("txt"), p.constant(pattern, P.VARCHAR)), p.scan("s"))
    return p.set_root(p.materialize(sel), request_all=True)
