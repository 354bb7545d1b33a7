"""Plans with string join keys of different kinds / lengths and string constants as keys, shared by the GPU parity test
(tests/test_gpu_string_join_kinds.py), the oracle-vs-reference test (tests/test_oracle_string_joins.py) and the script that
records the reference's answers (tests/golden/make_string_join_golden.py)."""
import itertools
import zlib

import numpy as np

from resql_amd import plan as P

T = P.TypeInit
KINDS = [("CHAR", 4), ("VARCHAR", 4), ("CHAR", 6), ("VARCHAR", 6), ("CHAR", 11), ("VARCHAR", 11)]
KIND_PAIRS = [(b, p) for b, p in itertools.product(KINDS, KINDS) if b != p]


def strs(vals, width):
    a = np.zeros((len(vals), width), dtype=np.uint8)
    for i, v in enumerate(vals):
        b = v.encode()[:width]
        a[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
    return a


def values(rng, n, width):
    out = []
    for _ in range(n):
        core = "".join(rng.choice(list("ab"), size=int(rng.integers(0, width + 1))))
        out.append((core + " " * int(rng.integers(0, 3)))[:width])
    return out


def pair_name(build, probe):
    return f"{build[0].lower()}{build[1]}_{probe[0].lower()}{probe[1]}"


def kinds_plan(build, probe, salt=0):
    """`salt` picks the data: the recording script skips data on which the REFERENCE loses matches - ht_get (qlib/hash.h:427-477)
    wraps around the table's end only behind a non-matching entry, so a probe that continues from a hash-equal entry in the LAST
    slot reads one entry past the table and stops; the strings over {a, b} used here hash alike whenever they are anagrams, and a
    chain of hash-equal entries across the table's end then hides its second half (seen: 'ab', 'ab ', 'ab  ' in slots 94-96 of
    97, 'ba', 'ba  ' in slots 0-1 never reached).  The engine and the oracle return every match."""
    rng = np.random.default_rng(zlib.crc32((pair_name(build, probe) + (f"#{salt}" if salt else "")).encode()))
    bt, pt = getattr(T, build[0])(build[1]), getattr(T, probe[0])(probe[1])
    bvals = sorted(set(values(rng, 60, build[1])))
    pvals = values(rng, 400, probe[1])
    dim = P.Table("dim", [P.Column("dk", bt, strs(bvals, build[1])), P.Column("dv", T.BIGINT(), np.arange(len(bvals), dtype=np.int64))], len(bvals))
    fact = P.Table("t", [P.Column("k", pt, strs(pvals, probe[1])), P.Column("v", T.BIGINT(), np.arange(len(pvals), dtype=np.int64))], len(pvals))
    p = P.Plan([dim, fact])
    j = p.hashjoin([p.eq(p.attr("dk"), p.attr("k"))], p.scan("dim"), p.scan("t"), single_match=False)
    return p.set_root(p.materialize(p.projection([p.attr("dv"), p.attr("v")], j)))


def fact_and_dim():
    rng = np.random.default_rng(1)
    n = 5000
    t = P.Table("t", [P.Column("k", T.VARCHAR(4), strs([["ab", "abcd", "x", "ab "][i % 4] for i in range(n)], 4)),
                      P.Column("g", T.INT(), rng.integers(0, 5, n).astype(np.int32)),
                      P.Column("v", T.BIGINT(), rng.integers(0, 100, n).astype(np.int64))], n)
    dim = P.Table("dim", [P.Column("dk", T.VARCHAR(4), strs(["ab", "zz", "abcd"], 4)), P.Column("dv", T.BIGINT(), np.arange(3, dtype=np.int64))], 3)
    return t, dim


def const_group_plan(kind, with_column):
    t, _ = fact_and_dim()
    p = P.Plan([t])
    s, c = p.sum(p.attr("v")), p.constant("abc", kind)
    keys = [c, p.attr("g")] if with_column else [c]
    return p.set_root(p.materialize(p.projection(keys + [p.as_("s", s)], p.aggregation([s], keys, p.scan("t")))))


def const_join_plan(kind, side):
    t, dim = fact_and_dim()
    p = P.Plan([dim, t])
    c = p.constant("ab", kind)
    eq = p.eq(p.attr("dk"), c) if side == "probe" else p.eq(c, p.attr("k"))
    j = p.hashjoin([eq], p.scan("dim"), p.scan("t"), single_match=False)
    s, cnt = p.sum(p.attr("v")), p.count(p.star())
    return p.set_root(p.materialize(p.projection([p.attr("dv"), p.as_("s", s), p.as_("c", cnt)], p.aggregation([s, cnt], [p.attr("dv")], j))))


def multi_match_agg_plan():
    rng = np.random.default_rng(3)
    n, m = 40_000, 300
    names = [f"k{i:03d}" for i in range(100)]
    dim = P.Table("dim", [P.Column("dk", T.VARCHAR(6), strs([names[i % 100] for i in range(m)], 6)),
                          P.Column("dv", T.BIGINT(), np.arange(m, dtype=np.int64))], m)
    fact = P.Table("t", [P.Column("k", T.VARCHAR(6), strs([names[int(i)] for i in rng.integers(0, 120, n) % 110 % 100], 6)),
                         P.Column("v", T.BIGINT(), rng.integers(0, 50, n).astype(np.int64))], n)
    p = P.Plan([dim, fact])
    j = p.hashjoin([p.eq(p.attr("dk"), p.attr("k"))], p.scan("dim"), p.scan("t"), single_match=False)
    s, cnt = p.sum(p.attr("v")), p.count(p.star())
    key = p.add(p.attr("dv"), p.constant("1", P.BIGINT))                 # a computed key: generic hash aggregation
    return p.set_root(p.materialize(p.projection([p.as_("key", key), p.as_("s", s), p.as_("c", cnt)], p.aggregation([s, cnt], [key], j))))


def all_cases(salts=None):
    """name -> plan builder, every case of this module; `salts`: case name -> data salt (from the golden file)"""
    salts = salts or {}
    cases = {"kinds_" + pair_name(b, pr): (lambda b=b, pr=pr: kinds_plan(b, pr, salts.get("kinds_" + pair_name(b, pr), 0))) for b, pr in KIND_PAIRS}
    for kind, kn in ((P.VARCHAR, "varchar"), (P.CHAR, "char")):
        for wc in (True, False):
            cases[f"const_group_{kn}_{'col' if wc else 'only'}"] = (lambda kind=kind, wc=wc: const_group_plan(kind, wc))
        for side in ("probe", "build"):
            cases[f"const_join_{kn}_{side}"] = (lambda kind=kind, side=side: const_join_plan(kind, side))
    cases["multi_match_agg"] = multi_match_agg_plan
    return cases


def canonical(text):
    """rows as a sorted multiset (join output order is the reference's hash-table order: compared as a multiset)"""
    lines = text.splitlines()
    return lines[:1] + sorted(lines[1:])
