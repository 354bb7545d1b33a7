"""Seeded random plans over random tables: the differential-testing half of the parity suite.

Every seed gives one (tables, plan) pair built with the same ExprGen / operator constructors the reference's
tests use (test/test_operators.h), restricted to the operators of the hot path: scan, selection,
hash join (single and multi match, strings carried across), aggregation (0..3 group keys incl. CHAR(n) / VARCHAR(n)
keys with trailing-space variants; sum/count/avg/min/max; CASE inside), projection,
materialize, order by, limit.  The same pair is fed to
  * the unmodified reference  vs  the oracle   (CPU suite, tests/test_oracle.py, when /root/reference is here)
  * the HIP engine            vs  the oracle   (GPU suite, tests/test_gpu_fuzz.py)
so the oracle is pinned on exactly the shapes the engine is later judged on.
"""
import random

import numpy as np

from resql_amd import plan as P

T = P.TypeInit
WORDS = ["AIR", "RAIL", "SHIP", "TRUCK", "MAIL", "FOB", "REG AIR", "BUILDING", "MACHINERY", "x", ""]


def _dates(rng, n):
    y = rng.integers(1992, 1999, n); m = rng.integers(1, 13, n); d = rng.integers(1, 29, n)
    return (y * 10000 + m * 100 + d).astype(np.uint32)


def _strings(rng, n, width, pad):
    out = []
    for i in rng.integers(0, len(WORDS), n):
        w = WORDS[i][:width]
        if pad and rng.integers(0, 3) == 0:
            w = (w + "   ")[:width]                   # CHAR compares equal up to trailing spaces (qlib/scalar.h:27-46)
        out.append(w.encode())
    return np.array(out, dtype=np.dtype(("S", width)))


def fact_table(rng, n, key_domain):
    cols = [
        P.Column("fk", T.BIGINT(), rng.integers(0, key_domain, n).astype(np.int64)),
        P.Column("fi", T.INT(), rng.integers(-5, 21, n).astype(np.int32)),
        P.Column("fc", T.CHAR(1), rng.choice(np.frombuffer(b"ABC", dtype=np.uint8), n)),
        P.Column("fd", T.DATE(), _dates(rng, n)),
        P.Column("fx", T.DECIMAL(12, 2), rng.integers(-100000, 1000000, n).astype(np.int64)),
        P.Column("fy", T.DECIMAL(12, 2), rng.integers(0, 11, n).astype(np.int64)),
        P.Column("fz", T.BIGINT(), rng.integers(-1000000, 1000000, n).astype(np.int64)),
        P.Column("fs", T.CHAR(10), _strings(rng, n, 10, True)),
        P.Column("fv", T.VARCHAR(12), _strings(rng, n, 12, False)),
        P.Column("fb", T.BOOL(), rng.integers(0, 2, n).astype(np.uint8)),
    ]
    return P.Table("f", cols, n)


def dim_table(rng, m, key_domain, unique):
    if unique:
        keys = rng.permutation(max(key_domain, m))[:m].astype(np.int64)
    else:
        keys = rng.integers(0, key_domain, m).astype(np.int64)
    cols = [
        P.Column("dk", T.BIGINT(), keys),
        P.Column("di", T.INT(), rng.integers(0, 7, m).astype(np.int32)),
        P.Column("dx", T.DECIMAL(10, 2), rng.integers(0, 5000, m).astype(np.int64)),
        P.Column("dd", T.DATE(), _dates(rng, m)),
        P.Column("dc", T.CHAR(1), rng.choice(np.frombuffer(b"XYZ", dtype=np.uint8), m)),
        P.Column("ds", T.CHAR(6), _strings(rng, m, 6, True)),
        P.Column("dv", T.VARCHAR(9), _strings(rng, m, 9, False)),
    ]
    return P.Table("d", cols, m)


class Gen:
    def __init__(self, seed):
        self.r = random.Random(seed)
        self.rng = np.random.default_rng(seed)
        self.seed = seed

    # ---- scalar expressions -----------------------------------------------------------------
    def numeric_leaf(self, p, cols):
        r = self.r
        k = r.random()
        if k < 0.6:
            return p.attr(r.choice(cols))
        if k < 0.8:
            return p.constant(str(r.randint(0, 40)), P.BIGINT)
        return p.constant(f"{r.randint(0, 9)}.{r.randint(0, 99):02d}", P.DECIMAL)

    def numeric(self, p, cols, depth):
        """Most expressions stay inside what the reference can compile (BIGINT / DECIMAL arithmetic; division only
        between BIGINTs: 'Decimal division not yet implemented', INT operands have no ADD/MUL emitters).  One in
        eight is unrestricted so that the refusals themselves are compared too."""
        r = self.r
        if r.random() < 0.125:
            return self._numeric(p, cols, depth, True)
        if r.random() < 0.3:
            big = [c for c in cols if c in ("fz", "fk", "dk")]
            return self._bigint(p, big, depth)
        return self._numeric(p, [c for c in cols if c not in ("fi", "di")], depth, False)

    def _bigint(self, p, cols, depth):
        r = self.r
        if depth == 0 or r.random() < 0.3:
            return p.attr(r.choice(cols)) if r.random() < 0.7 else p.constant(str(r.randint(0, 40)), P.BIGINT)
        op = r.choice(["add", "sub", "mul", "div"])
        l = self._bigint(p, cols, depth - 1)
        if op == "div":      # the reference traps on a zero divisor: keep divisors non-zero by construction
            return p.div(l, p.constant(str(r.choice([1, 2, 3, 7, 100])), P.BIGINT))
        return getattr(p, op)(l, self._bigint(p, cols, depth - 1))

    def _numeric(self, p, cols, depth, wild):
        r = self.r
        if depth == 0 or r.random() < 0.3:
            return self.numeric_leaf(p, cols)
        op = r.choice(["add", "sub", "mul", "add", "sub"] + (["div"] if wild else []))
        l = self._numeric(p, cols, depth - 1, wild)
        if op == "div":
            return p.div(l, p.constant(str(r.choice([1, 2, 3, 7, 100])), P.BIGINT))
        return getattr(p, op)(l, self._numeric(p, cols, depth - 1, wild))

    def predicate_leaf(self, p, side):
        r = self.r
        f = side == "f"
        kinds = ["num", "num", "date", "char1", "int"] + (["str", "vstr", "bool", "numnum"] if f else [])
        k = r.choice(kinds)
        cmp_ = getattr(p, r.choice(["lt", "le", "gt", "ge", "eq", "neq"]))
        if k == "num":
            col, lo, hi = r.choice([("fx", -100000, 1000000), ("fz", -1000000, 1000000), ("fk", 0, 50)] if f else [("dx", 0, 5000), ("dk", 0, 50)])
            c = r.randint(lo, hi)
            if col in ("fx", "dx"):
                return cmp_(p.attr(col), p.constant(f"{c // 100}.{abs(c) % 100:02d}" if c >= 0 else f"-{(-c) // 100}.{(-c) % 100:02d}", P.DECIMAL))
            return cmp_(p.attr(col), p.constant(str(c), P.BIGINT))
        if k == "numnum":
            return cmp_(self.numeric(p, ["fx", "fy", "fz", "fi"], 1), self.numeric(p, ["fx", "fy", "fz"], 1))
        if k == "int":
            return cmp_(p.attr("fi" if f else "di"), p.constant(str(r.randint(-5, 20)), P.BIGINT))
        if k == "date":
            return cmp_(p.attr("fd" if f else "dd"), p.constant(f"{r.randint(1992, 1998)}-{r.randint(1, 12):02d}-{r.randint(1, 28):02d}", P.DATE))
        if k == "char1":
            return getattr(p, r.choice(["eq", "neq"]))(p.attr("fc" if f else "dc"), p.constant(r.choice("ABCXYZ"), P.CHAR))
        if k == "str":
            return getattr(p, r.choice(["eq", "neq"]))(p.attr("fs"), p.constant(r.choice(WORDS[:-1]), P.VARCHAR))
        if k == "vstr":
            return getattr(p, r.choice(["eq", "neq"]))(p.attr("fv"), p.constant(r.choice(WORDS[:-1]), P.VARCHAR))
        return p.eq(p.attr("fb"), p.constant(r.choice(["true", "false"]), P.BOOL))

    def predicate(self, p, side, depth=2):
        r = self.r
        if depth == 0 or r.random() < 0.4:
            return self.predicate_leaf(p, side)
        op = r.choice([p.and_, p.and_, p.or_])
        return op(self.predicate(p, side, depth - 1), self.predicate(p, side, depth - 1))

    # ---- plan shapes ------------------------------------------------------------------------
    def aggregates(self, p, cols, joined):
        r = self.r
        aggs, names = [], []
        for i in range(r.randint(1, 4)):
            kind = r.choice(["sum", "sum", "count", "avg", "min", "max", "countstar", "sumcase"])
            if kind == "countstar":
                a = p.count(p.star())
            elif kind == "sumcase":
                zero = p.constant("0", P.BIGINT)
                a = p.sum(p.case(p.when_then(self.predicate_leaf(p, "f"), self.numeric(p, cols, 1)), zero))
            else:
                a = getattr(p, kind)(self.numeric(p, cols, 2))
            aggs.append(a); names.append(f"a{i}")
        return aggs, names

    def build(self):
        r = self.r
        n = r.choice([0, 1, 2, 63, 64, 129, 1000, 5000, 20000])
        key_domain = r.choice([3, 40, 1000, 100000])
        f = fact_table(self.rng, n, key_domain)
        shape = r.choice(["agg", "agg", "agg", "select", "joinagg", "joinagg", "joinmat"])
        if shape.startswith("join"):
            unique = r.random() < 0.6
            m = r.choice([0, 1, 30, 500, 3000])
            d = dim_table(self.rng, m, key_domain, unique)
            p = P.Plan([d, f])
            left = p.scan("d")
            if r.random() < 0.6:
                left = p.selection(self.predicate(p, "d", 1), left)
            right = p.scan("f")
            if r.random() < 0.6:
                right = p.selection(self.predicate(p, "f", 1), right)
            single = unique and r.random() < 0.7
            node = p.hashjoin([p.eq(p.attr("dk"), p.attr("fk"))], left, right, single_match=single)
            num_cols = ["fx", "fy", "fz", "fi", "dx", "di"]
            group_pool = ["fk", "fi", "fc", "di", "dc", "dd", "dk", "ds", "dv", "fs", "fv"]
            if shape == "joinmat":
                outs = [p.attr(c) for c in r.sample(["fk", "fx", "fd", "dx", "dd", "di", "dc", "fz", "ds", "dv", "fs"], r.randint(1, 5))]
                node = p.materialize(p.projection(outs, node))
                # matches of one probe row come out in hash-table order (the engine's table is not the reference's)
                limit = r.choice([None, None, 0, 5])
                self.kind = "exact" if single else ("multiset" if limit is None else "count")
                return p.set_root(node, limit=limit)
        else:
            p = P.Plan([f])
            node = p.scan("f")
            if r.random() < 0.75:
                node = p.selection(self.predicate(p, "f"), node)
            num_cols = ["fx", "fy", "fz", "fi", "fk"]
            group_pool = ["fk", "fi", "fc", "fd", "fb", "fs", "fv"]
            if shape == "select":
                if r.random() < 0.5:
                    outs = [p.attr(c) for c in r.sample(["fk", "fx", "fd", "fs", "fv", "fc", "fb", "fi"], r.randint(1, 5))]
                    if r.random() < 0.5:
                        outs.append(p.as_("e", self.numeric(p, num_cols, 2)))
                    node = p.projection(outs, node)
                    node = p.materialize(node)
                    self.kind = "exact"
                    return p.set_root(node, limit=r.choice([None, None, 0, 1, 100]))
                self.kind = "exact"
                if r.random() < 0.5:
                    order = [r.choice([p.asc, p.desc, lambda e: e])(p.attr(c)) for c in r.sample(["fk", "fi", "fd", "fx", "fc"], r.randint(1, 3))]
                    return p.set_root(p.orderby(order, node), limit=r.choice([None, 10, 300]), request_all=True)
                return p.set_root(p.materialize(node), limit=r.choice([None, None, 0, 7]), request_all=True)
        # aggregation on top of `node`
        ngroups = r.choice([0, 1, 1, 2, 3])
        gnames = r.sample(group_pool, ngroups)
        groups = [p.attr(g) for g in gnames]
        aggs, anames = self.aggregates(p, num_cols, shape != "agg")
        node = p.aggregation(aggs, groups, node)
        outs = list(groups) + [p.as_(nm, a) for nm, a in zip(anames, aggs)]
        if r.random() < 0.3 and len(aggs) >= 2:
            outs.append(p.as_("ratio", p.add(aggs[0], aggs[1])))
        node = p.projection(outs, node)
        self.kind = "exact"
        if shape == "joinagg" and not single:
            # groups first met by the same probe row are inserted in match order: emission order is the reference's
            # hash-table order there, so compare as a multiset and keep order by / limit out of it
            self.kind = "multiset"
            return p.set_root(p.materialize(node))
        if r.random() < 0.4:
            keys = [r.choice([p.asc, p.desc])(p.attr(nm)) for nm in r.sample(anames + gnames, min(2, len(anames) + len(gnames)))]
            node = p.orderby(keys, node)
            return p.set_root(node, limit=r.choice([None, 3, 10]))
        return p.set_root(p.materialize(node), limit=r.choice([None, None, 4]))


def make(seed):
    """-> (plan, comparison kind): "exact" = byte-identical serialisation, "multiset" = same rows in any order,
    "count" = same number of rows (a LIMIT over an order the reference leaves to its hash table)"""
    g = Gen(seed)
    plan = g.build()
    return plan, g.kind


def same(kind, got_text, want_text):
    from collections import Counter
    if kind == "exact":
        return got_text == want_text
    g, w = got_text.splitlines(), want_text.splitlines()
    if g[:1] != w[:1]:
        return False                                   # '#schema' line
    if kind == "multiset":
        return Counter(g) == Counter(w)
    return len(g) == len(w)


def digest(kind, text):
    """canonical sha256 of a serialised result under the comparison `kind`"""
    import hashlib
    lines = text.splitlines()
    if kind == "multiset":
        lines = lines[:1] + sorted(lines[1:])
    elif kind == "count":
        lines = lines[:1] + [str(len(lines) - 1)]
    return hashlib.sha256("\n".join(lines).encode("latin1")).hexdigest()
