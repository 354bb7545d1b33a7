"""Plan construction for the test / bench harness: a thin Python mirror of the reference's
`ExprGen::` helpers (reference src/expressions.h:518-705) and operator constructors
(reference src/operators/*.h) that produces the plain-C plan description of
include/resql_plan.h (ctypes structs below mirror that header field for field).

It also reads and writes the `resqlplan` text form of the same description, which is what
oracle/ref/ref_harness.cpp (the unmodified reference) consumes and what the golden fixtures
under tests/golden/ store.

This module only *describes* plans; it computes nothing.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

RSQ_SYMBOL_MAX = 64
RSQ_MAX_CHILDREN = 8
RSQ_MAX_OP_EXPRS = 32

# rsq_type_tag == SqlType::Tag (reference src/types.h:66-76)
VARCHAR, CHAR, BOOL, INT, BIGINT, DECIMAL, FLOAT, DATE, NT = range(9)
TYPE_NAMES = ["VARCHAR", "CHAR", "BOOL", "INT", "BIGINT", "DECIMAL", "FLOAT", "DATE", ""]

# rsq_expr_tag == Expr::Tag (reference src/expressions.h:25-63)
EXPR_TAGS = [
    "ADD", "SUB", "MUL", "DIV", "AND", "OR", "LT", "LE", "GT", "GE", "EQ", "NEQ", "LIKE",
    "SUM", "COUNT", "AVG", "MIN", "MAX", "ASC", "DESC", "CASE", "WHENTHEN",
    "ATTRIBUTE", "TYPECAST", "CONSTANT", "AS", "TYPE", "TABLE", "STAR", "UNDEFINED",
]
ETAG = {n: i for i, n in enumerate(EXPR_TAGS)}

# rsq_op_tag == RelOperator::OperatorTag (reference src/operators/RelOperator.h:25-35)
OP_TAGS = ["UNDEFINED", "SCAN", "PROJECTION", "SELECTION", "MATERIALIZE", "NESTEDLOOPSJOIN",
           "HASHJOIN", "AGGREGATION", "ORDERBY"]
OTAG = {n: i for i, n in enumerate(OP_TAGS)}


class rsq_type(C.Structure):
    _fields_ = [("tag", C.c_int32), ("precision", C.c_int32), ("scale", C.c_int32), ("len", C.c_int32)]


class rsq_expr(C.Structure):
    _fields_ = [("tag", C.c_int32), ("n_children", C.c_int32), ("child", C.c_int32 * RSQ_MAX_CHILDREN),
                ("const_category", C.c_int32), ("symbol", C.c_char * RSQ_SYMBOL_MAX)]


class rsq_op(C.Structure):
    _fields_ = [("tag", C.c_int32), ("child", C.c_int32 * 2), ("table", C.c_int32),
                ("n_exprs", C.c_int32), ("exprs", C.c_int32 * RSQ_MAX_OP_EXPRS),
                ("n_exprs2", C.c_int32), ("exprs2", C.c_int32 * RSQ_MAX_OP_EXPRS),
                ("single_match", C.c_int32)]


class rsq_plan_desc(C.Structure):
    _fields_ = [("exprs", C.POINTER(rsq_expr)), ("n_exprs", C.c_int32),
                ("ops", C.POINTER(rsq_op)), ("n_ops", C.c_int32),
                ("root", C.c_int32), ("request_all", C.c_int32), ("has_limit", C.c_int32),
                ("limit", C.c_int64)]


class rsq_column(C.Structure):
    _fields_ = [("name", C.c_char * RSQ_SYMBOL_MAX), ("type", rsq_type), ("data", C.c_void_p)]


class rsq_table_desc(C.Structure):
    _fields_ = [("name", C.c_char * RSQ_SYMBOL_MAX), ("n_rows", C.c_int64), ("n_cols", C.c_int32),
                ("cols", C.POINTER(rsq_column))]


class rsq_result_view(C.Structure):
    _fields_ = [("n_cols", C.c_int32), ("names", C.POINTER(C.c_char * RSQ_SYMBOL_MAX)),
                ("types", C.POINTER(rsq_type)), ("offsets", C.POINTER(C.c_int32)),
                ("tuple_size", C.c_int32), ("n_rows", C.c_int64), ("tuples", C.POINTER(C.c_uint8))]


# ------------------------------------------------------------------------------------------------
# SQL types (mirror of TypeInit::, reference src/types.h:177-206)
# ------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class SqlType:
    tag: int
    precision: int = 0
    scale: int = 0
    len: int = 0

    def c(self) -> rsq_type:
        return rsq_type(self.tag, self.precision, self.scale, self.len)

    def __str__(self) -> str:  # serializeType, reference src/types.h:121-150
        if self.tag == DECIMAL:
            return f"DECIMAL({self.precision},{self.scale})"
        if self.tag in (CHAR, VARCHAR):
            return f"{TYPE_NAMES[self.tag]}({self.len})"
        return TYPE_NAMES[self.tag]

    def text(self) -> str:  # resqlplan text form
        if self.tag == DECIMAL:
            return f"DECIMAL {self.precision} {self.scale}"
        if self.tag in (CHAR, VARCHAR):
            return f"{TYPE_NAMES[self.tag]} {self.len}"
        return TYPE_NAMES[self.tag]

    @property
    def width(self) -> int:
        """bytes per row in a columnar buffer (include/resql_plan.h, 'tables')"""
        return {INT: 4, DATE: 4, BIGINT: 8, DECIMAL: 8, BOOL: 1}.get(self.tag, self.len)

    @property
    def np_dtype(self):
        return {INT: np.int32, DATE: np.uint32, BIGINT: np.int64, DECIMAL: np.int64, BOOL: np.uint8}.get(
            self.tag, np.dtype(("S", self.len)) if self.len > 1 else np.uint8)


class TypeInit:
    INT = staticmethod(lambda: SqlType(INT))
    BIGINT = staticmethod(lambda: SqlType(BIGINT))
    DATE = staticmethod(lambda: SqlType(DATE))
    BOOL = staticmethod(lambda: SqlType(BOOL))
    DECIMAL = staticmethod(lambda p, s: SqlType(DECIMAL, p, s))
    CHAR = staticmethod(lambda n: SqlType(CHAR, len=n))
    VARCHAR = staticmethod(lambda n: SqlType(VARCHAR, len=n))


def parse_type(tokens: List[str]) -> SqlType:
    t = tokens.pop(0)
    if t == "DECIMAL":
        p = int(tokens.pop(0)); s = int(tokens.pop(0))
        return SqlType(DECIMAL, p, s)
    if t in ("CHAR", "VARCHAR"):
        return SqlType(CHAR if t == "CHAR" else VARCHAR, len=int(tokens.pop(0)))
    return SqlType(TYPE_NAMES.index(t))


def sqltype_from_c(t: rsq_type) -> SqlType:
    return SqlType(t.tag, t.precision, t.scale, t.len)


# ------------------------------------------------------------------------------------------------
# tables
# ------------------------------------------------------------------------------------------------
@dataclass
class Column:
    name: str
    type: SqlType
    data: Optional[np.ndarray] = None   # None: declared in the schema, never touched


@dataclass
class Table:
    name: str
    columns: List[Column]
    n_rows: int

    def col(self, name: str) -> Column:
        for c in self.columns:
            if c.name == name:
                return c
        raise KeyError(name)

    def to_c(self, keep: list) -> rsq_table_desc:
        """host-pointer table description; `keep` collects objects that must outlive the call"""
        cols = (rsq_column * len(self.columns))()
        for i, c in enumerate(self.columns):
            cols[i].name = c.name.encode()
            cols[i].type = c.type.c()
            if c.data is not None:
                arr = np.ascontiguousarray(c.data)
                expect = self.n_rows * c.type.width
                if arr.nbytes != expect:
                    raise ValueError(f"column {c.name}: {arr.nbytes} bytes, expected {expect}")
                keep.append(arr)
                cols[i].data = arr.ctypes.data
            else:
                cols[i].data = None
        keep.append(cols)
        td = rsq_table_desc()
        td.name = self.name.encode()
        td.n_rows = self.n_rows
        td.n_cols = len(self.columns)
        td.cols = cols
        return td


def tables_to_c(tables: Sequence[Table], keep: list):
    arr = (rsq_table_desc * max(1, len(tables)))()
    for i, t in enumerate(tables):
        arr[i] = t.to_c(keep)
    keep.append(arr)
    return arr


def column_from_strings(t: SqlType, values: Sequence[str]) -> np.ndarray:
    """Parse literal table cells the way the reference's test helper does
    (test/test_common.h:38-62 relationFromStrings -> parseConstant, expressions.h:369-515):
    a DECIMAL cell is its digits with the point removed."""
    if t.tag == DECIMAL:
        return np.array([int(v.replace(".", "")) for v in values], dtype=np.int64)
    if t.tag == BIGINT:
        return np.array([int(v) for v in values], dtype=np.int64)
    if t.tag == INT:
        return np.array([int(v) for v in values], dtype=np.int32)
    if t.tag == BOOL:
        return np.array([1 if v == "true" else 0 for v in values], dtype=np.uint8)
    if t.tag == DATE:
        out = []
        for v in values:
            y, m, d = v.replace("-", "/").split("/")
            out.append(int(y) * 10000 + int(m) * 100 + int(d))
        return np.array(out, dtype=np.uint32)
    if t.tag in (CHAR, VARCHAR):
        if t.len == 1:
            return np.array([ord(v[0]) if v else 0 for v in values], dtype=np.uint8)
        return np.array([v.encode()[: t.len] for v in values], dtype=np.dtype(("S", t.len)))
    raise ValueError(t)


def table_from_strings(name: str, schema: Sequence[Tuple[str, SqlType]], rows: Sequence[Sequence[str]]) -> Table:
    cols = []
    for j, (cn, ct) in enumerate(schema):
        cols.append(Column(cn, ct, column_from_strings(ct, [r[j] for r in rows])))
    return Table(name, cols, len(rows))


# ------------------------------------------------------------------------------------------------
# plan builder
# ------------------------------------------------------------------------------------------------
@dataclass
class ExprNode:
    tag: str
    children: List[int] = field(default_factory=list)
    symbol: str = ""
    category: int = NT


@dataclass
class OpNode:
    tag: str
    children: List[int] = field(default_factory=list)
    table: str = ""
    exprs: List[int] = field(default_factory=list)
    exprs2: List[int] = field(default_factory=list)
    single_match: bool = False


class Plan:
    """Mirror of how the reference's tests assemble plans: `ExprGen::` calls build the scalar
    expression DAG, operator constructors build the tree (test/test_operators.h)."""

    def __init__(self, tables: Sequence[Table] = ()):
        self.tables: List[Table] = list(tables)
        self.exprs: List[ExprNode] = []
        self.ops: List[OpNode] = []
        self.root: int = -1
        self.request_all: bool = False
        self.limit: Optional[int] = None

    # -- ExprGen:: (reference src/expressions.h:518-705) --
    def _e(self, tag, children=(), symbol="", category=NT) -> int:
        self.exprs.append(ExprNode(tag, list(children), symbol, category))
        return len(self.exprs) - 1

    def attr(self, name): return self._e("ATTRIBUTE", symbol=name)
    def constant(self, text, category): return self._e("CONSTANT", symbol=str(text), category=category)
    def star(self): return self._e("STAR", symbol="*")
    def as_(self, alias, child): return self._e("AS", [child], symbol=alias)
    def add(self, l, r): return self._e("ADD", [l, r])
    def sub(self, l, r): return self._e("SUB", [l, r])
    def mul(self, l, r): return self._e("MUL", [l, r])
    def div(self, l, r): return self._e("DIV", [l, r])
    def and_(self, l, r): return self._e("AND", [l, r])
    def or_(self, l, r): return self._e("OR", [l, r])
    def lt(self, l, r): return self._e("LT", [l, r])
    def le(self, l, r): return self._e("LE", [l, r])
    def gt(self, l, r): return self._e("GT", [l, r])
    def ge(self, l, r): return self._e("GE", [l, r])
    def eq(self, l, r): return self._e("EQ", [l, r])
    def neq(self, l, r): return self._e("NEQ", [l, r])
    def like(self, l, r): return self._e("LIKE", [l, r])
    def sum(self, c): return self._e("SUM", [c])
    def count(self, c): return self._e("COUNT", [c])
    def avg(self, c): return self._e("AVG", [c])
    def min(self, c): return self._e("MIN", [c])
    def max(self, c): return self._e("MAX", [c])
    def asc(self, c): return self._e("ASC", [c])
    def desc(self, c): return self._e("DESC", [c])
    def typecast(self, type_: "SqlType", c): return self._e("TYPECAST", [c], symbol=type_.text())
    def when_then(self, w, t): return self._e("WHENTHEN", [w, t])
    def case(self, *branches): return self._e("CASE", list(branches))

    def conjunction(self, conds: Sequence[int]) -> int:
        """planner.h:244-251: left-deep AND chain"""
        e = conds[0]
        for c in conds[1:]:
            e = self.and_(e, c)
        return e

    # -- operators --
    def _o(self, node: OpNode) -> int:
        self.ops.append(node)
        return len(self.ops) - 1

    def table_index(self, name: str) -> int:
        for i, t in enumerate(self.tables):
            if t.name == name:
                return i
        raise KeyError(name)

    def scan(self, table: str): return self._o(OpNode("SCAN", table=table))
    def selection(self, cond: int, child: int): return self._o(OpNode("SELECTION", [child], exprs=[cond]))
    def projection(self, exprs: Sequence[int], child: int): return self._o(OpNode("PROJECTION", [child], exprs=list(exprs)))
    def hashjoin(self, eqs: Sequence[int], left: int, right: int, single_match=False):
        return self._o(OpNode("HASHJOIN", [left, right], exprs=list(eqs), single_match=single_match))
    def aggregation(self, aggs: Sequence[int], groups: Sequence[int], child: int):
        return self._o(OpNode("AGGREGATION", [child], exprs=list(aggs), exprs2=list(groups)))
    def materialize(self, child: int): return self._o(OpNode("MATERIALIZE", [child]))
    def orderby(self, exprs: Sequence[int], child: int): return self._o(OpNode("ORDERBY", [child], exprs=list(exprs)))

    def set_root(self, op: int, limit: Optional[int] = None, request_all: bool = False):
        self.root = op
        self.limit = limit
        self.request_all = request_all
        return self

    # -- to the C structs of include/resql_plan.h --
    def to_c(self, keep: list) -> rsq_plan_desc:
        ex = (rsq_expr * max(1, len(self.exprs)))()
        for i, e in enumerate(self.exprs):
            ex[i].tag = ETAG[e.tag]
            ex[i].n_children = len(e.children)
            for k, ch in enumerate(e.children):
                ex[i].child[k] = ch
            ex[i].const_category = e.category
            ex[i].symbol = e.symbol.encode()
        ops = (rsq_op * max(1, len(self.ops)))()
        for i, o in enumerate(self.ops):
            ops[i].tag = OTAG[o.tag]
            ops[i].child[0] = o.children[0] if len(o.children) > 0 else -1
            ops[i].child[1] = o.children[1] if len(o.children) > 1 else -1
            ops[i].table = self.table_index(o.table) if o.tag == "SCAN" else -1
            ops[i].n_exprs = len(o.exprs)
            for k, v in enumerate(o.exprs):
                ops[i].exprs[k] = v
            ops[i].n_exprs2 = len(o.exprs2)
            for k, v in enumerate(o.exprs2):
                ops[i].exprs2[k] = v
            ops[i].single_match = 1 if o.single_match else 0
        keep += [ex, ops]
        d = rsq_plan_desc()
        d.exprs = ex; d.n_exprs = len(self.exprs)
        d.ops = ops; d.n_ops = len(self.ops)
        d.root = self.root
        d.request_all = 1 if self.request_all else 0
        d.has_limit = 1 if self.limit is not None else 0
        d.limit = self.limit or 0
        return d

    @staticmethod
    def from_c(d: rsq_plan_desc, table_names: Sequence[str], tables: Sequence[Table] = ()) -> "Plan":
        """the inverse of to_c: scan operators name their table through `table_names` (the array the description indexes)"""
        p = Plan(tables)
        for i in range(d.n_exprs):
            e = d.exprs[i]
            p.exprs.append(ExprNode(EXPR_TAGS[e.tag], [e.child[k] for k in range(e.n_children)], e.symbol.decode("latin1"),
                                    e.const_category))
        for i in range(d.n_ops):
            o = d.ops[i]
            tag = OP_TAGS[o.tag]
            kids = [o.child[k] for k in range(2) if o.child[k] >= 0]
            p.ops.append(OpNode(tag, kids, table_names[o.table] if tag == "SCAN" else "",
                                [o.exprs[k] for k in range(o.n_exprs)], [o.exprs2[k] for k in range(o.n_exprs2)],
                                bool(o.single_match)))
        p.root = d.root
        p.request_all = bool(d.request_all)
        p.limit = d.limit if d.has_limit else None
        return p

    # -- resqlplan text form --
    def to_text(self, table_sources: Optional[Dict[str, Dict[str, str]]] = None,
                tbl_files: Optional[Dict[str, str]] = None) -> str:
        """table_sources[table][column] = path of the raw little-endian column file ("bin"),
        columns without a path are declared "zero"; tbl_files[table] = '|' separated text rows."""
        out = ["resqlplan 1"]
        for t in self.tables:
            out.append(f"table {t.name} {t.n_rows}")
            for c in t.columns:
                src = (table_sources or {}).get(t.name, {}).get(c.name)
                out.append(f"col {c.name} {c.type.text()} " + (f"bin {src}" if src else "zero"))
            if tbl_files and t.name in tbl_files:
                out.append(f"tbl {tbl_files[t.name]}")
            out.append("end")
        for i, e in enumerate(self.exprs):
            if e.tag == "CONSTANT":
                out.append(f"expr {i} CONSTANT {TYPE_NAMES[e.category]} {e.symbol}")
            elif e.tag in ("ATTRIBUTE",):
                out.append(f"expr {i} ATTRIBUTE {e.symbol}")
            elif e.tag == "AS":
                out.append(f"expr {i} AS {e.symbol} {e.children[0]}")
            elif e.tag == "STAR":
                out.append(f"expr {i} STAR")
            elif e.tag == "TYPECAST":       # an explicit `expr :: type`: the symbol is the target type in text form
                out.append(f"expr {i} TYPECAST {e.children[0]} {e.symbol}")
            else:
                out.append(f"expr {i} {e.tag} " + " ".join(str(c) for c in e.children))
        for i, o in enumerate(self.ops):
            if o.tag == "SCAN":
                out.append(f"op {i} SCAN {o.table}")
            elif o.tag == "SELECTION":
                out.append(f"op {i} SELECTION {o.children[0]} {o.exprs[0]}")
            elif o.tag in ("PROJECTION", "ORDERBY"):
                out.append(f"op {i} {o.tag} {o.children[0]} {len(o.exprs)} " + " ".join(map(str, o.exprs)))
            elif o.tag == "HASHJOIN":
                out.append(f"op {i} HASHJOIN {o.children[0]} {o.children[1]} {int(o.single_match)} "
                           f"{len(o.exprs)} " + " ".join(map(str, o.exprs)))
            elif o.tag == "AGGREGATION":
                out.append((f"op {i} AGGREGATION {o.children[0]} {len(o.exprs)} " + " ".join(map(str, o.exprs))).rstrip()
                           + f" {len(o.exprs2)} " + " ".join(map(str, o.exprs2)))
            elif o.tag == "MATERIALIZE":
                out.append(f"op {i} MATERIALIZE {o.children[0]}")
            else:
                raise ValueError(o.tag)
        r = f"root {self.root}"
        if self.limit is not None:
            r += f" limit {self.limit}"
        if self.request_all:
            r += " requestall"
        out.append(r)
        return "\n".join(s.rstrip() for s in out) + "\n"

    @staticmethod
    def from_text(text: str, tables: Optional[Sequence[Table]] = None) -> "Plan":
        """Parse the resqlplan text form.  Table declarations give schemas; data must be supplied
        through `tables` (matched by name) or stays None."""
        p = Plan()
        given = {t.name: t for t in (tables or [])}
        cur: Optional[Table] = None
        exprs: Dict[int, ExprNode] = {}
        ops: Dict[int, OpNode] = {}
        for line in text.splitlines():
            tok = line.split()
            if not tok or tok[0].startswith("#") or tok[0] == "resqlplan":
                continue
            kw = tok.pop(0)
            if kw == "table":
                cur = Table(tok[0], [], int(tok[1]))
            elif kw == "col":
                name = tok.pop(0)
                cur.columns.append(Column(name, parse_type(tok)))
            elif kw == "tbl":
                pass
            elif kw == "end":
                p.tables.append(given.get(cur.name, cur))
                cur = None
            elif kw == "expr":
                i = int(tok.pop(0)); tag = tok.pop(0)
                if tag == "CONSTANT":
                    cat = TYPE_NAMES.index(tok.pop(0))
                    sym = line.split(None, 4)[4] if len(line.split(None, 4)) > 4 else ""
                    exprs[i] = ExprNode(tag, [], sym, cat)
                elif tag == "ATTRIBUTE":
                    exprs[i] = ExprNode(tag, [], tok[0])
                elif tag == "AS":
                    exprs[i] = ExprNode(tag, [int(tok[1])], tok[0])
                elif tag == "STAR":
                    exprs[i] = ExprNode(tag, [], "*")
                elif tag == "TYPECAST":
                    exprs[i] = ExprNode(tag, [int(tok[0])], " ".join(tok[1:]))
                else:
                    exprs[i] = ExprNode(tag, [int(x) for x in tok])
            elif kw == "op":
                i = int(tok.pop(0)); tag = tok.pop(0)
                if tag == "SCAN":
                    ops[i] = OpNode(tag, table=tok[0])
                elif tag == "SELECTION":
                    ops[i] = OpNode(tag, [int(tok[0])], exprs=[int(tok[1])])
                elif tag in ("PROJECTION", "ORDERBY"):
                    n = int(tok[1])
                    ops[i] = OpNode(tag, [int(tok[0])], exprs=[int(x) for x in tok[2:2 + n]])
                elif tag == "HASHJOIN":
                    n = int(tok[3])
                    ops[i] = OpNode(tag, [int(tok[0]), int(tok[1])], exprs=[int(x) for x in tok[4:4 + n]],
                                    single_match=bool(int(tok[2])))
                elif tag == "AGGREGATION":
                    n = int(tok[1]); aggs = [int(x) for x in tok[2:2 + n]]
                    m = int(tok[2 + n]); grps = [int(x) for x in tok[3 + n:3 + n + m]]
                    ops[i] = OpNode(tag, [int(tok[0])], exprs=aggs, exprs2=grps)
                elif tag == "MATERIALIZE":
                    ops[i] = OpNode(tag, [int(tok[0])])
                else:
                    raise ValueError(tag)
            elif kw == "root":
                p.root = int(tok.pop(0))
                while tok:
                    a = tok.pop(0)
                    if a == "limit":
                        p.limit = int(tok.pop(0))
                    elif a == "requestall":
                        p.request_all = True
            else:
                raise ValueError(f"unknown keyword {kw}")
        p.exprs = [exprs[i] for i in range(len(exprs))]
        p.ops = [ops[i] for i in range(len(ops))]
        return p


# ------------------------------------------------------------------------------------------------
# results
# ------------------------------------------------------------------------------------------------
@dataclass
class Result:
    names: List[str]
    types: List[SqlType]
    offsets: List[int]
    tuple_size: int
    n_rows: int
    tuples: bytes

    @staticmethod
    def from_view(v: rsq_result_view) -> "Result":
        n = v.n_cols
        names = [v.names[i].value.decode() for i in range(n)]
        types = [sqltype_from_c(v.types[i]) for i in range(n)]
        offs = [v.offsets[i] for i in range(n)]
        nbytes = v.tuple_size * v.n_rows
        data = C.string_at(v.tuples, nbytes) if nbytes else b""
        return Result(names, types, offs, v.tuple_size, v.n_rows, data)

    def value(self, row: int, col: int):
        """raw value: int for numeric/date/bool/char(1), bytes for strings"""
        t = self.types[col]
        base = row * self.tuple_size + self.offsets[col]
        b = self.tuples
        if t.tag in (BIGINT, DECIMAL):
            return int.from_bytes(b[base:base + 8], "little", signed=True)
        if t.tag == INT:
            return int.from_bytes(b[base:base + 4], "little", signed=True)
        if t.tag == DATE:
            return int.from_bytes(b[base:base + 4], "little", signed=False)
        if t.tag == BOOL or (t.tag == CHAR and t.len == 1):
            return b[base]
        raw = b[base:base + t.len + 1]
        return raw.split(b"\0", 1)[0]

    def rows(self) -> List[tuple]:
        return [tuple(self.value(r, c) for c in range(len(self.names))) for r in range(self.n_rows)]

    def serialize_value(self, row: int, col: int) -> str:
        """serializeSqlValue (reference src/values.h:30-127)"""
        t = self.types[col]
        v = self.value(row, col)
        if t.tag == CHAR:
            s = (chr(v) if v else "") if t.len == 1 else v.decode("latin1")
            return s + " " * (t.len - len(s))
        if t.tag == VARCHAR:
            return v.decode("latin1")
        if t.tag == DATE:
            return f"{v // 10000}/{v // 100 % 100:02d}/{v % 100:02d}"
        if t.tag in (INT, BIGINT):
            return str(v)
        if t.tag == BOOL:
            return "true" if v else "false"
        if t.tag == DECIMAL:
            neg = v < 0
            d = str(-v if neg else v)
            if len(d) <= t.scale:
                d = "0." + "0" * (t.scale - len(d)) + d
            elif t.scale > 0:
                d = d[:-t.scale] + "." + d[-t.scale:]
            return ("-" if neg else "") + d
        raise ValueError(t)

    def serialize(self, with_schema: bool = True) -> str:
        """serializeRelation (reference src/dbdata.h:688-701) preceded by a '#schema' line"""
        out = []
        if with_schema:
            out.append("#schema " + "".join(f"{n}:{t}|" for n, t in zip(self.names, self.types)))
        for r in range(self.n_rows):
            out.append("".join(self.serialize_value(r, c) + "|" for c in range(len(self.names))))
        return "\n".join(out) + "\n"
