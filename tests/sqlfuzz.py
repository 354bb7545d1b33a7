"""Random SQL statements for the front-end parity tests: text generated from the grammar of the reference's parser.y with
random operator mixes (so that the precedence ladder decides the shape), BETWEEN / IN / LIKE / CASE / typecast, negated
literals, asc / desc, plus damaged variants that must be refused.  Deterministic per seed."""
import random

IDENTS = ["a", "b", "c", "d", "l_quantity", "o_orderdate", "x1", "n_name"]
BINOPS = ["+", "-", "*", "/", "=", "<>", "<", "<=", ">", ">=", "and", "or"]
TYPES = ["int", "bigint", "date", "char(3)", "varchar(12)", "decimal(12,2)"]


class Gen:
    def __init__(self, seed: int):
        self.r = random.Random(seed)

    def const(self) -> str:
        k = self.r.randrange(7)
        if k == 0:
            return str(self.r.randrange(0, 100000))
        if k == 1:
            return f"{self.r.randrange(0, 1000)}.{self.r.randrange(0, 100):02d}"
        if k == 2:
            return "'" + self.r.choice(["A", "MAIL", "x y", "", "PROMO%", "it''s"[:4]]) + "'"
        if k == 3:
            return "date '19%02d-%02d-%02d'" % (self.r.randrange(92, 99), self.r.randrange(1, 13), self.r.randrange(1, 29))
        if k == 4:
            return "-" + str(self.r.randrange(1, 500))
        if k == 5:
            return "-" + f"{self.r.randrange(0, 10)}.{self.r.randrange(0, 1000):03d}"
        return '"' + self.r.choice(["q", "two words", "AIR REG"]) + '"'

    def expr(self, depth: int = 0) -> str:
        r = self.r
        if depth > 3 or r.random() < 0.3:
            return r.choice(IDENTS) if r.random() < 0.6 else self.const()
        k = r.randrange(12)
        if k < 5:
            return f"{self.expr(depth + 1)} {r.choice(BINOPS)} {self.expr(depth + 1)}"
        if k == 5:
            return f"({self.expr(depth + 1)})"
        if k == 6:
            return f"{self.expr(depth + 1)} between {self.expr(depth + 1)} and {self.expr(depth + 1)}"
        if k == 7:
            items = ", ".join(self.const() for _ in range(r.randrange(1, 4)))
            return f"{r.choice(IDENTS)} in ({items})"
        if k == 8:
            return f"{r.choice(IDENTS)} like '{r.choice(['%x', 'a_c%', '%', 'PROMO%'])}'"
        if k == 9:
            whens = " ".join(f"when {self.expr(depth + 1)} then {self.expr(depth + 1)}" for _ in range(r.randrange(1, 3)))
            els = f" else {self.expr(depth + 1)}" if r.random() < 0.7 else ""
            return f"case {whens}{els} end"
        if k == 10:
            return f"{r.choice(['sum', 'avg', 'min', 'max', 'count'])}({self.expr(depth + 1)})"
        return f"{self.expr(depth + 1)} :: {r.choice(TYPES)}"

    def select(self) -> str:
        r = self.r
        if r.random() < 0.1:
            sel = "*"
        else:
            items = []
            for _ in range(r.randrange(1, 4)):
                e = self.expr()
                if r.random() < 0.3:
                    e += f" as {r.choice(['s', 'revenue', 'k2'])}"
                items.append(e)
            sel = ", ".join(items)
        s = f"select {sel}"
        if r.random() < 0.9:
            s += " from " + ", ".join(r.sample(["t", "u", "lineitem", "orders"], r.randrange(1, 3)))
        if r.random() < 0.7:
            s += " where " + self.expr()
        if r.random() < 0.4:
            s += " group by " + ", ".join(self.expr(2) for _ in range(r.randrange(1, 3)))
        if r.random() < 0.4:
            keys = []
            for _ in range(r.randrange(1, 3)):
                k = r.choice(IDENTS)
                keys.append(k + r.choice(["", " asc", " desc"]))
            s += " order by " + ", ".join(keys)
        if r.random() < 0.3:
            s += f" limit {r.randrange(0, 50)}"
        return s

    def statement(self) -> str:
        r = self.r
        k = r.random()
        if k < 0.06:
            cols = ", ".join(f"c{i} {r.choice(TYPES)}" for i in range(r.randrange(1, 5)))
            return f"create table {r.choice(['t', 'nation'])} ( {cols} )"
        if k < 0.12:
            w = r.choice(["", ' with ( fieldterminator="|" )', " with (firstrow=2, fieldterminator=',')", ' with ( firstrow = 1 )'])
            return f'bulk insert {r.choice(["t", "part"])} from "data/{r.choice(["a", "b"])}.tbl"{w}'
        s = self.select()
        if r.random() < 0.25:        # damage it: drop / duplicate / swap a token-ish word
            words = s.split(" ")
            i = r.randrange(len(words))
            m = r.randrange(3)
            if m == 0:
                del words[i]
            elif m == 1:
                words.insert(i, words[i])
            else:
                j = r.randrange(len(words))
                words[i], words[j] = words[j], words[i]
            s = " ".join(words)
        return s


def statement(seed: int) -> str:
    return Gen(seed).statement()
