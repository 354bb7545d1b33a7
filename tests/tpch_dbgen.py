"""A restatement of the TPC-H data generator (dbgen 2.x: build.c / rnd.c / bm_utils.c as published by the TPC) for the
columns the reference's queries read — TEST INFRASTRUCTURE, not part of the product.

Why: the reference ships SF 0.01 data files (tpch/datasets/sf001/*.tbl) and known answers for seven queries
(test/reference/q*.tbl, compared in order by test/test_queries.h:5-110) — but the snapshot lacks lineitem.tbl, so those answers
cannot be replayed as they are.  dbgen is deterministic: every column draws from its own Lehmer stream (x <- 16807 x mod
2^31 - 1) with a fixed seed and a fixed number of draws per row, so lineitem can be regenerated exactly.  This module is
pinned two ways (tests/test_tpch_known_answers.py):
  * every column it generates for orders / customer / part / supplier / partsupp / nation / region equals the reference's
    own .tbl files (o_totalprice and o_orderstatus are functions of the order's lineitems: price, discount, tax, ship
    date — so orders.tbl checks lineitem too);
  * the reference's known answers for Q1 / Q3 / Q5 / Q6 / Q12 / Q19 come out of the oracle and of the HIP engine when they
    run the reference's query texts over these tables (BULK INSERT of the generated .tbl files).
Text columns drawn from dbgen's 300 MB pseudo-text pool (comments, p_name) are not generated (filled with a placeholder): no
query with a committed answer reads them except Q10, which prints c_comment.

Streams, per-row draw counts ("boundaries") and value ranges follow dbgen's seed table and dss.h; the weighted string sets
(dists.dss) used here all have uniform weights, their orders are checked against the reference data.
"""
from __future__ import annotations

M = 2147483647            # 2^31 - 1
A = 16807


class Stream:
    """one Lehmer stream of dbgen's Seed[] table; `boundary` draws are consumed per row (row_stop pads the rest)"""
    __slots__ = ("value", "boundary", "usage")

    def __init__(self, seed: int, boundary: int):
        self.value, self.boundary, self.usage = seed, boundary, 0

    def uniform(self, lo: int, hi: int) -> int:            # dss_random / UnifInt (rnd.c)
        self.value = (A * self.value) % M
        self.usage += 1
        span = hi - lo + 1
        if lo == 0 and hi == 2147483647:
            span = -2147483648          # UnifInt computes this range in 32-bit ints: MAX_LONG - 0 + 1 wraps, the draws are negative
        return lo + int((self.value / 2147483647.0) * span)

    def row_stop(self):                                     # NthElement(boundary - usage)
        n = self.boundary - self.usage
        if n < 0:
            raise AssertionError("a stream drew more values than its per-row boundary")
        if n:
            self.value = (self.value * pow(A, n, M)) % M
        self.usage = 0


# dbgen's seed table (rnd.h): (seed, draws per row)
SEEDS = {
    "P_MFG": (1, 1), "P_BRND": (46831694, 1), "P_TYPE": (1841581359, 1), "P_SIZE": (1193163244, 1), "P_CNTR": (727633698, 1),
    "PS_QTY": (1671059989, 4), "PS_SCST": (1051288424, 4),
    "O_CLRK": (1171034773, 1), "O_ODATE": (1066728069, 1), "O_PRIO": (591449447, 1), "O_CKEY": (851767375, 1), "O_LCNT": (1434868289, 1),
    "L_QTY": (209208115, 7), "L_DCNT": (554590007, 7), "L_TAX": (721958466, 7), "L_SHIP": (1371272478, 7), "L_SMODE": (675466456, 7),
    "L_PKEY": (1808217256, 7), "L_SKEY": (2095021727, 7), "L_SDTE": (1769349045, 7), "L_CDTE": (904914315, 7), "L_RDTE": (373135028, 7),
    "L_RFLG": (717419739, 7),
    "C_ADDR": (881155353, 9), "C_NTRG": (1489529863, 1), "C_PHNE": (1521138112, 3), "C_ABAL": (298370230, 1), "C_MSEG": (1140279430, 1),
    "S_ADDR": (706178559, 9), "S_NTRG": (110356601, 1), "S_PHNE": (884434366, 3), "S_ABAL": (962338209, 1),
}

# dists.dss string sets with uniform weights (pick_str: index = RANDOM(1, n) - 1)
PRIORITIES = ["1-URGENT", "2-HIGH", "3-MEDIUM", "4-NOT SPECIFIED", "5-LOW"]
SEGMENTS = ["AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY"]
INSTRUCT = ["DELIVER IN PERSON", "COLLECT COD", "NONE", "TAKE BACK RETURN"]
SHIPMODES = ["REG AIR", "AIR", "RAIL", "TRUCK", "MAIL", "FOB", "SHIP"]      # (this order reproduces the reference's Q12 answer)
RFLAGS = ["R", "A"]
CONTAINERS = [f"{a} {b}" for a in ("SM", "LG", "MED", "JUMBO", "WRAP") for b in ("CASE", "BOX", "BAG", "JAR", "PACK", "PKG", "CAN", "DRUM")]
TYPES = [f"{a} {b} {c}" for a in ("STANDARD", "SMALL", "MEDIUM", "LARGE", "ECONOMY", "PROMO")
         for b in ("ANODIZED", "BURNISHED", "PLATED", "POLISHED", "BRUSHED") for c in ("TIN", "NICKEL", "BRASS", "STEEL", "COPPER")]
NATIONS = [("ALGERIA", 0), ("ARGENTINA", 1), ("BRAZIL", 1), ("CANADA", 1), ("EGYPT", 4), ("ETHIOPIA", 0), ("FRANCE", 3), ("GERMANY", 3),
           ("INDIA", 2), ("INDONESIA", 2), ("IRAN", 4), ("IRAQ", 4), ("JAPAN", 2), ("JORDAN", 4), ("KENYA", 0), ("MOROCCO", 0),
           ("MOZAMBIQUE", 0), ("PERU", 1), ("CHINA", 2), ("ROMANIA", 3), ("SAUDI ARABIA", 4), ("VIETNAM", 2), ("RUSSIA", 3),
           ("UNITED KINGDOM", 3), ("UNITED STATES", 1)]
REGIONS = ["AFRICA", "AMERICA", "ASIA", "EUROPE", "MIDDLE EAST"]
ALPHA_NUM = "0123456789abcdefghijklmnopqrstuvwxyz ABCDEFGHIJKLMNOPQRSTUVWXYZ,"

STARTDATE, CURRENTDATE_INDEX, TOTDATE = 92001, 1263, 2557        # 1995-06-17 is day 1263 after 1992-01-01 (julian 95168)
MONTH_DAYS = [31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31]


def date_text(day_index: int) -> str:
    """asc_date[]: day 0 = 1992-01-01, as yyyy-mm-dd"""
    y, d = 1992, day_index
    while True:
        n = 366 if y % 4 == 0 else 365
        if d < n:
            break
        d -= n
        y += 1
    for m in range(12):
        n = MONTH_DAYS[m] + (1 if m == 1 and y % 4 == 0 else 0)
        if d < n:
            return f"{y:04d}-{m + 1:02d}-{d + 1:02d}"
        d -= n
    raise AssertionError


def money(cents: int) -> str:
    sign = "-" if cents < 0 else ""
    cents = abs(cents)
    return f"{sign}{cents // 100}.{cents % 100:02d}"


def retail_price(partkey: int) -> int:                      # rpb_routine: cents
    return 90000 + (partkey // 10) % 20001 + (partkey % 1000) * 100


def sparse_key(i: int) -> int:                              # mk_sparse with SPARSE_BITS 2, SPARSE_KEEP 3, seq 0
    return ((i >> 3) << 5) | (i & 7)


def v_string(st: Stream, lo: int, hi: int) -> str:          # a_rnd: one draw for the length, one per five characters
    n = st.uniform(lo, hi)
    out, bits = [], 0
    for i in range(n):
        if i % 5 == 0:
            bits = st.uniform(0, 2147483647)
        out.append(ALPHA_NUM[bits & 0o77])
        bits >>= 6
    return "".join(out)


def phone(st: Stream, nation: int) -> str:                  # gen_phone
    a, b, c = st.uniform(100, 999), st.uniform(100, 999), st.uniform(1000, 9999)
    return f"{10 + nation:02d}-{a:03d}-{b:03d}-{c:04d}"


class Tables:
    """the eight tables at a scale factor below 1 (dbgen keeps `scale` = 1 there and scales the row counts)"""

    def __init__(self, sf: float = 0.01):
        assert sf <= 1
        self.n_supp, self.n_part = int(10000 * sf), int(200000 * sf)
        self.n_cust, self.n_orders = int(150000 * sf), int(1500000 * sf)
        self.s = {k: Stream(*v) for k, v in SEEDS.items()}
        self.nation = [(i, n, r) for i, (n, r) in enumerate(NATIONS)]
        self.region = list(enumerate(REGIONS))
        self.supplier = self._suppliers()
        self.part, self.partsupp = self._parts()
        self.customer = self._customers()
        self.orders, self.lineitem = self._orders()

    def _stop(self, prefix: str):
        for k, st in self.s.items():
            if k.startswith(prefix):
                st.row_stop()

    def _suppliers(self):
        s, rows = self.s, []
        for k in range(1, self.n_supp + 1):
            addr = v_string(s["S_ADDR"], 10, 40)
            nat = s["S_NTRG"].uniform(0, 24)
            ph = phone(s["S_PHNE"], nat)
            bal = s["S_ABAL"].uniform(-99999, 999999)
            rows.append({"s_suppkey": k, "s_name": f"Supplier#{k:09d}", "s_address": addr, "s_nationkey": nat, "s_phone": ph, "s_acctbal": bal})
            self._stop("S_")
        return rows

    def supp_of(self, partkey: int, i: int) -> int:         # PART_SUPP_BRIDGE
        n = self.n_supp
        return (partkey + i * (n // 4 + (partkey - 1) // n)) % n + 1

    def _parts(self):
        s, parts, ps = self.s, [], []
        for k in range(1, self.n_part + 1):
            mfgr = s["P_MFG"].uniform(1, 5)
            brand = s["P_BRND"].uniform(1, 5)
            ptype = TYPES[s["P_TYPE"].uniform(1, len(TYPES)) - 1]
            size = s["P_SIZE"].uniform(1, 50)
            cntr = CONTAINERS[s["P_CNTR"].uniform(1, len(CONTAINERS)) - 1]
            parts.append({"p_partkey": k, "p_mfgr": f"Manufacturer#{mfgr}", "p_brand": f"Brand#{mfgr}{brand}", "p_type": ptype, "p_size": size,
                          "p_container": cntr, "p_retailprice": retail_price(k)})
            for i in range(4):
                qty = s["PS_QTY"].uniform(1, 9999)
                cost = s["PS_SCST"].uniform(100, 100000)
                ps.append({"ps_partkey": k, "ps_suppkey": self.supp_of(k, i), "ps_availqty": qty, "ps_supplycost": cost})
            self._stop("P_"); self._stop("PS_")
        return parts, ps

    def _customers(self):
        s, rows = self.s, []
        for k in range(1, self.n_cust + 1):
            addr = v_string(s["C_ADDR"], 10, 40)
            nat = s["C_NTRG"].uniform(0, 24)
            ph = phone(s["C_PHNE"], nat)
            bal = s["C_ABAL"].uniform(-99999, 999999)
            seg = SEGMENTS[s["C_MSEG"].uniform(1, 5) - 1]
            rows.append({"c_custkey": k, "c_name": f"Customer#{k:09d}", "c_address": addr, "c_nationkey": nat, "c_phone": ph, "c_acctbal": bal,
                         "c_mktsegment": seg})
            self._stop("C_")
        return rows

    def _orders(self):
        s, orders, lines = self.s, [], []
        odate_max = TOTDATE - (121 + 30) - 1                 # O_ODATE_MAX - STARTDATE
        for i in range(1, self.n_orders + 1):
            okey = sparse_key(i)
            ck = s["O_CKEY"].uniform(1, self.n_cust)
            delta = 1
            while ck % 3 == 0:                               # CUST_MORTALITY: a third of the customers never order
                ck += delta
                ck = min(ck, self.n_cust)
                delta *= -1
            odate = s["O_ODATE"].uniform(STARTDATE, STARTDATE + odate_max) - STARTDATE
            prio = PRIORITIES[s["O_PRIO"].uniform(1, 5) - 1]
            clerk = s["O_CLRK"].uniform(1, 1000)
            n_lines = s["O_LCNT"].uniform(1, 7)
            total, shipped = 0, 0
            for ln in range(1, n_lines + 1):
                qty = s["L_QTY"].uniform(1, 50)
                disc = s["L_DCNT"].uniform(0, 10)
                tax = s["L_TAX"].uniform(0, 8)
                instr = INSTRUCT[s["L_SHIP"].uniform(1, 4) - 1]
                mode = SHIPMODES[s["L_SMODE"].uniform(1, 7) - 1]
                pk = s["L_PKEY"].uniform(1, self.n_part)
                sk = self.supp_of(pk, s["L_SKEY"].uniform(0, 3))
                eprice = retail_price(pk) * qty
                total += ((eprice * (100 - disc)) // 100) * (100 + tax) // 100
                sdate = odate + s["L_SDTE"].uniform(1, 121)
                cdate = odate + s["L_CDTE"].uniform(30, 90)
                rdate = sdate + s["L_RDTE"].uniform(1, 30)
                rflag = RFLAGS[s["L_RFLG"].uniform(1, 2) - 1] if rdate <= CURRENTDATE_INDEX else "N"
                if sdate <= CURRENTDATE_INDEX:
                    shipped += 1
                    status = "F"
                else:
                    status = "O"
                lines.append({"l_orderkey": okey, "l_partkey": pk, "l_suppkey": sk, "l_linenumber": ln, "l_quantity": qty, "l_extendedprice": eprice,
                              "l_discount": disc, "l_tax": tax, "l_returnflag": rflag, "l_linestatus": status, "l_shipdate": sdate,
                              "l_commitdate": cdate, "l_receiptdate": rdate, "l_shipinstruct": instr, "l_shipmode": mode})
            ostatus = "F" if shipped == n_lines else ("P" if shipped > 0 else "O")
            orders.append({"o_orderkey": okey, "o_custkey": ck, "o_orderstatus": ostatus, "o_totalprice": total, "o_orderdate": odate,
                           "o_orderpriority": prio, "o_clerk": f"Clerk#{clerk:09d}", "o_shippriority": 0})
            self._stop("O_"); self._stop("L_")
        return orders, lines

    # ---- '.tbl' text, the layout of tpch/create.sql; columns dbgen takes from its text pool hold `filler` -----------------
    def tbl_lines(self, table: str, filler: str = "x"):
        d, m = date_text, money
        if table == "lineitem":
            for r in self.lineitem:
                yield (f"{r['l_orderkey']}|{r['l_partkey']}|{r['l_suppkey']}|{r['l_linenumber']}|{r['l_quantity']}|{m(r['l_extendedprice'])}|"
                       f"0.{r['l_discount']:02d}|0.{r['l_tax']:02d}|{r['l_returnflag']}|{r['l_linestatus']}|{d(r['l_shipdate'])}|{d(r['l_commitdate'])}|"
                       f"{d(r['l_receiptdate'])}|{r['l_shipinstruct']}|{r['l_shipmode']}|{filler}|")
        elif table == "orders":
            for r in self.orders:
                yield (f"{r['o_orderkey']}|{r['o_custkey']}|{r['o_orderstatus']}|{m(r['o_totalprice'])}|{d(r['o_orderdate'])}|{r['o_orderpriority']}|"
                       f"{r['o_clerk']}|{r['o_shippriority']}|{filler}|")
        elif table == "customer":
            for r in self.customer:
                yield (f"{r['c_custkey']}|{r['c_name']}|{r['c_address']}|{r['c_nationkey']}|{r['c_phone']}|{m(r['c_acctbal'])}|{r['c_mktsegment']}|{filler}|")
        elif table == "part":
            for r in self.part:
                yield (f"{r['p_partkey']}|{filler}|{r['p_mfgr']}|{r['p_brand']}|{r['p_type']}|{r['p_size']}|{r['p_container']}|{m(r['p_retailprice'])}|{filler}|")
        elif table == "supplier":
            for r in self.supplier:
                yield (f"{r['s_suppkey']}|{r['s_name']}|{r['s_address']}|{r['s_nationkey']}|{r['s_phone']}|{m(r['s_acctbal'])}|{filler}|")
        elif table == "partsupp":
            for r in self.partsupp:
                yield f"{r['ps_partkey']}|{r['ps_suppkey']}|{r['ps_availqty']}|{m(r['ps_supplycost'])}|{filler}|"
        elif table == "nation":
            for k, n, r in self.nation:
                yield f"{k}|{n}|{r}|{filler}|"
        elif table == "region":
            for k, n in self.region:
                yield f"{k}|{n}|{filler}|"
        else:
            raise KeyError(table)

    def write(self, directory: str):
        import os
        for t in TABLE_NAMES:
            with open(os.path.join(directory, f"{t}.tbl"), "w") as f:
                for line in self.tbl_lines(t):
                    f.write(line + "\n")


TABLE_NAMES = ["lineitem", "orders", "customer", "part", "supplier", "partsupp", "nation", "region"]
# which fields of the reference's .tbl files this module generates (the others come from dbgen's text pool)
GENERATED_FIELDS = {
    "orders": [0, 1, 2, 3, 4, 5, 6, 7], "customer": [0, 1, 2, 3, 4, 5, 6], "part": [0, 2, 3, 4, 5, 6, 7], "supplier": [0, 1, 2, 3, 4, 5],
    "partsupp": [0, 1, 2, 3], "nation": [0, 1, 2], "region": [0, 1],
}
