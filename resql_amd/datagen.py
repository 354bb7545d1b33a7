"""Deterministic, counter-based TPC-H-shaped synthetic data — numpy side.

Every value is a pure function of (seed, stream, index), so any row range can be generated
independently (morsel sharding across GPUs) and the device generator
(resql_amd/csrc/datagen.hip) produces bit-identical columns.

Shapes follow the TPC-H specification's distributions for the columns Q1/Q3/Q6 touch:
  * orders come in blocks of 7 with line counts that are a permutation of 1..7 (28 lines per
    block, i.e. exactly 4 lines per order on average, 1-7 lines per order);
  * o_orderkey is sparse (first 8 keys of every 32), lineitem is clustered by l_orderkey;
  * l_extendedprice = l_quantity * retailprice(l_partkey) in cents;
  * l_shipdate = o_orderdate + [1,121] days, l_receiptdate = l_shipdate + [1,30] days,
    l_returnflag = R/A (50/50) if receiptdate <= 1995-06-17 else N,
    l_linestatus = O if shipdate > 1995-06-17 else F;
  * a third of the customers (custkey % 3 == 0) have no orders; 5 market segments, uniform.
"""
from __future__ import annotations

import numpy as np

SEED = 20240613
LINES_PER_BLOCK = 28
ORDERS_PER_BLOCK = 7

# random streams
S_QTY, S_PKEY, S_DISC, S_TAX, S_SHIP, S_RCPT, S_RFLG, S_ODATE, S_OCUST, S_PERM, S_CSEG = range(1, 12)
# synthetic 4 x int64 table
S_A, S_B, S_C, S_D = range(21, 25)

_BASE_COUNTS = np.array([4, 1, 7, 3, 5, 2, 6], dtype=np.int64)
# 14 line-count patterns: 7 rotations x {forward, reversed}
_PATTERNS = np.array([np.roll(_BASE_COUNTS, -k) for k in range(7)] +
                     [np.roll(_BASE_COUNTS[::-1], -k) for k in range(7)], dtype=np.int64)
_PREFIX = np.concatenate([np.zeros((14, 1), dtype=np.int64), np.cumsum(_PATTERNS, axis=1)], axis=1)  # [14][8]
# for every pattern and every offset 0..27: which of the 7 orders, and the line number
_ORDER_OF = np.zeros((14, LINES_PER_BLOCK), dtype=np.int64)
_LINE_OF = np.zeros((14, LINES_PER_BLOCK), dtype=np.int64)
for _p in range(14):
    for _j in range(7):
        for _l in range(_PATTERNS[_p, _j]):
            _ORDER_OF[_p, _PREFIX[_p, _j] + _l] = _j
            _LINE_OF[_p, _PREFIX[_p, _j] + _l] = _l + 1

EPOCH_1992 = 8035            # days from 1970-01-01 to 1992-01-01
ORDERDATE_SPAN = 2406        # 1992-01-01 .. 1998-08-02 inclusive
CUTOFF_DAY = 1263            # 1995-06-17 as days since 1992-01-01

M1 = np.uint64(0xBF58476D1CE4E5B9)
M2 = np.uint64(0x94D049BB133111EB)
G1 = np.uint64(0x9E3779B97F4A7C15)
G2 = np.uint64(0xD1342543DE82EF95)


def mix(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser of a (seed, stream, index) key — uint64 wrap-around arithmetic"""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.uint64(stream) * G1 + idx.astype(np.uint64) * G2
        z = (z ^ (z >> np.uint64(30))) * M1
        z = (z ^ (z >> np.uint64(27))) * M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(seed: int, stream: int, idx: np.ndarray, n: int) -> np.ndarray:
    """integer in [0, n), n < 2**32: high 32 bits scaled (multiply-high, no division)"""
    z = mix(seed, stream, idx)
    return (((z >> np.uint64(32)) * np.uint64(n)) >> np.uint64(32)).astype(np.int64)


def yyyymmdd(days_since_1992: np.ndarray) -> np.ndarray:
    """civil-from-days (proleptic Gregorian), vectorised; returns uint32 yyyymmdd"""
    z = days_since_1992.astype(np.int64) + EPOCH_1992 + 719468
    era = z // 146097
    doe = z - era * 146097
    yoe = (doe - doe // 1460 + doe // 36524 - doe // 146096) // 365
    y = yoe + era * 400
    doy = doe - (365 * yoe + yoe // 4 - yoe // 100)
    mp = (5 * doy + 2) // 153
    d = doy - (153 * mp + 2) // 5 + 1
    m = np.where(mp < 10, mp + 3, mp - 9)
    y = np.where(m <= 2, y + 1, y)
    return (y * 10000 + m * 100 + d).astype(np.uint32)


def orderkey_of(order_idx: np.ndarray) -> np.ndarray:
    """sparse order keys: 8 used of every 32"""
    return ((order_idx // 8) * 32 + (order_idx % 8) + 1).astype(np.int32)


def n_orders(sf: float) -> int:
    n = int(round(1_500_000 * sf))
    return max(ORDERS_PER_BLOCK, (n // ORDERS_PER_BLOCK) * ORDERS_PER_BLOCK)


def n_lineitem(sf: float) -> int:
    return n_orders(sf) * 4


def n_customer(sf: float) -> int:
    return max(3, int(round(150_000 * sf)))


def order_dates(order_idx: np.ndarray, seed: int = SEED) -> np.ndarray:
    return uniform(seed, S_ODATE, order_idx, ORDERDATE_SPAN)


def lineitem_columns(row0: int, n: int, sf: float, seed: int = SEED, columns=None) -> dict:
    """columns of lineitem rows [row0, row0+n) as numpy arrays in the engine's column widths"""
    r = np.arange(row0, row0 + n, dtype=np.int64)
    block = r // LINES_PER_BLOCK
    off = r % LINES_PER_BLOCK
    pat = uniform(seed, S_PERM, block, 14)
    o = block * ORDERS_PER_BLOCK + _ORDER_OF[pat, off]
    want = (lambda c: True) if columns is None else (lambda c: c in columns)
    out = {}
    if want("l_orderkey"):
        out["l_orderkey"] = orderkey_of(o)
    if want("l_linenumber"):
        out["l_linenumber"] = _LINE_OF[pat, off].astype(np.int32)
    qty = 1 + uniform(seed, S_QTY, r, 50)
    if want("l_quantity"):
        out["l_quantity"] = qty
    if want("l_extendedprice") or want("l_partkey"):
        pk = 1 + uniform(seed, S_PKEY, r, max(1, int(round(200_000 * sf))))
        if want("l_partkey"):
            out["l_partkey"] = pk.astype(np.int32)
        retail = 90000 + ((pk // 10) % 20001) + 100 * (pk % 1000)
        out["l_extendedprice"] = qty * retail
    if want("l_discount"):
        out["l_discount"] = uniform(seed, S_DISC, r, 11)
    if want("l_tax"):
        out["l_tax"] = uniform(seed, S_TAX, r, 9)
    ship = order_dates(o, seed) + 1 + uniform(seed, S_SHIP, r, 121)
    if want("l_shipdate"):
        out["l_shipdate"] = yyyymmdd(ship)
    if want("l_returnflag") or want("l_receiptdate"):
        rcpt = ship + 1 + uniform(seed, S_RCPT, r, 30)
        if want("l_receiptdate"):
            out["l_receiptdate"] = yyyymmdd(rcpt)
        ra = np.where(uniform(seed, S_RFLG, r, 2) == 0, ord("R"), ord("A"))
        out["l_returnflag"] = np.where(rcpt <= CUTOFF_DAY, ra, ord("N")).astype(np.uint8)
    if want("l_linestatus"):
        out["l_linestatus"] = np.where(ship > CUTOFF_DAY, ord("O"), ord("F")).astype(np.uint8)
    return out


def orders_columns(o0: int, n: int, sf: float, seed: int = SEED) -> dict:
    o = np.arange(o0, o0 + n, dtype=np.int64)
    nc = n_customer(sf)
    k = 1 + uniform(seed, S_OCUST, o, nc)
    k = np.where((k % 3 == 0), k - 1, k)          # customers with custkey % 3 == 0 place no orders
    k = np.maximum(k, 1)
    return {
        "o_orderkey": orderkey_of(o),
        "o_custkey": k.astype(np.int32),
        "o_orderdate": yyyymmdd(order_dates(o, seed)),
        "o_shippriority": np.zeros(n, dtype=np.int32),
    }


SEGMENTS = [b"AUTOMOBILE", b"BUILDING", b"FURNITURE", b"MACHINERY", b"HOUSEHOLD"]


def customer_columns(c0: int, n: int, sf: float, seed: int = SEED) -> dict:
    c = np.arange(c0, c0 + n, dtype=np.int64)
    seg = uniform(seed, S_CSEG, c, 5)
    segs = np.array(SEGMENTS, dtype="S10")
    return {
        "c_custkey": (c + 1).astype(np.int32),
        "c_mktsegment": segs[seg],
    }


def synthetic_columns(row0: int, n: int, groups: int, seed: int = SEED) -> dict:
    """the 4 x int64 filter + hash-agg table: a uniform [0,2^31), b group key in [0,groups),
    c,d in [0,2^20)"""
    r = np.arange(row0, row0 + n, dtype=np.int64)
    return {
        "a": uniform(seed, S_A, r, 1 << 31),
        "b": uniform(seed, S_B, r, groups),
        "c": uniform(seed, S_C, r, 1 << 20),
        "d": uniform(seed, S_D, r, 1 << 20),
    }
