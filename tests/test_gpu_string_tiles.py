"""String columns the leading selection compares with constants arrive as whole tiles (128 rows x W bytes, the wave's 16-byte loads)
through the wave's LDS region (codegen_internal.h strStaged, rsq_device.h ld_str_chunk / st_str_chunk / wave_lds_order), and the
records of a dictionary built in row order leave the same way (flush_tile_records).  Every width the staging takes (2..32 bytes; the last
chunk round of a tile is partial unless W is a multiple of 8), widths it leaves alone (> 32, or a second column beyond the region's 40
bytes per row), tables that end inside a tile, fewer tiles than waves and several tiles per wave: == the oracle, twice (the second
execution runs the specialised kernels for sure).

Found by this shape: the compiler PROVES that a lane's own 16-byte chunk stores never touch the bytes it reads back (they are other
lanes' chunks) and kept the previous tile's words in registers - hence wave_lds_order() between the stores and the loads."""
import numpy as np
import pytest

from resql_amd import plan as P
from oracle import orc

pytestmark = pytest.mark.gpu
T = P.TypeInit

WORDS = [b"MAIL", b"SHIP", b"AIR", b"AIR REG", b"DELIVER IN PERSON", b"TAKE BACK RETURN", b"NONE", b"COLLECT COD", b"a", b"ab", b"RAIL", b"TRUCK-AND-TRAILER-COMBINATION"]


def _strings(n, w, seed):
    rng = np.random.default_rng(seed)
    vocab = [x[:w] for x in WORDS]
    return np.array([vocab[i] for i in rng.integers(0, len(vocab), n)], dtype=f"S{w}")


def _run_twice(gpu_ctx, plan):
    want = orc.execute(plan)
    tabs = [gpu_ctx.table(t) for t in plan.tables]
    q = gpu_ctx.compile(plan, tabs)
    try:
        q.await_kernels()
        for _ in range(2):
            q.execute()
            got = q.result()
            assert sorted(got.text.splitlines()) == sorted(want.text.splitlines()) and got.n_rows == want.n_rows
    finally:
        q.close()
        for t in tabs:
            t.close()
    return want


@pytest.mark.parametrize("w", [2, 3, 7, 8, 9, 10, 15, 16, 17, 24, 25, 31, 32, 33, 44])
def test_one_compared_column_of_every_width(gpu_ctx, w):
    for n in (100, 128 * 5, 1500, 70_000):                       # inside one tile / whole tiles only / 11 tiles + 92 rows / several tiles per wave
        s = _strings(n, w, seed=w * 1000 + n)
        v = (np.arange(n, dtype=np.int64) * 7919) % 1000
        t = P.Table("t", [P.Column("s", T.CHAR(w), s), P.Column("v", T.BIGINT(), v)], n)
        for const, neq in ((b"MAIL", False), (b"AIR REG", False), (b"DELIVER IN PERSON", False), (b"SHIP", True)):
            c = const[:w].decode()
            p = P.Plan([t])
            cmp_ = (p.neq if neq else p.eq)(p.attr("s"), p.constant(c, P.VARCHAR))
            node = p.selection(cmp_, p.scan("t"))
            sm, cnt = p.sum(p.attr("v")), p.count(p.star())
            node = p.aggregation([sm, cnt], [], node)
            _run_twice(gpu_ctx, p.set_root(p.materialize(p.projection([p.as_("sv", sm), p.as_("cn", cnt)], node))))


def test_two_columns_in_and_beyond_the_region(gpu_ctx):
    """CHAR(25) + CHAR(10) share the region (TPC-H Q19's shape); CHAR(30) + CHAR(12) do not fit 40 bytes per row: the second is read row
    by row; an OR over the second column inside the AND; the rows materialised (count / scan / write passes, the write pass skips tiles)"""
    n = 40_000
    v = (np.arange(n, dtype=np.int64) * 104729) % 5000
    for wa, wb in ((25, 10), (30, 12), (32, 8), (16, 24)):
        a, b = _strings(n, wa, 5), _strings(n, wb, 6)
        t = P.Table("t", [P.Column("a", T.CHAR(wa), a), P.Column("b", T.CHAR(wb), b), P.Column("v", T.BIGINT(), v)], n)
        p = P.Plan([t])
        ca, c1, c2 = b"DELIVER IN PERSON"[:wa].decode(), b"AIR"[:wb].decode(), b"AIR REG"[:wb].decode()
        cond = p.and_(p.eq(p.attr("a"), p.constant(ca, P.VARCHAR)),
                      p.or_(p.eq(p.attr("b"), p.constant(c1, P.VARCHAR)), p.eq(p.attr("b"), p.constant(c2, P.VARCHAR))))
        node = p.selection(cond, p.scan("t"))
        want = _run_twice(gpu_ctx, p.set_root(p.materialize(p.projection([p.attr("v"), p.attr("b")], node))))
        assert want.n_rows > 50


@pytest.mark.parametrize("n_payload", [1, 2, 3, 6, 9])
def test_dictionary_records_in_row_order_leave_as_whole_lines(gpu_ctx, n_payload):
    """build side: a bare scan in key order with unique keys (entry number = row number), 1..9 payload words, 11 tiles + a tail; probed by
    a fact table; twice"""
    m, n = 1500, 30_000
    rng = np.random.default_rng(n_payload)
    cols = [P.Column("dk", T.INT(), np.arange(10, 10 + 3 * m, 3, dtype=np.int32))]
    for k in range(n_payload):
        cols.append(P.Column(f"p{k}", T.BIGINT(), rng.integers(-1000, 1000, m).astype(np.int64)))
    dim = P.Table("dim", cols, m)
    fact = P.Table("t", [P.Column("k", T.INT(), rng.integers(0, 20 + 3 * m, n).astype(np.int32)), P.Column("g", T.INT(), rng.integers(0, 5, n).astype(np.int32))], n)
    p = P.Plan([dim, fact])
    j = p.hashjoin([p.eq(p.attr("dk"), p.attr("k"))], p.scan("dim"), p.scan("t"), single_match=True)
    sums = [p.sum(p.attr(f"p{k}")) for k in range(n_payload)]
    node = p.aggregation(sums + [p.count(p.star())], [p.attr("g")], j)
    proj = [p.attr("g")] + [p.as_(f"s{k}", sums[k]) for k in range(n_payload)]
    _run_twice(gpu_ctx, p.set_root(p.materialize(p.projection(proj, node))))


def test_staged_tiles_in_front_of_a_large_dense_aggregation(gpu_ctx, monkeypatch):
    """the count / scatter kernels of a partitioned aggregation run as 1024-thread workgroups (16 waves, 16 LDS regions) from the same source:
    6 M rows, 2^17 dense groups, a CHAR(10) equality in front; every form of the large aggregation (RSQ_PARTITION 0 / default / 2)"""
    n, g = 6_000_000, 1 << 17
    rng = np.random.default_rng(3)
    s = _strings(n, 10, 9)
    b = rng.integers(0, g, n).astype(np.int64)
    c = rng.integers(0, 1000, n).astype(np.int64)
    t = P.Table("t", [P.Column("s", T.CHAR(10), s), P.Column("b", T.BIGINT(), b), P.Column("c", T.BIGINT(), c)], n)
    p = P.Plan([t])
    node = p.selection(p.eq(p.attr("s"), p.constant("MAIL", P.VARCHAR)), p.scan("t"))
    sm, cnt = p.sum(p.attr("c")), p.count(p.star())
    node = p.aggregation([sm, cnt], [p.attr("b")], node)
    plan = p.set_root(p.materialize(p.projection([p.attr("b"), p.as_("sv", sm), p.as_("cn", cnt)], node)))
    want = orc.execute(plan)
    for mode in (None, "0", "2"):
        if mode is None: monkeypatch.delenv("RSQ_PARTITION", raising=False)
        else: monkeypatch.setenv("RSQ_PARTITION", mode)
        tabs = [gpu_ctx.table(t)]
        q = gpu_ctx.compile(plan, tabs)
        try:
            q.await_kernels()
            for _ in range(2):
                q.execute()
                got = q.result()
                assert got.n_rows == want.n_rows and sorted(got.text.splitlines()) == sorted(want.text.splitlines()), mode
        finally:
            q.close(); tabs[0].close()
