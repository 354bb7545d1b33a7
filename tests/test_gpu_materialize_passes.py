"""The two passes of a materialisation (codegen.cpp consumeMaterialize, codegen_loop.cpp "pairOk" / "liveWalk"): the count pass evaluates the
selection above the scan for BOTH rows of a lane and runs the rest once for the row that passed (a second time where both did); the write
pass looks at the counts of 64 of a wave's tiles at once and walks the tiles that counted something.  Output must be the reference's: every
tuple, in SCAN ORDER, also when rows emit several tuples (a join probed for all matches), when nearly every row passes (both rows of most
lanes), when nearly none does (few live tiles), and under LIMIT."""
import numpy as np
import pytest

from resql_amd import plan as P
from oracle import orc

pytestmark = pytest.mark.gpu
T = P.TypeInit


def _tables(n, m, pass_fraction, seed, unique_build=False):
    rng = np.random.default_rng(seed)
    tags = np.where(rng.random(n) < pass_fraction, b"AA", b"BB").astype("S4")
    fact = P.Table("t", [P.Column("rid", T.BIGINT(), np.arange(n, dtype=np.int64)), P.Column("k", T.INT(), rng.integers(0, m + 5, n).astype(np.int32)),
                         P.Column("s", T.CHAR(4), tags), P.Column("v", T.BIGINT(), rng.integers(-1000, 1000, n).astype(np.int64))], n)
    # build keys repeat: a probe for all matches emits several tuples per row (in the order of the reference's table: only the multiset is pinned)
    dk = rng.permutation(2 * m).astype(np.int32) if unique_build else rng.integers(0, m, 2 * m).astype(np.int32)
    dim = P.Table("dim", [P.Column("dk", T.INT(), dk), P.Column("dname", T.CHAR(6), np.array([f"d{i % 97:03d}".encode() for i in range(2 * m)], dtype="S6")),
                          P.Column("dw", T.BIGINT(), rng.integers(0, 50, 2 * m).astype(np.int64))], 2 * m)
    return dim, fact


def _plan(dim, fact, joined, limit=None):
    p = P.Plan([dim, fact])
    sel = p.selection(p.eq(p.attr("s"), p.constant("AA", P.VARCHAR)), p.scan("t"))
    if joined:
        node = p.hashjoin([p.eq(p.attr("dk"), p.attr("k"))], p.scan("dim"), sel, single_match=False)
        cols = [p.attr("rid"), p.attr("k"), p.attr("dname"), p.as_("x", p.add(p.attr("v"), p.attr("dw")))]
    else:
        node, cols = sel, [p.attr("rid"), p.attr("k"), p.attr("s"), p.attr("v")]
    p.set_root(p.materialize(p.projection(cols, node)))
    if limit is not None:
        p.limit = limit
    return p


@pytest.mark.parametrize("joined", [False, True])
@pytest.mark.parametrize("pass_fraction", [0.0005, 0.07, 0.5, 0.97])
def test_scan_order_is_kept_by_both_passes(gpu_ctx, pass_fraction, joined):
    dim, fact = _tables(400_000, 3_000, pass_fraction, seed=int(pass_fraction * 10_000) + joined)
    plan = _plan(dim, fact, joined)
    want = orc.execute(plan)
    tabs = [gpu_ctx.table(t) for t in plan.tables]
    q = gpu_ctx.compile(plan, tabs)
    try:
        for _ in range(3):                       # the first execution reads the total back, the next ones write with the remembered one
            q.execute()
            got = q.result()
            assert got.n_rows == want.n_rows
            if joined:       # the rows of the scan in scan order, each with all its matches (their order within a row is the build table's)
                rid = lambda text: [l.split("|")[0] for l in text.splitlines()[1:]]
                assert rid(got.text) == rid(want.text) and sorted(got.text.splitlines()) == sorted(want.text.splitlines())
            else:
                assert got.text == want.text
    finally:
        q.close()
        for t in tabs:
            t.close()


@pytest.mark.parametrize("limit", [1, 777])
def test_limit_takes_the_first_tuples_in_scan_order(gpu_ctx, limit):
    dim, fact = _tables(300_000, 2_000, 0.3, seed=91, unique_build=True)      # (one match per row: which tuples fall under the limit is then decided)
    plan = _plan(dim, fact, True, limit=limit)
    want = orc.execute(plan)
    tabs = [gpu_ctx.table(t) for t in plan.tables]
    q = gpu_ctx.compile(plan, tabs)
    try:
        for _ in range(2):
            q.execute()
            assert q.result().text == want.text
    finally:
        q.close()
        for t in tabs:
            t.close()
