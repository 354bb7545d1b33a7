"""BASELINE.json's full sizes (SF10: 59 999 996 lineitem rows, 15 M orders, 1.5 M customers; device-generated), checked
through size-independent properties — the oracle cannot run these sizes in seconds:
  * linearity: Q1 over the whole table == finalize(merge of the partial tables of two row-range shards), bit for bit;
  * a checksum of checksums: the per-group counts / sums of Q1 add up to an UNGROUPED aggregate with the same
    predicate, which runs through a different kernel shape;
  * Q3: the key-aligned two-shard run merges to exactly the unsharded top-10; revenue is sorted; repeatable.
Small-size parity against the oracle and the reference's goldens is in the other test files."""
import numpy as np
import pytest

from resql_amd import datagen, engine, plan as P, tpch
from resql_amd.dist import merge_ordered_results, shard_rows, shard_rows_on_key

pytestmark = pytest.mark.gpu
SF = 10.0


@pytest.fixture(scope="module")
def big(gpu_ctx):
    n = datagen.n_lineitem(SF)
    li = gpu_ctx.generate(engine.GEN_LINEITEM, n, SF, param=1)
    yield n, li
    li.close()


def _ungrouped_q1(table):
    p = P.Plan([table])
    qty, cnt = p.sum(p.attr("l_quantity")), p.count(p.star())
    price = p.sum(p.attr("l_extendedprice"))
    node = p.selection(p.le(p.attr("l_shipdate"), p.constant("1998-9-02", P.DATE)), p.scan("lineitem"))
    node = p.aggregation([qty, price, cnt], [], node)
    return p.set_root(p.materialize(p.projection([p.as_("q", qty), p.as_("p", price), p.as_("n", cnt)], node)))


def test_q1_sf10_linearity_and_checksums(gpu_ctx, big):
    n, li = big
    schema_only = tpch.lineitem_table(0.001, tpch.Q1_COLUMNS, n_rows=0)
    q = gpu_ctx.compile(tpch.q1_plan(schema_only), [li])
    q.execute()
    whole = q.result()
    assert whole.n_rows == 4 and q.report().kernel_time_ms < 1.0          # the headline kernel: 0.35 ms on MI355X
    # ---- linearity: two shards, partial tables merged as the multi-GPU path merges them ----
    n_min, n_max, n_sum = q.partial_layout()
    merged = None
    for rank in range(2):
        row0, cnt = shard_rows(n, 2, rank)
        shard = gpu_ctx.generate(engine.GEN_LINEITEM, cnt, SF, row0=row0, param=1)
        qs = gpu_ctx.compile(tpch.q1_plan(schema_only), [shard])
        assert qs.partial_layout() == (n_min, n_max, n_sum)
        import torch
        part = torch.zeros(n_min + n_max + n_sum, dtype=torch.int64, device="cuda:0")
        qs.bind_partial(part.data_ptr(), part.numel() * 8)
        qs.execute_partial()
        words = part.cpu().numpy()
        if merged is None:
            merged = words.copy()
        else:
            a, b = n_min, n_min + n_max
            merged[:a] = np.minimum(merged[:a], words[:a]); merged[a:b] = np.maximum(merged[a:b], words[a:b]); merged[b:] += words[b:]
        qs.close(); shard.close()
    q.finalize_host(merged)
    assert q.result().tuples == whole.tuples
    q.close()
    # ---- checksum of checksums against the ungrouped aggregate (register accumulators, one group) ----
    u = gpu_ctx.compile(_ungrouped_q1(schema_only), [li])
    u.execute()
    ur = u.result()
    u.close()
    col = {nm: i for i, nm in enumerate(whole.names)}
    assert sum(whole.value(r, col["count_order"]) for r in range(4)) == ur.value(0, 2)
    assert sum(whole.value(r, col["sum_qty"]) for r in range(4)) == ur.value(0, 0)
    assert sum(whole.value(r, col["sum_base_price"]) for r in range(4)) == ur.value(0, 1)
    assert 0.97 * n < ur.value(0, 2) <= n                                  # the predicate keeps ~98-99 % of the rows


def test_q3_sf10_sharded_equals_unsharded(gpu_ctx, big):
    n, li = big
    cu = gpu_ctx.generate(engine.GEN_CUSTOMER, datagen.n_customer(SF), SF)
    od = gpu_ctx.generate(engine.GEN_ORDERS, datagen.n_orders(SF), SF)
    plan = tpch.q3_plan(tpch.customer_table(0.001), tpch.orders_table(0.001), tpch.lineitem_table(0.001, tpch.Q3_LINEITEM_COLUMNS, n_rows=0))
    q = gpu_ctx.compile(plan, [cu, od, li])
    q.execute(); first = q.result()
    q.execute(); assert q.result().tuples == first.tuples                 # repeatable
    q.close()
    rev = [first.value(r, first.names.index("revenue")) for r in range(first.n_rows)]
    assert first.n_rows == 10 and rev == sorted(rev, reverse=True)
    # two key-aligned shards (keys of the rows around the nominal cut come from the host generator)
    def key_at(i):
        return int(datagen.lineitem_columns(i, 1, SF, columns={"l_orderkey"})["l_orderkey"][0])
    tuples = b""
    for rank in range(2):
        row0, cnt = shard_rows_on_key(n, 2, rank, key_at)
        shard = gpu_ctx.generate(engine.GEN_LINEITEM, cnt, SF, row0=row0, param=1)
        qs = gpu_ctx.compile(plan, [cu, od, shard])
        qs.execute()
        tuples += qs.result().tuples
        qs.close(); shard.close()
    both = P.Result(first.names, first.types, first.offsets, first.tuple_size, len(tuples) // first.tuple_size, tuples)
    merged = merge_ordered_results(None, both, [("revenue", False), ("o_orderdate", True)], 20, 1)
    assert merged.text.splitlines()[:11] == first.text.splitlines()
    cu.close(); od.close()
