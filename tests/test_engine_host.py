"""Host-side logic of the engine, on CPU: the C-ABI library loads and exports every symbol the header declares,
typing matches the reference's golden strings, plans lower to pipelines and compile for gfx950 (hiprtc
cross-compiles without a GPU), and the partial-table merge + finalisation step works from host memory.
No GPU compute here — a compile-only context (device = -1) cannot execute anything."""
import ctypes
import os
import re

import numpy as np
import pytest

from resql_amd import datagen, engine, plan as P, tpch
from oracle import orc

import refcases
from test_oracle import datatype_strings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "resql_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(rsq_[a-z_]+)\s*\(", header)))
    assert len(declared) >= 25
    lib = ctypes.CDLL(engine.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/resql_hip.h but not exported"
    assert sorted(engine.EXPORTED_SYMBOLS) == declared


def test_no_gpu_no_execution():
    """a context without a device can compile but must refuse to execute (no CPU fallback exists)"""
    ctx = engine.Context(device=-1)
    li = tpch.lineitem_table(0.01, tpch.Q6_COLUMNS, n_rows=1000)
    q = ctx.compile(tpch.q6_plan(li), [ctx.table(li)])
    with pytest.raises(engine.EngineError) as e:
        q.execute()
    assert e.value.status == 4      # RSQ_ERR_DEVICE


def test_engine_typing_matches_reference_strings(compile_ctx):
    want = refcases.LIT["datatypes"]
    for label, got in datatype_strings(lambda p, e, d: compile_ctx.serialize_expr(p, e, d)):
        assert got == want[label], label


def test_engine_typing_equals_oracle_on_tpch_expressions(compile_ctx):
    sf = 0.01
    li = tpch.lineitem_table(sf, tpch.Q1_COLUMNS + ["l_orderkey"], n_rows=128)
    tabs = [compile_ctx.table(li)]
    for plan in (tpch.q1_plan(li), tpch.q6_plan(li)):
        roots = [e for op in plan.ops for e in op.exprs + op.exprs2]
        for e in roots:
            assert compile_ctx.serialize_expr(plan, e, True, tabs) == orc.serialize_expr(plan, e, True)


def test_standard_plans_lower_and_compile_for_gfx950(compile_ctx):
    explains = []
    for plan in tpch.standard_plans(0.01):
        tabs = [compile_ctx.table(t) for t in plan.tables]
        q = compile_ctx.compile(plan, tabs)
        assert "__global__ void __launch_bounds__(RSQ_BLOCK_THREADS) rsq_p0_" in q.source and '#include "rsq_device.h"' in q.source
        explains.append(q.explain)
        r = q.report()
        assert r.jit_compiles + r.jit_cache_hits >= 1
    q1, q6, q3 = explains[0], explains[1], explains[2]
    assert "aggregation dense groups=6 accumulators=6 (of 11 in the reference) in registers" in q1
    assert "groups=1" in q6
    assert q3.count("pipeline") == 3 and "build hash table ht0" in q3 and "probe ht1 (single match)" in q3 \
        and "aggregation at the matched entry of ht1" in q3
    assert "workgroup LDS table" in explains[4] and "HBM table" in explains[5]


def test_code_object_cache_is_hit_on_recompile(compile_ctx):
    li = tpch.lineitem_table(0.01, tpch.Q6_COLUMNS, n_rows=256)
    t = compile_ctx.table(li)
    compile_ctx.compile(tpch.q6_plan(li), [t])
    r = compile_ctx.compile(tpch.q6_plan(li), [t]).report()
    assert r.jit_compiles == 0      # same pipeline shape: served from the in-memory / on-disk cache


def test_error_codes(compile_ctx):
    t = P.table_from_strings("rel", [("a", P.TypeInit.INT()), ("b", P.TypeInit.BIGINT())], [["1", "2"], ["2", "3"]])
    dt = compile_ctx.table(t)
    # SUM over INT: the reference throws "ADD code generation not implemented for datatype"
    p = P.Plan([t]); p.set_root(p.materialize(p.aggregation([p.sum(p.attr("a"))], [], p.scan("rel"))))
    with pytest.raises(engine.EngineError) as e:
        compile_ctx.compile(p, [dt])
    assert e.value.status == 2 and "ADD code generation" in e.value.message
    # unknown attribute
    p = P.Plan([t]); p.set_root(p.materialize(p.aggregation([p.sum(p.attr("zz"))], [], p.scan("rel"))))
    with pytest.raises(engine.EngineError) as e:
        compile_ctx.compile(p, [dt])
    assert e.value.status == 2 and "not found" in e.value.message
    # root must materialize
    p = P.Plan([t]); p.set_root(p.scan("rel"))
    with pytest.raises(engine.EngineError) as e:
        compile_ctx.compile(p, [dt])
    assert e.value.status == 1
    # LIKE between strings compiles (resql_amd/csrc/kernels/rsq_device.h: rsq::like)
    p = P.Plan([t])
    cond = p.like(p.constant("ab", P.VARCHAR), p.constant("a%", P.VARCHAR))
    p.set_root(p.materialize(p.aggregation([p.count(p.star())], [], p.selection(cond, p.scan("rel")))))
    compile_ctx.compile(p, [dt]).close()
    # one node twice below a parent would chain it to itself through the sibling pointer: refused, not walked for ever
    p = P.Plan([t])
    sb = p.sum(p.attr("b"))
    p.set_root(p.materialize(p.projection([p.as_("twice", p.add(sb, sb))], p.aggregation([sb], [], p.scan("rel")))))
    with pytest.raises(engine.EngineError) as e:
        compile_ctx.compile(p, [dt])
    assert e.value.status == 1 and "same child twice" in e.value.message
    # a nested-loops join is valid ReSQL but outside this engine's scope: refused, never emulated
    p = P.Plan([t])
    nl = p._o(P.OpNode("NESTEDLOOPSJOIN", [p.scan("rel"), p.scan("rel")], exprs=[p.eq(p.attr("a"), p.attr("a"))]))
    p.set_root(p.materialize(nl), request_all=True)
    with pytest.raises(engine.EngineError) as e:
        compile_ctx.compile(p, [dt])
    assert e.value.status == 3


def q1_partial_table_numpy(cols, row0, n_groups_rf=(65, 78, 82), n_groups_ls=(70, 79)):
    """numpy model of the dense partial aggregate table of Q1 for one row-range shard, in the layout the engine
    documents in `explain`: blocks [min:#firstrow | sum:qty | sum:price | sum:disc_price | sum:charge | sum:COUNT |
    sum:disc], word = block * 6 + group, group = rank(returnflag) * 2 + rank(linestatus)."""
    keep = cols["l_shipdate"] <= 19980902
    rf = np.searchsorted(np.array(n_groups_rf), cols["l_returnflag"])
    ls = np.searchsorted(np.array(n_groups_ls), cols["l_linestatus"])
    gid = rf * 2 + ls
    price, disc, tax, qty = cols["l_extendedprice"], cols["l_discount"], cols["l_tax"], cols["l_quantity"]
    disc_price = price * (100 - disc)
    charge = disc_price * (100 + tax)
    rows = np.arange(row0, row0 + len(qty), dtype=np.int64)
    table = np.zeros(7 * 6, dtype=np.int64)
    table[0:6] = np.iinfo(np.int64).max
    for g in range(6):
        m = keep & (gid == g)
        if m.any():
            table[g] = rows[m].min()
            for b, v in enumerate((qty, price, disc_price, charge, np.ones_like(qty), disc), start=1):
                table[b * 6 + g] = v[m].sum()
    return table


def test_finalize_from_host_partial_table(compile_ctx):
    """finalisation (emission-order replay, AVG, projection, ORDER BY) from a partial table in host memory"""
    n = 30_000
    li = tpch.lineitem_table(0.01, tpch.Q1_COLUMNS, n_rows=n)
    plan = tpch.q1_plan(li)
    q = compile_ctx.compile(plan, [compile_ctx.table(li)])
    assert "blocks=[min:#firstrow | sum:SUM" in q.explain and "l_returnflag{65,78,82} x l_linestatus{70,79}" in q.explain
    assert q.partial_layout() == (6, 0, 36)
    cols = {c.name: c.data for c in li.columns if c.data is not None}
    q.finalize_host(q1_partial_table_numpy(cols, 0))
    assert q.result().text == orc.execute(plan).text


def test_generator_shapes():
    sf = 0.01
    n = datagen.n_lineitem(sf)
    cols = datagen.lineitem_columns(0, n, sf)
    assert n == 4 * datagen.n_orders(sf)
    assert cols["l_quantity"].min() == 1 and cols["l_quantity"].max() == 50
    assert cols["l_discount"].min() == 0 and cols["l_discount"].max() == 10 and cols["l_tax"].max() == 8
    assert set(np.unique(cols["l_returnflag"])) == {ord("A"), ord("N"), ord("R")}
    assert set(np.unique(cols["l_linestatus"])) == {ord("F"), ord("O")}
    ok = cols["l_orderkey"]
    assert (np.diff(ok) >= 0).all()                                  # clustered by order key
    counts = np.unique(ok, return_counts=True)[1]
    assert counts.min() == 1 and counts.max() == 7                   # 1-7 lines per order
    assert ((ok - 1) % 32 < 8).all()                                 # sparse keys: 8 used of every 32
    assert 19920102 <= cols["l_shipdate"].min() and cols["l_shipdate"].max() <= 19981201
    # any row range is generated independently of the rest
    part = datagen.lineitem_columns(1000, 500, sf)
    for k in cols:
        assert (part[k] == cols[k][1000:1500]).all(), k
    od = datagen.orders_columns(0, datagen.n_orders(sf), sf)
    assert (od["o_custkey"] % 3 != 0).all() and od["o_custkey"].max() <= datagen.n_customer(sf)


def test_plan_text_roundtrip():
    sf = 0.01
    li = tpch.lineitem_table(sf, tpch.Q3_LINEITEM_COLUMNS, n_rows=64)
    plan = tpch.q3_plan(tpch.customer_table(sf), tpch.orders_table(sf), li)
    text = plan.to_text()
    again = P.Plan.from_text(text, plan.tables)
    assert again.to_text() == text
    assert orc.execute(again).text == orc.execute(plan).text


def test_plain_c_host_links_and_runs(tmp_path):
    """integration/examples/sql_host.c: the public headers are valid C11 and a C program can drive the statement loop through the
    C ABI alone (compile-only context here: a SELECT is parsed, planned and compiled, executing it needs a GPU)"""
    import subprocess
    exe = str(tmp_path / "sql_host")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "integration", "examples", "sql_host.c"), "-L" + os.path.join(ROOT, "resql_amd"),
                           "-lresql_hip", "-Wl,-rpath," + os.path.join(ROOT, "resql_amd"), "-o", exe])
    env = dict(os.environ, RSQ_DEVICE="-1")
    pr = subprocess.run([exe, "create table t ( a int, b char(3) )", "create table t ( a int )", "select a from t where",
                         "select a, count(*) from t group by a"], capture_output=True, text=True, env=env)
    out = pr.stdout.splitlines()
    assert out[0] == "create table ok"
    assert "Table t already exists." in out[1]
    assert "Syntax error." in out[2]
    assert out[3].startswith("error") and pr.returncode == 1          # compile-only context: no device to execute on


def test_config_struct_is_validated_without_a_gpu():
    """rsq_config.struct_size / emission_order / compat_flags are checked by rsq_ctx_create (compile-only context: no device needed)"""
    import ctypes as C
    L = engine.lib()
    h = C.c_void_p()
    cfg = engine.rsq_config.make(-1)
    cfg.struct_size = 0
    assert L.rsq_ctx_create(C.byref(cfg), C.byref(h)) == 1 and b"struct_size" in L.rsq_last_error(None)
    for bad in (engine.rsq_config.make(-1, emission_order=2), engine.rsq_config.make(-1, compat_flags=6)):
        assert L.rsq_ctx_create(C.byref(bad), C.byref(h)) == 1
    old_host = engine.rsq_config.make(-1, emission_order=2, compat_flags=6)      # fields its header never had are not read
    old_host.struct_size = engine.rsq_config.emission_order.offset
    assert L.rsq_ctx_create(C.byref(old_host), C.byref(h)) == 0
    L.rsq_ctx_destroy(h)
    ok = engine.Context(device=-1, compat_flags=engine.COMPAT_JIT_INT16_CAST)
    ok.close()


def test_multi_config_has_its_own_struct_size():
    """rsq_multi_config used to embed rsq_config BY VALUE, so a grown rsq_config moved devices / n_devices / merge for a
    host built against the older header.  It now leads with its own struct_size and points at the base config; rsq_multi_create reads
    struct_size bytes and validates them before it touches a device (no GPU here: the call fails on the devices, never on the layout)."""
    import ctypes as C
    L = engine.lib()
    h = C.c_void_p()
    devs = (C.c_int32 * 1)(0)
    base = engine.rsq_config.make(0)
    cfg = engine.rsq_multi_config(0, 1, C.pointer(base), devs, engine.MERGE_PEER_COPY, 0)
    assert L.rsq_multi_create(C.byref(cfg), C.byref(h)) == 1 and b"struct_size" in L.rsq_multi_last_error(None)
    cfg.struct_size = 5000
    assert L.rsq_multi_create(C.byref(cfg), C.byref(h)) == 1 and b"struct_size" in L.rsq_multi_last_error(None)
    # an "older host": its header ended before `merge`; what lies behind its struct is not read (a garbage merge mode would be refused)
    old = engine.rsq_multi_config(engine.rsq_multi_config.merge.offset, 1, C.pointer(base), devs, 77, 0)
    rc = L.rsq_multi_create(C.byref(old), C.byref(h))
    assert rc != 0 and b"merge mode" not in L.rsq_multi_last_error(None)          # (no GPU in this container: the devices fail, not the layout)
    full = engine.rsq_multi_config(C.sizeof(engine.rsq_multi_config), 1, C.pointer(base), devs, 77, 0)
    assert L.rsq_multi_create(C.byref(full), C.byref(h)) == 1 and b"merge mode" in L.rsq_multi_last_error(None)
    # a base config from an older header keeps working through the pointer: rsq_config's own struct_size covers it
    old_base = engine.rsq_config.make(0, emission_order=2)
    old_base.struct_size = engine.rsq_config.emission_order.offset
    cfg2 = engine.rsq_multi_config(C.sizeof(engine.rsq_multi_config), 1, C.pointer(old_base), devs, 77, 0)
    assert L.rsq_multi_create(C.byref(cfg2), C.byref(h)) == 1 and b"merge mode" in L.rsq_multi_last_error(None)


def test_engine_flags_and_memory_stats_without_a_gpu():
    import ctypes as C
    L = engine.lib()
    h = C.c_void_p()
    bad = engine.rsq_config.make(-1, engine_flags=8)
    assert L.rsq_ctx_create(C.byref(bad), C.byref(h)) == 1 and b"engine_flags" in L.rsq_last_error(None)
    ctx = engine.Context(device=-1, engine_flags=engine.ENGINE_DRIVER_ALLOC | engine.ENGINE_NO_PLAN_MEMO)
    st = ctx.memory_stats()
    assert st["device_slab_bytes"] == 0 and st["plan_memo_entries"] == 0
    small = engine.rsq_memory_stats()
    small.struct_size = engine.rsq_memory_stats.device_used_bytes.offset       # an older header: only the leading fields are written
    small.device_used_bytes = 12345
    assert L.rsq_ctx_memory_stats(ctx.h, C.byref(small)) == 0 and small.device_used_bytes == 12345
    small.struct_size = 4
    assert L.rsq_ctx_memory_stats(ctx.h, C.byref(small)) == 1
    ctx.close()
