"""ctypes binding of the engine's C ABI (include/resql_hip.h -> resql_amd/libresql_hip.so).

This is the whole Python side of the product: load the library, hand plans and columns across the
C ABI, read results back.  There is no Python or CPU implementation of any operator here — if the
library is missing the import fails, and on a machine without a GPU only compile-only contexts
(device = -1) can be created.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

from . import plan as P

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libresql_hip.so")

STATUS = {0: "OK", 1: "INVALID", 2: "TYPE", 3: "UNSUPPORTED", 4: "DEVICE", 5: "RUNTIME", 6: "NOMEM"}


class rsq_config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32),
                ("print_assembly", C.c_int32), ("print_flounder", C.c_int32), ("print_performance", C.c_int32),
                ("num_threads", C.c_int32), ("emit_machine_code", C.c_int32), ("optimize", C.c_int32),
                ("device", C.c_int32), ("kernel_cache_dir", C.c_char_p), ("emission_order", C.c_int32),
                ("compat_flags", C.c_uint32), ("engine_flags", C.c_uint32), ("reserved0", C.c_uint32),
                ("arena_reserve_bytes", C.c_int64), ("arena_keep_bytes", C.c_int64)]

    @classmethod
    def make(cls, device: int, cache_dir=None, print_source: bool = False, emission_order: int = 0, compat_flags: int = 0,
             engine_flags: int = 0, arena_reserve_bytes: int = 0, arena_keep_bytes: int = 0):
        return cls(C.sizeof(cls), 1 if print_source else 0, 0, 0, 1, 1, 0, device, cache_dir, emission_order, compat_flags,
                   engine_flags, 0, arena_reserve_bytes, arena_keep_bytes)


class rsq_report(C.Structure):
    _fields_ = [("compilation_time_ms", C.c_double), ("execution_time_ms", C.c_double),
                ("kernel_time_ms", C.c_double), ("finalize_time_ms", C.c_double),
                ("num_kernels", C.c_uint64), ("bytes_read", C.c_uint64), ("hbm_gbps", C.c_double),
                ("jit_cache_hits", C.c_int32), ("jit_compiles", C.c_int32)]


class rsq_memory_stats(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32), ("device_slab_bytes", C.c_uint64), ("device_used_bytes", C.c_uint64),
                ("device_slab_allocs", C.c_uint64), ("pinned_slab_bytes", C.c_uint64), ("pinned_used_bytes", C.c_uint64),
                ("pinned_slab_allocs", C.c_uint64), ("raw_driver_calls", C.c_uint64), ("arena_requests", C.c_uint64), ("driver_ms", C.c_double),
                ("plan_memo_entries", C.c_uint64), ("plan_memo_hits", C.c_uint64), ("key_index_entries", C.c_uint64), ("key_index_bytes", C.c_uint64)]


class rsq_multi_config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_devices", C.c_int32), ("base", C.POINTER(rsq_config)), ("devices", C.POINTER(C.c_int32)),
                ("merge", C.c_int32), ("reserved0", C.c_int32)]


MERGE_AUTO, MERGE_RCCL, MERGE_PEER_COPY = 0, 1, 2
EMIT_REFERENCE, EMIT_ANY = 0, 1
COMPAT_JIT_INT16_CAST = 1      # rsq_compat: TYPECAST INT -> BIGINT as the reference's asmjit JIT executes it (low 16 bits)
ENGINE_DRIVER_ALLOC, ENGINE_NO_PLAN_MEMO = 1, 2      # rsq_engine_flags


class EngineError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"[{STATUS.get(status, status)}] {message}")
        self.status = status
        self.message = message


_lib = None


def lib():
    """load libresql_hip.so; raises if it has not been built (no fallback)"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C resql_amd/csrc` "
                              f"(or __graft_entry__.build()); the engine has no Python/CPU fallback")
        # PyTorch-ROCm wheels bundle their own HIP runtime.  If this library brings the system runtime up first and
        # torch is imported later in the same process (multi-GPU merge, torch.distributed), torch's copy reports
        # "No HIP GPUs are available".  Loading torch's runtime first makes both share it.  Skipped when torch is not
        # installed or RSQ_NO_TORCH_PRELOAD is set (pure C/C++ hosts never see this).
        if "torch" not in __import__("sys").modules and not os.environ.get("RSQ_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = C.CDLL(LIB_PATH)
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        L.rsq_ctx_create.argtypes = [C.POINTER(rsq_config), C.POINTER(vp)]
        L.rsq_ctx_destroy.argtypes = [vp]
        L.rsq_last_error.restype = C.c_char_p
        L.rsq_ctx_memory_stats.argtypes = [vp, C.POINTER(rsq_memory_stats)]
        L.rsq_last_error.argtypes = [vp]
        L.rsq_table_create.argtypes = [vp, C.POINTER(P.rsq_table_desc), C.POINTER(vp)]
        L.rsq_table_create_device.argtypes = [vp, C.POINTER(P.rsq_table_desc), C.POINTER(vp)]
        L.rsq_table_from_rowstore.argtypes = [vp, C.POINTER(P.rsq_table_desc), C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_size_t), i32, C.POINTER(vp)]
        L.rsq_table_generate.argtypes = [vp, i32, i64, i64, C.c_double, i64, C.c_uint64, C.POINTER(vp)]
        L.rsq_table_load_tbl.argtypes = [vp, C.POINTER(P.rsq_table_desc), C.c_char_p, C.c_char, i32, C.POINTER(vp)]
        L.rsq_table_rows.restype = i64
        L.rsq_table_rows.argtypes = [vp]
        L.rsq_table_read_column.argtypes = [vp, vp, C.c_char_p, vp, C.c_size_t]
        L.rsq_table_set_first_row.argtypes = [vp, i64]
        L.rsq_table_refresh_stats.argtypes = [vp]
        L.rsq_table_append.argtypes = [vp, vp]
        L.rsq_table_stats_bytes.restype = i64
        L.rsq_table_stats_bytes.argtypes = [vp]
        L.rsq_table_stats_export.argtypes = [vp, vp, i64]
        L.rsq_table_unify_shard_stats.argtypes = [vp, vp, i32, i64]
        L.rsq_table_total_rows.restype = i64
        L.rsq_table_total_rows.argtypes = [vp]
        L.rsq_table_destroy.argtypes = [vp]
        L.rsq_query_compile.argtypes = [vp, C.POINTER(P.rsq_plan_desc), C.POINTER(vp), i32, C.POINTER(vp)]
        L.rsq_query_execute.argtypes = [vp]
        L.rsq_query_await_kernels.argtypes = [vp]
        L.rsq_query_execute_partial.argtypes = [vp, C.POINTER(vp), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
        L.rsq_query_execute_partial_async.argtypes = [vp]
        L.rsq_ctx_set_stream.argtypes = [vp, vp, i32]
        L.rsq_query_finalize.argtypes = [vp]
        L.rsq_query_bind_partial.argtypes = [vp, vp, C.c_size_t]
        L.rsq_query_merge_gathered.argtypes = [vp, vp, i32]
        L.rsq_query_finalize_host.argtypes = [vp, C.POINTER(i64), i64]
        L.rsq_query_partial_layout.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
        L.rsq_query_result.argtypes = [vp, C.POINTER(P.rsq_result_view)]
        L.rsq_query_report.argtypes = [vp, C.POINTER(rsq_report)]
        L.rsq_query_kernel_time_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int32]
        L.rsq_query_source.restype = C.c_char_p
        L.rsq_query_source.argtypes = [vp]
        L.rsq_query_explain.restype = C.c_char_p
        L.rsq_query_explain.argtypes = [vp]
        L.rsq_query_destroy.argtypes = [vp]
        L.rsq_serialize_expr.restype = vp
        L.rsq_serialize_expr.argtypes = [vp, C.POINTER(P.rsq_plan_desc), i32, i32, C.POINTER(vp), i32]
        L.rsq_result_serialize.restype = vp
        L.rsq_result_serialize.argtypes = [C.POINTER(P.rsq_result_view)]
        L.rsq_free.argtypes = [vp]
        L.rsq_ref_emission_order.argtypes = [vp, i64, C.c_uint64, i32, vp]
        L.rsq_ref_emission_order_device.argtypes = [vp, vp, i64, C.c_uint64, vp]
        L.rsq_measure_read_bandwidth.argtypes = [vp, C.c_size_t, i32, C.POINTER(C.c_double)]
        L.rsq_sql_plan_select.argtypes = [vp, C.c_char_p, C.POINTER(vp), i32, C.POINTER(vp)]
        L.rsq_sql_plan_desc.restype = C.POINTER(P.rsq_plan_desc)
        L.rsq_sql_plan_desc.argtypes = [vp]
        L.rsq_sql_plan_destroy.argtypes = [vp]
        L.rsq_sql_plan_text.restype = vp
        L.rsq_sql_plan_text.argtypes = [vp]
        L.rsq_sql_compile.argtypes = [vp, C.c_char_p, C.POINTER(vp), i32, C.POINTER(vp)]
        L.rsq_sql_describe.restype = vp
        L.rsq_sql_describe.argtypes = [vp, C.c_char_p, i32]
        L.rsq_db_create.argtypes = [vp, C.POINTER(vp)]
        L.rsq_db_execute.argtypes = [vp, C.c_char_p, C.POINTER(i32), C.POINTER(P.rsq_result_view)]
        L.rsq_db_adopt_table.argtypes = [vp, vp]
        L.rsq_db_message.restype = C.c_char_p
        L.rsq_db_message.argtypes = [vp]
        L.rsq_db_report.argtypes = [vp, C.POINTER(rsq_report)]
        L.rsq_db_destroy.argtypes = [vp]
        L.rsq_multi_create.argtypes = [C.POINTER(rsq_multi_config), C.POINTER(vp)]
        L.rsq_multi_destroy.argtypes = [vp]
        L.rsq_multi_last_error.restype = C.c_char_p
        L.rsq_multi_last_error.argtypes = [vp]
        L.rsq_multi_devices.restype = i32
        L.rsq_multi_devices.argtypes = [vp]
        L.rsq_multi_ctx.restype = vp
        L.rsq_multi_ctx.argtypes = [vp, i32]
        L.rsq_multi_merge_name.restype = C.c_char_p
        L.rsq_multi_merge_name.argtypes = [vp]
        L.rsq_multi_shard_rows.restype = None
        L.rsq_multi_shard_rows.argtypes = [i64, i32, i32, C.POINTER(i64), C.POINTER(i64)]
        L.rsq_multi_table_generate.argtypes = [vp, i32, i64, C.c_double, i64, C.c_uint64, C.POINTER(vp)]
        L.rsq_multi_table_generate_on_key.argtypes = [vp, i32, i64, C.c_double, i64, C.c_uint64, C.c_char_p, C.POINTER(vp)]
        L.rsq_multi_query_merge_name.restype = C.c_char_p
        L.rsq_multi_query_merge_name.argtypes = [vp]
        L.rsq_multi_query_compile.argtypes = [vp, C.POINTER(P.rsq_plan_desc), C.POINTER(vp), i32, C.POINTER(vp)]
        L.rsq_multi_query_execute.argtypes = [vp]
        L.rsq_multi_query_result.argtypes = [vp, C.POINTER(P.rsq_result_view)]
        L.rsq_multi_query_report.argtypes = [vp, C.POINTER(rsq_report), C.POINTER(C.c_double)]
        L.rsq_multi_query_collective_ms.restype = C.c_double
        L.rsq_multi_query_collective_ms.argtypes = [vp]
        L.rsq_multi_query_destroy.argtypes = [vp]
        _lib = L
    return _lib


EXPORTED_SYMBOLS = [
    "rsq_ctx_create", "rsq_ctx_destroy", "rsq_last_error", "rsq_table_create", "rsq_table_create_device",
    "rsq_table_from_rowstore", "rsq_table_load_tbl", "rsq_table_generate", "rsq_table_rows", "rsq_table_set_first_row", "rsq_table_refresh_stats", "rsq_table_append", "rsq_ctx_memory_stats", "rsq_table_read_column",
    "rsq_table_stats_bytes", "rsq_table_stats_export", "rsq_table_unify_shard_stats", "rsq_table_total_rows",
    "rsq_table_destroy", "rsq_query_compile", "rsq_query_execute", "rsq_query_await_kernels", "rsq_query_execute_partial",
    "rsq_query_execute_partial_async", "rsq_ctx_set_stream",
    "rsq_query_finalize", "rsq_query_merge_gathered", "rsq_query_finalize_host", "rsq_query_bind_partial", "rsq_query_partial_layout", "rsq_query_result", "rsq_query_report", "rsq_query_kernel_time_stats", "rsq_query_source", "rsq_query_explain",
    "rsq_query_destroy", "rsq_serialize_expr", "rsq_result_serialize", "rsq_free", "rsq_ref_emission_order", "rsq_ref_emission_order_device",
    "rsq_measure_read_bandwidth",
    "rsq_sql_plan_select", "rsq_sql_plan_desc", "rsq_sql_plan_destroy", "rsq_sql_plan_text", "rsq_sql_compile", "rsq_sql_describe",
    "rsq_db_create", "rsq_db_execute", "rsq_db_message", "rsq_db_adopt_table", "rsq_db_report", "rsq_db_destroy",
    "rsq_multi_create", "rsq_multi_destroy", "rsq_multi_last_error", "rsq_multi_devices", "rsq_multi_ctx", "rsq_multi_merge_name",
    "rsq_multi_shard_rows", "rsq_multi_table_generate", "rsq_multi_table_generate_on_key", "rsq_multi_query_merge_name", "rsq_multi_query_compile", "rsq_multi_query_execute",
    "rsq_multi_query_result", "rsq_multi_query_report", "rsq_multi_query_collective_ms", "rsq_multi_query_destroy",
]

GEN_LINEITEM, GEN_ORDERS, GEN_CUSTOMER, GEN_SYNTHETIC = 0, 1, 2, 3


class Context:
    """one engine context per GPU (device=-1: compile-only, no GPU needed)"""

    @classmethod
    def borrowed(cls, handle, device: int) -> "Context":
        """a context owned by someone else (a MultiContext's shard): close() does not destroy it"""
        self = cls.__new__(cls)
        self._L = lib()
        self._cache = None
        self.h = C.c_void_p(handle)
        self.device = device
        self._borrowed = True
        return self

    def __init__(self, device: int = 0, cache_dir: Optional[str] = None, print_source: bool = False, emission_order: int = 0,
                 compat_flags: int = 0, engine_flags: int = 0, arena_reserve_bytes: int = 0, arena_keep_bytes: int = 0):
        """emission_order: EMIT_REFERENCE (0: rows of an unsorted aggregation in the reference's hash-table order) or EMIT_ANY;
        compat_flags: rsq_compat bits (COMPAT_JIT_INT16_CAST); engine_flags: rsq_engine_flags bits (ENGINE_DRIVER_ALLOC,
        ENGINE_NO_PLAN_MEMO); arena_*: see rsq_config"""
        self._L = lib()
        self._cache = cache_dir.encode() if cache_dir else None
        cfg = rsq_config.make(device, self._cache, print_source, emission_order, compat_flags, engine_flags, arena_reserve_bytes, arena_keep_bytes)
        h = C.c_void_p()
        rc = self._L.rsq_ctx_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise EngineError(rc, self._L.rsq_last_error(None).decode())
        self.h = h
        self.device = device

    def _check(self, rc: int):
        if rc != 0:
            raise EngineError(rc, self._L.rsq_last_error(self.h).decode())

    def memory_stats(self) -> dict:
        """rsq_ctx_memory_stats as a dict: what the context's arenas hold, driver calls, plan-memo entries / hits"""
        m = rsq_memory_stats()
        m.struct_size = C.sizeof(rsq_memory_stats)
        self._check(self._L.rsq_ctx_memory_stats(self.h, C.byref(m)))
        return {k: getattr(m, k) for k, _ in rsq_memory_stats._fields_ if k not in ("struct_size", "reserved0")}

    def set_stream(self, hip_stream: Optional[int]):
        """launch on the caller's HIP stream (an integer handle, 0 = the null stream; e.g.
        torch.cuda.current_stream().cuda_stream), or None to go back to the context's own stream"""
        self._check(self._L.rsq_ctx_set_stream(self.h, hip_stream or None, 0 if hip_stream is None else 1))

    def _adopt(self, child):
        """queries, tables and databases hold device memory of this context: they are closed before it is destroyed"""
        import weakref
        if not hasattr(self, "_children"):
            self._children = weakref.WeakSet()
        self._children.add(child)

    def close(self):
        if getattr(self, "h", None):
            for child in list(getattr(self, "_children", ())):
                try:
                    child.close()
                except Exception:
                    pass
            if not getattr(self, "_borrowed", False):
                self._L.rsq_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- tables ----
    def table(self, t: P.Table) -> "DeviceTable":
        keep: list = []
        d = t.to_c(keep)
        h = C.c_void_p()
        self._check(self._L.rsq_table_create(self.h, C.byref(d), C.byref(h)))
        return DeviceTable(self, h, t.name)

    def table_from_device(self, name: str, n_rows: int, columns) -> "DeviceTable":
        """columns: list of (name, SqlType, device_pointer_or_None); pointers stay owned by the caller"""
        cols = (P.rsq_column * len(columns))()
        for i, (cn, ct, ptr) in enumerate(columns):
            cols[i].name = cn.encode(); cols[i].type = ct.c(); cols[i].data = ptr
        d = P.rsq_table_desc(); d.name = name.encode(); d.n_rows = n_rows; d.n_cols = len(columns); d.cols = cols
        h = C.c_void_p()
        self._check(self._L.rsq_table_create_device(self.h, C.byref(d), C.byref(h)))
        return DeviceTable(self, h, name)

    def load_tbl(self, schema: P.Table, path: str, terminator: str = "|", threads: int = 0) -> "DeviceTable":
        """BULK INSERT of a '.tbl' text file into a table with `schema`'s column names and types"""
        keep: list = []
        bare = P.Table(schema.name, [P.Column(c.name, c.type) for c in schema.columns], 0)
        d = bare.to_c(keep)
        h = C.c_void_p()
        self._check(self._L.rsq_table_load_tbl(self.h, C.byref(d), path.encode(), terminator.encode()[:1], threads, C.byref(h)))
        return DeviceTable(self, h, schema.name)

    def generate(self, kind: int, n_rows: int, sf: float, row0: int = 0, param: int = 0,
                 seed: int = 20240613) -> "DeviceTable":
        h = C.c_void_p()
        self._check(self._L.rsq_table_generate(self.h, kind, row0, n_rows, sf, param, seed, C.byref(h)))
        return DeviceTable(self, h, ["lineitem", "orders", "customer", "t"][kind])

    # ---- queries ----
    def compile(self, plan: P.Plan, tables: Sequence["DeviceTable"]) -> "Query":
        keep: list = []
        d = plan.to_c(keep)
        arr = (C.c_void_p * max(1, len(tables)))(*[t.h for t in tables])
        h = C.c_void_p()
        self._check(self._L.rsq_query_compile(self.h, C.byref(d), arr, len(tables), C.byref(h)))
        return Query(self, h)

    def run(self, plan: P.Plan, tables: Optional[Sequence["DeviceTable"]] = None) -> P.Result:
        """upload plan.tables (unless device tables are given), compile, execute, fetch the result"""
        own = tables is None
        tabs = [self.table(t) for t in plan.tables] if own else list(tables)
        try:
            q = self.compile(plan, tabs)
            try:
                q.execute()
                return q.result()
            finally:
                q.close()
        finally:
            if own:
                for t in tabs:
                    t.close()

    def serialize_expr(self, plan: P.Plan, expr: int, derive: bool, tables: Sequence["DeviceTable"] = ()) -> str:
        keep: list = []
        d = plan.to_c(keep)
        arr = (C.c_void_p * max(1, len(tables)))(*[t.h for t in tables])
        p = self._L.rsq_serialize_expr(self.h, C.byref(d), expr, 1 if derive else 0, arr, len(tables))
        if not p:
            raise EngineError(2, self._L.rsq_last_error(self.h).decode())
        s = C.string_at(p).decode("latin1")
        self._L.rsq_free(p)
        return s

    # ---- SQL text (the reference's parseSql + buildQuery in front of the path) ----
    def sql_describe(self, sql: str, what: int = 1) -> str:
        """what=0: token names, what=1: the parsed statement (see resql_hip.h rsq_sql_describe)"""
        p = self._L.rsq_sql_describe(self.h, sql.encode("latin1"), what)
        if not p:
            raise EngineError(2, self._L.rsq_last_error(self.h).decode())
        s = C.string_at(p).decode("latin1")
        self._L.rsq_free(p)
        return s

    def sql_plan(self, sql: str, tables: Sequence["DeviceTable"], host_tables: Sequence[P.Table] = ()) -> P.Plan:
        """plan of a select statement over the database `tables`, as a P.Plan (scan operators name their table;
        `host_tables` — same order — become plan.tables so that the oracle / the reference harness can run it)"""
        arr = (C.c_void_p * max(1, len(tables)))(*[t.h for t in tables])
        h = C.c_void_p()
        self._check(self._L.rsq_sql_plan_select(self.h, sql.encode("latin1"), arr, len(tables), C.byref(h)))
        try:
            d = self._L.rsq_sql_plan_desc(h).contents
            return P.Plan.from_c(d, [t.name for t in tables], list(host_tables))
        finally:
            self._L.rsq_sql_plan_destroy(h)

    def sql_plan_text(self, sql: str, tables: Sequence["DeviceTable"]) -> str:
        arr = (C.c_void_p * max(1, len(tables)))(*[t.h for t in tables])
        h = C.c_void_p()
        self._check(self._L.rsq_sql_plan_select(self.h, sql.encode("latin1"), arr, len(tables), C.byref(h)))
        try:
            p = self._L.rsq_sql_plan_text(h)
            if not p:
                raise EngineError(2, "rsq_sql_plan_text failed")
            s = C.string_at(p).decode("latin1")
            self._L.rsq_free(p)
            return s
        finally:
            self._L.rsq_sql_plan_destroy(h)

    def sql_compile(self, sql: str, tables: Sequence["DeviceTable"]) -> "Query":
        arr = (C.c_void_p * max(1, len(tables)))(*[t.h for t in tables])
        h = C.c_void_p()
        self._check(self._L.rsq_sql_compile(self.h, sql.encode("latin1"), arr, len(tables), C.byref(h)))
        return Query(self, h)

    def read_bandwidth(self, nbytes: int = 8 << 30, iters: int = 5) -> float:
        v = C.c_double()
        self._check(self._L.rsq_measure_read_bandwidth(self.h, nbytes, iters, C.byref(v)))
        return v.value


class DeviceTable:
    def __init__(self, ctx: Context, h, name: str):
        self.ctx, self.h, self.name = ctx, h, name
        ctx._adopt(self)

    @property
    def n_rows(self) -> int:
        return self.ctx._L.rsq_table_rows(self.h)

    def set_row0(self, row0: int):
        """this table is rows [row0, row0 + n_rows) of a larger one (a shard); before queries are compiled over it"""
        self.ctx._check(self.ctx._L.rsq_table_set_first_row(self.h, row0))

    def append(self, more: "DeviceTable"):
        """the rows of `more` go behind this table's rows (BULK INSERT appends); `more` is consumed"""
        self.ctx._check(self.ctx._L.rsq_table_append(self.h, more.h))
        more.h = None

    def refresh_stats(self):
        """gather the column statistics again (adopted columns whose content changed); compile statements anew afterwards"""
        self.ctx._check(self.ctx._L.rsq_table_refresh_stats(self.h))

    @property
    def total_rows(self) -> int:
        """rows of the whole table this one is a shard of (its own count before unify_shard_stats)"""
        return self.ctx._L.rsq_table_total_rows(self.h)

    def stats_blob(self) -> bytes:
        """this shard's column statistics + row range as a fixed-size blob (the same size on every shard of a schema)"""
        n = self.ctx._L.rsq_table_stats_bytes(self.h)
        buf = C.create_string_buffer(n)
        self.ctx._check(self.ctx._L.rsq_table_stats_export(self.h, buf, n))
        return buf.raw

    def unify_shard_stats(self, blobs: Sequence[bytes]):
        """plan this shard as the whole table: `blobs` = stats_blob() of ALL shards (this one's included).  Before compiling."""
        joined = b"".join(blobs)
        self.ctx._check(self.ctx._L.rsq_table_unify_shard_stats(self.h, joined, len(blobs), len(blobs[0]) if blobs else 0))

    def read_column(self, name: str, dtype, count: Optional[int] = None) -> np.ndarray:
        n = self.n_rows if count is None else count
        out = np.empty(n, dtype=dtype)
        self.ctx._check(self.ctx._L.rsq_table_read_column(self.ctx.h, self.h, name.encode(), out.ctypes.data, out.nbytes))
        return out

    def close(self):
        if self.h:
            self.ctx._L.rsq_table_destroy(self.h)
            self.h = None


class Query:
    def __init__(self, ctx: Context, h):
        self.ctx, self.h = ctx, h
        ctx._adopt(self)

    def execute(self):
        self.ctx._check(self.ctx._L.rsq_query_execute(self.h))

    def await_kernels(self):
        """block until the query runs on its specialised kernels (it may have started on the pre-compiled generic pipeline)"""
        self.ctx._check(self.ctx._L.rsq_query_await_kernels(self.h))

    def execute_partial(self):
        """returns (device_pointer, n_min_words, n_max_words, n_sum_words)"""
        p = C.c_void_p(); a = C.c_int64(); b = C.c_int64(); c = C.c_int64()
        self.ctx._check(self.ctx._L.rsq_query_execute_partial(self.h, C.byref(p), C.byref(a), C.byref(b), C.byref(c)))
        return p.value, a.value, b.value, c.value

    def execute_partial_async(self):
        """enqueue the pipelines on the context's stream and return; finalize() synchronises (see resql_hip.h)"""
        self.ctx._check(self.ctx._L.rsq_query_execute_partial_async(self.h))

    def finalize_host(self, words: np.ndarray):
        """finalise from a partial aggregate table held in host memory (int64 words)"""
        w = np.ascontiguousarray(words, dtype=np.int64)
        self.ctx._check(self.ctx._L.rsq_query_finalize_host(self.h, w.ctypes.data_as(C.POINTER(C.c_int64)), w.size))

    def bind_partial(self, dev_ptr: int, nbytes: int):
        self.ctx._check(self.ctx._L.rsq_query_bind_partial(self.h, dev_ptr, nbytes))

    def merge_gathered(self, gathered_dev_ptr: int, n_ranks: int):
        """reduce the gathered partial tables of n_ranks ranks (device memory, back to back) into the bound partial table"""
        self.ctx._check(self.ctx._L.rsq_query_merge_gathered(self.h, gathered_dev_ptr, n_ranks))

    def partial_layout(self):
        """(n_min_words, n_max_words, n_sum_words) of the partial aggregate table"""
        a = C.c_int64(); b = C.c_int64(); c = C.c_int64()
        self.ctx._check(self.ctx._L.rsq_query_partial_layout(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def finalize(self):
        self.ctx._check(self.ctx._L.rsq_query_finalize(self.h))

    def result(self, text: bool = True) -> P.Result:
        """text=False skips the (pure Python) serialisation — for results with millions of rows"""
        v = P.rsq_result_view()
        self.ctx._check(self.ctx._L.rsq_query_result(self.h, C.byref(v)))
        res = P.Result.from_view(v)
        res.text = res.serialize() if text else None
        return res

    def report(self) -> rsq_report:
        r = rsq_report()
        self.ctx._check(self.ctx._L.rsq_query_report(self.h, C.byref(r)))
        return r

    def kernel_time_stats(self, reset: bool = False):
        """(device ms summed over the executions since the last reset, number of executions)"""
        s, n = C.c_double(0), C.c_uint64(0)
        self.ctx._check(self.ctx._L.rsq_query_kernel_time_stats(self.h, C.byref(s), C.byref(n), 1 if reset else 0))
        return s.value, n.value

    @property
    def source(self) -> str:
        return self.ctx._L.rsq_query_source(self.h).decode()

    @property
    def explain(self) -> str:
        return self.ctx._L.rsq_query_explain(self.h).decode()

    def close(self):
        if self.h:
            self.ctx._L.rsq_query_destroy(self.h)
            self.h = None


class Database:
    """The statement loop of the reference's executeStatement (execute.h:508-545) over the engine — the C ABI's rsq_db_*:
    CREATE TABLE records a schema, BULK INSERT loads a '.tbl' file into device columns, SELECT is parsed, planned, compiled
    and executed.  execute_script splits at ';' like expandExecStatements (execute.h:470-505)."""

    KINDS = {1: "SELECT", 2: "CREATE_TABLE", 3: "BULK_INSERT", 4: "CONTROL"}

    def __init__(self, ctx: Context):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._check(ctx._L.rsq_db_create(ctx.h, C.byref(h)))
        self.h = h
        ctx._adopt(self)

    def add_table(self, t: "DeviceTable"):
        """hand a device table to the database (which owns it from now on)"""
        self.ctx._check(self.ctx._L.rsq_db_adopt_table(self.h, t.h))
        t.h = None

    def execute(self, sql: str):
        """returns a P.Result for a select, None otherwise"""
        kind = C.c_int32(0)
        v = P.rsq_result_view()
        self.ctx._check(self.ctx._L.rsq_db_execute(self.h, sql.encode("latin1"), C.byref(kind), C.byref(v)))
        self.last_kind = self.KINDS.get(kind.value)
        if kind.value != 1:
            return None
        res = P.Result.from_view(v)
        res.text = res.serialize()
        return res

    @property
    def message(self) -> str:
        """what the reference's printQueryResult prints for the last statement in front of the relation"""
        return self.ctx._L.rsq_db_message(self.h).decode("utf8", "replace")

    def report(self) -> rsq_report:
        r = rsq_report()
        self.ctx._check(self.ctx._L.rsq_db_report(self.h, C.byref(r)))
        return r

    def execute_script(self, text: str):
        res = None
        for stmt in text.split(";"):
            if stmt.strip():
                res = self.execute(stmt.strip())
        return res

    def close(self):
        if getattr(self, "h", None):
            self.ctx._L.rsq_db_destroy(self.h)
            self.h = None


class MultiContext:
    """one host process, N GPUs (include/resql_hip.h rsq_multi_*): shard contexts + the RCCL (or peer-copy) group-by merge"""

    def __init__(self, devices: Sequence[int], merge: int = MERGE_AUTO, cache_dir: Optional[str] = None, emission_order: int = 0,
                 compat_flags: int = 0):
        self._L = lib()
        self._devs = (C.c_int32 * len(devices))(*devices)
        self._cache = cache_dir.encode() if cache_dir else None
        self._base = rsq_config.make(0, self._cache, False, emission_order, compat_flags)
        cfg = rsq_multi_config(C.sizeof(rsq_multi_config), len(devices), C.pointer(self._base), self._devs, merge, 0)
        h = C.c_void_p()
        rc = self._L.rsq_multi_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise EngineError(rc, self._L.rsq_multi_last_error(None).decode())
        self.h = h
        self.n = len(devices)
        self.shards = [Context.borrowed(self._L.rsq_multi_ctx(h, i), devices[i]) for i in range(self.n)]

    def _check(self, rc: int):
        if rc != 0:
            raise EngineError(rc, self._L.rsq_multi_last_error(self.h).decode())

    @property
    def merge_name(self) -> str:
        return self._L.rsq_multi_merge_name(self.h).decode()

    def shard_rows(self, n_total: int, shard: int):
        a, b = C.c_int64(), C.c_int64()
        self._L.rsq_multi_shard_rows(n_total, self.n, shard, C.byref(a), C.byref(b))
        return a.value, b.value

    def generate(self, kind: int, n_rows_total: int, sf: float, param: int = 0, seed: int = 20240613) -> List["DeviceTable"]:
        arr = (C.c_void_p * self.n)()
        self._check(self._L.rsq_multi_table_generate(self.h, kind, n_rows_total, sf, param, seed, arr))
        name = ["lineitem", "orders", "customer", "t"][kind]
        return [DeviceTable(self.shards[i], C.c_void_p(arr[i]), name) for i in range(self.n)]

    def generate_on_key(self, kind: int, n_rows_total: int, sf: float, key_column: str, param: int = 0,
                        seed: int = 20240613) -> List["DeviceTable"]:
        """like generate, with every shard boundary moved to the next change of `key_column`"""
        arr = (C.c_void_p * self.n)()
        self._check(self._L.rsq_multi_table_generate_on_key(self.h, kind, n_rows_total, sf, param, seed, key_column.encode(), arr))
        name = ["lineitem", "orders", "customer", "t"][kind]
        return [DeviceTable(self.shards[i], C.c_void_p(arr[i]), name) for i in range(self.n)]

    def compile(self, plan: P.Plan, tables_per_shard: Sequence[Sequence["DeviceTable"]]) -> "MultiQuery":
        keep: list = []
        d = plan.to_c(keep)
        nt = len(tables_per_shard[0])
        flat = [t.h for shard in tables_per_shard for t in shard]
        arr = (C.c_void_p * max(1, len(flat)))(*flat)
        h = C.c_void_p()
        self._check(self._L.rsq_multi_query_compile(self.h, C.byref(d), arr, nt, C.byref(h)))
        return MultiQuery(self, h)

    def _adopt(self, query):
        import weakref
        if not hasattr(self, "_queries"):
            self._queries = weakref.WeakSet()
        self._queries.add(query)

    def close(self):
        """queries first (rsq_multi_query_destroy frees device memory of the shard contexts and reads their tables), then the
        shards' tables, then the handle (which deletes the contexts)"""
        if getattr(self, "h", None):
            for q in list(getattr(self, "_queries", ())):
                q.close()
            for s in self.shards:
                s.close()
            self._L.rsq_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiQuery:
    def __init__(self, m: MultiContext, h):
        self.m, self.h = m, h
        m._adopt(self)

    def execute(self):
        self.m._check(self.m._L.rsq_multi_query_execute(self.h))

    def result(self) -> P.Result:
        v = P.rsq_result_view()
        self.m._check(self.m._L.rsq_multi_query_result(self.h, C.byref(v)))
        res = P.Result.from_view(v)
        res.text = res.serialize()
        return res

    @property
    def merge_name(self) -> str:
        return self.m._L.rsq_multi_query_merge_name(self.h).decode()

    @property
    def collective_ms(self) -> float:
        """device time of the last execution's group-by merge (event pair on the root's stream)"""
        return self.m._L.rsq_multi_query_collective_ms(self.h)

    def report(self):
        """(rsq_report, [kernel ms of every shard])"""
        r = rsq_report()
        k = (C.c_double * self.m.n)()
        self.m._check(self.m._L.rsq_multi_query_report(self.h, C.byref(r), k))
        return r, list(k)

    def close(self):
        if self.h:
            self.m._L.rsq_multi_query_destroy(self.h)
            self.h = None
