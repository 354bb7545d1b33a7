// api.cpp — the extern "C" surface of include/resql_hip.h.  No exception crosses it.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>

#include "engine.h"
#include "sqlfront.h"
#include "hostref.h"

using namespace rsq;

namespace {
thread_local std::string g_createError;

template <typename F>
int guarded(Context* ctx, F&& f) {
    try { f(); return RSQ_OK; }
    catch (const Error& e) { if (ctx) ctx->lastError = e.what(); else g_createError = e.what(); return e.status; }
    catch (const std::bad_alloc&) { if (ctx) ctx->lastError = "out of host memory"; return RSQ_ERR_NOMEM; }
    catch (const std::exception& e) { if (ctx) ctx->lastError = e.what(); else g_createError = e.what(); return RSQ_ERR_INVALID; }
}

Context* C(rsq_ctx* c) { return reinterpret_cast<Context*>(c); }
Table* T(rsq_table* t) { return reinterpret_cast<Table*>(t); }
Query* Q(rsq_query* q) { return reinterpret_cast<Query*>(q); }

struct QueryHandle { Context* ctx; Query* q; };

Table* makeTable(Context& ctx, const rsq_table_desc& d, bool adopt) {
    std::unique_ptr<Table> t(new Table());
    t->ctx = &ctx;
    t->name = std::string(d.name, strnlen(d.name, RSQ_SYMBOL_MAX));
    t->nRows = d.n_rows;
    if (d.n_rows < 0 || d.n_cols < 0) failInvalid("negative table size");
    for (int i = 0; i < d.n_cols; i++) {
        const rsq_column& c = d.cols[i];
        TableColumn tc;
        tc.name = std::string(c.name, strnlen(c.name, RSQ_SYMBOL_MAX));
        tc.type = Type::fromC(c.type);
        size_t bytes = (size_t)d.n_rows * (size_t)columnWidth(tc.type);
        if (c.data) {
            if (adopt) { tc.dptr = const_cast<void*>(c.data); tc.owned = false; }
            else if (ctx.device >= 0) {
                tc.dptr = ctx.allocRaw(bytes); tc.owned = true;
                if (bytes) RSQ_HIP(hipMemcpy(tc.dptr, c.data, bytes, hipMemcpyHostToDevice));
            } else {   // compile-only context: keep a host copy for the statistics
                tc.dptr = malloc(bytes ? bytes : 1); tc.owned = true;
                if (!tc.dptr) throw std::bad_alloc();
                memcpy(tc.dptr, c.data, bytes);
            }
        }
        t->cols.push_back(tc);
    }
    computeColumnStats(ctx, *t);
    return t.release();
}
}  // namespace

// rsq_config as the host's header declared it: struct_size bytes are the host's, everything behind them reads as 0
namespace rsq {
rsq_config readConfig(const rsq_config* cfg) {
    rsq_config c{};
    c.struct_size = (uint32_t)sizeof(rsq_config);
    if (!cfg) return c;
    const uint32_t have = cfg->struct_size;
    if (have < offsetof(rsq_config, kernel_cache_dir) || have > 4096)
        failInvalid("rsq_config.struct_size is " + std::to_string(have) + ": set it to sizeof(rsq_config) (" + std::to_string(sizeof(rsq_config)) + " in this library)");
    memcpy(&c, cfg, std::min<size_t>(have, sizeof c));
    c.struct_size = (uint32_t)sizeof(rsq_config);
    if (c.emission_order != RSQ_EMIT_REFERENCE && c.emission_order != RSQ_EMIT_ANY) failInvalid("rsq_config.emission_order must be RSQ_EMIT_REFERENCE (0) or RSQ_EMIT_ANY (1)");
    if (c.compat_flags & ~(uint32_t)RSQ_COMPAT_JIT_INT16_CAST) failInvalid("rsq_config.compat_flags has bits this library does not know");
    if (c.engine_flags & ~(uint32_t)(RSQ_ENGINE_DRIVER_ALLOC | RSQ_ENGINE_NO_PLAN_MEMO)) failInvalid("rsq_config.engine_flags has bits this library does not know");
    return c;
}
}  // namespace rsq

// ---- shard statistics -------------------------------------------------------------------------------------------------------
// One blob per table: [magic | columns | row0 | rows] then per column [valid | ascending | min | max | number of byte values | 256 byte
// values].  Fixed size for a schema, plain little-endian words: what one all-gather between the rank processes moves.
namespace rsq {
namespace {
const uint64_t STATS_MAGIC = 0x3174617473717372ull;      // "rsqstat1"
struct StatsHead { uint64_t magic; int64_t nCols, row0, nRows; };
struct StatsCol { int32_t valid, ascending; int64_t min, max; int32_t nBytes, strict; uint8_t bytes[256]; };
}  // namespace
size_t tableStatsBytes(const Table& t) { return sizeof(StatsHead) + t.cols.size() * sizeof(StatsCol); }
void exportTableStats(const Table& t, void* buf, size_t bytes) {
    if (bytes < tableStatsBytes(t)) failInvalid("statistics buffer too small");
    StatsHead h{STATS_MAGIC, (int64_t)t.cols.size(), t.row0, t.nRows};
    memcpy(buf, &h, sizeof h);
    for (size_t i = 0; i < t.cols.size(); i++) {
        const ColumnStats& st = t.shardStats(i);
        StatsCol c{};
        c.valid = st.valid; c.ascending = st.ascending; c.strict = st.strictlyAscending; c.min = st.min; c.max = st.max; c.nBytes = (int32_t)st.distinctBytes.size();
        for (size_t k = 0; k < st.distinctBytes.size() && k < 256; k++) c.bytes[k] = st.distinctBytes[k];
        memcpy((char*)buf + sizeof h + i * sizeof c, &c, sizeof c);
    }
}
void unifyShardStats(Table& t, const void* blobs, int nShards, size_t blobBytes) {
    if (!blobs || nShards < 1 || blobBytes < tableStatsBytes(t)) failInvalid("shard statistics: " + std::to_string(nShards) + " blob(s) of " + std::to_string(blobBytes) + " bytes do not describe table " + t.name);
    if (t.ownStats.empty()) for (auto& c : t.cols) t.ownStats.push_back(c.stats);
    int64_t total = 0;
    bool mine = false;
    std::vector<ColumnStats> u(t.cols.size());
    std::vector<bool> unknown(t.cols.size(), false), any(t.cols.size(), false);
    for (int s = 0; s < nShards; s++) {
        const char* b = (const char*)blobs + (size_t)s * blobBytes;
        StatsHead h; memcpy(&h, b, sizeof h);
        if (h.magic != STATS_MAGIC || h.nCols != (int64_t)t.cols.size() || h.nRows < 0) failInvalid("shard statistics: blob " + std::to_string(s) + " is not a statistics blob of table " + t.name);
        if (h.row0 == t.row0 && h.nRows == t.nRows) mine = true;
        total += h.nRows;
        if (h.nRows == 0) continue;                      // an empty shard says nothing about the values
        for (size_t i = 0; i < t.cols.size(); i++) {
            StatsCol c; memcpy(&c, b + sizeof h + i * sizeof c, sizeof c);
            if (!c.valid) { unknown[i] = true; continue; }
            if (c.nBytes < 0 || c.nBytes > 256) failInvalid("shard statistics: malformed byte-value set");
            if (c.min > c.max) failInvalid("shard statistics: blob " + std::to_string(s) + " has min > max in column " + t.cols[i].name);
            ColumnStats& o = u[i];
            if (!any[i]) { o.min = c.min; o.max = c.max; any[i] = true; }
            else { o.min = std::min(o.min, c.min); o.max = std::max(o.max, c.max); }
            for (int k = 0; k < c.nBytes; k++) o.distinctBytes.push_back(c.bytes[k]);
        }
    }
    if (!mine) failInvalid("shard statistics: none of the blobs is this shard's own (rows " + std::to_string(t.row0) + " + " + std::to_string(t.nRows) + " of " + t.name + ")");
    for (size_t i = 0; i < t.cols.size(); i++) {
        ColumnStats& o = u[i];
        // this shard's OWN statistics are folded in whatever the blobs said about it: device code trusts the union without a range check
        // for engine-owned columns (codegen_join.cpp checkKey), so a stale or foreign blob must never narrow it below what the rows here hold
        const ColumnStats& own = t.ownStats[i];
        if (own.valid && t.nRows > 0) {
            if (!any[i]) { o.min = own.min; o.max = own.max; any[i] = true; }
            else { o.min = std::min(o.min, own.min); o.max = std::max(o.max, own.max); }
            o.distinctBytes.insert(o.distinctBytes.end(), own.distinctBytes.begin(), own.distinctBytes.end());
        }
        std::sort(o.distinctBytes.begin(), o.distinctBytes.end());
        o.distinctBytes.erase(std::unique(o.distinctBytes.begin(), o.distinctBytes.end()), o.distinctBytes.end());
        o.valid = any[i] && !unknown[i];
        o.ascending = t.ownStats[i].ascending;           // (properties of this shard's own rows: they shape kernels, never a layout)
        o.strictlyAscending = t.ownStats[i].strictlyAscending;
        t.cols[i].stats = o;
    }
    t.nRowsTotal = total;
    t.bumpVersion();
}
}  // namespace rsq

extern "C" {

int64_t rsq_table_stats_bytes(const rsq_table* t) { return t ? (int64_t)tableStatsBytes(*reinterpret_cast<const Table*>(t)) : -1; }
int rsq_table_stats_export(const rsq_table* t, void* buf, int64_t bytes) {
    if (!t || !buf || bytes < 0) return RSQ_ERR_INVALID;
    const Table* tb = reinterpret_cast<const Table*>(t);
    return guarded(tb->ctx, [&] { exportTableStats(*tb, buf, (size_t)bytes); });
}
int rsq_table_unify_shard_stats(rsq_table* t, const void* blobs, int32_t n_shards, int64_t blob_bytes) {
    if (!t || !blobs || blob_bytes < 0) return RSQ_ERR_INVALID;
    return guarded(T(t)->ctx, [&] { unifyShardStats(*T(t), blobs, n_shards, (size_t)blob_bytes); });
}
int64_t rsq_table_total_rows(const rsq_table* t) { return t ? reinterpret_cast<const Table*>(t)->totalRows() : -1; }

int rsq_ctx_create(const rsq_config* cfg, rsq_ctx** out) {
    if (!out) return RSQ_ERR_INVALID;
    *out = nullptr;
    return guarded(nullptr, [&] { *out = reinterpret_cast<rsq_ctx*>(new Context(readConfig(cfg))); });
}

void rsq_ctx_destroy(rsq_ctx* ctx) { delete C(ctx); }

const char* rsq_last_error(const rsq_ctx* ctx) {
    if (!ctx) return g_createError.c_str();
    return reinterpret_cast<const Context*>(ctx)->lastError.c_str();
}

int rsq_ctx_memory_stats(const rsq_ctx* ctx, rsq_memory_stats* out) {
    if (!ctx || !out) return RSQ_ERR_INVALID;
    const Context& c = *reinterpret_cast<const Context*>(ctx);
    const uint32_t have = out->struct_size;
    if (have < offsetof(rsq_memory_stats, device_used_bytes) || have > 4096) return RSQ_ERR_INVALID;
    rsq_memory_stats m{};
    m.struct_size = (uint32_t)sizeof m;
    if (c.devArena) { m.device_slab_bytes = c.devArena->slabBytes(); m.device_used_bytes = c.devArena->usedBytes(); m.device_slab_allocs = c.devArena->nSlabAllocs; m.arena_requests += c.devArena->nAllocs; m.driver_ms += c.devArena->slabAllocMs; }
    for (const Arena* a : {c.pinArena.get(), c.pinNcArena.get()})
        if (a) { m.pinned_slab_bytes += a->slabBytes(); m.pinned_used_bytes += a->usedBytes(); m.pinned_slab_allocs += a->nSlabAllocs; m.arena_requests += a->nAllocs; m.driver_ms += a->slabAllocMs; }
    m.raw_driver_calls = c.allocStats.rawCalls;
    m.driver_ms += c.allocStats.rawMs;
    m.plan_memo_entries = c.planMemo.size();
    m.plan_memo_hits = c.planMemoHits;
    m.key_index_entries = c.keyIndexes.size();
    for (auto& kv : c.keyIndexes) m.key_index_bytes += (uint64_t)kv.second.bmBlocks * 32;
    memcpy(out, &m, std::min<size_t>(have, sizeof m));
    out->struct_size = have;
    return RSQ_OK;
}

int rsq_table_create(rsq_ctx* ctx, const rsq_table_desc* desc, rsq_table** out) {
    if (!ctx || !desc || !out) return RSQ_ERR_INVALID;
    return guarded(C(ctx), [&] { *out = reinterpret_cast<rsq_table*>(makeTable(*C(ctx), *desc, false)); });
}

int rsq_table_create_device(rsq_ctx* ctx, const rsq_table_desc* desc, rsq_table** out) {
    if (!ctx || !desc || !out) return RSQ_ERR_INVALID;
    return guarded(C(ctx), [&] {
        if (C(ctx)->device < 0) throw Error(RSQ_ERR_DEVICE, "rsq_table_create_device needs a device context");
        *out = reinterpret_cast<rsq_table*>(makeTable(*C(ctx), *desc, true));
    });
}

int rsq_table_from_rowstore(rsq_ctx* ctx, const rsq_table_desc* schema, const uint8_t* const* blocks,
                            const size_t* content_size, int32_t n_blocks, rsq_table** out) {
    if (!ctx || !schema || !out || n_blocks < 0) return RSQ_ERR_INVALID;
    if (schema->n_cols <= 0 || (n_blocks > 0 && (!blocks || !content_size))) return RSQ_ERR_INVALID;
    return guarded(C(ctx), [&] {
        // transpose packed ReSQL tuples (strings by value, schema.h:76-106 offsets) into columns on the host,
        // then upload: the bridge is ingest, not the hot path
        std::vector<Type> types; std::vector<int> offs; int ts = 0;
        for (int i = 0; i < schema->n_cols; i++) {
            Type t = Type::fromC(schema->cols[i].type);
            types.push_back(t); offs.push_back(ts); ts += sizeInTuple(t, true);
        }
        int64_t n = 0;
        for (int b = 0; b < n_blocks; b++) n += (int64_t)(content_size[b] / (size_t)ts);
        std::vector<std::vector<uint8_t>> cols((size_t)schema->n_cols);
        for (int i = 0; i < schema->n_cols; i++) cols[(size_t)i].assign((size_t)n * (size_t)columnWidth(types[(size_t)i]), 0);
        int64_t r = 0;
        for (int b = 0; b < n_blocks; b++) {
            size_t cnt = content_size[b] / (size_t)ts;
            for (size_t k = 0; k < cnt; k++, r++) {
                const uint8_t* tup = blocks[b] + k * (size_t)ts;
                for (int i = 0; i < schema->n_cols; i++) {
                    int w = columnWidth(types[(size_t)i]);
                    uint8_t* dst = &cols[(size_t)i][(size_t)r * (size_t)w];
                    if (types[(size_t)i].tag == RSQ_CHAR || types[(size_t)i].tag == RSQ_VARCHAR) {
                        const uint8_t* s = tup + offs[(size_t)i];
                        for (int c = 0; c < w && s[c]; c++) dst[c] = s[c];
                    } else memcpy(dst, tup + offs[(size_t)i], (size_t)w);
                }
            }
        }
        std::vector<rsq_column> cd((size_t)schema->n_cols);
        for (int i = 0; i < schema->n_cols; i++) { cd[(size_t)i] = schema->cols[i]; cd[(size_t)i].data = cols[(size_t)i].data(); }
        rsq_table_desc d = *schema; d.n_rows = n; d.cols = cd.data();
        *out = reinterpret_cast<rsq_table*>(makeTable(*C(ctx), d, false));
    });
}

int rsq_table_load_tbl(rsq_ctx* ctx, const rsq_table_desc* schema, const char* path, char field_terminator,
                       int32_t n_threads, rsq_table** out) {
    if (!ctx || !schema || !path || !out || schema->n_cols < 0) return RSQ_ERR_INVALID;
    return guarded(C(ctx), [&] {
        std::vector<Type> types;
        for (int i = 0; i < schema->n_cols; i++) types.push_back(Type::fromC(schema->cols[i].type));
        std::vector<std::vector<uint8_t>> cols;
        int64_t n = 0;
        int threads = n_threads > 0 ? n_threads : (int)std::max(1u, std::thread::hardware_concurrency());
        parseTblFile(path, types, field_terminator, threads, cols, n);
        static const uint8_t kEmpty[16] = {0};       // an empty file still gives columns WITH data (of zero rows)
        std::vector<rsq_column> cd((size_t)schema->n_cols);
        for (int i = 0; i < schema->n_cols; i++) {
            cd[(size_t)i] = schema->cols[i];
            cd[(size_t)i].data = cols[(size_t)i].empty() ? (const void*)kEmpty : (const void*)cols[(size_t)i].data();
        }
        rsq_table_desc d = *schema; d.n_rows = n; d.cols = cd.data();
        *out = reinterpret_cast<rsq_table*>(makeTable(*C(ctx), d, false));
    });
}

int rsq_table_generate(rsq_ctx* ctx, int32_t kind, int64_t row0, int64_t n_rows, double scale_factor,
                       int64_t param, uint64_t seed, rsq_table** out) {
    if (!ctx || !out || n_rows < 0) return RSQ_ERR_INVALID;
    return guarded(C(ctx), [&] {
        std::unique_ptr<Table> t(new Table());
        generateTable(*C(ctx), *t, kind, row0, n_rows, scale_factor, param, seed);
        *out = reinterpret_cast<rsq_table*>(t.release());
    });
}

int64_t rsq_table_rows(const rsq_table* t) { return t ? reinterpret_cast<const Table*>(t)->nRows : -1; }
int rsq_table_set_first_row(rsq_table* t, int64_t row0) {
    if (!t || row0 < 0) return RSQ_ERR_INVALID;
    reinterpret_cast<Table*>(t)->row0 = row0;
    reinterpret_cast<Table*>(t)->bumpVersion();
    return RSQ_OK;
}

int rsq_table_refresh_stats(rsq_table* t) {
    if (!t) return RSQ_ERR_INVALID;
    Table& tab = *reinterpret_cast<Table*>(t);
    if (!tab.ctx) return RSQ_ERR_INVALID;
    return guarded(tab.ctx, [&] {
        if (!tab.ownStats.empty()) failInvalid("rsq_table_refresh_stats: the table plans with unified shard statistics; refresh the shards and unify again");
        computeColumnStats(*tab.ctx, tab);
        tab.bumpVersion();          // (the context's plan memo: what queries learnt over the old content does not describe the new)
    });
}

int rsq_table_read_column(rsq_ctx* ctx, const rsq_table* t, const char* name, void* host_dst, size_t bytes) {
    if (!ctx || !t || !name || !host_dst) return RSQ_ERR_INVALID;
    return guarded(C(ctx), [&] {
        const Table* tb = reinterpret_cast<const Table*>(t);
        int ci = tb->findCol(name);
        if (ci < 0 || !tb->cols[(size_t)ci].dptr) failInvalid(std::string("no such column: ") + name);
        size_t have = (size_t)tb->nRows * (size_t)columnWidth(tb->cols[(size_t)ci].type);
        if (bytes > have) failInvalid("read beyond the column");
        if (C(ctx)->device >= 0) RSQ_HIP(hipMemcpy(host_dst, tb->cols[(size_t)ci].dptr, bytes, hipMemcpyDeviceToHost));
        else memcpy(host_dst, tb->cols[(size_t)ci].dptr, bytes);
    });
}

void rsq_table_destroy(rsq_table* t) { delete T(t); }

int rsq_query_compile(rsq_ctx* ctx, const rsq_plan_desc* plan, rsq_table* const* tables, int32_t n_tables, rsq_query** out) {
    if (!ctx || !plan || !out || n_tables < 0) return RSQ_ERR_INVALID;
    *out = nullptr;
    return guarded(C(ctx), [&] {
        std::unique_ptr<QueryHandle> h(new QueryHandle{C(ctx), nullptr});
        h->q = compileQuery(*C(ctx), *plan, tables, n_tables);
        *out = reinterpret_cast<rsq_query*>(h.release());
    });
}

#define QH(q) reinterpret_cast<QueryHandle*>(q)

int rsq_query_execute(rsq_query* q) {
    if (!q) return RSQ_ERR_INVALID;
    return guarded(QH(q)->ctx, [&] { executeQuery(*QH(q)->q, false); });
}

int rsq_query_await_kernels(rsq_query* q) {
    if (!q) return RSQ_ERR_INVALID;
    return guarded(QH(q)->ctx, [&] { awaitKernels(*QH(q)->q); });
}

int rsq_query_execute_partial(rsq_query* q, void** dev_ptr, int64_t* n_min_words, int64_t* n_max_words, int64_t* n_sum_words) {
    if (!q || !dev_ptr || !n_min_words || !n_max_words || !n_sum_words) return RSQ_ERR_INVALID;
    return guarded(QH(q)->ctx, [&] {
        executeQuery(*QH(q)->q, true);
        partialBuffer(*QH(q)->q, dev_ptr, n_min_words, n_max_words, n_sum_words);
    });
}

int rsq_query_execute_partial_async(rsq_query* q) {
    if (!q) return RSQ_ERR_INVALID;
    return guarded(QH(q)->ctx, [&] { executeQuery(*QH(q)->q, true, true); });
}

int rsq_ctx_set_stream(rsq_ctx* ctx, void* hip_stream, int32_t use_callers_stream) {
    if (!ctx) return RSQ_ERR_INVALID;
    return guarded(C(ctx), [&] { C(ctx)->setStream((hipStream_t)hip_stream, use_callers_stream != 0); });
}

int rsq_query_partial_layout(const rsq_query* q, int64_t* n_min_words, int64_t* n_max_words, int64_t* n_sum_words) {
    if (!q || !n_min_words || !n_max_words || !n_sum_words) return RSQ_ERR_INVALID;
    const QueryHandle* h = reinterpret_cast<const QueryHandle*>(q);
    return guarded(h->ctx, [&] { void* p = nullptr; partialBuffer(*h->q, &p, n_min_words, n_max_words, n_sum_words); });
}

int rsq_query_bind_partial(rsq_query* q, void* dev_ptr, size_t bytes) {
    if (!q) return RSQ_ERR_INVALID;
    return guarded(QH(q)->ctx, [&] { bindPartial(*QH(q)->q, dev_ptr, bytes); });
}

int rsq_query_merge_gathered(rsq_query* q, const void* gathered_dev, int32_t n_ranks) {
    if (!q) return RSQ_ERR_INVALID;
    return guarded(QH(q)->ctx, [&] { mergeGathered(*QH(q)->q, gathered_dev, n_ranks); });
}

int rsq_query_finalize_host(rsq_query* q, const int64_t* words, int64_t n_words) {
    if (!q || !words || n_words < 0) return RSQ_ERR_INVALID;
    return guarded(QH(q)->ctx, [&] { finalizeQueryHost(*QH(q)->q, words, (size_t)n_words); });
}

int rsq_query_finalize(rsq_query* q) {
    if (!q) return RSQ_ERR_INVALID;
    return guarded(QH(q)->ctx, [&] { finalizeQuery(*QH(q)->q); });
}

int rsq_query_result(rsq_query* q, rsq_result_view* out) {
    if (!q || !out) return RSQ_ERR_INVALID;
    return guarded(QH(q)->ctx, [&] { queryResult(*QH(q)->q, out); });
}

int rsq_query_report(const rsq_query* q, rsq_report* out) {
    if (!q || !out) return RSQ_ERR_INVALID;
    const QueryHandle* h = reinterpret_cast<const QueryHandle*>(q);
    return guarded(h->ctx, [&] { queryReport(*h->q, out); });
}

int rsq_query_kernel_time_stats(rsq_query* q, double* sum_ms, uint64_t* executions, int32_t reset) {
    if (!q) return RSQ_ERR_INVALID;
    return guarded(QH(q)->ctx, [&] { queryKernelTimeStats(*QH(q)->q, sum_ms, executions, reset != 0); });
}

const char* rsq_query_source(const rsq_query* q) { return q ? querySource(*reinterpret_cast<const QueryHandle*>(q)->q) : ""; }
const char* rsq_query_explain(const rsq_query* q) { return q ? queryExplain(*reinterpret_cast<const QueryHandle*>(q)->q) : ""; }

void rsq_query_destroy(rsq_query* q) {
    if (!q) return;
    destroyQuery(QH(q)->q);
    delete QH(q);
}

char* rsq_serialize_expr(rsq_ctx* ctx, const rsq_plan_desc* plan, int32_t expr, int32_t derive,
                         rsq_table* const* tables, int32_t n_tables) {
    if (!ctx || !plan) return nullptr;
    char* res = nullptr;
    guarded(C(ctx), [&] {
        ExprPool pool;
        for (int i = 0; i < n_tables; i++)
            for (auto& c : T(tables[i])->cols) pool.identTypes[c.name] = c.type;
        std::vector<Expr*> v = pool.build(*plan);
        if (expr < 0 || expr >= (int)v.size()) failInvalid("bad expression index");
        if (derive) pool.derive(v[(size_t)expr]);
        res = strdup(serializeExpr(v[(size_t)expr]).c_str());
    });
    return res;
}

char* rsq_result_serialize(const rsq_result_view* view) {
    if (!view) return nullptr;
    try { return strdup(serializeResultView(*view).c_str()); } catch (...) { return nullptr; }
}

void rsq_free(void* p) { free(p); }

int rsq_ref_emission_order_device(rsq_ctx* ctx, const uint64_t* hashes, int64_t n, uint64_t min_size, uint32_t* out) {
    if (!ctx || n < 0 || (n > 0 && (!hashes || !out))) return RSQ_ERR_INVALID;
    return guarded(C(ctx), [&] {
        Context& c = *C(ctx);
        if (c.device < 0) throw Error(RSQ_ERR_DEVICE, "this context has no device (compile-only)");
        if (n == 0) return;
        std::vector<std::pair<uint64_t, uint64_t>> levels;
        if (!replayLevels((uint64_t)n, min_size, levels)) failUnsupported("the device replay does not take tables of this size");
        RSQ_HIP(hipSetDevice(c.device));
        uint64_t* dH = (uint64_t*)c.alloc((size_t)n * 8);
        uint32_t* dO = (uint32_t*)c.alloc((size_t)n * 4);
        void* work = c.alloc(replayDeviceBytes((uint64_t)n, levels.back().first));
        RSQ_HIP(hipMemcpy(dH, hashes, (size_t)n * 8, hipMemcpyHostToDevice));
        replayEmissionOrderDevice(c, dH, (uint64_t)n, levels, work, dO);
        RSQ_HIP(hipStreamSynchronize(c.stream));
        RSQ_HIP(hipMemcpy(out, dO, (size_t)n * 4, hipMemcpyDeviceToHost));
        c.free(dH); c.free(dO); c.free(work);
    });
}

int rsq_ref_emission_order(const uint64_t* hashes, int64_t n, uint64_t min_size, int32_t parallel, uint32_t* out) {
    if (n < 0 || (n > 0 && (!hashes || !out))) return RSQ_ERR_INVALID;
    try {
        if (parallel) {
            std::vector<uint32_t> order; ReplayScratch scratch;
            refEmissionOrderParallel(hashes, (size_t)n, min_size, order, scratch);
            for (int64_t i = 0; i < n; i++) out[i] = order[(size_t)i];
        } else {
            std::vector<size_t> order = refEmissionOrder(std::vector<uint64_t>(hashes, hashes + n), min_size);
            for (int64_t i = 0; i < n; i++) out[i] = (uint32_t)order[(size_t)i];
        }
        return RSQ_OK;
    } catch (const Error& e) { return e.status; }
    catch (const std::exception&) { return RSQ_ERR_RUNTIME; }
}

// ---- SQL front end (sqlfront.cpp) ----
struct rsq_sql_plan { rsq::ExprPool pool; rsq::sql::Statement st; rsq::sql::PlanDesc plan; std::vector<Table*> db; };

int rsq_sql_plan_select(rsq_ctx* ctx, const char* sqlText, rsq_table* const* tables, int32_t n_tables, rsq_sql_plan** out) {
    if (!ctx || !sqlText || !out || n_tables < 0) return RSQ_ERR_INVALID;
    *out = nullptr;
    return guarded(C(ctx), [&] {
        std::unique_ptr<rsq_sql_plan> p(new rsq_sql_plan());
        std::vector<Table*> db;
        for (int i = 0; i < n_tables; i++) { if (!tables[i]) failInvalid("null table"); db.push_back(T(tables[i])); }
        rsq::sql::parse(sqlText, p->pool, p->st);
        if (p->st.kind != rsq::sql::Statement::SELECT) failInvalid("not a select statement");
        rsq::sql::planSelect(p->st, p->pool, db, p->plan);
        p->db = db;
        *out = p.release();
    });
}
const rsq_plan_desc* rsq_sql_plan_desc(const rsq_sql_plan* plan) { return plan ? &plan->plan.desc : nullptr; }
void rsq_sql_plan_destroy(rsq_sql_plan* plan) { delete plan; }
char* rsq_sql_plan_text(const rsq_sql_plan* plan) {
    if (!plan) return nullptr;
    try { return strdup(rsq::sql::dumpPlan(plan->plan.desc, plan->db).c_str()); }
    catch (const std::exception& e) { return strdup((std::string("error: ") + e.what()).c_str()); }
    catch (...) { return nullptr; }
}

int rsq_sql_compile(rsq_ctx* ctx, const char* sqlText, rsq_table* const* tables, int32_t n_tables, rsq_query** out) {
    if (!out) return RSQ_ERR_INVALID;
    *out = nullptr;
    rsq_sql_plan* p = nullptr;
    int st = rsq_sql_plan_select(ctx, sqlText, tables, n_tables, &p);
    if (st != RSQ_OK) return st;
    st = rsq_query_compile(ctx, &p->plan.desc, tables, n_tables, out);
    delete p;
    return st;
}

// ---- statement loop (executeStatement, execute.h:508-545) ----
struct rsq_db {
    Context* ctx = nullptr;
    std::map<std::string, std::vector<std::pair<std::string, Type>>> schemas;      // CREATE TABLE
    std::map<std::string, Table*> tables;                                          // name order = the planner's table order
    rsq_query* last = nullptr;
    rsq_report lastReport{};
    // DBConfig / JitConfig as the control variables set them (execute.h:23-27, JitContextFlounder.h:86-109)
    bool showPlan = false, writeResultsToFile = false;
    uint16_t numThreads = 1;
    bool printPerformance = false, printAssembly = false, printFlounder = false, optimizeFlounder = false, emitMachineCode = false;
    std::string message;          // what the reference's printQueryResult would print in front of the relation (execute.h:173-200)
};

}  // extern "C"

namespace {

// setBoolVar / setIntVar (execute.h:407-451): spaces are removed from the line, the variable name may stand anywhere in it
// (rfind), the line is either the bare name (prints the value) or name=value
bool controlVar(std::string cmd, const std::string& name, std::string& value, bool& bare) {
    cmd.erase(std::remove(cmd.begin(), cmd.end(), ' '), cmd.end());
    if (cmd.rfind(name) == std::string::npos) return false;
    bare = cmd.length() == name.length();
    if (bare) return true;
    if (cmd.at(name.length()) != '=') failInvalid("Expected varname=value");
    value = cmd.substr(name.length() + 1);
    return true;
}
void boolVar(const std::string& line, const char* name, bool& var, bool& done, std::string& out) {
    std::string v; bool bare = false;
    if (!controlVar(line, name, v, bare)) return;
    done = true;
    if (bare) { out += var ? "true\n" : "false\n"; return; }
    if (v == "true") var = true;
    else if (v == "false") var = false;
    else failInvalid("Expected true or false");
}
void intVar(const std::string& line, const char* name, uint16_t& var, bool& done, std::string& out) {
    std::string v; bool bare = false;
    if (!controlVar(line, name, v, bare)) return;
    done = true;
    if (bare) { out += std::to_string(var) + "\n"; return; }
    // std::stoi semantics (leading blanks, sign, digits; trailing characters ignored) without its exceptions: the reference lets
    // std::invalid_argument escape executeStatement (it catches runtime_error and ResqlError only) and dies
    size_t i = 0;
    bool neg = false;
    if (i < v.size() && (v[i] == '+' || v[i] == '-')) neg = v[i++] == '-';
    if (i >= v.size() || !isdigit((unsigned char)v[i])) failInvalid("Expected an integer value");
    long long x = 0;
    for (; i < v.size() && isdigit((unsigned char)v[i]); i++) { x = x * 10 + (v[i] - '0'); if (x > 0x7fffffffLL) failInvalid("Integer value out of range"); }
    var = (uint16_t)(int)(neg ? -x : x);
}

// printStringTable (dbdata.h:578-626): cells are right-aligned in columns two wider than their longest string
std::string stringTable(const std::vector<std::string>& cells, int nCols, int headerRows, const std::string& subTitle) {
    std::vector<size_t> w((size_t)nCols, 0);
    for (size_t i = 0; i < cells.size(); i++) w[i % (size_t)nCols] = std::max(w[i % (size_t)nCols], cells[i].size() + 2);
    auto rule = [&](const char* l, const char* m, const char* x, const char* r) {
        std::string o = l;
        for (int c = 0; c < nCols; c++) { if (c) o += x; for (size_t k = 0; k < w[(size_t)c]; k++) o += m; }
        return o + r + "\n";
    };
    auto row = [&](size_t at) {
        std::string o;
        for (int c = 0; c < nCols; c++) {
            std::string cell = " " + cells[at + (size_t)c] + " ";
            o += "\xe2\x94\x82" + std::string(w[(size_t)c] > cell.size() ? w[(size_t)c] - cell.size() : 0, ' ') + cell;
        }
        return o + "\xe2\x94\x82\n";
    };
    std::string out = rule("\xe2\x94\x8c", "\xe2\x94\x80", "\xe2\x94\xac", "\xe2\x94\x90");
    size_t at = 0;
    for (int h = 0; h < headerRows; h++, at += (size_t)nCols) out += row(at);
    out += rule("\xe2\x94\x9c", "\xe2\x94\x80", "\xe2\x94\xbc", "\xe2\x94\xa4");
    for (; at < cells.size(); at += (size_t)nCols) out += row(at);
    out += rule("\xe2\x94\x94", "\xe2\x94\x80", "\xe2\x94\xb4", "\xe2\x94\x98");
    if (!subTitle.empty()) {
        size_t width = (size_t)nCols;
        for (size_t x : w) width += x;
        out += std::string(width > subTitle.size() ? width - subTitle.size() : 0, ' ') + subTitle + "\n";
    }
    return out;
}

// processControl (execute.h:454-474): every variable is tried in the reference's order; "tables" must be the whole line
bool processControl(rsq_db& db, const std::string& line, std::string& out) {
    bool done = false;
    boolVar(line, "showplan", db.showPlan, done, out);
    boolVar(line, "tofile", db.writeResultsToFile, done, out);
    intVar(line, "threads", db.numThreads, done, out);
    boolVar(line, "showperf", db.printPerformance, done, out);
    boolVar(line, "showasm", db.printAssembly, done, out);
    boolVar(line, "showfln", db.printFlounder, done, out);
    boolVar(line, "optimize", db.optimizeFlounder, done, out);
    boolVar(line, "emitmc", db.emitMachineCode, done, out);
    if (line == "tables") {          // showTables (execute.h:390-404); std::map order = by name
        std::vector<std::string> cells = {"Table name", "Number of attributes", "Number of tuples"};
        for (auto& t : db.tables) { cells.push_back(t.first); cells.push_back(std::to_string(t.second->cols.size())); cells.push_back(std::to_string(t.second->nRows)); }
        out += stringTable(cells, 3, 1, std::to_string(db.tables.size()) + " tables");
        done = true;
    }
    return done;
}

// showReport (JitContextFlounder.h:132-150) for this engine: the code (HIP source for showasm, the pipeline description for
// showfln), then the performance lines in the reference's format plus the device-side figures
std::string reportText(const rsq_db& db, rsq_query* q, const rsq_report& r) {
    std::string o;
    if (db.printFlounder) { o += rsq_query_explain(q); }
    if (db.printAssembly) { o += rsq_query_source(q); }
    if (db.printPerformance) {
        char b[256];
        snprintf(b, sizeof b, "Launched %llu kernels (%d compiled by hiprtc, %d from the code-object cache). \n", (unsigned long long)r.num_kernels, r.jit_compiles, r.jit_cache_hits);
        o += b;
        snprintf(b, sizeof b, "compile: %.3f ms\n", r.compilation_time_ms); o += b;
        snprintf(b, sizeof b, "execute: %.3f ms\n", r.execution_time_ms); o += b;
        snprintf(b, sizeof b, "device:  %.3f ms, %llu bytes scanned, %.1f GB/s\n", r.kernel_time_ms, (unsigned long long)r.bytes_read, r.hbm_gbps); o += b;
    }
    return o;
}

// concatenate `more` behind `t` (BULK INSERT appends, execute.h:332-388): new device columns, statistics recomputed
void appendTable(Context& ctx, Table& t, Table& more) {
    if (t.cols.size() != more.cols.size()) failInvalid("append: different schemas");
    for (size_t c = 0; c < t.cols.size(); c++)
        if (t.cols[c].type.tag != more.cols[c].type.tag || columnWidth(t.cols[c].type) != columnWidth(more.cols[c].type)) failInvalid("append: column " + t.cols[c].name + " has another type");
    const int64_t n0 = t.nRows, n1 = more.nRows;
    for (size_t c = 0; c < t.cols.size(); c++) {
        TableColumn& a = t.cols[c]; TableColumn& b = more.cols[c];
        const size_t w = (size_t)columnWidth(a.type);
        if (ctx.device >= 0) {
            char* nu = (char*)ctx.allocRaw((size_t)(n0 + n1) * w);
            if (n0 && a.dptr) RSQ_HIP(hipMemcpy(nu, a.dptr, (size_t)n0 * w, hipMemcpyDeviceToDevice));
            if (n1 && b.dptr) RSQ_HIP(hipMemcpy(nu + (size_t)n0 * w, b.dptr, (size_t)n1 * w, hipMemcpyDeviceToDevice));
            if (a.owned && a.dptr) ctx.freeRaw(a.dptr);
            a.dptr = nu; a.owned = true;
        } else {
            char* nu = (char*)malloc(std::max<size_t>(1, (size_t)(n0 + n1) * w));
            if (!nu) throw std::bad_alloc();
            if (n0 && a.dptr) memcpy(nu, a.dptr, (size_t)n0 * w);
            if (n1 && b.dptr) memcpy(nu + (size_t)n0 * w, b.dptr, (size_t)n1 * w);
            if (a.owned && a.dptr) ::free(a.dptr);
            a.dptr = nu; a.owned = true;
        }
    }
    t.nRows = n0 + n1;
    t.layoutVersion++;
    t.bumpVersion();
    computeColumnStats(ctx, t);
}

}  // namespace

extern "C" {

int rsq_table_append(rsq_table* t, rsq_table* more) {
    if (!t || !more || t == more) return RSQ_ERR_INVALID;
    Table& a = *T(t); Table& b = *T(more);
    if (!a.ctx || a.ctx != b.ctx) return RSQ_ERR_INVALID;
    return guarded(a.ctx, [&] {
        if (!a.ownStats.empty() || a.nRowsTotal >= 0) failInvalid("rsq_table_append: the table is a shard that plans with unified statistics");
        appendTable(*a.ctx, a, b);
        delete &b;
    });
}

int rsq_db_create(rsq_ctx* ctx, rsq_db** out) {
    if (!ctx || !out) return RSQ_ERR_INVALID;
    *out = new rsq_db();
    (*out)->ctx = C(ctx);
    return RSQ_OK;
}

void rsq_db_destroy(rsq_db* db) {
    if (!db) return;
    if (db->last) rsq_query_destroy(db->last);
    for (auto& t : db->tables) delete t.second;
    delete db;
}

int rsq_db_adopt_table(rsq_db* db, rsq_table* table) {
    if (!db || !table) return RSQ_ERR_INVALID;
    return guarded(db->ctx, [&] {
        Table* t = T(table);
        if (db->tables.count(t->name) || db->schemas.count(t->name)) failInvalid("Table " + t->name + " already exists.");
        std::vector<std::pair<std::string, Type>> sch;
        for (auto& c : t->cols) sch.emplace_back(c.name, c.type);
        db->schemas[t->name] = sch;
        db->tables[t->name] = t;
    });
}

int rsq_db_report(const rsq_db* db, rsq_report* out) {
    if (!db || !out) return RSQ_ERR_INVALID;
    *out = db->lastReport;
    return RSQ_OK;
}

const char* rsq_db_message(const rsq_db* db) { return db ? db->message.c_str() : ""; }

int rsq_db_execute(rsq_db* db, const char* sqlText, int32_t* kind, rsq_result_view* result) {
    if (!db || !sqlText) return RSQ_ERR_INVALID;
    if (kind) *kind = 0;
    db->message.clear();
    rsq_ctx* cx = reinterpret_cast<rsq_ctx*>(db->ctx);
    int selectStatus = RSQ_OK;
    bool isSelect = false;
    int st = guarded(db->ctx, [&] {
        // control statements first (executeStatement, execute.h:512-516)
        std::string ctl;
        if (processControl(*db, sqlText, ctl)) { db->message = ctl; if (kind) *kind = 4; return; }
        rsq::ExprPool pool;
        rsq::sql::Statement stmt;
        rsq::sql::parse(sqlText, pool, stmt);
        if (kind) *kind = stmt.kind;
        if (stmt.kind == rsq::sql::Statement::CREATE_TABLE) {
            if (db->schemas.count(stmt.tableName)) failInvalid("Table " + stmt.tableName + " already exists.");
            if (stmt.schema.empty()) failInvalid("Create table needs at least one schema element.");
            db->schemas[stmt.tableName] = stmt.schema;
            // an empty relation that can be scanned (db.relations.try_emplace(name, schema), execute.h:279)
            static const uint8_t kEmpty[16] = {0};       // columns WITH data, of zero rows
            std::vector<rsq_column> cols(stmt.schema.size());
            for (size_t i = 0; i < cols.size(); i++) {
                memset(&cols[i], 0, sizeof cols[i]);
                snprintf(cols[i].name, RSQ_SYMBOL_MAX, "%s", stmt.schema[i].first.c_str());
                cols[i].type = stmt.schema[i].second.toC();
                cols[i].data = kEmpty;
            }
            rsq_table_desc d;
            memset(&d, 0, sizeof d);
            snprintf(d.name, RSQ_SYMBOL_MAX, "%s", stmt.tableName.c_str());
            d.n_rows = 0; d.n_cols = (int32_t)cols.size(); d.cols = cols.data();
            db->tables[stmt.tableName] = makeTable(*db->ctx, d, false);
            db->message = "Created table " + stmt.tableName + "\n";
        } else if (stmt.kind == rsq::sql::Statement::BULK_INSERT) {
            auto it = db->schemas.find(stmt.tableName);
            if (it == db->schemas.end()) failInvalid("Table " + stmt.tableName + " does not exist.");
            if (stmt.fieldTerminator.length() > 1) failInvalid("Bulk insert only supports single-character field terminators.");
            if (stmt.fieldTerminator.empty()) failInvalid("empty field terminator");
            auto have = db->tables.find(stmt.tableName);
            std::vector<rsq_column> cols(it->second.size());
            for (size_t i = 0; i < cols.size(); i++) {
                memset(&cols[i], 0, sizeof cols[i]);
                snprintf(cols[i].name, RSQ_SYMBOL_MAX, "%s", it->second[i].first.c_str());
                cols[i].type = it->second[i].second.toC();
            }
            rsq_table_desc d;
            memset(&d, 0, sizeof d);
            snprintf(d.name, RSQ_SYMBOL_MAX, "%s", stmt.tableName.c_str());
            d.n_cols = (int32_t)cols.size(); d.cols = cols.data();
            rsq_table* t = nullptr;
            int rc = rsq_table_load_tbl(cx, &d, stmt.fileName.c_str(), stmt.fieldTerminator[0], 0, &t);
            if (rc != RSQ_OK) throw Error(rc, db->ctx->lastError);
            const int64_t inserted = T(t)->nRows;
            if (have != db->tables.end() && have->second->nRows > 0) {
                // the reference appends to the relation (AppendIterator on the existing table, execute.h:348-350)
                std::unique_ptr<Table> more(T(t));
                appendTable(*db->ctx, *have->second, *more);
            } else {
                if (have != db->tables.end()) delete have->second;
                db->tables[stmt.tableName] = T(t);
            }
            db->message = "Inserted " + std::to_string(inserted) + " tuples\n";
        } else isSelect = true;
    });
    if (st != RSQ_OK || !isSelect) return st;
    // SELECT: executeSelect (execute.h:250-260)
    if (db->last) { rsq_query_destroy(db->last); db->last = nullptr; }
    std::vector<rsq_table*> arr;
    for (auto& t : db->tables) arr.push_back(reinterpret_cast<rsq_table*>(t.second));
    std::string planText;
    if (db->showPlan) {            // root->print(plan) before anything else (execute.h:217-220)
        rsq_sql_plan* p = nullptr;
        selectStatus = rsq_sql_plan_select(cx, sqlText, arr.data(), (int32_t)arr.size(), &p);
        if (selectStatus != RSQ_OK) return selectStatus;
        char* txt = rsq_sql_plan_text(p);
        if (txt) { planText = txt; free(txt); }
        rsq_sql_plan_destroy(p);
    }
    selectStatus = rsq_sql_compile(cx, sqlText, arr.data(), (int32_t)arr.size(), &db->last);
    if (selectStatus != RSQ_OK) return selectStatus;
    selectStatus = rsq_query_execute(db->last);
    if (selectStatus != RSQ_OK) return selectStatus;
    rsq_query_report(db->last, &db->lastReport);
    rsq_result_view view;
    selectStatus = rsq_query_result(db->last, &view);
    if (selectStatus != RSQ_OK) return selectStatus;
    if (db->writeResultsToFile) {            // writeRelationToFile(*rel, "qres.tbl") (execute.h:203-210, 243-245)
        selectStatus = guarded(db->ctx, [&] {
            FILE* f = fopen("qres.tbl", "w");
            if (!f) failInvalid("Could not open file qres.tbl");
            std::string text = serializeResultView(view);
            fwrite(text.data(), 1, text.size(), f);
            fclose(f);
        });
        if (selectStatus != RSQ_OK) return selectStatus;
    }
    // printQueryResult (execute.h:178-183): the plan line, then showReport
    db->message = planText + "\n" + reportText(*db, db->last, db->lastReport);
    if (result) *result = view;
    return RSQ_OK;
}

char* rsq_sql_describe(rsq_ctx* ctx, const char* sqlText, int32_t what) {
    if (!ctx || !sqlText) return nullptr;
    char* res = nullptr;
    guarded(C(ctx), [&] {
        if (what == 0) {
            bool err = false;
            std::string out;
            for (auto& t : rsq::sql::tokenize(sqlText, err)) { out += t.name; out += " "; out += t.text; out += "\n"; }
            if (err) out += "ERROR\n";
            res = strdup(out.c_str());
        } else {
            rsq::ExprPool pool;
            rsq::sql::Statement stmt;
            rsq::sql::parse(sqlText, pool, stmt);
            res = strdup(rsq::sql::dumpStatement(stmt).c_str());
        }
    });
    return res;
}

int rsq_measure_read_bandwidth(rsq_ctx* ctx, size_t bytes, int32_t iters, double* gb_per_s) {
    if (!ctx || !gb_per_s || iters <= 0) return RSQ_ERR_INVALID;
    return guarded(C(ctx), [&] { *gb_per_s = measureReadBandwidth(*C(ctx), bytes, iters); });
}

}  // extern "C"
