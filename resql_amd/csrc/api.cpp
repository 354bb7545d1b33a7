// api.cpp — the extern "C" surface of include/resql_hip.h.  No exception crosses it.
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
                tc.dptr = ctx.alloc(bytes); tc.owned = true;
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

extern "C" {

int rsq_ctx_create(const rsq_config* cfg, rsq_ctx** out) {
    if (!out) return RSQ_ERR_INVALID;
    *out = nullptr;
    rsq_config c{};
    if (cfg) c = *cfg;
    return guarded(nullptr, [&] { *out = reinterpret_cast<rsq_ctx*>(new Context(c)); });
}

void rsq_ctx_destroy(rsq_ctx* ctx) { delete C(ctx); }

const char* rsq_last_error(const rsq_ctx* ctx) {
    if (!ctx) return g_createError.c_str();
    return reinterpret_cast<const Context*>(ctx)->lastError.c_str();
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
    void* p = nullptr;
    partialBuffer(*reinterpret_cast<const QueryHandle*>(q)->q, &p, n_min_words, n_max_words, n_sum_words);
    return RSQ_OK;
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
};

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

int rsq_db_execute(rsq_db* db, const char* sqlText, int32_t* kind, rsq_result_view* result) {
    if (!db || !sqlText) return RSQ_ERR_INVALID;
    if (kind) *kind = 0;
    rsq_ctx* cx = reinterpret_cast<rsq_ctx*>(db->ctx);
    int selectStatus = RSQ_OK;
    bool isSelect = false;
    int st = guarded(db->ctx, [&] {
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
        } else if (stmt.kind == rsq::sql::Statement::BULK_INSERT) {
            auto it = db->schemas.find(stmt.tableName);
            if (it == db->schemas.end()) failInvalid("Table " + stmt.tableName + " does not exist.");
            auto have = db->tables.find(stmt.tableName);
            if (have != db->tables.end() && have->second->nRows > 0) failUnsupported("BULK INSERT into a table that already holds data");
            if (stmt.fieldTerminator.empty()) failInvalid("empty field terminator");
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
            if (have != db->tables.end()) delete have->second;
            db->tables[stmt.tableName] = T(t);
        } else isSelect = true;
    });
    if (st != RSQ_OK || !isSelect) return st;
    // SELECT: executeSelect (execute.h:250-260)
    if (db->last) { rsq_query_destroy(db->last); db->last = nullptr; }
    std::vector<rsq_table*> arr;
    for (auto& t : db->tables) arr.push_back(reinterpret_cast<rsq_table*>(t.second));
    selectStatus = rsq_sql_compile(cx, sqlText, arr.data(), (int32_t)arr.size(), &db->last);
    if (selectStatus != RSQ_OK) return selectStatus;
    selectStatus = rsq_query_execute(db->last);
    if (selectStatus != RSQ_OK) return selectStatus;
    rsq_query_report(db->last, &db->lastReport);
    if (result) return rsq_query_result(db->last, result);
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
