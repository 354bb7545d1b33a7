/*
 * resql_hip.h — C ABI of the MI355X-native execution engine for ReSQL operator pipelines.
 *
 * This library takes the place of ReSQL's JIT context: where the reference's
 * `executeSelectPlan` (reference src/execute.h:213-247) does
 *
 *     JitContextFlounder ctx ( config.jit );            // src/JitContextFlounder.h:228
 *     root->produceFlounder ( ctx, {} );                // operators emit Flounder IR
 *     ctx.compile();                                    // src/JitContextFlounder.h:410-456
 *     ctx.execute();                                    // src/JitContextFlounder.h:459-487
 *     rel = root->retrieveResult();                     // operators/materialize.h:48, orderby.h:87
 *
 * a host links this library and does
 *
 *     rsq_ctx_create      -> one engine context per GPU (replaces the JitContextFlounder ctor)
 *     rsq_table_create*   -> columns resident in HBM (replaces Relation / DataBlock row store,
 *                            src/dbdata.h:23-461, as the scan source; scan.h:85-166)
 *     rsq_query_compile   -> describe + compile: type derivation, pipeline extraction, HIP kernel
 *                            specialisation (replaces produceFlounder + ctx.compile())
 *     rsq_query_execute   -> launch the pipelines' kernels, synchronise (replaces ctx.execute())
 *     rsq_query_result    -> packed result tuples in ReSQL's own layout (replaces retrieveResult())
 *
 * Everything is plain C: opaque handles, plain pointers and sizes, int status codes.  No
 * exceptions cross the boundary and the library never calls exit() (the reference's
 * query_error() does, src/qlib/error.h:29-68): errors come back as a status code plus
 * rsq_last_error().
 *
 * The plan / table / result structs are in resql_plan.h.
 */
#ifndef RESQL_HIP_H
#define RESQL_HIP_H

#include "resql_plan.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rsq_ctx   rsq_ctx;
typedef struct rsq_table rsq_table;
typedef struct rsq_query rsq_query;

enum rsq_status {
    RSQ_OK = 0,
    RSQ_ERR_INVALID = 1,        /* malformed plan / arguments */
    RSQ_ERR_TYPE = 2,           /* what the reference reports as ResqlError during type derivation / codegen */
    RSQ_ERR_UNSUPPORTED = 3,    /* valid ReSQL plan, pipeline shape not implemented by this engine */
    RSQ_ERR_DEVICE = 4,         /* HIP runtime / compiler failure, no usable GPU */
    RSQ_ERR_RUNTIME = 5,        /* query-time error (division by zero, hash table full, ...) */
    RSQ_ERR_NOMEM = 6
};

/* Mirrors JitConfig (reference src/JitContextFlounder.h:86-109) field for field — behind `struct_size` — plus the device and the
 * engine's own settings.  struct_size = sizeof(rsq_config) of the header the HOST was compiled against: the library reads that many
 * bytes and takes every later field as 0, so the struct can grow without a host built against an older header handing over
 * garbage (0 is refused: an uninitialised struct). */
typedef struct rsq_config {
    uint32_t struct_size;        /* sizeof(rsq_config) */
    int32_t print_assembly;      /* JitConfig::printAssembly  -> dump generated HIP source */
    int32_t print_flounder;      /* JitConfig::printFlounder  -> dump the pipeline description */
    int32_t print_performance;   /* JitConfig::printPerformance */
    int32_t num_threads;         /* JitConfig::numThreads: host threads of the CPU path; unused by the GPU engine */
    int32_t emit_machine_code;   /* JitConfig::emitMachineCode: ignored (always in-process code objects) */
    int32_t optimize;            /* JitConfig::optimizeFlounder: ignored */
    int32_t device;              /* HIP device ordinal this context drives (one context per GPU / process) */
    const char* kernel_cache_dir;/* directory with pre-built code objects (NULL: <library dir>/../_kcache) */
    int32_t emission_order;      /* rsq_emission_order: order of an aggregation's rows when the plan does not sort them */
    uint32_t compat_flags;       /* rsq_compat bits: where "as the reference's source says" and "as its JIT executes" differ */
    uint32_t engine_flags;       /* rsq_engine_flags bits (0: the defaults) */
    uint32_t reserved0;
    int64_t  arena_reserve_bytes;/* device memory the context takes from the driver when it is created, for the tables and buffers of its
                                  * queries (0: 2 GiB, and 40 MiB of pinned host memory; < 0: nothing up front - slabs are taken when first needed) */
    int64_t  arena_keep_bytes;   /* free arena memory the context keeps between queries before slabs go back to the driver (0: 1/8 of the
                                  * device's memory) */
} rsq_config;

/* A ReSQL host compiles a statement, executes it ONCE and deletes the plan (reference src/execute.h:213-247).  Two things make that one
 * execution cost what a repeated one costs, both owned by the context and both on by default:
 *   - memory arenas: join / aggregation tables, group rows and pinned read-back buffers of a query are ranges of slabs the context took
 *     from the driver once (hipMalloc / hipFree / hipHostMalloc cost more than a whole TPC-H Q3 at SF10).  RSQ_ENGINE_DRIVER_ALLOC takes
 *     every buffer from the driver instead;
 *   - the plan memo: what an execution learns about a plan over a table version - build cardinalities, whether build keys are unique
 *     (rank dictionary or hash form), group counts, region layouts of a staged aggregation, result sizes, which kernel form ran - is
 *     kept by (kernel texts, table ids + versions) and a later query of the same shape starts with it; everything in it is re-checked by
 *     the execution (a stale entry costs a repeated execution, never a wrong answer).  RSQ_ENGINE_NO_PLAN_MEMO starts every query cold. */
enum rsq_engine_flags { RSQ_ENGINE_DRIVER_ALLOC = 1u, RSQ_ENGINE_NO_PLAN_MEMO = 2u };

/* Semantics switches.  The default (0) computes what the reference's SOURCE specifies; a bit selects what its asmjit back end
 * actually executes where the two differ, for a drop-in host that must return the JIT's own answers:
 *   RSQ_COMPAT_JIT_INT16_CAST — TYPECAST INT -> BIGINT is a 32 -> 64-bit sign extension in ExpressionsJitFlounder.h:818-824 (`movsx`);
 *     the asmjit translation encodes the 16-bit form, so the JIT extends the LOW 16 BITS (`l_orderkey < 3` also keeps key 65537).
 *     With the bit set device code, interpreter and host tail do the same.
 * Not selectable: ht_get's missing wrap-around (src/qlib/hash.h:427-477) reads one entry past the table's allocation, so what the
 * reference returns for a probe chain that crosses the table's end depends on heap contents; the engine always walks the chain
 * (tests/test_reference_defect.py pins the difference on the live reference). */
enum rsq_compat { RSQ_COMPAT_JIT_INT16_CAST = 1u };

/* Without ORDER BY the reference emits an aggregation's groups in the slot order of its hash table (operators/aggregation.h:
 * 298-343), which depends on the order the groups first occur in the input, its hash function, prime table sizes and growth rule.
 * RSQ_EMIT_REFERENCE (0, the default) reproduces that order exactly — the result relation is byte-identical to ReSQL's — at the
 * price of replaying that table on the host (a few ms per million groups).  RSQ_EMIT_ANY returns the same rows in whatever
 * order the device tables hold them (the reference's own tests compare such results as multisets, test/test_common.h:152-190);
 * ORDER BY still sorts them with the reference's quicksort, but rows that tie on all sort keys may then come in another order. */
enum rsq_emission_order { RSQ_EMIT_REFERENCE = 0, RSQ_EMIT_ANY = 1 };

/* Mirrors JitExecutionReport (reference src/JitContextFlounder.h:114-129), times in ms, plus the
 * GPU-side figures SURVEY.md §8(b) asks for. */
typedef struct rsq_report {
    double   compilation_time_ms;    /* describe + kernel specialisation (JIT or cache hit) */
    double   execution_time_ms;      /* host wall time of rsq_query_execute, launches to final sync */
    double   kernel_time_ms;         /* sum of device kernel durations (HIP events on the engine's stream) */
    double   finalize_time_ms;       /* host-side AVG / projection / order-by / limit over the group rows */
    uint64_t num_kernels;            /* launches issued by the last execute (cf. numMachineInstructions) */
    uint64_t bytes_read;             /* ALGORITHMIC bytes: column bytes the pipelines must read once */
    double   hbm_gbps;               /* bytes_read / kernel_time */
    int32_t  jit_cache_hits;         /* pipelines served from the code-object cache */
    int32_t  jit_compiles;           /* pipelines compiled with hiprtc in this call */
} rsq_report;

/* ---- context ----------------------------------------------------------------------------- */
int  rsq_ctx_create(const rsq_config* cfg, rsq_ctx** out);
void rsq_ctx_destroy(rsq_ctx* ctx);
/* Message of the last failing call on this context (or of rsq_ctx_create when ctx == NULL). */
const char* rsq_last_error(const rsq_ctx* ctx);
/* What the context holds (its arenas, see rsq_engine_flags) and how often it went to the driver: a host that runs statement after
 * statement sees driver calls stop growing once the slabs cover its working set.  struct_size = sizeof(rsq_memory_stats), as in rsq_config. */
typedef struct rsq_memory_stats {
    uint32_t struct_size;
    uint32_t reserved0;
    uint64_t device_slab_bytes;      /* device memory held by the arena */
    uint64_t device_used_bytes;      /* ... of it handed out to live queries */
    uint64_t device_slab_allocs;     /* hipMalloc calls made for slabs since the context was created */
    uint64_t pinned_slab_bytes;      /* pinned host memory held (coherent + non-coherent arenas) */
    uint64_t pinned_used_bytes;
    uint64_t pinned_slab_allocs;     /* hipHostMalloc calls made for slabs */
    uint64_t raw_driver_calls;       /* hipMalloc / hipFree calls outside the arenas (table columns; every buffer with RSQ_ENGINE_DRIVER_ALLOC) */
    uint64_t arena_requests;         /* buffers handed out by the arenas */
    double   driver_ms;              /* host time spent inside those driver calls */
    uint64_t plan_memo_entries;      /* plans the context remembers */
    uint64_t plan_memo_hits;         /* queries that were compiled with a remembered entry */
    uint64_t key_index_entries;      /* key bitmaps of engine-owned key columns kept for every query over the same table version (joins whose build side is that table as it stands) */
    uint64_t key_index_bytes;        /* ... device memory they hold (part of device_used_bytes; freed with the table or when it changes) */
} rsq_memory_stats;
int  rsq_ctx_memory_stats(const rsq_ctx* ctx, rsq_memory_stats* out);

/* ---- tables ------------------------------------------------------------------------------ */
/* Copy host columns to the device (H2D is outside every timed region). Column statistics the
 * planner uses (row count as Relation::tupleNum(), min/max, byte-value sets) are gathered here. */
int  rsq_table_create(rsq_ctx* ctx, const rsq_table_desc* desc, rsq_table** out);
/* Adopt columns that already live in this GPU's memory (e.g. torch tensors): `data` pointers are
 * device pointers, not copied, and must outlive the table.  The column statistics gathered here (min / max, byte-value sets,
 * whether a column ascends) shape the kernels compiled against the table (dense group ids, key bitmaps, rank dictionaries), so
 * the CONTENT of adopted columns should not change while queries compiled over the table are in use.  If it does, nothing is
 * answered wrongly in silence: kernels over adopted columns check every value they derive from the statistics - a key outside
 * [min, max], a CHAR(1) / BOOL value that is not in the byte-value set (also one INSIDE the range: 'B' where the set was
 * {A, N, R}), two build rows with one key where the keys were unique - and the execution fails with RSQ_ERR_RUNTIME ("column
 * statistics"), or falls back to the general form of the table, instead of touching memory it should not or counting a row into a
 * neighbouring group.  (Columns the engine owns - uploaded, generated, loaded from '.tbl' - cannot change and skip those checks.)
 * After changing adopted data call rsq_table_refresh_stats and compile the statements anew. */
int  rsq_table_create_device(rsq_ctx* ctx, const rsq_table_desc* desc, rsq_table** out);
/* Gather the column statistics of `t` again (adopted columns whose content the host changed in place; the reference has no
 * counterpart - its operators take whatever values arrive, operators/aggregation.h:240-295).  Queries compiled BEFORE the call keep the
 * statistics they were compiled with (and their checks); compile the statement again to plan with the new ones. */
int  rsq_table_refresh_stats(rsq_table* t);
/* Transpose a ReSQL row store (reference src/dbdata.h: DataBlocks of packed tuples, strings by
 * value) into device columns: the bridge for a host that keeps ReSQL's own Relation objects. */
int  rsq_table_from_rowstore(rsq_ctx* ctx, const rsq_table_desc* schema /* data pointers ignored */,
                             const uint8_t* const* blocks, const size_t* content_size, int32_t n_blocks,
                             rsq_table** out);
/* BULK INSERT: load a '.tbl' text file (one tuple per line, fields ended by `field_terminator`, dbgen style with or
 * without the trailing terminator) straight into device columns, with the parsing rules of the reference's
 * executeBulkInsert / parseConstant (reference src/execute.h:332-388, src/expressions.h:369-515): see
 * resql_amd/csrc/tbl.cpp for the rules that matter (DECIMAL = digits with the point removed, BIGINT through int32,
 * dates as yyyy-mm-dd or yyyy/mm/dd).  `schema` gives names and types (data pointers ignored).
 * n_threads <= 0: all host threads.  Malformed lines give RSQ_ERR_INVALID with the line number. */
int  rsq_table_load_tbl(rsq_ctx* ctx, const rsq_table_desc* schema, const char* path, char field_terminator,
                        int32_t n_threads, rsq_table** out);
/* Fill a lineitem / orders / customer / synthetic table on the device with the deterministic
 * generator of resql_amd/datagen.py (same bits), rows [row0, row0 + n_rows).  kind: 0 lineitem,
 * 1 orders, 2 customer, 3 synthetic 4 x int64 (param = number of groups). */
int  rsq_table_generate(rsq_ctx* ctx, int32_t kind, int64_t row0, int64_t n_rows, double scale_factor,
                        int64_t param, uint64_t seed, rsq_table** out);
int64_t rsq_table_rows(const rsq_table* t);
/* BULK INSERT appends (executeBulkInsert adds tuples to the Relation it finds, reference src/execute.h:332-388; a Relation also grows
 * under its AppendIterator, src/dbdata.h:246-330): the rows of `more` - same context, same column types, made by any of the functions
 * above from the NEW tuples only - go behind the rows of `t`; `more` is consumed (destroyed) on success.  Column statistics are gathered
 * again and the columns move: a statement compiled BEFORE the call is refused afterwards (rsq_query_execute returns RSQ_ERR_INVALID,
 * "compile it again") - a ReSQL host compiles per statement anyway (execute.h:213-247). */
int  rsq_table_append(rsq_table* t, rsq_table* more);
/* A table that is a row range [row0, row0 + n_rows) of a larger one (a shard): rows are numbered from row0 wherever a row number
 * is observable — the order in which groups first occur decides the emission order of an aggregation (operators/aggregation.h:
 * 298-343 scans the hash table the groups entered in input order).  Set before queries are compiled over the table. */
int  rsq_table_set_first_row(rsq_table* t, int64_t row0);
/* Shards of ONE table must plan as that table.  The reference has one Relation and one hash table all its workers reach
 * (src/operators/aggregation.h:240-295, src/JitContextFlounder.h:459-487), so it never sees a "shard"; here the planner reads column
 * statistics (dense group ids come from a column's byte-value set or [min, max] range) and the row count (the reference sizes its
 * aggregation table from Relation::tupleNum(), which decides the emission order) of the table it is handed.  Before compiling
 * a plan over a shard, give it every shard's statistics: rsq_table_stats_export writes this table's fixed-size blob
 * (rsq_table_stats_bytes; the same size on every shard of a schema), the host moves the blobs (one all-gather between rank
 * processes), and rsq_table_unify_shard_stats(t, blobs of ALL shards incl. t's own, back to back) makes t plan with the union of the
 * value sets / ranges and the summed row count.  All shards then derive the same partial-table layout whatever their rows hold
 * (a shard without any 'R' row still keeps a cell for 'R'); where the union is not dense the plan takes the hash aggregation on
 * every shard alike.  rsq_multi_query_compile does this itself for the shards it is given.  rsq_table_total_rows: the summed count
 * (before unification: this table's own). */
int64_t rsq_table_stats_bytes(const rsq_table* t);
int  rsq_table_stats_export(const rsq_table* t, void* buf, int64_t bytes);
int  rsq_table_unify_shard_stats(rsq_table* t, const void* blobs, int32_t n_shards, int64_t blob_bytes);
int64_t rsq_table_total_rows(const rsq_table* t);
/* Copy one device column back (tests). */
int  rsq_table_read_column(rsq_ctx* ctx, const rsq_table* t, const char* name, void* host_dst, size_t bytes);
void rsq_table_destroy(rsq_table* t);

/* ---- queries ----------------------------------------------------------------------------- */
/* tables[i] is the table SCAN operators refer to by index i. */
int  rsq_query_compile(rsq_ctx* ctx, const rsq_plan_desc* plan, rsq_table* const* tables, int32_t n_tables,
                       rsq_query** out);
/* Run all pipelines and finalise the result.  Blocking, like JitContextFlounder::execute(). */
int  rsq_query_execute(rsq_query* q);
/* Low compile latency (the reference compiles a query in 0.6-3 ms, src/JitContextFlounder.h:410-456): a plan whose specialised
 * kernels are not in the code-object cache is served by pre-compiled interpreter kernels from its first execution on - whole
 * pipelines: scans, selections, projections, join builds and probes, strings, dense / hash / at-the-join-entry aggregation,
 * materialisation - while hiprtc builds the kernels on host threads in two tiers (a quick one with stage 2 of a compacted
 * pipeline as a call, then the inlined one); an execution that finds a tier ready switches over.  Results are identical on
 * every tier.  rsq_query_await_kernels blocks until the query runs on its final kernels (benchmarks call it before timing);
 * it returns at once for queries that never were on the interpreter.  Environment: RSQ_GENERIC=0 compiles blocking,
 * RSQ_FORCE_GENERIC=1 keeps eligible plans on the interpreter (tests). */
int  rsq_query_await_kernels(rsq_query* q);
/* Multi-GPU (row-range sharded scans): run the pipelines up to the aggregation and stop.  The
 * dense partial aggregate table then sits in device memory at *dev_ptr as int64 words laid out
 * [ n_min_words | n_max_words | n_sum_words ]: the first segment merges with MIN (first-row
 * trackers, MIN aggregates), the second with MAX, the third with SUM (sums, counts) — one RCCL
 * all-reduce per non-empty segment, issued by the host (one process per GPU).
 * rsq_query_finalize then produces the result from the reduced table. */
int  rsq_query_execute_partial(rsq_query* q, void** dev_ptr, int64_t* n_min_words, int64_t* n_max_words,
                               int64_t* n_sum_words);
/* Same step without the host synchronisation: the pipelines are enqueued on the context's stream and the call
 * returns.  Work the caller enqueues on that stream afterwards (the merge collective over the bound partial table,
 * rsq_query_bind_partial) is ordered behind the kernels; rsq_query_finalize() synchronises once, checks the device
 * error word and fills the report.  Dense aggregations without join build pipelines only (everything else needs the
 * host between kernels); pair it with rsq_ctx_set_stream so that the caller's collectives share the stream. */
int  rsq_query_execute_partial_async(rsq_query* q);
/* Make the context launch on the caller's HIP stream (use_callers_stream != 0; hip_stream may be the null stream) or go
 * back to its own non-blocking stream (use_callers_stream == 0).  Synchronises the stream it leaves. */
int  rsq_ctx_set_stream(rsq_ctx* ctx, void* hip_stream, int32_t use_callers_stream);
int  rsq_query_finalize(rsq_query* q);
/* Same, from a partial aggregate table that is already in host memory (n_words int64 words in the
 * layout above).  Needs no device: it is the merge + finalisation step of the multi-GPU path in
 * isolation, which is how the CPU-only tests cover it (gloo). */
int  rsq_query_finalize_host(rsq_query* q, const int64_t* words, int64_t n_words);
/* Word counts of the partial aggregate table ([min | max | sum] segments), known after compile. */
int  rsq_query_partial_layout(const rsq_query* q, int64_t* n_min_words, int64_t* n_max_words, int64_t* n_sum_words);
/* Make the query keep its partial aggregate table in caller-owned device memory (e.g. a torch
 * tensor the host hands to RCCL) instead of its own allocation; `bytes` must cover all words. */
int  rsq_query_bind_partial(rsq_query* q, void* dev_ptr, size_t bytes);
/* The kernel behind the merge collective of small tables: `gathered_dev` holds the partial tables of `n_ranks` ranks back to
 * back (the receive buffer of ONE all-gather, device memory); they are reduced segment by segment (min | max | sum) into this
 * query's own partial table (the bound one).  Enqueued on the context's stream; no synchronisation. */
int  rsq_query_merge_gathered(rsq_query* q, const void* gathered_dev, int32_t n_ranks);
int  rsq_query_result(rsq_query* q, rsq_result_view* out);
int  rsq_query_report(const rsq_query* q, rsq_report* out);
/* Device time (HIP events around the launches) summed over the executions since the last reset, and their number: what a
 * benchmark loop reads ONCE after its timed region instead of a report per step (the reference prints executionTime per
 * query, JitContextFlounder.h:132-150; a loop of 50 steps would otherwise time its own bookkeeping). */
int  rsq_query_kernel_time_stats(rsq_query* q, double* sum_ms, uint64_t* executions, int32_t reset);
/* Generated HIP source and pipeline description of the compiled query (debugging). */
const char* rsq_query_source(const rsq_query* q);
const char* rsq_query_explain(const rsq_query* q);
void rsq_query_destroy(rsq_query* q);

/* ---- host-side helpers that mirror reference free functions ------------------------------ */
/* serializeExpr after deriveExpressionTypes (reference src/expressions.h:177-204, 1367-1392): lets a
 * host check the engine's typing against the reference's (test/test_datatypes.h).  Returns a
 * malloc'ed string (free with rsq_free) or NULL. */
char* rsq_serialize_expr(rsq_ctx* ctx, const rsq_plan_desc* plan, int32_t expr, int32_t derive,
                         rsq_table* const* tables, int32_t n_tables);
/* serializeRelation (reference src/dbdata.h:688-701) of a result view. */
char* rsq_result_serialize(const rsq_result_view* view);
void  rsq_free(void* p);
/* The slot order of the reference's aggregation hash table (allocateHashTable / ht_put / growHashTable, src/qlib/hash.h:225-287,
 * 330-419) after inserting n groups with the given Values::hash values in this order into a table allocated for min_size
 * entries: out[k] = index of the group in the k-th occupied slot.  parallel != 0 takes the cluster-parallel replay the engine's
 * tail uses for many groups, 0 the sequential one; both give the same permutation (tests compare them). */
int   rsq_ref_emission_order(const uint64_t* hashes, int64_t n, uint64_t min_size, int32_t parallel, uint32_t* out);
/* The same on the GPU of `ctx` (the form the tail of a large dense aggregation uses): hashes and the result travel through device
 * memory.  RSQ_ERR_UNSUPPORTED for tables beyond the device path's range (2^31 slots). */
int   rsq_ref_emission_order_device(rsq_ctx* ctx, const uint64_t* hashes, int64_t n, uint64_t min_size, uint32_t* out);

/* ---- SQL text in front of the path --------------------------------------
 * The reference turns SQL text into an operator tree with parseSql (src/parser/parseSql.h:130-166:
 * tokens of src/parser/lexer.y, grammar of src/parser/parser.y) and buildQuery (src/planner.h:409-497),
 * then calls executeSelectPlan (src/execute.h:213-247) — the boundary above.  These entry points are
 * that front end for a host that has no ReSQL planner of its own: the same tokens, grammar, desugaring
 * (BETWEEN, IN, CASE) and planning rules, producing the plan description of resql_plan.h.
 *
 * `tables` is the database: every table a FROM clause may name (rsq_table carries name, schema and
 * row count — what Database::relations holds, src/dbdata.h).  Scan operators of the plan index into it,
 * so the same array goes to rsq_query_compile.  Errors: RSQ_ERR_INVALID "Syntax error." (execute.h:520-523),
 * "Table x does not exist." (planner.h:437-439); RSQ_ERR_UNSUPPORTED for plans that need a nested-loops join. */
typedef struct rsq_sql_plan rsq_sql_plan;
int  rsq_sql_plan_select(rsq_ctx* ctx, const char* sql, rsq_table* const* tables, int32_t n_tables, rsq_sql_plan** out);
const rsq_plan_desc* rsq_sql_plan_desc(const rsq_sql_plan* plan);
void rsq_sql_plan_destroy(rsq_sql_plan* plan);
/* the operator tree as text (what `showplan` prints in spirit; tests compare it with the reference's planner);
 * malloc'ed, rsq_free.  The tables passed to rsq_sql_plan_select must still be alive. */
char* rsq_sql_plan_text(const rsq_sql_plan* plan);
/* parse + plan + rsq_query_compile in one call (executeSelect, src/execute.h:250-260, up to ctx.compile()) */
int  rsq_sql_compile(rsq_ctx* ctx, const char* sql, rsq_table* const* tables, int32_t n_tables, rsq_query** out);
/* The statement as text, for hosts that dispatch on its kind (executeStatement, src/execute.h:508-545) and for
 * tests: what = 0 the token names one per line ("NAME text"), what = 1 the parsed statement ("SELECT" with its
 * clause expressions, "CREATE_TABLE name" + "column name TYPE" lines, "BULK_INSERT name" + file / fieldterminator /
 * firstrow lines).  Returns a malloc'ed string (rsq_free) or NULL with rsq_last_error set. */
char* rsq_sql_describe(rsq_ctx* ctx, const char* sql, int32_t what);

/* ---- statement loop: executeStatement (src/execute.h:508-545) over the engine ----------------
 * A database handle keeps what the reference's `Database` keeps (src/dbdata.h: relations by name): schemas from CREATE TABLE
 * (executeCreateTable, execute.h:263-281: "Table x already exists."), device-resident tables from BULK INSERT
 * (executeBulkInsert, execute.h:332-388: "Table x does not exist.", field terminator = first character of
 * `fieldterminator`, default ','), and runs SELECT statements through rsq_sql_compile + rsq_query_execute.
 * rsq_db_execute: `result` may be NULL; for a SELECT it receives a view of the result relation that stays valid until the
 * next statement on the same handle.  *kind receives 1 SELECT, 2 CREATE TABLE, 3 BULK INSERT, 4 CONTROL (may be NULL).
 * A second BULK INSERT into a table appends, as the reference's does.
 * rsq_db_adopt_table hands an existing table (rsq_table_create / _generate / ...) to the database, which then owns it.
 * Destroy a database before the context it was created on (its tables live in that context's device memory). */
typedef struct rsq_db rsq_db;
int  rsq_db_create(rsq_ctx* ctx, rsq_db** out);
int  rsq_db_execute(rsq_db* db, const char* sql, int32_t* kind, rsq_result_view* result);
/* Control statements (processControl, execute.h:454-474; *kind receives 4): `name=value` or the bare name (prints the value)
 * for showplan, tofile, threads, showperf, showasm, showfln, optimize, emitmc — spaces are ignored and, as in the reference,
 * the name may stand anywhere in the line — and `tables`.  Effects on later SELECTs: showplan puts the operator tree,
 * showperf the `compile:` / `execute:` lines of showReport (JitContextFlounder.h:132-150) plus the device figures, showasm the
 * generated HIP source, showfln the pipeline description into the message; tofile=true writes the result to "qres.tbl"
 * (serializeRelation format, execute.h:203-210, 243-245); threads / optimize / emitmc are stored and reported only.
 * rsq_db_message: what the reference's printQueryResult prints for the last statement in front of the relation
 * (execute.h:173-200): the control statement's answer, "Created table x", "Inserted n tuples", or plan + report. */
const char* rsq_db_message(const rsq_db* db);
int  rsq_db_adopt_table(rsq_db* db, rsq_table* table);
int  rsq_db_report(const rsq_db* db, rsq_report* out);          /* of the last SELECT */
void rsq_db_destroy(rsq_db* db);

/* ---- one host process, N GPUs ------------------------------------------------------------------
 * JitContextFlounder::execute() fans ONE compiled function out to config.numThreads workers that pull morsels from a shared
 * iterator and joins them (reference src/JitContextFlounder.h:459-487).  The multi-GPU form of that call: a host process
 * (ReSQL's executeSelectPlan, src/execute.h:213-247, unchanged in shape) holds one handle over N GPUs; the morsels are
 * row-range shards of the scanned table, one per GPU and resident in its HBM; rsq_multi_query_execute runs the same compiled
 * pipelines on every GPU and merges the partial aggregate tables with ONE exchange step — an RCCL reduce over xGMI to the
 * root GPU (ncclReduce per [min | max | sum] segment, all segments and GPUs in one ncclGroup; communicators from
 * ncclCommInitAll; librccl is dlopen'ed by rsq_multi_create) — then finalises on the root.  Integer min / max / sum:
 * the result is bit-identical to the single-GPU one.
 *
 * Plans that do not end in a dense partial table (joins with many groups, hash aggregation, materialisation) run their
 * pipelines on every shard concurrently (every shard holds the full build-side tables) and merge on the host:
 * the GENERAL merge reads all shards' group rows (or materialised rows, in shard = scan order) back, re-aggregates groups that
 * occur in several shards by key (sum / min / max per accumulator, the smallest first row — the reference has ONE hash table
 * all its workers reach, src/operators/aggregation.h:240-295) and runs ONE tail (AVG, projection, emission order, ORDER BY,
 * LIMIT) over them: the result is the single-GPU result whatever the sharding.  Rows are numbered over the whole table for
 * that: shard tables carry their first row's number (rsq_table_generate's row0, rsq_table_set_first_row).  When the column
 * statistics prove that a group-by attribute has disjoint value ranges on the shards (the caller sharded on a boundary of
 * that key, rsq_multi_table_generate_on_key) and the plan ends in ORDER BY ... LIMIT k, every shard runs its own tail and only
 * its k leading rows are merged by the sort keys.
 *
 * devices may list a GPU more than once (shards that share a GPU: how a one-GPU box runs N shards); RCCL cannot hold a device
 * twice, so such handles — and any handle created with RSQ_MERGE_PEER_COPY — move the partial tables with peer copies to
 * the root and reduce them with the engine's merge kernel. */
enum rsq_merge_mode { RSQ_MERGE_AUTO = 0, RSQ_MERGE_RCCL = 1, RSQ_MERGE_PEER_COPY = 2 };
typedef struct rsq_multi_config {
    uint32_t struct_size;       /* sizeof(rsq_multi_config) of the header the host was compiled against (as rsq_config.struct_size) */
    int32_t n_devices;
    const rsq_config* base;     /* per-GPU context settings (NULL: the defaults); base->device is ignored.  A pointer, not a member: rsq_config
                                 * grows behind its own struct_size without moving the fields of this struct */
    const int32_t* devices;     /* HIP device ordinals, one shard each; devices[0] is the root */
    int32_t merge;              /* rsq_merge_mode; AUTO = RCCL when the devices are distinct */
    int32_t reserved0;
} rsq_multi_config;
typedef struct rsq_multi rsq_multi;
typedef struct rsq_multi_query rsq_multi_query;
int  rsq_multi_create(const rsq_multi_config* cfg, rsq_multi** out);
void rsq_multi_destroy(rsq_multi* m);                            /* after its queries and tables */
const char* rsq_multi_last_error(const rsq_multi* m);            /* m == NULL: of rsq_multi_create */
int32_t rsq_multi_devices(const rsq_multi* m);
rsq_ctx* rsq_multi_ctx(rsq_multi* m, int32_t shard);             /* the shard's context: create its tables on it */
const char* rsq_multi_merge_name(const rsq_multi* m);
/* rows [*row0, *row0 + *n_rows) of shard `shard` of n_shards: equal shards on 128-row tile boundaries, remainder to the last */
void rsq_multi_shard_rows(int64_t n_total, int32_t n_shards, int32_t shard, int64_t* row0, int64_t* n_rows);
/* rsq_table_generate over all shards: out_tables[i] = rows of shard i of an n_rows_total table, on GPU i */
int  rsq_multi_table_generate(rsq_multi* m, int32_t kind, int64_t n_rows_total, double scale_factor, int64_t param, uint64_t seed,
                              rsq_table** out_tables /* [n_devices] */);
/* The same with every shard boundary moved forward to the next change of `key_column` (an integer column the table is clustered
 * by, e.g. lineitem's l_orderkey): no key value spans two shards, which lets plans grouped by that key take the short merge
 * below.  Correctness never depends on it. */
int  rsq_multi_table_generate_on_key(rsq_multi* m, int32_t kind, int64_t n_rows_total, double scale_factor, int64_t param, uint64_t seed,
                                     const char* key_column, rsq_table** out_tables /* [n_devices] */);
/* tables[shard * n_tables + t] is table t of the plan on that shard's context */
int  rsq_multi_query_compile(rsq_multi* m, const rsq_plan_desc* plan, rsq_table* const* tables, int32_t n_tables, rsq_multi_query** out);
int  rsq_multi_query_execute(rsq_multi_query* q);                /* blocking, like JitContextFlounder::execute() */
int  rsq_multi_query_result(rsq_multi_query* q, rsq_result_view* out);
/* kernel_time_ms = the slowest shard's; shard_kernel_ms (may be NULL) receives every shard's */
int  rsq_multi_query_report(const rsq_multi_query* q, rsq_report* out, double* shard_kernel_ms /* [n_devices] */);
/* device time of the last execution's group-by merge (RCCL reduce / peer copies + merge kernel): an event pair on the root GPU's
 * stream around it, so it includes the root's wait for the slowest shard; 0 for one shard without a collective and for host merges */
double rsq_multi_query_collective_ms(const rsq_multi_query* q);
/* which merge the query takes and why (dense partial tables / ordered merge of LIMIT-ed rows / general merge), for logs and tests */
const char* rsq_multi_query_merge_name(const rsq_multi_query* q);
void rsq_multi_query_destroy(rsq_multi_query* q);

/* Sustained read-only streaming bandwidth of this GPU (grid-stride int64 sum over `bytes` of
 * resident memory): the measured roofline SURVEY.md §8(d) quotes fractions against. */
int  rsq_measure_read_bandwidth(rsq_ctx* ctx, size_t bytes, int32_t iters, double* gb_per_s);

#ifdef __cplusplus
}
#endif
#endif /* RESQL_HIP_H */
