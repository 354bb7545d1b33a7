// engine.h — internal structures of the MI355X execution engine behind include/resql_hip.h.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <map>
#include <tuple>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "expr.h"
#include "hostref.h"
#include "mempool.h"
#include "resql_hip.h"

#define RSQ_RANK_CHUNK_BLOCKS 1024       /* 32-byte bitmap blocks one workgroup of the rank index handles (aot_kernels.hip) */

namespace rsq {

#define RSQ_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t _e = (call);                                                                    \
        if (_e != hipSuccess)                                                                      \
            throw ::rsq::Error(RSQ_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)

// ---- column statistics gathered when a table is created (planner input, like the reference's
// Relation::tupleNum(); lets the planner pick dense group ids) -----------------------------------
struct ColumnStats {
    bool valid = false;
    int64_t min = 0, max = 0;                  // numeric / date columns
    bool ascending = false;                    // ... and no row is smaller than the row before it (the table is clustered by this column)
    bool strictlyAscending = false;            // ... nor equal to it: the column's values are unique
    std::vector<uint8_t> distinctBytes;        // 1-byte columns (CHAR(1), BOOL): sorted distinct values
};

struct TableColumn {
    std::string name;
    Type type;
    void* dptr = nullptr;                      // device pointer, nullptr if the column was declared without data
    bool owned = false;
    ColumnStats stats;
};

struct Context;

inline std::atomic<uint64_t> g_nextTableUid{1};

struct Table {
    Context* ctx = nullptr;
    // identity for the context's plan memo: uid is unique in the process, version counts everything that changes what a query over the
    // table may have learnt (rows appended, statistics unified / refreshed, first row set)
    uint64_t uid = g_nextTableUid.fetch_add(1);
    uint64_t version = 0;
    void bumpVersion();                        // version++, and the context's key indexes over the old content are retired (runtime.cpp)
    uint64_t layoutVersion = 0;                // counts the times the columns MOVED (rsq_table_append allocates them anew): a statement compiled
                                               // before holds the old addresses and is refused (engine.cpp executeQuery)
    std::string name;
    int64_t nRows = 0;
    int64_t row0 = 0;                          // global index of the first row (row-range shards)
    // A shard of a larger table plans as that table: rsq_table_unify_shard_stats / rsq_multi_query_compile give every shard the row
    // count of the whole table (the reference sizes its hash tables from Relation::tupleNum(), and first rows are numbered over the
    // whole table) and the UNION of the shards' column statistics in cols[i].stats, so that all shards derive one dense group layout.
    // ownStats keeps what this shard's own rows say (empty: never unified) - the proof that shards are disjoint in a key needs it.
    int64_t nRowsTotal = -1;
    std::vector<ColumnStats> ownStats;
    int64_t totalRows() const { return nRowsTotal >= 0 ? nRowsTotal : nRows; }
    const ColumnStats& shardStats(size_t col) const { return ownStats.empty() ? cols[col].stats : ownStats[col]; }
    std::vector<TableColumn> cols;
    int findCol(const std::string& n) const {
        for (size_t i = 0; i < cols.size(); i++) if (cols[i].name == n) return (int)i;
        return -1;
    }
    ~Table();
};

// ---- JIT: hiprtc + code object cache ----------------------------------------------------------
struct Kernel {
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    bool fromCache = false;
    std::map<int, int> residentPerCU;          // workgroup size -> workgroups of this kernel one CU holds at a time (registers, LDS)
};

struct Context {
    rsq_config cfg{};
    int device = 0;
    int numCUs = 256;
    hipStream_t stream = nullptr;              // the stream every launch / copy of this context goes to
    hipStream_t ownStream = nullptr;           // created by the context; `stream` is a caller's stream after setStream()
    std::string lastError;
    std::string cacheDir;
    std::string includeDir;                    // where kernels/rsq_device.h lives
    std::string headerText;                    // its content (part of the code-object cache key)
    std::map<std::string, Kernel> kernels;     // by source hash
    uint32_t* dErr = nullptr;                  // device error word
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int jitCompiles = 0, jitCacheHits = 0;
    bool errWordClean = false;                 // *dErr is known to be 0 (fused steps reset it themselves and rely on that)
    uint64_t execEpoch = 0;                    // counts executions on this context (a query's "readied for the next execution" state is good for the very next one only)

    explicit Context(const rsq_config& c);
    ~Context();
    Kernel& getKernel(const std::string& source, const std::string& entry);
    // is the code object of `source` at hand (loaded, or in the on-disk cache) — i.e. would getKernel return without compiling?
    bool kernelCached(const std::string& source);
    bool kernelCachedOnDisk(const std::string& source);      // (no access to the table of loaded kernels: for the compiler thread)
    // compile `source` into the on-disk cache without touching the context's tables: safe on another host thread
    void compileToCache(const std::string& source);
    // ... several at once, in parallel helper processes (runtime.cpp); returns with every source in the on-disk cache
    void compileManyToCache(const std::vector<std::string>& sources);
    std::set<std::string> freshlyCompiled;      // keys compileManyToCache built and getKernel has not loaded yet
    std::mutex freshMutex;
    std::string cacheKey(const std::string& source);
    // Buffers of queries (join / aggregation tables, group rows, scratch) come out of the context's arenas (mempool.h): a statement
    // that is compiled, executed once and destroyed - all a ReSQL host ever does, reference src/execute.h:213-247 - must not pay
    // hipMalloc / hipFree / hipHostMalloc per execution.  A freed range is handed out again once the stream has drained
    // (alloc asks hipStreamQuery; waitForStream tells).  RSQ_ENGINE_DRIVER_ALLOC in rsq_config.engine_flags takes every buffer
    // from the driver as before (measurement, tests).
    void* alloc(size_t bytes);
    void free(void* p);
    void* allocPinned(size_t bytes, bool nonCoherent = false);      // host memory the device reads / writes (hipHostMalloc)
    void freePinned(void* p);
    void* allocRaw(size_t bytes);              // straight from the driver: table columns (large, long-lived)
    void freeRaw(void* p);
    void streamDrained();                      // the caller has just seen the stream idle: pending ranges are free
    // timing events of queries: a query takes them here and gives them back when it is destroyed (a fresh query of a one-launch step
    // created 512 of them in its first execution: 0.12 ms of a 0.35 ms statement)
    std::vector<hipEvent_t> eventPool;
    hipEvent_t takeEvent();
    void giveEvent(hipEvent_t e) { if (e) eventPool.push_back(e); }
    std::unique_ptr<Arena> devArena, pinArena, pinNcArena;
    bool driverAlloc = false;
    size_t arenaKeepBytes = 0;
    struct AllocStats { uint64_t devCalls = 0, pinCalls = 0, rawCalls = 0; double devMs = 0, pinMs = 0, rawMs = 0; } allocStats;
    void setStream(hipStream_t s, bool callers);   // callers == false: back to the context's own stream
    // large transient device buffers (partition records): freed buffers are kept and handed out again, because
    // hipMalloc / hipFree of multi-GB buffers costs tens to hundreds of ms
    void* scratchAlloc(size_t bytes);
    void scratchFree(void* p);
    std::vector<std::pair<void*, size_t>> scratchFreeList, scratchLive;
    // The device tail of a large dense aggregation (engine.cpp runDenseDeviceTail) works in one device arena and one pinned host
    // arena per query, tens of MB each.  A query gives its arenas back when it is destroyed and the next one takes them: a host
    // that runs shard after shard (or statement after statement) pays hipMalloc / hipHostMalloc once, not per query.  The host
    // side of the replay (hostref.h) keeps its scratch here for the same reason (fresh pages cost more than the work in them).
    struct TailArena { void* dev = nullptr; size_t devBytes = 0; void* pinned = nullptr; size_t pinnedBytes = 0; };
    // chain words of the group-row compaction's look-back (aot_kernels.hip k_compact_entries): [value:32 | launch number:30 | state:2];
    // a word of an earlier launch carries an earlier number and reads as "nothing yet", so the buffer is zeroed once, when it is made
    uint64_t* dCompactChain = nullptr;
    size_t compactChainWords = 0;
    uint32_t compactLaunch = 0;
    TailArena spareTailArena;
    ReplayScratch replayScratch;
    std::vector<uint32_t> replayOrder;
    // The plan memo (include/resql_hip.h rsq_engine_flags): what executions have learnt, by (kernel texts, table ids + versions).  A ReSQL
    // host compiles, executes once and deletes (reference src/execute.h:213-247); the reference's operators size their tables from
    // getSize() estimates and grow them while they run (qlib/hash.h:385-419) - here sizes are found by a counting pass, a read-back and
    // an allocation, which a second query over the same tables must not repeat.  Every entry is re-checked by the execution it serves.
    struct PlanMemo {
        struct Join { int64_t buildRows = -1; bool dupKeys = false; uint32_t lastCount = 0; };
        struct Pipe { int64_t stage2Rows = -1; bool stagedExact = false; std::vector<uint32_t> stagedCaps; int64_t stagedCapsRows = -1; };
        std::vector<Join> joins;
        std::vector<Pipe> pipes;
        int64_t hashCapacity = 0; uint32_t hashCount = 0; bool charGroupsNeedMerge = false;
        int64_t matLastTotal = -1;
        bool narrowRowsOff = false, fusedSelectOff = false, chainedIndexOff = false, scanChainedOff = false;
        uint32_t stageWorkgroups = 0;
        uint64_t stamp = 0;
    };
    std::map<std::string, PlanMemo> planMemo;
    // KEY INDEXES: the key bitmap with its rank words (kernels/rsq_device.h rank_of) of an engine-owned, strictly ascending key column - what a
    // DIRECT join table over the bare scan of that table consists of (HashTable::direct).  It is a function of the column alone, so the first
    // query that builds one leaves it here and every later execution - of that query or of any other over the same table version - probes
    // it as it stands (TPC-H Q12 at SF10: 46 us of key bits for 15 M orders + 11 us of index per execution).  Entries are shared by
    // reference count; a table that changes (bumpVersion) or goes retires its entries, and a retired entry is freed with its last user.
    struct KeyIndex { uint32_t* dBitmap = nullptr; int64_t bmBlocks = 0, bmMin = 0, bmBits = 0; int refs = 0; bool retired = false; uint64_t uid = 0; };
    typedef std::tuple<uint64_t, uint64_t, int64_t, int64_t, int> KeyIndexKey;      // table uid, version, rows, first row, column
    std::map<KeyIndexKey, KeyIndex> keyIndexes;
    void retireKeyIndexes(uint64_t uid);
    void releaseKeyIndex(KeyIndex* k);
    uint64_t planMemoClock = 0, planMemoHits = 0;
    bool planMemoOff = false;
};

rsq_config readConfig(const rsq_config* cfg);      // api.cpp: the host's struct (struct_size bytes), validated
inline bool jitInt16Cast(const Context& c) { return (c.cfg.compat_flags & RSQ_COMPAT_JIT_INT16_CAST) != 0; }

// launch helper: kernel takes one struct of 8-byte slots by value
// start / stop (optional): events that take the kernel's own begin and end (hipExtModuleLaunchKernel)
void launch(Context& ctx, Kernel& k, unsigned grid, unsigned block, const std::vector<uint64_t>& args, hipEvent_t start = nullptr,
            hipEvent_t stop = nullptr);

// AOT kernels (aot_kernels.hip)
void computeColumnStats(Context& ctx, Table& t);
void generateTable(Context& ctx, Table& t, int kind, int64_t row0, int64_t nRows, double sf, int64_t param, uint64_t seed);
double measureReadBandwidth(Context& ctx, size_t bytes, int iters);
size_t scanTempBytes(int64_t n);
void exclusiveScanCounts(Context& ctx, const uint32_t* counts, uint64_t* offs, int64_t n, void* temp, size_t tempBytes);
void exclusiveScanCountsChained(Context& ctx, const uint32_t* counts, uint64_t* offs, int64_t n, void* temp, size_t tempBytes);      // one launch; bit 512 of *ctx.dErr: repeat with the above
void fillU64Async(Context& ctx, uint64_t* dptr, size_t n, uint64_t value);
// several fills in one launch: whole 4-byte words, `value` a 64-bit pattern (aot_kernels.hip k_fill_batch)
struct FillItem { void* p; size_t bytes; uint64_t value; };
// the words an execution's host side reads first, written into host-mapped pinned memory by one kernel: [0] error word, [1] group count,
// [2] candidate count, [8 + i] row counter of pipeline i (null sources are skipped)
// ... and, when `rows` is given and the group count is at most maxInline, the first group-count rows of `rowWords` words into hostRows
// (host-mapped too): a handful of group rows travel with the status words
void publishStatusAsync(Context& ctx, uint64_t* hostWords, const uint32_t* err, const uint32_t* groupCount, const uint32_t* candCount,
                        const uint64_t* pipeStats, int nPipelines, const int64_t* rows = nullptr, int rowWords = 0, uint32_t maxInline = 0,
                        int64_t* hostRows = nullptr,
                        const uint64_t* word3 = nullptr,       // ... and *word3 into word [3] (a warm materialisation's row total)
                        const uint64_t* copySrc = nullptr, uint64_t* copyDst = nullptr, uint32_t copyWords = 0);      // ... and copyWords words besides (a small dense table)
void fillBatchAsync(Context& ctx, const FillItem* items, int count);
// bitmap-rank dictionary (aot_kernels.hip): the rank words of a bitmap laid out in nBlocks 32-byte blocks [rank | 224 bits]
// (chunkTotal / chunkBase[ceil(nBlocks / RSQ_RANK_CHUNK_BLOCKS) (+ 1)] are scratch), and the placement of appended build records at the rank of
// their key
void rankTableIndex(Context& ctx, uint32_t* bitmap, int64_t nBlocks, uint32_t* chunkTotal, uint32_t* chunkBase);
// the same index in one launch: chunk totals chained through `chain` (the chunkTotal scratch, zero before the launch); a workgroup
// that waits too long raises bit 128 of the device error word and the caller repeats with rankTableIndex
void rankTableIndexChained(Context& ctx, uint32_t* bitmap, int64_t nBlocks, uint32_t* chain, uint32_t* chunkBase);
void rankTablePlace(Context& ctx, const int64_t* temp, const uint32_t* used, uint32_t nWaves, uint32_t region, const uint32_t* nRecords, int nWords,
                    const uint32_t* bitmap, int64_t bmMin, int64_t bmBits, const uint32_t* chunkBase, int64_t nBlocks, int64_t* words,
                    int64_t capacity);
// multi-GPU group-by merge: out[w] = min | max | sum over nParts partial tables (`stride` words apart) by segment
// hostOut / hostFlag / seq (optional, tables of up to 2048 words): the merged table is also stored into host-mapped memory and
// announced by `seq` in *hostFlag
void mergePartialsAsync(Context& ctx, const int64_t* parts, int nParts, int64_t stride, int64_t nMin, int64_t nMax, int64_t nSum, int64_t* out,
                        int64_t* hostOut = nullptr, uint64_t* hostFlag = nullptr, uint64_t seq = 0);
// one launch: fill[0..nFill) = fillValue (u64), zeroA / zeroB cleared (u32 words), *count = 0 (any of them may be empty / null)
void prepareTableAsync(Context& ctx, uint64_t* fill, size_t nFill, uint64_t fillValue, uint32_t* zeroA, size_t nZeroA, uint32_t* zeroB, size_t nZeroB,
                       uint32_t* count);
// partitioned aggregation: counts[workgroup][partition] -> exclusive prefix inside each partition (in place), partition
// bounds partStart[0..P] and the record total
void partitionOffsets(Context& ctx, uint32_t* counts, int nWorkgroups, int nPartitions, uint64_t* totals, uint32_t* partStart, uint64_t* total);
// gather the occupied entries (first-row word != INT64_MAX) of a hash table that carries aggregates into
// packed rows [first row | table words | accumulator blocks]; *count receives the number of rows, at most maxRows are written
void compactEntries(Context& ctx, const int64_t* firstRow, int64_t capacity, const int64_t* words, int nWords, bool wordsAos,
                    const int64_t* acc, int nAcc, int64_t* outRows, uint32_t maxRows, uint32_t* count, bool unmix = false,
                    // optional: while writing the rows, collect the range of the images of row word `keyWord` (a sort key) into
                    // imageRange[0] = max(image), imageRange[1] = max(~image) — both must be 0 before the launch
                    int keyWord = -1, bool keyIs32 = false, bool keyDesc = false, uint64_t* imageRange = nullptr,
                    // narrow: rows of TWO words [slot | word `keyWord` of the packed row] instead (wide rows of which an ORDER BY ... LIMIT
                    // wants a few: selectTopCandidatesRangePublish fetches those from the table, TableEntries)
                    bool narrow = false,
                    // deref (device memory, nWords ints, or null): table words that are rebuilt from a string kept by address (entryDerefCode)
                    const int* deref = nullptr, int tabStride = 0);
// word w of a table entry = bytes [off, off + len) of the string whose address stands in table word src (len 1..8)
inline int entryDerefCode(int src, int off, int len) { return (int)(0x40000000u | ((unsigned)src << 16) | ((unsigned)off << 4) | (unsigned)len); }
// ... or simply table word src (entries that keep one word per carried value: the row's word numbers are not the entry's)
inline int entryPlainCode(int src) { return (int)(0x20000000u | ((unsigned)src << 16)); }
struct TableEntries {
    const int64_t* firstRow; int64_t capacity; const int64_t* words; int nWords; bool wordsAos; const int64_t* acc; int nAcc; bool unmix;
    const int* deref = nullptr;
    int tabStride = 0;       // words between the entries of `words` (0: nWords)
};
// ORDER BY ... LIMIT pre-selection: the rows of `rows` ([*nRows][stride] words) whose word `keyWord` is among the `want`
// leading values of the requested order (ties of the last one included) are copied to `cand`.  `scratch` (topkHistBytes() bytes)
// holds [image range: 2 x u64][candidate count: u32][pad][histograms]; the candidate count is read from scratch + 16.
// selectTopCandidates is the exact radix select (7 launches); selectTopCandidatesRange needs the image range collected by
// compactEntries (prepareTopCandidatesRange before that compaction) and selects a superset in 2 launches.
size_t topkHistBytes();
void selectTopCandidates(Context& ctx, const int64_t* rows, int stride, int keyWord, bool is32, bool desc, const uint32_t* nRows,
                         uint32_t rowsUpperBound, uint32_t want, uint64_t* images, void* scratch, int64_t* cand, uint32_t capacity);
void prepareTopCandidatesRange(Context& ctx, void* scratch);
size_t topkRangeScratchBytes();      // the part of the scratch prepareTopCandidatesRange zeroes (an execution may fold it into its batched fill)
void selectTopCandidatesRange(Context& ctx, const int64_t* rows, int stride, int keyWord, bool is32, bool desc, const uint32_t* nRows,
                              uint32_t rowsUpperBound, uint32_t want, void* scratch, int64_t* cand, uint32_t capacity);

// ... and in ONE launch that also delivers: the candidates are written into host-mapped pinned memory (`candHostMapped`, the device's
// view of it) and the last workgroup publishes the execution's status words like publishStatusAsync (candidate count = word 2), then `seq` in
// word 4 (system-scope release: whoever sees it sees candidates and status words).
// A meeting point that timed out sets bit 256 of *err.
void selectTopCandidatesRangePublish(Context& ctx, const int64_t* rows, int stride, int keyWord, bool is32, bool desc, const uint32_t* nRows,
                                     uint32_t rowsUpperBound, uint32_t want, void* scratch, int64_t* candHostMapped, uint32_t capacity,
                                     uint64_t* hostWords, uint64_t seq, uint32_t* err, const uint32_t* groupCount, const uint64_t* pipeStats, int nPipelines,
                                     // rows are compactEntries' narrow rows (stride 2, keyWord 1): the candidates are written as full packed rows read from these entries
                                     const TableEntries* entries = nullptr);

// devtail.hip: the tail of a large dense aggregation on the device (present groups, order by first row, the reference's hashes,
// packed result tuples); tail.cpp planDenseDeviceTail says whether a plan qualifies and describes keys and columns
struct DenseTailKey { int64_t min, card, stride; int32_t byteSet, typeTag; uint8_t values[32]; };
struct DenseTailKeys { int32_t n; DenseTailKey k[4]; };
struct DenseTailCol { int32_t kind, a, b, width, offset; };      // kind 0: group value of key a; 1: table block a; 2: AVG = block a * 100 / block b
struct DenseTailCols { int32_t n; DenseTailCol c[24]; };
void densePresentGroups(Context& ctx, const int64_t* firstBlock, int64_t D, uint32_t* flags /* [D + 1] */, uint64_t* offs /* [D + 1], offs[D] = count */,
                        void* scanTemp /* scanTempBytes(D + 1) */, uint64_t* outFirst, uint32_t* outGid);
size_t radixSortTempBytes(int64_t n);
bool radixSortPairs(Context& ctx, uint64_t* keysA, uint32_t* valsA, uint64_t* keysB, uint32_t* valsB, int64_t n, int keyBits, void* temp, size_t tempBytes);
void denseGroupHashes(Context& ctx, const uint32_t* gids, int64_t n, const DenseTailKeys& keys, uint64_t* hashes);
void denseResultRows(Context& ctx, const uint64_t* table, int64_t D, const uint32_t* gids, const uint32_t* order, int64_t nRows, const DenseTailKeys& keys,
                     const DenseTailCols& cols, int tupleSize, uint8_t* out);

// ... and the same tail for the GROUP ROWS of a hash / join-entry aggregation ([first row | table words | accumulator blocks], as
// compactEntries leaves them): order by first row, the reference's hashes from the group values, the replay, packed tuples.
// tail.cpp planRowsDeviceTail describes where the group values and output columns sit in a row.
struct RowTailKey { int32_t word, typeTag, len, pad; };      // group value k: row word `word`; strings: `len` bytes in consecutive words (little-endian words = the bytes in order)
struct RowTailKeys { int32_t n; RowTailKey k[16]; };
struct RowTailCol { int32_t kind, a, b, width, offset, len; };      // kind 0: the value at row word a (len > 0: a string); 1: row word a; 2: AVG = word a * 100 / word b
struct RowTailCols { int32_t n; RowTailCol c[24]; };
void rowTailFirstKeys(Context& ctx, const int64_t* rows, int stride, int64_t n, uint64_t* keys, uint32_t* idx);
void rowTailHashes(Context& ctx, const int64_t* rows, int stride, const uint32_t* idx, int64_t n, const RowTailKeys& keys, uint64_t* hashes);
void rowTailResultRows(Context& ctx, const int64_t* rows, int stride, const uint32_t* idx, const uint32_t* order, int64_t nRows, const RowTailCols& cols,
                       int tupleSize, uint8_t* out);

// the replay of the reference's aggregation hash table on the device (devtail.hip): level sizes on the host, everything else enqueued
bool replayLevels(uint64_t n, uint64_t minSize, std::vector<std::pair<uint64_t, uint64_t>>& levels);
size_t replayDeviceBytes(uint64_t n, uint64_t nMax);
void replayEmissionOrderDevice(Context& ctx, const uint64_t* hashes, uint64_t n, const std::vector<std::pair<uint64_t, uint64_t>>& levels, void* work, uint32_t* order);

// tbl.cpp: '.tbl' text -> columns with the reference's BULK INSERT semantics (execute.h:332-388)
void parseTblFile(const std::string& path, const std::vector<Type>& types, char terminator, int nThreads,
                  std::vector<std::vector<uint8_t>>& cols, int64_t& nRows);

// ---- query ------------------------------------------------------------------------------------
struct Query;
Query* compileQuery(Context& ctx, const rsq_plan_desc& plan, rsq_table* const* tables, int nTables);
void executeQuery(Query& q, bool partialOnly, bool async = false);
void finalizeQuery(Query& q);
void finalizeQueryHost(Query& q, const int64_t* words, size_t nWords);
void settleAsync(Query& q);
void awaitKernels(Query& q);
void resolveKernelTime(Query& q);
void mergeShardResults(Query& into, const std::vector<Query*>& parts);
// shards of one plan that does not end in a dense partial table (engine.cpp / tail.cpp; multi.cpp explains when each is used)
void setHoldTail(Query& q, bool hold);                  // execute stops in front of the host tail
void runTailMerged(Query& root, const std::vector<Query*>& parts);   // all parts' groups (or materialised rows) merged by key, then root's tail
bool shardGroupsDisjoint(const std::vector<Query*>& parts, std::string& why);   // provably no group in two shards (column statistics)
// shard statistics (api.cpp): one fixed-size blob per table; unify = plan this shard as the whole table (see Table::nRowsTotal)
size_t tableStatsBytes(const Table& t);
void exportTableStats(const Table& t, void* buf, size_t bytes);
void unifyShardStats(Table& t, const void* blobs, int nShards, size_t blobBytes);
bool queryOrderedWithLimit(const Query& q);             // ORDER BY ... LIMIT k at the root
bool queryAsyncCapable(const Query& q);                 // every pipeline can be enqueued without the host in between
bool queryIsDense(const Query& q);              // its aggregation ends in a dense partial table ([min | max | sum] words)
void queryDenseLayout(const Query& q, int64_t* nMin, int64_t* nMax, int64_t* nSum, void** dptr);
std::string queryPartialLayoutText(const Query& q);   // the "partial table: ..." line of explain (same on every mergeable shard)
void bindPartial(Query& q, void* dptr, size_t bytes);
void mergeGathered(Query& q, const void* gathered, int nRanks);
void partialBuffer(Query& q, void** dptr, int64_t* nMin, int64_t* nMax, int64_t* nSum);
void queryResult(Query& q, rsq_result_view* out);
void queryReport(const Query& q, rsq_report* out);
void queryKernelTimeStats(Query& q, double* sumMs, uint64_t* executions, bool reset);
const char* querySource(const Query& q);
const char* queryExplain(const Query& q);
void destroyQuery(Query* q);

std::string serializeResultView(const rsq_result_view& v);

}  // namespace rsq
