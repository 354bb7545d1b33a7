// engine_internal.h — plan / pipeline structures shared by the planner (engine.cpp), the kernel
// generator (codegen.cpp) and the host tail (tail.cpp).
#pragma once

#include <atomic>
#include <chrono>
#include <thread>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "engine.h"
#include "hostref.h"

namespace rsq {

struct Attr { std::string name; Type type; };
typedef std::vector<Attr> Schema;
int schemaTupleSize(const Schema& s);
int schemaOffset(const Schema& s, const std::string& name);

struct OpNode {
    int tag = RSQ_OP_UNDEFINED;
    OpNode* parent = nullptr;
    OpNode* child[2] = {nullptr, nullptr};
    int nChildren = 0;
    std::vector<Expr*> exprs, exprs2;
    bool singleMatch = false;
    bool hasLimit = false;
    int64_t limit = 0;
    Table* table = nullptr;
    std::vector<Expr*> splitAgg;     // aggregation.h:167-179
    Schema schema;                   // RelOperator::_schema
    int hashTable = -1;              // HASHJOIN: index into Query::hashTables
};

// ---- device hash table (join build side; optionally carries aggregates "at the entry") ----------
// Open addressing, linear probing, power-of-two capacity.  Struct of arrays in HBM:
//   state[cap]            u32   0 = empty, 1 = occupied
//   words[(k) * cap + s]  i64   key words first, then payload words (every value widened to 8 bytes)
//   acc[(w) * cap + s]    i64   aggregate words when an aggregation is fused onto the entries
struct HashTable {
    int id = 0;
    std::vector<Attr> keys;          // build-side key values (names of the build key expressions)
    std::vector<Attr> payload;       // build-side attributes carried to the probe side
    std::vector<std::pair<Attr, int>> keyAlias;   // build-side attributes that ARE a key value: (attribute, key word) — not stored twice
    int64_t capacity = 0;
    bool unique = false;             // probed single-match
    bool aos = false;                // join tables: words[slot][k] (a slot's words share a cache line) instead of words[k][slot]
    bool keyCas = false;             // one integer key word that is never INT64_MIN: the key word is the slot state
                                     // (EMPTY = INT64_MIN, claimed and published by one 64-bit CAS); dState is unused
    uint32_t* dState = nullptr;
    int64_t* dWords = nullptr;
    int64_t* dAcc = nullptr;         // [nAccBlocks][capacity] when aggregated at the entry
    int nAccBlocks = 0;
    uint32_t* dCount = nullptr;      // number of occupied slots (set by the build kernel)
    uint32_t lastCount = 0;          // generic aggregation: groups found by the previous execution (keeps the load below 1/2)
    // key-domain bitmap (single integer key whose column statistics give a modest range): one bit per possible key
    // value, set by the build, tested by the probe BEFORE it touches the table.  A probe that cannot match costs one
    // cached 4-byte load instead of two random line fetches from a table many times larger.
    bool hasBitmap = false;
    int64_t bmMin = 0, bmBits = 0;
    uint32_t* dBitmap = nullptr;
    // Bitmap-rank dictionary (kernels/rsq_device.h rank_of): a table probed single-match over one integer key with a bitmap
    // can skip hashing altogether when its build keys are unique — decided by the sizing pass of the first execution, which
    // sets the key bits and notices a bit that was already set.  The build pipeline then appends {key, payload} records in
    // arrival order (dTemp), two tiny kernels turn the bitmap into rank prefixes and a placement kernel writes every record to
    // entry number rank(key): no CAS, no scattered read-modify-writes, nearly sequential stores for input clustered by the key.
    // `capacity` is then the number of entries; words[entry][k] and acc[block][entry] as for the hash form.
    // Component bitmap of a table with SEVERAL key words: one bit per possible value of one integer key component (a plain column of
    // the build pipeline's scan with a modest range).  A probe whose value for that component is a column of ITS scan tests the bit at
    // the top of stage 2, before any other table is touched: a necessary condition of the match (TPC-H Q5: four of five lineitem rows
    // that found their order have a supplier outside ASIA).  Bits are only ever set - the build side's columns are immutable and a
    // stale bit merely lets a row through to the real probe -, so the bitmap is zeroed once, when it is made.
    bool hasCompBitmap = false;
    int compWord = 0;                // which key word
    int64_t cbMin = 0, cbBits = 0;
    uint32_t* dCompBitmap = nullptr;
    bool identityCapable = false;    // rank dictionary over a bare scan in key order: entry number == row number when every row is inserted (codegen.cpp consumeBuild)
    bool uniqueKnown = false;        // ... and the column statistics say its values are unique (strictly ascending): every row is inserted under its own key, no sizing pass
    bool identity = false;           // ... and this is so in this query
    bool dense = false;              // ... and the keys fill their whole range: entry number = key - min, probes skip bit and rank block (the sizing pass saw unique keys and as many entries as rows)
    // DIRECT: such a table whose key statistics are known beforehand (uniqueKnown) and whose payload values are all plain columns of the scanned
    // table is that table: entry number key - min is row number key - min, and a probe reads the payload from the build table's own columns
    // (a string's address is computed, not loaded).  Nothing is built, readied or indexed (TPC-H Q12: 240 MB of {o_orderkey, o_orderpriority}
    // records for 15 M orders were 87 us and three launches of the statement's 420 us).  Decided with the form, sizeJoinTable.
    bool directCapable = false;
    bool direct = false;
    std::vector<int> directCols;     // per payload word: the column of directSrc
    int directKeyCol = -1;           // the key's column of directSrc
    Context::KeyIndex* keyIndex = nullptr;      // the context's key index this table probes (dBitmap is then the index's, not the table's: Context::keyIndexes)
    bool keyIndexReady = false;      // dBitmap holds the bits and rank words of the build table at keyIndexVersion: nothing to build
    uint64_t keyIndexVersion = 0;
    const Table* directSrc = nullptr;
    bool rankCapable = false;
    bool setOnly = false;            // ... probed for all matches and carrying nothing but its key: in the rank form the bitmap alone (no entries)
    bool bmInterleaved = false;      // the bitmap's layout: 32-byte blocks of [rank word | 7 words = 224 bits] (rank-capable tables)
    bool rank = false;
    int64_t* dTemp = nullptr;        // arrival-order buffer: [wave of the build grid][tempRegion] records
    uint32_t* dTempUsed = nullptr;   // [wave] records the wave appended
    int64_t tempWaves = 0, tempRegion = 0;
    uint32_t* dChunkTotal = nullptr; // [chunks]
    uint32_t* dChunkBase = nullptr;  // [chunks + 1]
    int64_t bmBlocks = 0;            // 32-byte blocks the bitmap is allocated in (256 bits, or 224 bits + the rank word)
    // hash aggregation whose string group values are functions of its key: while the dependencies hold (every table of derefCondTables is a
    // rank dictionary in this execution) the kernel stores the string's ADDRESS in the value's first word and nothing else; the kernels
    // that make group rows rebuild the words (aot_kernels.hip table_word).  derefCodes[w]: engine.h entryDerefCode, 0 = the word itself.
    std::vector<int> derefCodes;
    std::vector<int> derefCondTables;
    int compactStride = 0;           // words per entry while the values stand by address (< keys + payload: one word per carried value)
    int* dDeref = nullptr;
    int64_t buildRows = -1;          // build rows and duplicate keys as the sizing pass saw them (what the plan memo keeps)
    bool dupKeys = false;
    bool prepared = false;           // this execution's first fill launch has readied the table (engine.cpp: the prologue); buildHashTable then skips its own
};

// one accumulator the aggregation keeps per group
struct Accum {
    int kind;            // RSQ_E_SUM / RSQ_E_MIN / RSQ_E_MAX / RSQ_E_COUNT
    std::string key;     // structural key of (kind, typed input): identical aggregates share an accumulator
    std::string input;   // device expression of the input value
    Type type;
    int merge;           // rsq::Merge on the device: 0 sum, 2 min (i64), 3 max (i64)
    Expr* inputExpr = nullptr;   // the input expression itself (null: the row index / a count), for the pre-compiled generic pipeline
};

struct DenseKey {
    Expr* expr = nullptr;
    Type type;
    bool byteSet = false;
    std::vector<uint8_t> values;     // byteSet: sorted distinct values
    int64_t min = 0;
    int64_t card = 1;
    int64_t stride = 1;
};

enum class AggMode { NONE, DENSE_REG, DENSE_LDS_PRIVATE, DENSE_LDS_SHARED, DENSE_GLOBAL, AT_JOIN_ENTRY, HASH };
enum class SinkKind { AGGREGATE, BUILD, MATERIALIZE };

struct ArgSlot { std::string name; std::string ctype; uint64_t value; };

// ---- the pre-compiled generic pipeline (generic.cpp, aot_kernels.hip k_generic_aggregate) ------------------------------
// One AOT kernel that interprets a register program per row: column loads, the reference's typed arithmetic / comparisons /
// CASE / TYPECAST, the filter, dense group ranks and the accumulator updates.  It serves a plan shape whose specialised
// kernel is not in the code-object cache yet, while hiprtc builds that kernel on a host thread (engine.cpp).
struct GenericInstr { uint8_t op, dst, a, b; uint32_t c; int64_t imm; };      // 16 bytes
enum GenericOp : uint8_t { G_COL = 1, G_CONST, G_ADD, G_SUB, G_MUL, G_DIV, G_LT, G_LE, G_GT, G_GE, G_EQ, G_NE, G_AND, G_OR, G_MULI, G_DIVI,
                           G_SELECT, G_FILTER };
enum { G_MAX_COLS = 16, G_MAX_KEYS = 4, G_MAX_ACCS = 16, G_MAX_SET = 16, G_REGS = 32, G_MAX_INSTR = 256 };
struct GenericProgram {
    std::vector<GenericInstr> code;
    struct Col { const void* ptr; int kind; };          // kind: 1 u8, 2 i32 (INT, and DATE read as signed 32-bit like the reference), 3 i64
    std::vector<Col> cols;
    struct Key { int reg; int byteSet; int64_t min, card, stride; uint8_t values[G_MAX_SET]; int nValues; };
    std::vector<Key> keys;
    struct Acc { int reg; int merge; int64_t block; };   // reg -1: the row index, -2: the constant 1
    std::vector<Acc> accs;
};
// ---- the interpreter for whole pipelines (generic2.cpp, generic_kernels.hip k_generic_pipeline) -------------------------
// Further instructions of the same 16-byte form: string values are device addresses (of the column's bytes, of a constant in the
// program's constant pool, or a payload word of a join table — strings travel by address there too).
enum GenericOp2 : uint8_t { G_COLADDR = 32, G_CONSTADDR, G_MOV, G_CAST16, G_STREQ, G_LIKE, G_STRWORD, G_PROBE };
enum { G2_REGS = 40, G2_MAX_COLS = 24, G2_MAX_TABLES = 8, G2_MAX_KEYW = 40, G2_MAX_DEPTH = 4, G2_MAX_OUT = 24, G2_MAX_ACCS = 16, G2_MAX_KEYS = 4, G2_MAX_SET = 32,
       G2_MAX_PAYLOAD = 24, G2_MAX_CHARKEYS = 8 };
enum { G2_SINK_BUILD = 1, G2_SINK_DENSE, G2_SINK_ENTRY, G2_SINK_HASH, G2_SINK_MATERIALIZE };
struct GenericProbeDesc {
    int32_t table, nKeys, nPayload, slotReg, single;
    uint8_t keyReg[8]; uint8_t payloadReg[G2_MAX_PAYLOAD];
};
struct GenericSinkDesc {
    int32_t kind, table, nKeys, nPayload, slotReg, nAccs, nOut, nCharKeys;
    uint8_t keyReg[G2_MAX_KEYW];               // BUILD: key words; HASH: all table words (compared keys, then carried values); DENSE: the group values
    uint8_t wordStr[G2_MAX_KEYW], wordOff[G2_MAX_KEYW], wordN[G2_MAX_KEYW];      // HASH: word w = bytes [off, off + n) of the string whose address is in keyReg[w]
    uint8_t payloadReg[G2_MAX_PAYLOAD];
    int32_t accReg[G2_MAX_ACCS], accMerge[G2_MAX_ACCS], accBlock[G2_MAX_ACCS];      // reg -1: the row number, -2: the constant 1
    int32_t keyByteSet[G2_MAX_KEYS], keyNValues[G2_MAX_KEYS]; int64_t keyMin[G2_MAX_KEYS], keyCard[G2_MAX_KEYS], keyStride[G2_MAX_KEYS];
    uint8_t keyValues[G2_MAX_KEYS][G2_MAX_SET];
    uint8_t charFirst[G2_MAX_CHARKEYS], charLast[G2_MAX_CHARKEYS];                   // HASH: word ranges of CHAR(n) group values (trailing-space note)
    uint8_t outReg[G2_MAX_OUT]; int32_t outWidth[G2_MAX_OUT], outString[G2_MAX_OUT], outSrcCap[G2_MAX_OUT];
};
struct GenericProgram2 {
    std::vector<GenericInstr> code;
    struct Col { const void* ptr; int width; };
    std::vector<Col> cols;
    std::vector<GenericProbeDesc> probes;
    std::vector<char> constPool;
    GenericSinkDesc sink{};
    int nRegs = 0;                   // registers the program names (the size of the kernel's register file in LDS)
    // device copies (made when the query is compiled)
    GenericInstr* dCode = nullptr; GenericProbeDesc* dProbes = nullptr; char* dConstPool = nullptr;
};
struct GenericTableRef { uint32_t* state; void* words; void* acc; uint64_t cap; uint32_t* count; int nWords; };
struct GenericPipelineLaunch {
    const GenericProgram2* prog; const GenericInstr* dCode; const GenericProbeDesc* dProbes; const char* dConstPool;
    GenericTableRef tables[G2_MAX_TABLES];
    int64_t nRows, row0;
    uint32_t* matCnt; const uint64_t* matOffs; uint64_t matLimit; void* matOut[G2_MAX_OUT]; int matPass;
    uint64_t* dense; int64_t denseGroups;
};
void launchGenericPipeline(Context& ctx, const GenericPipelineLaunch& L);
struct Query;
// generic2.cpp: one program per pipeline of the compiled query, or false (+ why) when some shape is not interpreted
bool buildGenericPlan(Query& q, std::vector<GenericProgram2>& out, std::string& why);

struct Pipeline {
    Table* src = nullptr;
    std::vector<int> cols;           // scanned columns (indices into src->cols)
    SinkKind sink = SinkKind::AGGREGATE;
    int buildTable = -1;             // SinkKind::BUILD
    std::string source;              // generated HIP source
    std::string sourceFlat;          // register-mode aggregation: the variant that flushes into the unpadded table
    Kernel* kernelFlat = nullptr;
    std::string sourcePass1;         // SinkKind::MATERIALIZE: the counting pass of the same pipeline
    Kernel* kernelPass1 = nullptr;
    // partitioned aggregation (large dense group domains, see emitDenseAggregation): the same pipeline compiled as a
    // per-(workgroup, partition) counting pass and as a record scatter pass, plus the per-partition LDS aggregation
    bool partitioned = false;
    std::string sourcePartCount, sourcePartScatter, sourcePartAgg;
    Kernel* kernelPartCount = nullptr; Kernel* kernelPartScatter = nullptr; Kernel* kernelPartAgg = nullptr;
    std::vector<ArgSlot> argsPartAgg;
    std::vector<int> partRecordInputs;   // accumulator indices whose input travels in the records (not row, not constant)
    int partCount = 0;                   // P: number of partitions
    int partGroups = 0;                  // groups per partition (power of two)
    int partAtomicsPerRow = 0;           // HBM atomics the direct form issues per passing row (sum accumulators)
    // form 3, staged partitioning (rsq_device.h): packed records through LDS rings, regions sized from a sample
    bool staged = false;
    double leadPass = -1.0;              // fraction of the rows the selection directly above the scan is expected to pass (column statistics), < 0: none
    bool lateLoads = false;              // the tile loop loads the columns behind the leading selection only for lanes with a passing row
    int stagedRecWords = 1, stagedRows = 4;
    std::string sourceStagedScatter, sourceStagedAgg;
    Kernel* kernelStagedScatter = nullptr; Kernel* kernelStagedAgg = nullptr;
    std::vector<ArgSlot> argsStagedAgg;
    bool stagedExact = false;            // the last execution overflowed a sampled region: size the regions by counting
    std::vector<uint32_t> stagedCaps;    // region capacities that held the last execution's records (reused while the row count stays)
    int64_t stagedCapsRows = -1;
    std::string entry = "rsq_pipeline";
    std::vector<ArgSlot> args;
    Kernel* kernel = nullptr;
    int64_t bytesPerRow = 0;
    bool hasStage2 = false;          // its kernel text has a stage 2 that RSQ_STAGE2_CALL turns into a real call (the quick tier of a cold compile)
    bool compact = false;            // wave-level selection compaction (codegen.cpp compactThen): carried 8-byte values
    int compactWords = 0;
    int compactWordsLazy = 0;        // ... in the late-load form (its queues are smaller)
    int gridPerCULazy = 2;
    std::vector<int> lazyCols;       // scanned columns only stage 2 needs (codegen.cpp compactThen): RSQ_LAZY 1 reads them by row
    std::string sourceLazy;
    Kernel* kernelLazy = nullptr;    // compiled when first chosen
    bool matSkip = false;            // a materialisation whose write pass skips the tiles that counted nothing (codegen.cpp)
    unsigned lastGrid = 0;           // workgroups of the most recent launch
    int64_t stage2Rows = -1;         // rows the previous execution sent to stage 2 (-1: not known yet)
    int ldsSlots = 0, ldsSlotBytes = 0;   // hash aggregation: slots of the LDS front table (the macro RSQ_LC_SLOTS) and bytes per slot
    std::map<std::string, Kernel*> fewGroupKernels;      // ... the same source compiled with 64 slots, per form (engine.cpp launchPipeline)
    int extraLdsBytes = 0;           // LDS a pipeline takes besides the compaction queues (hash aggregation's front table)
    int blockThreads = 256;
    int unroll = 2;
    unsigned maxGrid = 0;            // 256-thread workgroups per launch; 0 = gridPerCU per CU
    int gridPerCU = 2;               // 2 for pure streaming pipelines (measured optimum), 8 when the row path
                                     // does dependent random accesses (hash tables, HBM atomics): latency wants waves
    std::string explain;
};

struct TailState;
void destroyTailState(TailState* t);

struct Query {
    Context& ctx;
    ExprPool pool;
    std::vector<Expr*> exprs;
    std::vector<std::unique_ptr<OpNode>> ops;
    OpNode* root = nullptr;
    bool requestAll = false;
    std::vector<Table*> tables;
    std::vector<uint64_t> tableLayouts;    // Table::layoutVersion of each when the statement was compiled

    // device side
    std::vector<Pipeline> pipelines;
    std::vector<std::unique_ptr<HashTable>> hashTables;
    OpNode* agg = nullptr;                 // the aggregation whose input pipeline runs on the device (may be null)
    AggMode aggMode = AggMode::NONE;
    std::vector<DenseKey> denseKeys;
    int64_t denseGroups = 1;
    std::vector<Accum> accums;             // [0] is the first-row tracker
    std::vector<int> splitToAccum;         // splitAgg index -> accums index
    std::vector<int> accumSlot;            // accums index -> word-block index: blocks ordered [min | max | sum]
    int64_t nMinBlocks = 0, nMaxBlocks = 0, nSumBlocks = 0;
    int aggTable = -1;                     // AT_JOIN_ENTRY: hash table whose entries carry the aggregates
    std::vector<int> groupSource;          // AT_JOIN_ENTRY: per group expr, word index in the table (keys then payload)

    int aggPad = 1;                        // > 1 (register mode): kernels flush into dAggWork, cells `aggPad` words apart
    uint64_t* dAggWork = nullptr;          // padded working table and its identity image
    uint64_t* dAggWorkInit = nullptr;
    size_t padWords = 0;                   // cells * aggPad
    bool flatRun = false;                  // this execution is partial: kernels flush straight into dAgg
    size_t tableWords = 0;                 // accumulators * dense groups (the [block][group] table)
    bool dAggOwned = true;
    uint64_t* dAgg = nullptr;              // dense modes: [blocks][denseGroups]
    uint64_t* dAggInit = nullptr;          // identity image copied over dAgg at the start of every execute
    uint64_t* hPinned = nullptr;           // pinned read-back: aggregate words + error word
    size_t pinnedWords = 0;
    std::vector<uint64_t> hAgg;
    const uint64_t* hAggView = nullptr;    // where the tail reads the dense table: hAgg, or straight from the pinned read-back buffer (large unpadded tables)
    struct TailState* tailState = nullptr; // tail.cpp: group arrays and scratch kept between executions

    // partitioned aggregation buffers
    uint32_t* dPartCounts = nullptr;       // [workgroups][P] counts, turned into offsets in place
    size_t partCountsWords = 0;
    uint32_t* dPartStart = nullptr;        // [P + 1]
    uint64_t* dPartTotals = nullptr;       // [P] records per partition, [P] = grand total
    std::vector<void*> dPartRecords;       // [0] keys (group-in-partition << 40 | row - row0), then one array per record input
    uint64_t partRecordCapacity = 0;
    int64_t partTileStep = 1;              // > 1: the counting pass samples every n-th tile (selectivity estimate)
    // staged partitioning (form 3): region layout [P] base / capacity per workgroup, tracker control block, per-(workgroup, partition) counts
    double kernelTimeSumMs = 0; uint64_t kernelTimeLaunches = 0;      // device time of the executions since the last reset (rsq_query_kernel_time_stats)
    uint64_t* dDebugStamps = nullptr;      // RSQ_DEBUG_TAIL (measurement only)
    uint64_t* dPinnedDev = nullptr;        // hPinned as the device addresses it (status words are published by a kernel)
    uint64_t* dStageBase = nullptr; uint32_t* dStageCap = nullptr; void* dStageCtl = nullptr; void* hStageLayout = nullptr;
    uint32_t* dStageCounts = nullptr; size_t stageCountsWords = 0;
    size_t stageRecBytes = 0;              // bytes of dPartRecords[0] when it was allocated for form 3
    uint32_t stageMode = 0, stageWorkgroups = 0;

    uint64_t* dPipeStats = nullptr;        // per pipeline: rows that reached stage 2 (behind the wave compaction)

    // device-side materialisation (plans without aggregation)
    OpNode* matOp = nullptr;
    Schema matSchema;
    uint32_t* dMatCnt = nullptr;           // per lane-tile slot: tuples emitted
    uint32_t* dMatTileCnt = nullptr;       // per 128-row tile: tuples emitted (the sum of its 64 lane counts)
    uint64_t* dMatOffs = nullptr;          // exclusive scan of dMatTileCnt (+ total at the end)
    void* dScanTemp = nullptr; size_t scanTempBytes = 0;
    int64_t matSlots = 0;
    std::vector<void*> dMatCols;           // output columns (struct of arrays)
    std::vector<void*> hMatMapped;         // ... a small result's columns are host-mapped pinned memory (the write pass stores over PCIe; these are the host's pointers): no copy after the final wait
    int64_t matCapacity = 0;               // rows the output columns can hold
    uint64_t matLimit = 0;                 // rows pass 2 may write
    int64_t matRows = 0;
    int64_t matLastTotal = -1;             // rows the previous execution's count pass found (-1: unknown): a warm execution runs its write pass without
    bool matWarmRun = false;               // waiting for the total; the total then arrives with the status words and must equal the remembered one
    const uint64_t* dMatTotal = nullptr;   // (where it stands: the last offset of the scan)
    std::vector<std::vector<uint8_t>> hMatCols;

    // compacted group rows read back from a join-entry aggregation: [nGroups][groupWords]
    int64_t* dGroupRows = nullptr;
    uint32_t* dGroupCount = nullptr;
    int64_t* hGroupRows = nullptr;         // pinned (a pageable target makes the 6 MB read-back of Q3 SF10 cost ~1 ms)
    size_t hGroupRowsWords = 0;            // (allocated when rows are first read back: ensureHostGroupRows)
    size_t dGroupRowsWords = 0;
    int64_t nGroupRows = 0;
    int groupRowWords = 0;

    // ORDER BY ... LIMIT k above a join-entry / hash aggregation: candidate rows pre-selected on the device
    // (aot_kernels.hip selectTopCandidates; planDeviceTopK in tail.cpp decides whether the first sort key is a word of
    // the group row).  A candidate run hands the tail `nGroupRows` candidates of `totalGroups` groups; the tail sets
    // tailNeedsAllGroups when the candidates do not determine the answer (ties on all keys among the leading rows).
    int topkWord = -2;                     // -2 not analysed yet, -1 not applicable, else word index in the group row
    bool topkIs32 = false, topkDesc = false;
    uint32_t topkWant = 0;                 // k + 1
    uint64_t* dTopkImages = nullptr; size_t topkImageRows = 0;
    uint32_t* dTopkHists = nullptr;
    int64_t* dCandRows = nullptr; uint32_t candCapacity = 0; int candRowWords = 0;
    uint32_t* dCandCount = nullptr;
    bool candidateRun = false;
    int64_t totalGroups = 0;
    bool tailNeedsAllGroups = false;
    // hash aggregation with CHAR(n) group values: set by the kernel when a group value ends with a space — only then can
    // two device groups be one group of the reference (CHAR equality ignores trailing spaces) and the host has to merge
    bool charGroupsNeedMerge = false;
    bool topkNeedsNoMerge = false;         // the candidate pre-selection is valid only while no merge is needed

    // the tail of a large dense aggregation on the device (engine.cpp runDenseDeviceTail): buffers sized for all D groups
    int devTail = -1;                      // -1 not analysed yet, 0 no, 1 yes
    DenseTailKeys dtKeys{}; DenseTailCols dtCols{}; int dtTupleSize = 0; int64_t dtLimitRows = -1;
    Context::TailArena dtArena;            // everything below is carved out of it
    uint32_t* dtFlags = nullptr; uint64_t* dtOffs = nullptr; void* dtScanTemp = nullptr;
    uint64_t* dtFirst[2] = {nullptr, nullptr}; uint32_t* dtGid[2] = {nullptr, nullptr};
    void* dtSortTemp = nullptr; size_t dtSortTempBytes = 0;
    uint64_t* dtHashes = nullptr; uint32_t* dtOrder = nullptr; uint8_t* dtRows = nullptr;
    void* dtReplayWork = nullptr; size_t dtReplayBytes = 0;           // work area of the replay on the device (0: host replay)
    uint64_t* hDtHashes = nullptr; uint32_t* hDtOrder = nullptr;      // pinned
    uint8_t* resultPinned = nullptr;       // pinned copy of the result tuples (device tail)
    // ... of the group rows of a hash / join-entry aggregation (engine.cpp runRowsDeviceTail): buffers sized for rtCapacity rows
    int rowTail = -1;                      // -1 not analysed yet, 0 no, 1 yes
    RowTailKeys rtKeys{}; RowTailCols rtCols{}; int rtTupleSize = 0; int64_t rtLimitRows = -1; bool rtSorts = false;
    int64_t rtCapacity = 0;
    void* rtDev = nullptr; void* rtPinned = nullptr; size_t rtPinnedBytes = 0;
    bool resultInPinned = false;

    // result
    Schema resultSchema;
    std::vector<uint8_t> resultTuples;
    int64_t resultRows = 0;
    std::vector<char> rvNames;
    std::vector<rsq_type> rvTypes;
    std::vector<int32_t> rvOffsets;

    rsq_report report{};
    // a single register-mode pipeline carries the whole step in its one launch (engine.cpp, the fused step): the last
    // workgroup writes the finished table to finOut (host-mapped pinned memory, or the device partial table of a multi-GPU
    // step) and the error word to finErr, and resets the working table and the ticket
    uint32_t* dFinTicket = nullptr;
    uint64_t* dFinHost = nullptr;          // device-side address of hPinned
    uint64_t* finOut = nullptr;            // set around the launch; null: the kernel skips the hand-over
    uint64_t* finErr = nullptr;
    bool fusedReady = false;               // working table, ticket and error word are at their identities
    uint64_t finSeq = 0;                   // > 0 around a launch the host polls for: the number the last workgroup writes behind the error word
    uint64_t finSeqCounter = 0;
    // the one-launch candidate selection delivers the candidates into coherent pinned memory (the host reads them as soon as the
    // kernel's sequence number arrives, before the stream reports completion)
    int64_t* hCandRows = nullptr;        // [candidate capacity][groupRowWords]
    int64_t* dHostCandRows = nullptr;    // ... as the device sees it
    size_t hCandRowsWords = 0;
    const int64_t* hRowsView = nullptr;  // the tail reads group rows from here when set (else hGroupRows)
    // An execution that ended on its candidates enqueues the NEXT execution's clears (error word, counters, join tables, the
    // aggregates beside their entries) behind its last kernel, where they run while the host does its tail; the next execution of
    // this query skips its prologue fill if it is the very next execution on the context and wants exactly those clears.
    bool readied = false;
    uint64_t readiedEpoch = 0;
    std::vector<FillItem> readiedFill;
    bool scanChainedOff = false;         // a look-back of the one-launch offset scan timed out once: three launches from now on
    int64_t* hInlineRows = nullptr;      // a handful of group rows delivered with the status words (publishStatusAsync): host-mapped, and the device's view
    int64_t* dHostInlineRows = nullptr;
    size_t inlineRowsWords = 0;
    int64_t* dNarrowRows = nullptr;      // group rows as [slot | sort key] for the candidate selection over wide rows (executeQuery)
    uint32_t narrowRowsCap = 0;
    bool narrowRowsOff = false;          // an execution needed every group row after all: full rows from now on
    bool fusedSelectOff = false;         // a meeting point of the one-launch candidate selection timed out once: separate launches from now on
    uint64_t mergePublishedSeq = 0;        // > 0: rsq_query_merge_gathered also published the merged table to hPinned; finalize polls for this number
    bool kernelTimePending = false;        // the fused step's events have not been read yet (resolveKernelTime)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evRing; size_t evHead = 0, evTail = 0;      // event pairs of the one-launch steps not read yet
    hipEvent_t gev0 = nullptr, gev1 = nullptr;   // start / stop of the fused step's kernel (hipExtModuleLaunchKernel)
    bool pendingFused = false;             // the enqueued asynchronous step was a fused one
    // generic pipeline in front of the specialised kernel (see GenericProgram)
    bool genericActive = false, genericForced = false;
    GenericProgram generic;
    GenericInstr* dGenericCode = nullptr;
    // ... or the interpreter for whole pipelines, one program per pipeline (joins, hash aggregation, materialisation)
    bool generic2 = false;
    std::vector<GenericProgram2> generic2Progs;
    std::vector<bool> savedAos;            // the join tables' own layout flags while the interpreter's (words[slot][w]) are in force
    uint32_t* dG2Cnt = nullptr; uint64_t* dG2Offs = nullptr; void* dG2ScanTemp = nullptr; int64_t g2CntRows = 0;
    std::thread bgCompiler;                // builds the specialised kernels into the code-object cache
    std::atomic<int> bgState{0};           // 0 none, 1 running, 2 the quick tier is in the cache (the full one still compiling), 3 failed, 4 done
    bool quickTier = false;                // the pipelines run on the quick tier's kernels (stage 2 called, not inlined)
    std::string bgError;
    bool pendingAsync = false;             // rsq_query_execute_partial_async enqueued a step; finalize accounts for it
    bool chainedIndexOff = false;          // the one-launch rank index timed out once on this query: two launches from then on
    bool firstRowsForeign = false;         // the partial table may hold first rows of shards whose size this query's table does not know (bound / gathered partials, table never unified)
    bool holdTail = false;                 // a shard of a multi-GPU plan: execute reads the group rows / materialised columns back and stops (tail.cpp runTailMerged)
    std::string allSource, explainText;
    std::string memoKey;                   // the context's plan memo entry of this query (empty: none)
    bool memoApplied = false;              // ... and an earlier query's entry was found when this one was compiled

    explicit Query(Context& c) : ctx(c) {}
    ~Query();
};

bool buildGenericProgram(Query& q, GenericProgram& out, std::string& why);
void launchGenericAggregate(Context& ctx, const GenericProgram& prog, const GenericInstr* dCode, int64_t nRows, int64_t row0, uint64_t* dTable,
                            int64_t denseGroups, int64_t tableWords);

uint64_t opSize(OpNode* o, bool local = false);      // getSize() estimates of the reference's operators

inline double nowMs() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// codegen.cpp: turns the operator tree below the last pipeline breaker into device pipelines
void buildPipelines(Query& q);

// tail.cpp: aggregate table / group rows -> result relation (AVG, projection, materialize, order by, limit)
void runTail(Query& q);
// tail.cpp: the tail's buffers that live as long as the query (see TailState)
std::vector<uint32_t>& tailOrderBuffer(Query& q);
ReplayScratch& tailReplayScratch(Query& q);
// tail.cpp: can the rows of this dense aggregation be produced on the device (devtail.hip)?  fills the kernels' descriptions
bool planDenseDeviceTail(Query& q, DenseTailKeys& keys, DenseTailCols& cols, int& tupleSize, int64_t& limitRows);
// tail.cpp: the same for the group rows of a hash / join-entry aggregation; `sorts`: an ORDER BY follows (the host sorts the delivered tuples)
bool planRowsDeviceTail(Query& q, RowTailKeys& keys, RowTailCols& cols, int& tupleSize, int64_t& limitRows, bool& sorts);
void runRowsTailSort(Query& q, uint8_t* tuples, int64_t& rows);
// tail.cpp: is the first ORDER BY key of an `ORDER BY ... LIMIT k` above the aggregation one word of the group rows?
void planDeviceTopK(Query& q);

// ---- engine_pipelines.cpp: one pipeline on the device, by sink (called from executeQuery) --------------------------------------
inline int64_t nextPow2(int64_t v) { int64_t p = 1; while (p < v) p <<= 1; return p; }
bool denseMode(const Query& q);
void allocMatCols(Query& q, int64_t capacity);
void freeMatCols(Query& q);
uint64_t argValue(Query& q, const Pipeline& p, const ArgSlot& a, int countOnlyTable);
int residentWorkgroupsPerCU(Kernel* k, int blockThreads);
unsigned pipelineGrid(const Query& q, const Pipeline& p, bool lazyForm = false);
void launchPipelineKernel(Query& q, Pipeline& p, Kernel& k, int countOnlyTable, unsigned grid = 0, unsigned block = 0,
                                 hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
void waitForStream(Context& ctx);
void debugStamps(Query& q, Pipeline& p);
Kernel* fewGroupsKernel(Query& q, Pipeline& p, const std::string& source, const char* form, Kernel* large);
unsigned fewGroupsGrid(Query& q, Pipeline& p, Kernel* k);
void launchPipeline(Query& q, Pipeline& p, int countOnlyTable, bool pass1 = false);
void prepareStageBuffers(Query& q, const Pipeline& p);
bool runStagedAggregation(Query& q, Pipeline& p, const std::vector<uint64_t>& estimate, bool tentative = false);
void runLargeDenseAggregation(Query& q, Pipeline& p);
void materializePipeline(Query& q, Pipeline& p);
std::string tierSource(const Pipeline& p, const std::string& source, bool quick);
void buildHashTable(Query& q, Pipeline& p);
void sizeJoinTable(Query& q, Pipeline& p, HashTable& h, uint32_t n, bool dupKeys);
void checkDeviceError(uint32_t err);
void checkAsyncDeviceError(uint32_t err);
void enqueueTableInit(Query& q);
void enqueueTableReadback(Query& q);
void tableFromPinned(Query& q);
void dropTable(Context& ctx, HashTable& h);
void leaveGeneric2(Query& q);
void generic2Launch(Query& q, size_t pi, int matPass);
void runGeneric2Pipeline(Query& q, size_t pi);
// ---- engine_devtail.cpp: the tails that stay on the device ----------------------------------------------------------------------
bool denseDeviceTailWanted(Query& q);
double runDenseDeviceTail(Query& q);
bool rowsDeviceTailWanted(Query& q, int64_t n);
double runRowsDeviceTail(Query& q, int64_t n);

}  // namespace rsq
