// engine.cpp — describe -> compile -> execute -> retrieve for one query
// (the call sequence of the reference's executeSelectPlan, reference src/execute.h:213-247).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <sstream>

#include "engine_internal.h"

namespace rsq {

namespace {

// multiplicative inverse of an odd number modulo 2^64 (Newton: every step doubles the correct low bits)
uint64_t inverseOdd64(uint64_t a) { uint64_t x = a; for (int i = 0; i < 6; i++) x *= 2 - a * x; return x; }

// ================================================================================================
// plan construction + typing
// ================================================================================================
void buildOps(Query& q, const rsq_plan_desc& p) {
    int extra = 0;
    for (int i = 0; i < p.n_ops; i++) if (p.ops[i].tag == RSQ_OP_ORDERBY) extra++;
    for (int i = 0; i < p.n_ops + extra; i++) q.ops.emplace_back(new OpNode());
    int nextExtra = p.n_ops;
    for (int i = 0; i < p.n_ops; i++) {
        const rsq_op& d = p.ops[i];
        OpNode* o = q.ops[(size_t)i].get();
        o->tag = d.tag;
        if (d.n_exprs < 0 || d.n_exprs > RSQ_MAX_OP_EXPRS || d.n_exprs2 < 0 || d.n_exprs2 > RSQ_MAX_OP_EXPRS) failInvalid("bad expression count");
        for (int k = 0; k < d.n_exprs; k++) { if (d.exprs[k] < 0 || d.exprs[k] >= p.n_exprs) failInvalid("bad expression index"); o->exprs.push_back(q.exprs[(size_t)d.exprs[k]]); }
        for (int k = 0; k < d.n_exprs2; k++) { if (d.exprs2[k] < 0 || d.exprs2[k] >= p.n_exprs) failInvalid("bad expression index"); o->exprs2.push_back(q.exprs[(size_t)d.exprs2[k]]); }
        o->singleMatch = d.single_match != 0;
        switch (d.tag) {
            case RSQ_OP_SCAN:
                if (d.table < 0 || d.table >= (int)q.tables.size()) failInvalid("bad table index");
                o->table = q.tables[(size_t)d.table]; o->nChildren = 0; break;
            case RSQ_OP_HASHJOIN: o->nChildren = 2; if (d.n_exprs < 1) failInvalid("hash join needs an equality"); break;
            case RSQ_OP_SELECTION: if (d.n_exprs != 1) failInvalid("selection needs one condition"); o->nChildren = 1; break;
            case RSQ_OP_PROJECTION: case RSQ_OP_AGGREGATION: case RSQ_OP_MATERIALIZE: case RSQ_OP_ORDERBY: o->nChildren = 1; break;
            case RSQ_OP_NESTEDLOOPSJOIN: failUnsupported("NestedLoopsJoin is outside the hot path (SURVEY §2)");
            default: failInvalid("unknown operator tag");
        }
        for (int k = 0; k < o->nChildren; k++) {
            int ci = d.child[k];
            if (ci < 0 || ci >= p.n_ops || ci == i) failInvalid("bad child operator index");
            o->child[k] = q.ops[(size_t)ci].get();
        }
        if (d.tag == RSQ_OP_ORDERBY) {   // OrderByOp wraps its child into a MaterializeOp (orderby.h:32-38)
            OpNode* m = q.ops[(size_t)nextExtra++].get();
            m->tag = RSQ_OP_MATERIALIZE; m->nChildren = 1; m->child[0] = o->child[0];
            o->child[0] = m;
        }
    }
    for (auto& o : q.ops)
        for (int k = 0; k < o->nChildren; k++) {
            if (o->child[k]->parent && o->child[k]->parent != o.get()) failInvalid("operator has two parents");
            o->child[k]->parent = o.get();
        }
    if (p.root < 0 || p.root >= p.n_ops) failInvalid("bad root");
    q.root = q.ops[(size_t)p.root].get();
    if (q.root->tag != RSQ_OP_MATERIALIZE && q.root->tag != RSQ_OP_ORDERBY)
        failInvalid("Calling retrieveResult on non-materialized operator");
    if (p.has_limit) { q.root->hasLimit = true; q.root->limit = p.limit; }
}

// defineExpressionsForPlan (RelOperator.h:203-208) + deriveExpressionTypes in definition order
void defineAndDerive(Query& q, OpNode* o) {
    for (int i = 0; i < o->nChildren; i++) defineAndDerive(q, o->child[i]);
    auto all = [&](std::vector<Expr*>& v) { for (Expr* e : v) q.pool.derive(e); };
    switch (o->tag) {
        case RSQ_OP_SCAN: case RSQ_OP_MATERIALIZE: break;
        case RSQ_OP_SELECTION: case RSQ_OP_PROJECTION: case RSQ_OP_HASHJOIN: all(o->exprs); break;
        case RSQ_OP_AGGREGATION:
            for (Expr* e : o->exprs) {
                if (e->tag == RSQ_E_AVG) {
                    o->splitAgg.push_back(q.pool.unary(RSQ_E_SUM, "sum", e->child));
                    o->splitAgg.push_back(q.pool.unary(RSQ_E_COUNT, "count", e->child));
                } else o->splitAgg.push_back(e);
            }
            all(o->exprs2); all(o->splitAgg); all(o->exprs);
            break;
        case RSQ_OP_ORDERBY:
            for (auto& e : o->exprs) if (e->tag != RSQ_E_ASC && e->tag != RSQ_E_DESC) e = q.pool.unary(RSQ_E_ASC, "asc", e);
            for (Expr* e : o->exprs)
                if (e->child->tag != RSQ_E_ATTRIBUTE) failType("Order by only supports attribute expressions currently." + serializeExpr(e->child));
            all(o->exprs);
            break;
        default: failInvalid("unsupported operator in plan");
    }
}

}  // namespace

// getSize() estimates (reference src/operators/*.h).  local: over this shard's own rows (sizing of the device's tables) instead of the
// whole table's (everything that replays the reference: its table sizes decide the emission order)
uint64_t opSize(OpNode* o, bool local) {
    switch (o->tag) {
        case RSQ_OP_SCAN: return (uint64_t)(local ? o->table->nRows : o->table->totalRows());      // a shard sizes like the table it is a range of (Table::nRowsTotal)
        case RSQ_OP_SELECTION: return opSize(o->child[0], local) / 2;
        case RSQ_OP_PROJECTION: case RSQ_OP_ORDERBY: return opSize(o->child[0], local);
        case RSQ_OP_HASHJOIN: return opSize(o->child[0], local) + opSize(o->child[1], local) / 2;
        case RSQ_OP_AGGREGATION: {
            if (o->exprs2.empty()) return 1;
            int red = 512;
            for (size_t i = 1; i < o->exprs2.size() && red > 2; i++) red /= 2;
            return opSize(o->child[0], local) / (uint64_t)red;
        }
        case RSQ_OP_MATERIALIZE: { uint64_t s = opSize(o->child[0], local); if (o->hasLimit && (uint64_t)o->limit < s) s = (uint64_t)o->limit; return s; }
        default: return 0;
    }
}

Query::~Query() {
    if (bgCompiler.joinable()) bgCompiler.join();
    destroyTailState(tailState);
    if (dtArena.dev || dtArena.pinned) {
        // the arenas go back to the context for the next query, unless it already holds a pair
        if (!ctx.spareTailArena.dev && !ctx.spareTailArena.pinned) ctx.spareTailArena = dtArena;
        else { if (dtArena.dev) ctx.free(dtArena.dev); if (dtArena.pinned) ctx.freePinned(dtArena.pinned); }
    }
    if (rtDev) ctx.free(rtDev);
    if (rtPinned) ctx.freePinned(rtPinned);
    if (dGenericCode) ctx.free(dGenericCode);
    for (auto& gp : generic2Progs) { if (gp.dCode) ctx.free(gp.dCode); if (gp.dProbes) ctx.free(gp.dProbes); if (gp.dConstPool) ctx.free(gp.dConstPool); }
    if (dG2Cnt) ctx.free(dG2Cnt);
    if (dG2Offs) ctx.free(dG2Offs);
    if (dG2ScanTemp) ctx.free(dG2ScanTemp);
    if (dAgg && dAggOwned) ctx.free(dAgg);
    if (dAggInit) ctx.free(dAggInit);
    if (dAggWork) ctx.free(dAggWork);
    if (dAggWorkInit) ctx.free(dAggWorkInit);
    if (hPinned) ctx.freePinned(hPinned);
    if (hGroupRows) ctx.freePinned(hGroupRows);
    if (hCandRows) ctx.freePinned(hCandRows);
    if (dFinTicket) ctx.free(dFinTicket);
    // (events not read yet belong to work that may still run: they go back to the pool all the same - a record on a recorded event
    // simply replaces it, and nobody reads the old pair any more)
    ctx.giveEvent(gev0); ctx.giveEvent(gev1);
    for (auto& e : evRing) { ctx.giveEvent(e.first); ctx.giveEvent(e.second); }
    if (dMatCnt) ctx.free(dMatCnt);
    if (dMatOffs) ctx.free(dMatOffs);
    if (dMatTileCnt) ctx.free(dMatTileCnt);
    if (dScanTemp) ctx.free(dScanTemp);
    freeMatCols(*this);
    if (dGroupRows) ctx.free(dGroupRows);
    if (dNarrowRows) ctx.free(dNarrowRows);
    if (hInlineRows) ctx.freePinned(hInlineRows);
    if (dPipeStats) ctx.free(dPipeStats);
    if (dGroupCount) ctx.free(dGroupCount);
    if (dTopkImages) ctx.free(dTopkImages);
    if (dTopkHists) ctx.free(dTopkHists);
    if (dCandRows) ctx.free(dCandRows);
    if (dPartCounts) ctx.free(dPartCounts);
    if (dPartStart) ctx.free(dPartStart);
    if (dDebugStamps) ctx.free(dDebugStamps);
    if (dStageBase) ctx.free(dStageBase);
    if (dStageCap) ctx.free(dStageCap);
    if (dStageCtl) ctx.free(dStageCtl);
    if (hStageLayout) ctx.freePinned(hStageLayout);
    if (dStageCounts) ctx.free(dStageCounts);
    if (dPartTotals) ctx.free(dPartTotals);
    for (void* r : dPartRecords) if (r) ctx.scratchFree(r);
    for (auto& h : hashTables) {
        if (h->dState) ctx.free(h->dState);
        if (h->dWords) ctx.free(h->dWords);
        if (h->dAcc) ctx.free(h->dAcc);
        if (h->dCount) ctx.free(h->dCount);
        if (h->keyIndex) ctx.releaseKeyIndex(h->keyIndex);      // (the bitmap is the context's key index then)
        else if (h->dBitmap) ctx.free(h->dBitmap);
        if (h->dCompBitmap) ctx.free(h->dCompBitmap);
        if (h->dDeref) ctx.free(h->dDeref);
        if (h->dTemp) ctx.free(h->dTemp);
        if (h->dTempUsed) ctx.free(h->dTempUsed);
        if (h->dChunkTotal) ctx.free(h->dChunkTotal);
        if (h->dChunkBase) ctx.free(h->dChunkBase);
    }
}

// ================================================================================================
// compile
// ================================================================================================
bool denseMode(const Query& q) {
    return q.aggMode == AggMode::DENSE_REG || q.aggMode == AggMode::DENSE_LDS_PRIVATE ||
           q.aggMode == AggMode::DENSE_LDS_SHARED || q.aggMode == AggMode::DENSE_GLOBAL;
}

// Device-resident identity image of the dense aggregate table (0 for sums, +/-inf for min/max) and a pinned
// host buffer for the read-back: one execute is then {D2D init, kernel(s), D2H} on one stream with a single
// host synchronisation at the end.
static void prepareDenseBuffers(Query& q) {
    const int64_t D = q.denseGroups;
    const size_t words = q.accums.size() * (size_t)D;
    std::vector<uint64_t> init(words);
    for (size_t w = 0; w < q.accums.size(); w++) {
        uint64_t idv = q.accums[w].merge == 0 ? 0ull : q.accums[w].merge == 2 ? 0x7fffffffffffffffull : 0x8000000000000000ull;
        for (int64_t g = 0; g < D; g++) init[(size_t)(q.accumSlot[w] * D + g)] = idv;
    }
    q.dAgg = (uint64_t*)q.ctx.alloc(words * 8);
    q.dAggInit = (uint64_t*)q.ctx.alloc(words * 8);
    RSQ_HIP(hipMemcpy(q.dAggInit, init.data(), words * 8, hipMemcpyHostToDevice));
    q.hAgg.assign(words, 0);
    q.pinnedWords = words;
    q.tableWords = words;
    if (q.aggPad > 1) {
        // padded working table (one cell per 64-byte line) the register-mode kernels flush into
        q.padWords = words * (size_t)q.aggPad;
        std::vector<uint64_t> pinit(q.padWords + 1, 0);        // + one word: the device error word of graph replays
        for (size_t i = 0; i < words; i++) pinit[i * (size_t)q.aggPad] = init[i];
        q.dAggWork = (uint64_t*)q.ctx.alloc((q.padWords + 1) * 8);
        q.dAggWorkInit = (uint64_t*)q.ctx.alloc((q.padWords + 1) * 8);
        RSQ_HIP(hipMemcpy(q.dAggWorkInit, pinit.data(), (q.padWords + 1) * 8, hipMemcpyHostToDevice));
        q.pinnedWords = q.padWords;          // the read-back of a full execution takes the padded table
    }
}

// every kernel source a query's execution may need right away (lazily chosen forms are compiled when first chosen).
// quick: the tier a statement with a cold code-object cache runs on first - the pipelines that have a stage 2 with RSQ_STAGE2_CALL
// (stage 2 a real call instead of inlined at every drain site: a third of the compile time, 40-65 % slower to run; codegen_loop.cpp)
std::string tierSource(const Pipeline& p, const std::string& source, bool quick) {
    return quick && p.hasStage2 ? "#define RSQ_STAGE2_CALL 1\n" + source : source;
}
static std::vector<std::string> kernelSources(const Query& q, bool quick = false) {
    std::vector<std::string> v;
    for (auto& p : q.pipelines) {
        auto T = [&](const std::string& s) { return tierSource(p, s, quick); };
        if (!p.sourcePass1.empty()) v.push_back(T(p.sourcePass1));
        if (!p.sourceFlat.empty()) v.push_back(T(p.sourceFlat));
        if (p.partitioned) { v.push_back(T(p.sourcePartCount)); v.push_back(T(p.sourcePartScatter)); v.push_back(p.sourcePartAgg); }
        if (p.staged) { v.push_back(T(p.sourceStagedScatter)); v.push_back(p.sourceStagedAgg); }
        v.push_back(T(p.source));
        if (!p.sourceLazy.empty()) v.push_back(T(p.sourceLazy));      // (chosen at run time; compiled with the others so that choosing it never blocks an execution)
        // ... and so is the 64-slot front table of a hash aggregation that turns out to have a handful of groups (engine_pipelines.cpp
        // fewGroupsKernel): compiled when first wanted it cost TPC-H Q12's second query 0.57 s of hiprtc in the middle of an execution
        if (p.ldsSlots > 64 && q.aggMode == AggMode::HASH && p.sink == SinkKind::AGGREGATE && q.aggTable >= 0) {
            v.push_back(T("#define RSQ_LC_SLOTS 64\n" + p.source));
            if (!p.sourceLazy.empty()) v.push_back(T("#define RSQ_LC_SLOTS 64\n" + p.sourceLazy));
        }
    }
    return v;
}
static void resolveKernels(Query& q, bool quick = false) {
    Context& ctx = q.ctx;
    // (compiling a query's kernels on one host thread each was tried: hiprtc serialises internally, TPC-H Q3's three kernels took
    // 548 ms in parallel against 448 ms one after the other.  What is not in the cache is therefore compiled in helper PROCESSES
    // first, all at once - runtime.cpp compileManyToCache - and the loop below finds it there)
    ctx.compileManyToCache(kernelSources(q, quick));
    q.quickTier = quick;
    for (auto& p : q.pipelines) {
        auto T = [&](const std::string& s) { return tierSource(p, s, quick); };
        p.fewGroupKernels.clear();                // (kernels of the other tier)
        if (!p.sourcePass1.empty()) p.kernelPass1 = &ctx.getKernel(T(p.sourcePass1), p.entry);
        if (!p.sourceFlat.empty()) p.kernelFlat = &ctx.getKernel(T(p.sourceFlat), p.entry);
        if (p.partitioned) {
            p.kernelPartCount = &ctx.getKernel(T(p.sourcePartCount), p.entry);
            p.kernelPartScatter = &ctx.getKernel(T(p.sourcePartScatter), p.entry);
            p.kernelPartAgg = &ctx.getKernel(p.sourcePartAgg, "rsq_part_agg");
        }
        if (p.staged) {
            p.kernelStagedScatter = &ctx.getKernel(T(p.sourceStagedScatter), p.entry);
            p.kernelStagedAgg = &ctx.getKernel(p.sourceStagedAgg, "rsq_staged_agg");
        }
        p.kernel = &ctx.getKernel(T(p.source), p.entry);
        // both forms are loaded now: the first execution that chooses the late-load form must not read and load a code object
        // (TPC-H Q3 at SF10: 0.86 ms for the execution that first chose it, 0.32 ms after)
        if (!p.sourceLazy.empty()) p.kernelLazy = &ctx.getKernel(T(p.sourceLazy), p.entry);
        if (ctx.device < 0 && p.ldsSlots > 64 && q.aggMode == AggMode::HASH && p.sink == SinkKind::AGGREGATE && q.aggTable >= 0) {      // build(): the few-groups forms too
            (void)ctx.getKernel("#define RSQ_LC_SLOTS 64\n" + p.source, p.entry);
            if (!p.sourceLazy.empty()) (void)ctx.getKernel("#define RSQ_LC_SLOTS 64\n" + p.sourceLazy, p.entry);
        }
    }
}


// ================================================================================================
// the context's plan memo (engine.h Context::PlanMemo)
// ================================================================================================
static uint64_t fnv1a64(const std::string& s, uint64_t h = 1469598103934665603ull) {
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}
// (kernel texts - they carry every constant of the plan and everything the planner took from the column statistics -, then identity
// and version of every table in plan order)
static std::string planMemoKey(const Query& q) {
    uint64_t h = 1469598103934665603ull;
    for (auto& p : q.pipelines) h = fnv1a64(p.source, h) * 0x9E3779B97F4A7C15ull;
    std::string k = std::to_string(h);
    for (Table* t : q.tables) k += "|" + std::to_string(t->uid) + "." + std::to_string(t->version) + "." + std::to_string(t->nRows) + "." + std::to_string(t->row0) + "." + std::to_string(t->nRowsTotal);
    return k;
}

// a freshly compiled query takes what an earlier one with the same key has learnt: its join tables are sized and allocated (no counting
// pass, no read-back), its hash aggregation starts with the table that held the groups, late-load / staged / materialisation choices and
// the forms that timed out once are the earlier query's.  Runs at compile time, where the arenas make the allocations cheap.
static void applyPlanMemo(Query& q) {
    Context& ctx = q.ctx;
    if (ctx.planMemoOff || ctx.device < 0 || q.genericActive) return;
    q.memoKey = planMemoKey(q);
    auto it = ctx.planMemo.find(q.memoKey);
    if (it == ctx.planMemo.end()) return;
    Context::PlanMemo& m = it->second;
    if (m.joins.size() != q.hashTables.size() || m.pipes.size() != q.pipelines.size()) return;
    m.stamp = ++ctx.planMemoClock;
    for (size_t i = 0; i < q.pipelines.size(); i++) {
        Pipeline& p = q.pipelines[i];
        p.stage2Rows = m.pipes[i].stage2Rows;
        p.stagedExact = m.pipes[i].stagedExact; p.stagedCaps = m.pipes[i].stagedCaps; p.stagedCapsRows = m.pipes[i].stagedCapsRows;
        if (p.sink != SinkKind::BUILD) continue;
        HashTable& h = *q.hashTables[(size_t)p.buildTable];
        const Context::PlanMemo::Join& j = m.joins[(size_t)p.buildTable];
        if (j.buildRows < 0 || h.capacity != 0) continue;
        sizeJoinTable(q, p, h, (uint32_t)j.buildRows, j.dupKeys);
    }
    for (size_t i = 0; i < q.hashTables.size(); i++) q.hashTables[i]->lastCount = m.joins[i].lastCount;
    if (q.aggMode == AggMode::HASH && q.aggTable >= 0 && m.hashCapacity > 0) {
        HashTable& h = *q.hashTables[(size_t)q.aggTable];
        if (h.capacity == 0 && !h.dState) {
            h.capacity = m.hashCapacity; h.lastCount = m.hashCount;
            h.dState = (uint32_t*)ctx.alloc((size_t)h.capacity * 4);
            h.dWords = (int64_t*)ctx.alloc((size_t)h.capacity * 8 * (h.keys.size() + h.payload.size()));
            h.dAcc = (int64_t*)ctx.alloc((size_t)h.capacity * 8 * (size_t)h.nAccBlocks);
        }
        q.charGroupsNeedMerge = m.charGroupsNeedMerge;
        for (auto& p : q.pipelines)         // a handful of groups: the 64-slot front table's kernel is loaded now (engine_pipelines.cpp fewGroupsKernel)
            if (p.sink == SinkKind::AGGREGATE) {
                const bool lazy = !p.sourceLazy.empty() && p.stage2Rows >= 0 && p.stage2Rows * 32 < p.src->nRows;
                (void)fewGroupsKernel(q, p, lazy ? p.sourceLazy : p.source, lazy ? "lazy" : "eager", lazy ? p.kernelLazy : p.kernel);
            }
    }
    q.matLastTotal = m.matLastTotal;
    if (q.matOp && !q.agg && m.matLastTotal >= 0) {
        uint64_t keep = (uint64_t)m.matLastTotal;
        if (q.matOp->hasLimit) keep = std::min<uint64_t>(keep, (uint64_t)std::max<int64_t>(q.matOp->limit, 1));
        if (q.dMatCols.empty()) allocMatCols(q, std::max<int64_t>((int64_t)keep, 1));
    }
    q.narrowRowsOff = m.narrowRowsOff; q.fusedSelectOff = m.fusedSelectOff; q.chainedIndexOff = m.chainedIndexOff; q.scanChainedOff = m.scanChainedOff;
    q.stageWorkgroups = m.stageWorkgroups;
    q.memoApplied = true;
    ctx.planMemoHits++;
}

// after an execution that ran to its end: the key bitmaps of its direct join tables stay as they are for the next execution, and go to the
// context for every other query over the same table version (Context::keyIndexes)
static void keepKeyIndexes(Query& q) {
    Context& ctx = q.ctx;
    if (q.genericActive) return;
    for (auto& p : q.pipelines) {
        if (p.sink != SinkKind::BUILD) continue;
        HashTable& h = *q.hashTables[(size_t)p.buildTable];
        if (!h.direct || h.dense || !h.rank || !h.dBitmap || h.directKeyCol < 0 || h.directSrc != p.src) continue;
        if (!h.keyIndexReady) { h.keyIndexReady = true; h.keyIndexVersion = p.src->version; }
        if (h.keyIndex || ctx.planMemoOff || h.keyIndexVersion != p.src->version) continue;
        const Context::KeyIndexKey key{p.src->uid, p.src->version, p.src->nRows, p.src->row0, h.directKeyCol};
        if (ctx.keyIndexes.count(key)) continue;      // (another query was first: this one keeps its own)
        Context::KeyIndex& k = ctx.keyIndexes[key];
        k.dBitmap = h.dBitmap; k.bmBlocks = h.bmBlocks; k.bmMin = h.bmMin; k.bmBits = h.bmBits; k.refs = 1; k.uid = p.src->uid;
        h.keyIndex = &k;
    }
}

// after an execution that ran to its end: what it knows now
static void rememberPlan(Query& q) {
    Context& ctx = q.ctx;
    if (ctx.planMemoOff || q.memoKey.empty() || q.genericActive) return;
    Context::PlanMemo& m = ctx.planMemo[q.memoKey];
    m.stamp = ++ctx.planMemoClock;
    m.joins.resize(q.hashTables.size()); m.pipes.resize(q.pipelines.size());
    for (size_t i = 0; i < q.hashTables.size(); i++) {
        const HashTable& h = *q.hashTables[i];
        m.joins[i].buildRows = h.capacity != 0 ? h.buildRows : -1; m.joins[i].dupKeys = h.dupKeys; m.joins[i].lastCount = h.lastCount;
    }
    for (size_t i = 0; i < q.pipelines.size(); i++) {
        const Pipeline& p = q.pipelines[i];
        Context::PlanMemo::Pipe& mp = m.pipes[i];
        mp.stage2Rows = p.stage2Rows; mp.stagedExact = p.stagedExact; mp.stagedCapsRows = p.stagedCapsRows;
        if (mp.stagedCaps != p.stagedCaps) mp.stagedCaps = p.stagedCaps;
    }
    if (q.aggMode == AggMode::HASH && q.aggTable >= 0) {
        const HashTable& h = *q.hashTables[(size_t)q.aggTable];
        m.hashCapacity = h.dState ? h.capacity : 0; m.hashCount = h.lastCount; m.charGroupsNeedMerge = q.charGroupsNeedMerge;
    }
    m.matLastTotal = q.matLastTotal;
    m.narrowRowsOff = q.narrowRowsOff; m.fusedSelectOff = q.fusedSelectOff; m.chainedIndexOff = q.chainedIndexOff; m.scanChainedOff = q.scanChainedOff;
    m.stageWorkgroups = q.stageWorkgroups;
    if (ctx.planMemo.size() > 512) {          // the least recently used half goes
        std::vector<std::pair<uint64_t, std::string>> byAge;
        for (auto& kv : ctx.planMemo) byAge.emplace_back(kv.second.stamp, kv.first);
        std::sort(byAge.begin(), byAge.end());
        for (size_t i = 0; i < byAge.size() / 2; i++) ctx.planMemo.erase(byAge[i].second);
    }
}

Query* compileQuery(Context& ctx, const rsq_plan_desc& plan, rsq_table* const* tables, int nTables) {
    double t0 = nowMs();
    std::unique_ptr<Query> q(new Query(ctx));
    int hits0 = ctx.jitCacheHits, comp0 = ctx.jitCompiles;
    for (int i = 0; i < nTables; i++) {
        Table* t = reinterpret_cast<Table*>(tables[i]);
        if (!t) failInvalid("null table");
        q->tables.push_back(t);
        q->tableLayouts.push_back(t->layoutVersion);
        for (auto& c : t->cols) q->pool.identTypes[c.name] = c.type;
    }
    q->requestAll = plan.request_all != 0;
    q->exprs = q->pool.build(plan);
    buildOps(*q, plan);
    defineAndDerive(*q, q->root);
    const double tBuilt0 = nowMs();
    buildPipelines(*q);
    const double tBuilt = nowMs();
    // A plan shape whose specialised kernels are not in the code-object cache starts on the pre-compiled generic pipeline
    // (generic.cpp) while hiprtc builds them on a host thread; RSQ_FORCE_GENERIC=1 keeps every eligible plan there (tests),
    // RSQ_GENERIC=0 restores the blocking compile.
    const bool forceGeneric = getenv("RSQ_FORCE_GENERIC") && atoi(getenv("RSQ_FORCE_GENERIC")) != 0;
    const bool allowGeneric = ctx.device >= 0 && !(getenv("RSQ_GENERIC") && atoi(getenv("RSQ_GENERIC")) == 0);
    bool cached = true;
    if (allowGeneric) for (const std::string& src : kernelSources(*q)) cached = cached && ctx.kernelCached(src);
    if (getenv("RSQ_TRACE")) fprintf(stderr, "[rsq trace] compile: code-object cache looked up at +%.3f ms\n", nowMs() - tBuilt);
    if (allowGeneric && (forceGeneric || !cached)) {
        std::string why, why2;
        bool ok = buildGenericProgram(*q, q->generic, why);
        if (ok) {
            q->dGenericCode = (GenericInstr*)ctx.alloc(std::max<size_t>(1, q->generic.code.size()) * sizeof(GenericInstr));
            RSQ_HIP(hipMemcpy(q->dGenericCode, q->generic.code.data(), q->generic.code.size() * sizeof(GenericInstr), hipMemcpyHostToDevice));
            q->explainText += "generic pre-compiled pipeline (" + std::to_string(q->generic.code.size()) + " instructions, " + std::to_string(q->generic.cols.size()) +
                              " columns)" + (forceGeneric ? " forced" : " until hiprtc has built the specialised kernel") + "\n";
        } else if (!(getenv("RSQ_GENERIC2") && atoi(getenv("RSQ_GENERIC2")) == 0) && buildGenericPlan(*q, q->generic2Progs, why2)) {
            // joins, strings, hash aggregation, materialisation: the interpreter for whole pipelines, one program per pipeline
            ok = true; q->generic2 = true;
            size_t instr = 0;
            if (getenv("RSQ_TRACE")) fprintf(stderr, "[rsq trace] compile: interpreter programs built at +%.3f ms\n", nowMs() - tBuilt);
            for (auto& gp : q->generic2Progs) {
                instr += gp.code.size();
                gp.dCode = (GenericInstr*)ctx.alloc(std::max<size_t>(1, gp.code.size()) * sizeof(GenericInstr));
                RSQ_HIP(hipMemcpy(gp.dCode, gp.code.data(), gp.code.size() * sizeof(GenericInstr), hipMemcpyHostToDevice));
                gp.dProbes = (GenericProbeDesc*)ctx.alloc(std::max<size_t>(1, gp.probes.size()) * sizeof(GenericProbeDesc));
                if (!gp.probes.empty()) RSQ_HIP(hipMemcpy(gp.dProbes, gp.probes.data(), gp.probes.size() * sizeof(GenericProbeDesc), hipMemcpyHostToDevice));
                gp.dConstPool = (char*)ctx.alloc(std::max<size_t>(8, gp.constPool.size()));
                if (!gp.constPool.empty()) RSQ_HIP(hipMemcpy(gp.dConstPool, gp.constPool.data(), gp.constPool.size(), hipMemcpyHostToDevice));
            }
            for (auto& h : q->hashTables) q->savedAos.push_back(h->aos);
            if (getenv("RSQ_TRACE")) fprintf(stderr, "[rsq trace] compile: interpreter programs on the device at +%.3f ms\n", nowMs() - tBuilt);
            q->explainText += "generic pre-compiled interpreter for " + std::to_string(q->generic2Progs.size()) + " pipeline(s) (" + std::to_string(instr) +
                              " instructions)" + (forceGeneric ? " forced" : " until hiprtc has built the specialised kernels") + "\n";
        } else if (!why2.empty() && getenv("RSQ_TRACE")) fprintf(stderr, "[rsq trace] not interpreted: %s\n", why2.c_str());
        if (ok) {
            q->genericActive = true; q->genericForced = forceGeneric;
            if (!cached) {
                (void)ctx.cacheKey("");            // fills the header text the compiler thread reads
                const std::vector<std::string> sources = kernelSources(*q);
                // ... in two tiers where a pipeline has a stage 2: the quick tier's kernels first (ready in a third of the time: the
                // executions leave the interpreter for them), then the full ones (the executions move on to those when they are ready)
                bool twoTiers = false;
                for (auto& p : q->pipelines) twoTiers = twoTiers || p.hasStage2;
                const std::vector<std::string> quick = twoTiers ? kernelSources(*q, true) : std::vector<std::string>();
                Query* qp = q.get();
                q->bgState = 1;
                q->bgCompiler = std::thread([qp, sources, quick] {
                    // (one helper process per kernel: hiprtc serialises inside a process.  Both tiers' helpers start together - the quick
                    // tier is ready after its own slowest kernel, not after the full tier's)
                    std::string fullError;
                    std::thread full([&] {
                        try { qp->ctx.compileManyToCache(sources); }
                        catch (const std::exception& e) { fullError = e.what(); if (fullError.empty()) fullError = "kernel compilation failed"; }
                    });
                    try {
                        if (!quick.empty()) { qp->ctx.compileManyToCache(quick); qp->bgState = 2; }
                        full.join();
                        if (!fullError.empty()) throw Error(RSQ_ERR_DEVICE, fullError);
                        qp->bgState = 4;
                    }
                    catch (const std::exception& e) { if (full.joinable()) full.join(); qp->bgError = e.what(); qp->bgState = 3; }
                });
            }
        }
    }
    const double tGeneric = nowMs();
    if (!q->genericActive) resolveKernels(*q);
    if (getenv("RSQ_TRACE")) fprintf(stderr, "[rsq trace] compile: typing %.3f ms, pipelines + kernel text %.3f ms, cache lookup + interpreter programs %.3f ms, kernels %.3f ms\n",
                                     tBuilt0 - t0, tBuilt - tBuilt0, tGeneric - tBuilt, nowMs() - tGeneric);
    for (auto& p : q->pipelines) {
        q->allSource += p.source + "\n";
        q->explainText += p.explain + "\n";
    }
    if (denseMode(*q)) {
        // layout of the (partial) aggregate table, for hosts that merge it across GPUs
        std::ostringstream d;
        d << "partial table: groups=" << q->denseGroups << " keys=[";
        for (size_t i = 0; i < q->denseKeys.size(); i++) {
            const DenseKey& k = q->denseKeys[i];
            d << (i ? " x " : "") << k.expr->symbol;
            if (k.byteSet) { d << "{"; for (size_t v = 0; v < k.values.size(); v++) d << (v ? "," : "") << (int)k.values[v]; d << "}"; }
            else d << "[" << k.min << ".." << (k.min + k.card - 1) << "]";
        }
        d << "] blocks=[";
        std::vector<std::string> names(q->accums.size());
        for (size_t w = 0; w < q->accums.size(); w++)
            names[(size_t)q->accumSlot[w]] = std::string(q->accums[w].merge == 0 ? "sum:" : q->accums[w].merge == 2 ? "min:" : "max:") + q->accums[w].key;
        for (size_t b = 0; b < names.size(); b++) d << (b ? " | " : "") << names[b];
        d << "] (word = block * groups + group)";
        q->explainText += d.str() + "\n";
    }
    if (ctx.device >= 0) {
        if (denseMode(*q)) prepareDenseBuffers(*q);
        for (auto& p : q->pipelines) if (p.partitioned) prepareStageBuffers(*q, p);
        for (auto& h : q->hashTables) {
            h->dCount = (uint32_t*)ctx.alloc(sizeof(uint32_t));
            if (!h->derefCodes.empty()) {           // (string group values kept by address: what the row-making kernels rebuild, aot_kernels.hip table_word)
                h->dDeref = (int*)ctx.alloc(h->derefCodes.size() * sizeof(int));
                RSQ_HIP(hipMemcpyAsync(h->dDeref, h->derefCodes.data(), h->derefCodes.size() * sizeof(int), hipMemcpyHostToDevice, ctx.stream));
            }
        }
        if (q->aggMode == AggMode::AT_JOIN_ENTRY || q->aggMode == AggMode::HASH) q->dGroupCount = (uint32_t*)ctx.alloc(sizeof(uint32_t));
        q->dPipeStats = (uint64_t*)ctx.alloc(std::max<size_t>(1, q->pipelines.size()) * 8);
        size_t pw = q->pinnedWords + 8 + q->pipelines.size();
        q->hPinned = (uint64_t*)ctx.allocPinned(pw * 8);
        memset(q->hPinned, 0, pw * 8);
        {   // the device's view of the pinned words: status words are published by one kernel instead of one copy each
            void* dv = nullptr;
            if (hipHostGetDevicePointer(&dv, q->hPinned, 0) == hipSuccess && dv) q->dPinnedDev = (uint64_t*)dv; else (void)hipGetLastError();
        }
        if (q->aggMode == AggMode::DENSE_REG && q->aggPad > 1) {
            void* dv = q->dPinnedDev;
            if (dv) {
                q->dFinHost = (uint64_t*)dv;
                q->dFinTicket = (uint32_t*)ctx.alloc(16);        // [0] the ticket, [2..3] the resident step's doorbell as the workgroups see it
                RSQ_HIP(hipMemset(q->dFinTicket, 0, 16));
            }
        }
    }
    applyPlanMemo(*q);
    q->report.compilation_time_ms = nowMs() - t0;
    q->report.jit_cache_hits = ctx.jitCacheHits - hits0;
    q->report.jit_compiles = ctx.jitCompiles - comp0;
    if (ctx.cfg.print_assembly) fprintf(stderr, "%s\n", q->allSource.c_str());
    if (ctx.cfg.print_flounder) fprintf(stderr, "%s\n", q->explainText.c_str());
    return q.release();
}

// blocks until the query runs on its specialised kernels (joins the compiler thread of a query that started on the generic pipeline)
void awaitKernels(Query& q) {
    if ((!q.genericActive && !q.quickTier) || q.genericForced) return;
    if (q.bgCompiler.joinable()) q.bgCompiler.join();
    if (q.bgState.load() == 3) throw Error(RSQ_ERR_DEVICE, q.bgError);
    const bool wasGeneric = q.genericActive;
    resolveKernels(q);
    q.genericActive = false;
    if (wasGeneric && q.generic2) leaveGeneric2(q);
    if (wasGeneric) applyPlanMemo(q);
    q.explainText += "kernel tier: full (awaited)\n";
    q.report.jit_compiles = (int32_t)kernelSources(q).size();
}

// kernel time of a fused step: its events are read when somebody asks (the report) or before they are recorded again
void resolveKernelTime(Query& q) {
    // the one-launch steps that ran to their end record into a ring of event pairs, read here in one go: reading a pair costs the
    // host 2.4 us, which a benchmark loop would otherwise pay between every two steps
    if (q.evHead < q.evTail) RSQ_HIP(hipSetDevice(q.ctx.device));
    for (; q.evHead < q.evTail; q.evHead++) {
        auto& e = q.evRing[q.evHead % q.evRing.size()];
        RSQ_HIP(hipEventSynchronize(e.second));
        float ems = 0;
        RSQ_HIP(hipEventElapsedTime(&ems, e.first, e.second));
        q.report.kernel_time_ms = ems; q.kernelTimeSumMs += ems; q.kernelTimeLaunches++;
        q.report.hbm_gbps = ems > 0 ? (double)q.report.bytes_read / (ems * 1e-3) / 1e9 : 0;
    }
    if (!q.kernelTimePending || q.pendingAsync) return;
    q.kernelTimePending = false;
    RSQ_HIP(hipSetDevice(q.ctx.device));
    RSQ_HIP(hipEventSynchronize(q.gev1));
    float ms = 0;
    RSQ_HIP(hipEventElapsedTime(&ms, q.gev0, q.gev1));
    q.report.kernel_time_ms = ms; q.kernelTimeSumMs += ms; q.kernelTimeLaunches++;
    q.report.hbm_gbps = ms > 0 ? (double)q.report.bytes_read / (ms * 1e-3) / 1e9 : 0;
}

// The pinned host copy of the group rows, made when rows are first read back and as large as that read-back: a statement that ends on
// a handful of candidates (TPC-H Q10: 380 K groups of 40 words, LIMIT 20) never pins the 121 MB its full rows would take (7 ms of
// hipHostMalloc in the first execution on a fresh context).  Non-coherent: cached on the host (the tail reads every word), valid after
// the copy's synchronisation.
static void ensureHostGroupRows(Query& q, size_t words) {
    if (q.hGroupRowsWords >= words && q.hGroupRows) return;
    if (q.hGroupRows) q.ctx.freePinned(q.hGroupRows);
    q.hGroupRows = nullptr; q.hGroupRowsWords = 0;
    q.hGroupRows = (int64_t*)q.ctx.allocPinned(std::max<size_t>(words, 8) * 8, true);
    q.hGroupRowsWords = words;
}

// a single register-mode pipeline: its kernel carries the whole step (codegen.cpp, "The step in ONE launch")
static bool fusedEligible(const Query& q) {
    const bool off = getenv("RSQ_FUSED_STEP") && atoi(getenv("RSQ_FUSED_STEP")) == 0;
    return !off && q.aggMode == AggMode::DENSE_REG && q.pipelines.size() == 1 && q.pipelines[0].sink == SinkKind::AGGREGATE &&
           !q.pipelines[0].partitioned && q.dFinTicket != nullptr;
}

// a shard of a multi-GPU plan that does not end in a dense partial table runs its pipelines and reads the group rows (or the
// materialised columns) back, but leaves the tail to the root, which merges all shards' groups first (tail.cpp runTailMerged)
static void tailUnlessHeld(Query& q) { if (!q.holdTail) runTail(q); }

static void executeQueryBody(Query& q, bool partialOnly, bool async);
void executeQuery(Query& q, bool partialOnly, bool async) {
    // rows appended to a table move its columns (rsq_table_append): the kernels' arguments of a statement compiled before point at memory
    // that has been given back.  A ReSQL host compiles per statement (execute.h:213-247); whoever keeps a compiled query across a
    // BULK INSERT is told to compile it again instead of being answered from freed memory.
    for (size_t i = 0; i < q.tables.size() && i < q.tableLayouts.size(); i++)
        if (q.tables[i]->layoutVersion != q.tableLayouts[i])
            throw Error(RSQ_ERR_INVALID, "rows were appended to table " + q.tables[i]->name + " after this statement was compiled: compile it again");
    executeQueryBody(q, partialOnly, async);
    keepKeyIndexes(q);
    rememberPlan(q);
}

static void executeQueryBody(Query& q, bool partialOnly, bool async) {
    Context& ctx = q.ctx;
    if (ctx.device < 0) throw Error(RSQ_ERR_DEVICE, "this context has no device (compile-only)");
    RSQ_HIP(hipSetDevice(ctx.device));
    const uint64_t epochAtEntry = ctx.execEpoch++;
    q.hRowsView = nullptr;          // (set around a candidate run of the tail only; an execution that did not come back must not leave it)
    if (async) {
        if (!partialOnly || !denseMode(q)) failUnsupported("asynchronous execution is available for dense partial aggregation only");
        for (auto& p : q.pipelines)
            if (p.sink != SinkKind::AGGREGATE) failUnsupported("asynchronous execution needs a plan without join / materialize pipelines");
    }
    double t0 = nowMs();
    const size_t words = q.pinnedWords;
    q.report.num_kernels = 0; q.report.bytes_read = 0;
    if (!q.genericActive && q.quickTier && q.bgState.load() >= 3) {
        // the full kernels are in the cache now: the quick tier's are replaced (same arguments, same tables - nothing else changes).
        // (A full tier that failed to compile leaves the query on the quick one.)
        const int st = q.bgState.load();
        if (q.bgCompiler.joinable()) q.bgCompiler.join();
        if (st == 4) { resolveKernels(q); q.explainText += "kernel tier: full\n"; }
        else q.bgState = 0;
    }
    if (q.genericActive) {
        if (!q.genericForced && q.bgState.load() >= 2) {
            // the specialised kernels are in the code-object cache now (or their compilation failed): switch over
            const int st = q.bgState.load();
            if (st != 2 && q.bgCompiler.joinable()) q.bgCompiler.join();      // (2: the thread is still compiling the full tier)
            if (st == 3) throw Error(RSQ_ERR_DEVICE, q.bgError);
            resolveKernels(q, st == 2);
            q.explainText += st == 2 ? "kernel tier: quick (stage 2 called) until the inlined kernels are ready\n" : "kernel tier: full\n";
            q.genericActive = false;
            if (q.generic2) leaveGeneric2(q);
            applyPlanMemo(q);
            q.report.jit_compiles = (int32_t)kernelSources(q).size();       // built by the compiler thread since rsq_query_compile returned
        } else if (!q.generic2) {
            // ---- the generic pipeline: table init, ONE interpreter launch, read-back ----
            Pipeline& p = q.pipelines[0];
            q.fusedReady = false;
            ctx.errWordClean = false;
            RSQ_HIP(hipMemcpyAsync(q.dAgg, q.dAggInit, q.tableWords * 8, hipMemcpyDeviceToDevice, ctx.stream));
            RSQ_HIP(hipMemsetAsync(ctx.dErr, 0, 4, ctx.stream));
            RSQ_HIP(hipEventRecord(ctx.ev0, ctx.stream));
            launchGenericAggregate(ctx, q.generic, q.dGenericCode, p.src->nRows, p.src->row0, q.dAgg, q.denseGroups, (int64_t)q.tableWords);
            RSQ_HIP(hipEventRecord(ctx.ev1, ctx.stream));
            q.report.num_kernels = 1;
            q.report.bytes_read = (uint64_t)(p.bytesPerRow * p.src->nRows);
            RSQ_HIP(hipMemcpyAsync(q.hPinned + words, ctx.dErr, 4, hipMemcpyDeviceToHost, ctx.stream));
            if (!partialOnly) RSQ_HIP(hipMemcpyAsync(q.hPinned, q.dAgg, q.tableWords * 8, hipMemcpyDeviceToHost, ctx.stream));
            if (async && partialOnly) { q.pendingAsync = true; q.pendingFused = false; q.report.execution_time_ms = nowMs() - t0; return; }
            waitForStream(ctx);
            float gms = 0; RSQ_HIP(hipEventElapsedTime(&gms, ctx.ev0, ctx.ev1));
            q.report.kernel_time_ms = gms; q.kernelTimeSumMs += gms; q.kernelTimeLaunches++;
            q.report.hbm_gbps = gms > 0 ? (double)q.report.bytes_read / (gms * 1e-3) / 1e9 : 0;
            ctx.errWordClean = (uint32_t)q.hPinned[words] == 0;
            checkDeviceError((uint32_t)q.hPinned[words]);
            if (!partialOnly) {
                double t1 = nowMs();
                q.hAggView = nullptr;
                memcpy(q.hAgg.data(), q.hPinned, q.tableWords * 8);
                tailUnlessHeld(q);
                q.report.finalize_time_ms = nowMs() - t1;
            }
            q.report.execution_time_ms = nowMs() - t0;
            return;
        }
    }
    const bool interp = q.genericActive && q.generic2;      // this execution's pipelines run on the interpreter for whole pipelines
    q.flatRun = (partialOnly || interp) && q.aggPad > 1;      // (the interpreter aggregates into the unpadded table)
    const bool trace0 = getenv("RSQ_TRACE") != nullptr;
    uint32_t topkCapacity = 0, topkSpec = 0;      // > 0: this execution pre-selects ORDER BY ... LIMIT candidates on the device
    bool topkRange = false;                       // ... with the short form (one histogram over the images' range)
    uint32_t groupRowsAllocated = 0;              // rows the group-row buffers of this execution can take
    uint64_t selectSeq = 0;                       // ... announced by this sequence number in pinned word 4
    bool selectPublished = false;                 // ... and the selection's launch has delivered candidates and status words to the host
    bool narrowRows = false;                      // ... from narrow group rows [slot | sort key]: q.dGroupRows was NOT written by this execution
    auto ensureCandHost = [&]() {        // coherent pinned rows for topkCapacity candidates, and the device's view of them
        const size_t need = (size_t)topkCapacity * (size_t)q.groupRowWords;
        if (q.hCandRowsWords >= need && q.dHostCandRows) return;
        if (q.hCandRows) ctx.freePinned(q.hCandRows);
        q.hCandRows = nullptr; q.dHostCandRows = nullptr; q.hCandRowsWords = 0;
        q.hCandRows = (int64_t*)ctx.allocPinned(std::max<size_t>(need, 8) * 8);
        q.hCandRowsWords = need;
        void* dv = nullptr;
        if (hipHostGetDevicePointer(&dv, q.hCandRows, 0) == hipSuccess && dv) q.dHostCandRows = (int64_t*)dv; else (void)hipGetLastError();
    };
    auto fusedSelectOk = [&]() {
        const bool off = getenv("RSQ_FUSED_SELECT") && atoi(getenv("RSQ_FUSED_SELECT")) == 0;
        const bool publish = q.dPinnedDev && !(getenv("RSQ_PUBLISH_STATUS") && atoi(getenv("RSQ_PUBLISH_STATUS")) == 0);
        if (off || q.fusedSelectOff || !publish || partialOnly || async || getenv("RSQ_TRACE")) return false;
        ensureCandHost();
        return q.dHostCandRows != nullptr;
    };
    // ---- the step in one launch: a single register-mode pipeline whose last workgroup publishes the table ----
    if (fusedEligible(q) && !trace0 && !interp) {
        Pipeline& p = q.pipelines[0];
        // the kernel leaves its working table, the error word and the ticket at their identities; make them so the first
        // time, after an execution that did not come back (an exception between launch and synchronisation), and whenever
        // another query of this context may have left the shared error word set
        if (!q.fusedReady || !ctx.errWordClean) {
            RSQ_HIP(hipMemcpyAsync(q.dAggWork, q.dAggWorkInit, q.padWords * 8, hipMemcpyDeviceToDevice, ctx.stream));
            RSQ_HIP(hipMemsetAsync(q.dFinTicket, 0, 4, ctx.stream));
            RSQ_HIP(hipMemsetAsync(ctx.dErr, 0, 4, ctx.stream));
            ctx.errWordClean = true;
        }
        static const bool stepTrace0 = getenv("RSQ_TRACE") && atoi(getenv("RSQ_TRACE")) >= 2;
        // a step that runs to its end here takes the next pair of the event ring (read when somebody asks, or when the ring is
        // full); an asynchronous partial step keeps the single pair finalize / settle read
        const bool ringEvents = !(async && partialOnly);
        if (ringEvents) {
            // the ring grows with the steps nobody has read yet: 8 pairs, doubled when full, up to 256 (a statement that is executed once
            // takes 16 events from the context's pool, a loop of steps ends up with the 512 it had from the start before round 5)
            const bool full = q.evTail - q.evHead == q.evRing.size();      // (an empty ring is full too)
            if (q.kernelTimePending || full) resolveKernelTime(q);          // everything recorded so far is read: head == tail
            if (full && q.evRing.size() < 256) {
                const size_t more = q.evRing.empty() ? 8 : q.evRing.size();
                q.evHead = q.evTail = 0;
                for (size_t i = 0; i < more; i++) { hipEvent_t a = ctx.takeEvent(), b = ctx.takeEvent(); q.evRing.emplace_back(a, b); }
            }
        } else resolveKernelTime(q);                           // the previous step's events, before they are recorded again
        q.fusedReady = false;
        q.flatRun = false;                                     // always the padded kernel: the last workgroup unpads
        const bool pollOk = !(getenv("RSQ_POLL") && atoi(getenv("RSQ_POLL")) == 0);
        const bool poll = pollOk && !partialOnly;
        const uint64_t seq = poll ? ++q.finSeqCounter : 0;
        q.finSeq = seq;
        q.finOut = partialOnly ? q.dAgg : q.dFinHost;          // device partial table | host-mapped pinned read-back buffer
        q.finErr = q.dFinHost + q.pinnedWords;
        if (!q.gev0) { q.gev0 = ctx.takeEvent(); q.gev1 = ctx.takeEvent(); }
        hipEvent_t evA = q.gev0, evB = q.gev1;
        if (ringEvents) { auto& e = q.evRing[q.evTail % q.evRing.size()]; evA = e.first; evB = e.second; }
        // two event records around the launch.  (The extended launch that takes the events itself - RSQ_EXT_EVENTS=1 - costs the
        // host 3 us more per step and the tail another 1.5: measured 16 us of step overhead against 11.5.)
        const bool extEvents = false;
        if (extEvents) launchPipelineKernel(q, p, *p.kernel, -1, 0, 0, evA, evB);
        else {
            const double tB = stepTrace0 ? nowMs() : 0;
            RSQ_HIP(hipEventRecord(evA, ctx.stream));
            const double tC = stepTrace0 ? nowMs() : 0;
            launchPipelineKernel(q, p, *p.kernel, -1);
            const double tD = stepTrace0 ? nowMs() : 0;
            RSQ_HIP(hipEventRecord(evB, ctx.stream));
            if (stepTrace0) {
                static double a[3] = {0, 0, 0}; static int n = 0;
                a[0] += tC - tB; a[1] += tD - tC; a[2] += nowMs() - tD;
                if (++n == 64) { fprintf(stderr, "[rsq step]   event record %.1f us, launch %.1f us, event record %.1f us\n", a[0] / 64 * 1e3, a[1] / 64 * 1e3, a[2] / 64 * 1e3); a[0] = a[1] = a[2] = 0; n = 0; }
            }
        }
        q.finOut = nullptr;
        q.finSeq = 0;
        if (ringEvents) q.evTail++; else q.kernelTimePending = true;
        static const bool stepTrace = getenv("RSQ_TRACE") && atoi(getenv("RSQ_TRACE")) >= 2;        // host-side phases of the one-launch step, averaged over 64 steps
        const double tLaunched = stepTrace ? nowMs() : 0;
        q.report.bytes_read = (uint64_t)(p.bytesPerRow * p.src->nRows);
        if (async && partialOnly) {
            // (the kernel leaves its working table reset, and whatever comes next on this stream is ordered behind it)
            q.fusedReady = true;
            q.pendingAsync = true; q.pendingFused = true; q.report.execution_time_ms = nowMs() - t0; return;
        }
        if (poll) {
            // the finished table is in host memory as soon as the last workgroup's stores have landed: watch the sequence
            // number instead of waiting for the stream's completion signal (saves the interrupt path, ~8 us per step);
            // hipStreamQuery now and then notices a failed launch, and a kernel that takes long is waited for properly
            volatile uint64_t* flag = q.hPinned + q.pinnedWords + 1;
            const double deadline = nowMs() + 5.0;
            unsigned spins = 0;
            while (*flag != seq) {
                if ((++spins & 1023u) == 0) {
                    hipError_t e = hipStreamQuery(ctx.stream);
                    if (e != hipSuccess && e != hipErrorNotReady) RSQ_HIP(e);
                    if (e == hipSuccess || nowMs() > deadline) { waitForStream(ctx); break; }
                }
                __builtin_ia32_pause();
            }
            if (*flag != seq) failRuntime("internal error: the fused step finished without publishing its table");
            std::atomic_thread_fence(std::memory_order_acquire);
        } else waitForStream(ctx);
        q.fusedReady = true;
        const double tSeen = stepTrace ? nowMs() : 0;
        if (q.dDebugStamps) {
            static int nPrinted = 0;
            waitForStream(ctx);
            std::vector<uint64_t> st((size_t)p.lastGrid * 8);
            RSQ_HIP(hipMemcpy(st.data(), q.dDebugStamps, st.size() * 8, hipMemcpyDeviceToHost));
            uint64_t t0s = ~0ull, endLoop = 0, endRed = 0, endFlush = 0, endTicket = 0, last5 = 0, last6 = 0, last7 = 0;
            for (unsigned w = 0; w < p.lastGrid; w++) {
                t0s = std::min(t0s, st[w * 8 + 0]); endLoop = std::max(endLoop, st[w * 8 + 1]); endRed = std::max(endRed, st[w * 8 + 2]);
                endFlush = std::max(endFlush, st[w * 8 + 3]); endTicket = std::max(endTicket, st[w * 8 + 4]);
                last5 = std::max(last5, st[w * 8 + 5]); last6 = std::max(last6, st[w * 8 + 6]); last7 = std::max(last7, st[w * 8 + 7]);
            }
            if ((nPrinted + 1) % 50 == 0) {
                std::vector<uint64_t> ends;
                for (unsigned w = 0; w < p.lastGrid; w++) ends.push_back(st[w * 8 + 2]);
                std::sort(ends.begin(), ends.end());
                fprintf(stderr, "[rsq tail] workgroups done with their rows and reductions at: first %.2f, 10 %% %.2f, median %.2f, 90 %% %.2f, last %.2f us\n",
                        (ends.front() - t0s) / 100.0, (ends[ends.size() / 10] - t0s) / 100.0, (ends[ends.size() / 2] - t0s) / 100.0,
                        (ends[ends.size() * 9 / 10] - t0s) / 100.0, (ends.back() - t0s) / 100.0);
            }
            if (++nPrinted % 50 == 0)
                fprintf(stderr, "[rsq tail] %u workgroups; us since the first workgroup started: last loop end %.2f, wave reductions %.2f, flush issued %.2f, "
                                "ticket taken %.2f, cells exchanged + stored to host %.2f, stores acknowledged %.2f, flag stored %.2f\n", p.lastGrid,
                        (endLoop - t0s) / 100.0, (endRed - t0s) / 100.0, (endFlush - t0s) / 100.0, (endTicket - t0s) / 100.0, (last5 - t0s) / 100.0,
                        (last6 - t0s) / 100.0, (last7 - t0s) / 100.0);
            RSQ_HIP(hipMemset(q.dDebugStamps, 0, st.size() * 8));
        }
        checkDeviceError((uint32_t)q.hPinned[q.pinnedWords]);
        if (!partialOnly) {
            double t1 = nowMs();
            q.hAggView = nullptr;
            memcpy(q.hAgg.data(), q.hPinned, q.tableWords * 8);
            tailUnlessHeld(q);
            q.report.finalize_time_ms = nowMs() - t1;
        }
        q.report.execution_time_ms = nowMs() - t0;
        if (stepTrace) {
            static double acc[4] = {0, 0, 0, 0}; static int n = 0; static double lastEnd = 0;
            const double tEnd = nowMs();
            acc[0] += tLaunched - t0; acc[1] += tSeen - tLaunched; acc[2] += tEnd - tSeen; if (lastEnd > 0) acc[3] += t0 - lastEnd;
            lastEnd = tEnd;
            if (++n == 64) {
                fprintf(stderr, "[rsq step] enqueue %.1f us, launch -> result seen %.1f us, tail %.1f us, between executions %.1f us\n",
                        acc[0] / 64 * 1e3, acc[1] / 64 * 1e3, acc[2] / 64 * 1e3, acc[3] / 63 * 1e3);
                acc[0] = acc[1] = acc[2] = acc[3] = 0; n = 0; lastEnd = 0;
            }
        }
        return;
    }
    if (denseMode(q)) enqueueTableInit(q);
    ctx.errWordClean = false;      // until this execution has read the word back as 0
    bool anyCompaction = false;
    for (auto& p : q.pipelines) anyCompaction |= p.compact;
    const bool trace = getenv("RSQ_TRACE") != nullptr;      // per-pipeline wall time (synchronises after each one)
    // everything small this execution wants cleared, in one launch (aot_kernels.hip k_fill_batch): the error word, the pipelines'
    // row counters, the group counter and the candidate selection's scratch of a compaction behind the last pipeline
    bool groupCountCleared = false, topkScratchCleared = false;
    bool accumulatorsCleared = false;       // the aggregates beside a join table's entries are at their identities (AT_JOIN_ENTRY)
    // A hash aggregation that has run before: its table is readied with everything else in the first fill (or behind the previous
    // execution's last kernel), the group count is the remembered one, and whether the table overflowed or CHAR groups need merging
    // is read from the FINAL status words - no synchronisation between the pipelines and the group rows.  (TPC-H Q10 at SF10: two 4-byte
    // read-backs with a stream synchronisation each and three clearing launches, ~40 of 590 us.)  Whatever turns out different from the
    // remembered state starts the execution over on the careful path (lastCount = 0).
    bool hashWarm = false;
    std::vector<FillItem> f;                // (kept: an execution that ends on its candidates enqueues the same clears for the next one)
    {
        f.push_back(FillItem{ctx.dErr, 4, 0});
        if (anyCompaction) f.push_back(FillItem{q.dPipeStats, q.pipelines.size() * 8, 0});
        if (q.dGroupCount) { f.push_back(FillItem{q.dGroupCount, 4, 0}); groupCountCleared = true; }
        if (q.dTopkHists) { f.push_back(FillItem{q.dTopkHists, topkRangeScratchBytes(), 0}); topkScratchCleared = true; }
        // Join tables that have been sized (every execution but a query's first) are readied HERE, all of them in this one launch,
        // instead of one launch in front of every build (TPC-H Q5 builds five tables, Q3 two); likewise the aggregates kept beside
        // a join table's entries.  Nothing touches a table between this fill and its build pipeline.
        const bool prologueOk = true;
        for (auto& hp : q.hashTables) hp->prepared = false;       // (an execution that did not come back must not leave a table marked as readied)
        if (prologueOk && !trace && !interp)
            for (auto& p : q.pipelines) {
                if (p.sink != SinkKind::BUILD) continue;
                HashTable& h = *q.hashTables[(size_t)p.buildTable];
                h.prepared = false;
                if (h.capacity == 0) continue;
                if (h.direct && (h.dense || (h.keyIndexReady && h.keyIndexVersion == p.src->version))) continue;      // nothing is built: nothing to ready
                const size_t nWords = h.keys.size() + h.payload.size();
                const size_t bmBytes = h.hasBitmap && h.dBitmap ? (size_t)h.bmBlocks * 32 : 0;
                if (h.rank && h.setOnly) { if (!bmBytes) continue; }
                else if (h.rank) {
                    if (!h.dTempUsed || !h.dChunkTotal) continue;
                    f.push_back(FillItem{h.dTempUsed, (size_t)h.tempWaves * 4, 0});
                    f.push_back(FillItem{h.dChunkTotal, (size_t)((h.bmBlocks + RSQ_RANK_CHUNK_BLOCKS - 1) / RSQ_RANK_CHUNK_BLOCKS) * 4, 0});      // the chain of the one-launch index
                } else if (h.keyCas) {
                    if (!h.dWords) continue;
                    f.push_back(FillItem{h.dWords, (size_t)h.capacity * (h.aos ? std::max<size_t>(1, nWords) : 1) * 8, 0x8000000000000000ull});
                } else {
                    if (!h.dState) continue;
                    f.push_back(FillItem{h.dState, (size_t)h.capacity * 4, 0});
                }
                if (bmBytes) f.push_back(FillItem{h.dBitmap, bmBytes, 0});
                f.push_back(FillItem{h.dCount, 4, 0});
                h.prepared = true;
            }
        if (prologueOk && !trace && !interp && q.aggMode == AggMode::AT_JOIN_ENTRY) {
            HashTable& h = *q.hashTables[(size_t)q.aggTable];
            if (h.capacity > 0 && h.dAcc && h.prepared) {       // (sized, and not about to be re-sized by its build)
                for (int b = 0; b < h.nAccBlocks; b++) {
                    const uint64_t idv = b < q.nMinBlocks ? 0x7fffffffffffffffull : b < q.nMinBlocks + q.nMaxBlocks ? 0x8000000000000000ull : 0ull;
                    f.push_back(FillItem{(uint64_t*)h.dAcc + (size_t)b * (size_t)h.capacity, (size_t)h.capacity * 8, idv});
                }
                accumulatorsCleared = true;
            }
        }
        if (!trace && !interp && q.aggMode == AggMode::HASH && !partialOnly && !async && !q.holdTail && q.aggTable >= 0) {
            HashTable& h = *q.hashTables[(size_t)q.aggTable];
            if (h.capacity > 0 && h.dState && h.dAcc && h.lastCount > 0 && (int64_t)h.lastCount * 2 <= h.capacity) {
                f.push_back(FillItem{h.dState, (size_t)h.capacity * 4, 0});
                f.push_back(FillItem{h.dCount, 4, 0});
                for (int b = 0; b < h.nAccBlocks; b++) {
                    const uint64_t idv = b < q.nMinBlocks ? 0x7fffffffffffffffull : b < q.nMinBlocks + q.nMaxBlocks ? 0x8000000000000000ull : 0ull;
                    f.push_back(FillItem{(uint64_t*)h.dAcc + (size_t)b * (size_t)h.capacity, (size_t)h.capacity * 8, idv});
                }
                hashWarm = true;
            }
        }
        // ... unless the previous execution of this query has already enqueued exactly these clears behind its last kernel and
        // nothing else ran on the context since (RSQ_POST_CLEAR=0: never)
        auto same = [](const std::vector<FillItem>& a, const std::vector<FillItem>& b) {
            if (a.size() != b.size()) return false;
            for (size_t i = 0; i < a.size(); i++) if (a[i].p != b[i].p || a[i].bytes != b[i].bytes || a[i].value != b[i].value) return false;
            return true;
        };
        const bool readied = q.readied && q.readiedEpoch == epochAtEntry && same(f, q.readiedFill);
        q.readied = false;
        if (!readied) {
            fillBatchAsync(ctx, f.data(), (int)f.size());
            q.report.num_kernels += (f.size() + 23) / 24;
        }
    }
    // (the execution's event pair is the query's own, read when somebody asks: an execution that ends on the candidate selection's
    // sequence number returns before the stream has reported the second event)
    resolveKernelTime(q);
    if (!q.gev0) { q.gev0 = ctx.takeEvent(); q.gev1 = ctx.takeEvent(); }
    RSQ_HIP(hipEventRecord(q.gev0, ctx.stream));
    double tPipe = nowMs();
    auto tracePoint = [&](const Pipeline& p) {
        if (!trace) return;
        (void)hipStreamSynchronize(ctx.stream);
        double t = nowMs();
        fprintf(stderr, "[rsq trace] %.3f ms  %s\n", t - tPipe, p.explain.c_str());
        if (p.sink == SinkKind::BUILD) {
            HashTable& h = *q.hashTables[(size_t)p.buildTable];
            uint32_t n = 0;
            (void)hipMemcpy(&n, h.dCount, 4, hipMemcpyDeviceToHost);
            fprintf(stderr, "[rsq trace]     ht%d: %u entries in %lld slots\n", h.id, n, (long long)h.capacity);
        }
        tPipe = nowMs();
    };
    for (auto& p : q.pipelines) {
        if (interp) { runGeneric2Pipeline(q, (size_t)(&p - q.pipelines.data())); tracePoint(p); continue; }
        if (p.sink == SinkKind::BUILD) { buildHashTable(q, p); tracePoint(p); continue; }
        if (p.sink == SinkKind::MATERIALIZE) { materializePipeline(q, p); tracePoint(p); continue; }
        auto resetAccumulators = [&](HashTable& h) {
            // aggregate words beside the entries: first-row / min blocks to +inf, max blocks to -inf, sums to 0
            std::vector<FillItem> f;
            for (int b = 0; b < h.nAccBlocks; b++) {
                uint64_t idv = b < q.nMinBlocks ? 0x7fffffffffffffffull : b < q.nMinBlocks + q.nMaxBlocks ? 0x8000000000000000ull : 0ull;
                f.push_back(FillItem{(uint64_t*)h.dAcc + (size_t)b * (size_t)h.capacity, (size_t)h.capacity * 8, idv});
            }
            fillBatchAsync(ctx, f.data(), (int)f.size());
        };
        if (q.aggMode == AggMode::AT_JOIN_ENTRY && !accumulatorsCleared) { resetAccumulators(*q.hashTables[(size_t)q.aggTable]); q.report.num_kernels++; }
        if (q.aggMode == AggMode::HASH && hashWarm) {
            launchPipeline(q, p, -1);
            q.report.bytes_read += (uint64_t)(p.bytesPerRow * p.src->nRows);
            tracePoint(p);
            continue;
        }
        if (q.aggMode == AggMode::HASH) {
            // The number of groups is not known before the scan: start from the reference's own estimate
            // (AggregationOp::getSize, aggregation.h:81-92) and re-run the pipeline with a 4x larger table while
            // the kernel reports a full table.
            HashTable& h = *q.hashTables[(size_t)q.aggTable];
            if (h.capacity == 0) h.capacity = nextPow2(std::max<int64_t>(4096, 4 * (int64_t)opSize(q.agg, true)));
            if ((int64_t)h.lastCount * 2 > h.capacity) {        // the previous execution filled more than half of the table
                if (h.dState) { ctx.free(h.dState); ctx.free(h.dWords); ctx.free(h.dAcc); }
                h.dState = nullptr; h.dWords = nullptr; h.dAcc = nullptr;
                while ((int64_t)h.lastCount * 2 > h.capacity) h.capacity *= 2;
            }
            for (;;) {
                if (!h.dState) {
                    h.dState = (uint32_t*)ctx.alloc((size_t)h.capacity * 4);
                    h.dWords = (int64_t*)ctx.alloc((size_t)h.capacity * 8 * (h.keys.size() + h.payload.size()));      // key words, then carried group values
                    h.dAcc = (int64_t*)ctx.alloc((size_t)h.capacity * 8 * (size_t)h.nAccBlocks);
                }
                RSQ_HIP(hipMemsetAsync(h.dState, 0, (size_t)h.capacity * 4, ctx.stream));
                RSQ_HIP(hipMemsetAsync(h.dCount, 0, 4, ctx.stream));
                resetAccumulators(h);
                launchPipeline(q, p, -1);
                uint32_t err = 0;
                RSQ_HIP(hipMemcpyAsync(&err, ctx.dErr, 4, hipMemcpyDeviceToHost, ctx.stream));
                waitForStream(ctx);
                q.charGroupsNeedMerge = (err & 32u) != 0;
                if (trace) fprintf(stderr, "[rsq trace]     hash aggregation table ht%d: %lld slots%s\n", h.id, (long long)h.capacity, (err & 2) ? " - too small, four times as many next" : "");
                if (!(err & 2)) break;
                if (h.capacity >= ((int64_t)1 << 31)) failRuntime("Hash table full");
                ctx.free(h.dState); ctx.free(h.dWords); ctx.free(h.dAcc);
                h.dState = nullptr; h.dWords = nullptr; h.dAcc = nullptr;
                h.capacity *= 4;
                RSQ_HIP(hipMemsetAsync(ctx.dErr, 0, 4, ctx.stream));
            }
            q.report.bytes_read += (uint64_t)(p.bytesPerRow * p.src->nRows);
            tracePoint(p);
            continue;
        }
        if (p.partitioned) runLargeDenseAggregation(q, p);
        else launchPipeline(q, p, -1);
        q.report.bytes_read += (uint64_t)(p.bytesPerRow * p.src->nRows);
        tracePoint(p);
    }
    if ((q.aggMode == AggMode::AT_JOIN_ENTRY || q.aggMode == AggMode::HASH) && !partialOnly) {
        HashTable& h = *q.hashTables[(size_t)q.aggTable];
        const int nTab = (int)(h.keys.size() + h.payload.size());
        q.groupRowWords = 1 + nTab + h.nAccBlocks;
        // Rows to provide for: the number of occupied entries.  It is read back (one synchronisation) on the first
        // execution; later executions reuse it — the build reads the same table (hash aggregation: hashWarm above) — and the
        // compaction never writes beyond the buffer: a larger count is noticed after the final synchronisation and
        // the execution is repeated with a fresh count.
        uint32_t nEntries = h.lastCount;
        if ((q.aggMode != AggMode::AT_JOIN_ENTRY && !hashWarm) || nEntries == 0 || getenv("RSQ_TRACE")) {
            RSQ_HIP(hipMemcpyAsync(&nEntries, h.dCount, 4, hipMemcpyDeviceToHost, ctx.stream));
            waitForStream(ctx);
            h.lastCount = nEntries;
        }
        groupRowsAllocated = std::max<uint32_t>(1, nEntries);
        size_t need = (size_t)groupRowsAllocated * (size_t)q.groupRowWords;
        if (q.dGroupRowsWords < need) {
            if (q.dGroupRows) ctx.free(q.dGroupRows);
            q.dGroupRows = nullptr; q.dGroupRowsWords = 0;          // (nothing dangles if the allocation below throws)
            q.dGroupRows = (int64_t*)ctx.alloc(need * 8);
            q.dGroupRowsWords = need;
        }
        // ORDER BY ... LIMIT k over many groups: select the candidate rows on the device and read back only those
        if (q.topkWord == -2) planDeviceTopK(q);
        const bool preselect = !q.holdTail && q.topkWord >= 0 && nEntries >= 2048 && (uint64_t)nEntries > 4ull * q.topkWant && !(q.topkNeedsNoMerge && q.charGroupsNeedMerge);
        if (preselect) {
            topkCapacity = std::min<uint32_t>(nEntries, std::max<uint32_t>(1024, 4 * q.topkWant));
            if (!q.dTopkHists) { q.dTopkHists = (uint32_t*)ctx.alloc(topkHistBytes()); q.dCandCount = q.dTopkHists + 4; }
            if (q.candCapacity < topkCapacity || q.candRowWords != q.groupRowWords) {
                if (q.dCandRows) ctx.free(q.dCandRows);
                q.dCandRows = (int64_t*)ctx.alloc((size_t)topkCapacity * (size_t)q.groupRowWords * 8);
                q.candCapacity = topkCapacity; q.candRowWords = q.groupRowWords;
            }
            if (!topkScratchCleared) prepareTopCandidatesRange(ctx, q.dTopkHists);
        }
        if (!groupCountCleared) RSQ_HIP(hipMemsetAsync(q.dGroupCount, 0, 4, ctx.stream));
        // Wide group rows of which the statement wants a few (TPC-H Q10: 380 K groups of 40 words - string group values -, LIMIT 20): the
        // compaction writes [slot | sort key] per group, the one-launch selection fetches its candidates' rows from the table.  Whatever
        // then turns out to need every row (candidates that overflow or do not decide the answer) starts the execution over with full rows.
        narrowRows = preselect && q.groupRowWords > 8 && !q.narrowRowsOff && fusedSelectOk();
        if (narrowRows && q.narrowRowsCap < groupRowsAllocated) {
            if (q.dNarrowRows) ctx.free(q.dNarrowRows);
            q.dNarrowRows = nullptr; q.narrowRowsCap = 0;
            q.dNarrowRows = (int64_t*)ctx.alloc((size_t)groupRowsAllocated * 16);
            q.narrowRowsCap = groupRowsAllocated;
        }
        // string group values that stand in the entries by address (this execution's kernels wrote them so: not the interpreter's, and only
        // while every table of the dependency chain is a rank dictionary - the kernel's own run-time condition, codegen_agg.cpp)
        const int* deref = nullptr;
        int tabStride = nTab;
        if (h.dDeref && !interp) {
            bool holds = true;
            for (int t : h.derefCondTables) holds = holds && q.hashTables[(size_t)t]->rank;
            if (holds) { deref = h.dDeref; tabStride = h.compactStride > 0 ? h.compactStride : nTab; }
        }
        compactEntries(ctx, (const int64_t*)h.dAcc /* block 0 = first row */, h.capacity, h.dWords, nTab, h.aos, h.dAcc, h.nAccBlocks,
                       narrowRows ? q.dNarrowRows : q.dGroupRows, groupRowsAllocated, q.dGroupCount,
                       h.rank,
                       q.topkWord, q.topkIs32, q.topkDesc, preselect ? (uint64_t*)q.dTopkHists : nullptr, narrowRows, deref, tabStride);
        q.report.num_kernels++;
        if (narrowRows) {
            topkRange = true;
            const TableEntries te{(const int64_t*)h.dAcc, h.capacity, h.dWords, nTab, h.aos, h.dAcc, h.nAccBlocks, h.rank, deref, tabStride};
            selectTopCandidatesRangePublish(ctx, q.dNarrowRows, 2, 1, q.topkIs32, q.topkDesc, q.dGroupCount, groupRowsAllocated,
                                            q.topkWant, q.dTopkHists, q.dHostCandRows, topkCapacity, q.dPinnedDev + words, selectSeq = ++q.finSeqCounter, ctx.dErr, q.dGroupCount,
                                            anyCompaction ? q.dPipeStats : nullptr, (int)q.pipelines.size(), &te);
            q.report.num_kernels += 1;
            selectPublished = true;
            topkSpec = topkCapacity;
        } else
        if (preselect) {
            // the short form: the compaction collected the range of the sort key's images, ONE histogram over that range finds
            // the candidates (aot_kernels.hip); the exact radix select runs only if they overflow the buffer (below)
            topkRange = true;
            if (fusedSelectOk()) {
                // ... in one launch that also delivers the candidates and the status words to the host (aot_kernels.hip k_topk_range_select)
                selectTopCandidatesRangePublish(ctx, q.dGroupRows, q.groupRowWords, q.topkWord, q.topkIs32, q.topkDesc, q.dGroupCount, groupRowsAllocated,
                                                q.topkWant, q.dTopkHists, q.dHostCandRows, topkCapacity, q.dPinnedDev + words, selectSeq = ++q.finSeqCounter, ctx.dErr, q.dGroupCount,
                                                anyCompaction ? q.dPipeStats : nullptr, (int)q.pipelines.size());
                q.report.num_kernels += 1;
                selectPublished = true;
                topkSpec = topkCapacity;
            } else {
            selectTopCandidatesRange(ctx, q.dGroupRows, q.groupRowWords, q.topkWord, q.topkIs32, q.topkDesc, q.dGroupCount, groupRowsAllocated, q.topkWant,
                                     q.dTopkHists, q.dCandRows, topkCapacity);
            q.report.num_kernels += 2;
            // the leading candidates travel with the same synchronisation as the counts (usually that is all of them)
            topkSpec = std::min<uint32_t>(topkCapacity, std::max<uint32_t>(q.topkWant + 64, 65536u / (uint32_t)(q.groupRowWords * 8)));
            }
        }
    }
    // ORDER BY ... LIMIT k over a large DENSE aggregate table: the groups present are compacted into rows [first row | group id |
    // accumulator blocks] and go through the same candidate pre-selection; the table itself (tens of MB) is only read back
    // when the candidates do not decide the answer
    bool denseTopk = false;
    if (!partialOnly && !q.holdTail && q.aggMode == AggMode::DENSE_GLOBAL && q.aggPad == 1 && q.denseGroups >= 65536 && q.denseGroups <= (1 << 21)) {
        if (q.topkWord == -2) planDeviceTopK(q);
        if (q.topkWord >= 0) {
            const uint32_t D = (uint32_t)q.denseGroups;
            const int W = (int)q.accums.size();
            q.groupRowWords = 2 + W;
            groupRowsAllocated = D;
            const size_t need = (size_t)D * (size_t)q.groupRowWords;
            if (q.dGroupRowsWords < need) {
                if (q.dGroupRows) ctx.free(q.dGroupRows);
                q.dGroupRows = nullptr; q.dGroupRowsWords = 0;
                q.dGroupRows = (int64_t*)ctx.alloc(need * 8);
                q.dGroupRowsWords = need;
            }
            if (!q.dGroupCount) q.dGroupCount = (uint32_t*)ctx.alloc(sizeof(uint32_t));
            topkCapacity = std::min<uint32_t>(D, std::max<uint32_t>(1024, 4 * q.topkWant));
            if (!q.dTopkHists) { q.dTopkHists = (uint32_t*)ctx.alloc(topkHistBytes()); q.dCandCount = q.dTopkHists + 4; }
            if (q.candCapacity < topkCapacity || q.candRowWords != q.groupRowWords) {
                if (q.dCandRows) ctx.free(q.dCandRows);
                q.dCandRows = (int64_t*)ctx.alloc((size_t)topkCapacity * (size_t)q.groupRowWords * 8);
                q.candCapacity = topkCapacity; q.candRowWords = q.groupRowWords;
            }
            if (!topkScratchCleared) prepareTopCandidatesRange(ctx, q.dTopkHists);
            if (!groupCountCleared) RSQ_HIP(hipMemsetAsync(q.dGroupCount, 0, 4, ctx.stream));
            compactEntries(ctx, (const int64_t*)q.dAgg /* block 0 = first row */, (int64_t)D, nullptr, 1, false, (const int64_t*)q.dAgg, W,
                           q.dGroupRows, D, q.dGroupCount, false, q.topkWord, q.topkIs32, q.topkDesc, (uint64_t*)q.dTopkHists);
            topkRange = true;
            if (fusedSelectOk()) {
                selectTopCandidatesRangePublish(ctx, q.dGroupRows, q.groupRowWords, q.topkWord, q.topkIs32, q.topkDesc, q.dGroupCount, D, q.topkWant,
                                                q.dTopkHists, q.dHostCandRows, topkCapacity, q.dPinnedDev + words, selectSeq = ++q.finSeqCounter, ctx.dErr, q.dGroupCount,
                                                anyCompaction ? q.dPipeStats : nullptr, (int)q.pipelines.size());
                q.report.num_kernels += 2;
                selectPublished = true;
                topkSpec = topkCapacity;
            } else {
            selectTopCandidatesRange(ctx, q.dGroupRows, q.groupRowWords, q.topkWord, q.topkIs32, q.topkDesc, q.dGroupCount, D, q.topkWant,
                                     q.dTopkHists, q.dCandRows, topkCapacity);
            q.report.num_kernels += 3;
            topkSpec = std::min<uint32_t>(topkCapacity, std::max<uint32_t>(q.topkWant + 64, 65536u / (uint32_t)(q.groupRowWords * 8)));
            }
            denseTopk = true;
        }
    }
    // The execution's end event.  Where a status kernel follows, the event stands BEHIND it: an event between two kernels is a barrier
    // packet of its own and kept the status kernel waiting ~6 us (per-dispatch trace); behind the last kernel it delays nobody.
    const bool statusKernelFollows = !selectPublished && q.dPinnedDev && !(getenv("RSQ_PUBLISH_STATUS") && atoi(getenv("RSQ_PUBLISH_STATUS")) == 0);
    if (!statusKernelFollows) RSQ_HIP(hipEventRecord(q.gev1, ctx.stream));
    q.kernelTimePending = true;
    bool devTail = false;                   // the rows of a large dense aggregation are made on the device (runDenseDeviceTail)
    bool inlineRows = false;                // up to kInlineRows group rows arrive with the status words
    constexpr uint32_t kInlineRows = 64;
    // a small dense aggregate table (TPC-H Q14: one group behind a join) travels with the status words instead of a copy of its own
    const size_t tableInlineWords = q.aggPad > 1 && !q.flatRun ? (size_t)q.padWords : (size_t)q.tableWords;
    bool tableInline = false;
    if (!selectPublished) {
        const bool wantGroups = !partialOnly && (q.aggMode == AggMode::AT_JOIN_ENTRY || q.aggMode == AggMode::HASH || denseTopk);
        tableInline = statusKernelFollows && !partialOnly && denseMode(q) && !denseTopk && !trace && tableInlineWords > 0 && tableInlineWords <= 1024 && !(!async && denseDeviceTailWanted(q));
        if (q.dPinnedDev && !(getenv("RSQ_PUBLISH_STATUS") && atoi(getenv("RSQ_PUBLISH_STATUS")) == 0)) {
            // error word, group count, candidate count and the pipelines' row counters: one kernel writes them into the pinned words
            if (wantGroups && !denseTopk && !topkCapacity && !q.holdTail && !trace && q.dGroupRows && q.groupRowWords > 0) {
                const size_t need = (size_t)kInlineRows * (size_t)q.groupRowWords;
                if (q.inlineRowsWords < need) {
                    if (q.hInlineRows) ctx.freePinned(q.hInlineRows);
                    q.hInlineRows = nullptr; q.dHostInlineRows = nullptr; q.inlineRowsWords = 0;
                    q.hInlineRows = (int64_t*)ctx.allocPinned(need * 8);
                    void* dv = nullptr;
                    if (hipHostGetDevicePointer(&dv, q.hInlineRows, 0) == hipSuccess && dv) { q.dHostInlineRows = (int64_t*)dv; q.inlineRowsWords = need; }
                    else (void)hipGetLastError();
                }
                inlineRows = q.dHostInlineRows != nullptr;
            }
            publishStatusAsync(ctx, q.dPinnedDev + words, ctx.dErr, wantGroups ? q.dGroupCount : nullptr, topkCapacity ? q.dCandCount : nullptr,
                               anyCompaction ? q.dPipeStats : nullptr, (int)q.pipelines.size(),
                               inlineRows ? q.dGroupRows : nullptr, q.groupRowWords, std::min<uint32_t>(kInlineRows, groupRowsAllocated), q.dHostInlineRows,
                               q.matWarmRun ? q.dMatTotal : nullptr,
                               tableInline ? (const uint64_t*)(q.aggPad > 1 && !q.flatRun ? q.dAggWork : q.dAgg) : nullptr, q.dPinnedDev, tableInline ? (uint32_t)tableInlineWords : 0u);
            RSQ_HIP(hipEventRecord(q.gev1, ctx.stream));
            q.report.num_kernels++;
        } else {
            RSQ_HIP(hipMemcpyAsync(q.hPinned + words, ctx.dErr, 4, hipMemcpyDeviceToHost, ctx.stream));
            if (wantGroups) RSQ_HIP(hipMemcpyAsync(q.hPinned + words + 1, q.dGroupCount, 4, hipMemcpyDeviceToHost, ctx.stream));
            if (anyCompaction) RSQ_HIP(hipMemcpyAsync(q.hPinned + words + 8, q.dPipeStats, q.pipelines.size() * 8, hipMemcpyDeviceToHost, ctx.stream));
            if (topkCapacity) RSQ_HIP(hipMemcpyAsync(q.hPinned + words + 2, q.dCandCount, 4, hipMemcpyDeviceToHost, ctx.stream));
        }
        devTail = !partialOnly && !async && denseMode(q) && !denseTopk && denseDeviceTailWanted(q);
        if (!partialOnly && denseMode(q) && !denseTopk && !devTail && !tableInline) enqueueTableReadback(q);
        if (topkCapacity) {
            ensureHostGroupRows(q, (size_t)topkCapacity * (size_t)q.groupRowWords);
            RSQ_HIP(hipMemcpyAsync(q.hGroupRows, q.dCandRows, (size_t)topkSpec * (size_t)q.groupRowWords * 8, hipMemcpyDeviceToHost, ctx.stream));
        }
    }
    if (async && partialOnly) {
        // everything is enqueued; the caller orders its own work (the group-by merge collective) behind it on the same
        // stream and finalizeQuery() does the one host synchronisation of the step
        q.pendingAsync = true; q.pendingFused = false;
        q.report.execution_time_ms = nowMs() - t0;
        return;
    }
    static const bool execTraceOn = getenv("RSQ_TRACE") && atoi(getenv("RSQ_TRACE")) >= 2;       // host-side phases of an execution of the general path, averaged over 8
    const double tEnqueued = execTraceOn ? nowMs() : 0;
    bool sawSequence = false;
    if (selectPublished) {
        // candidates and status words are in host memory once the selection's sequence number is: watch that word instead of the
        // stream's completion (as the one-launch step does); hipStreamQuery now and then notices a failed launch
        volatile uint64_t* flag = q.hPinned + words + 4;
        const double deadline = nowMs() + 5.0;
        unsigned spins = 0;
        sawSequence = true;
        while (*flag != selectSeq) {
            if ((++spins & 1023u) == 0) {
                hipError_t e = hipStreamQuery(ctx.stream);
                if (e != hipSuccess && e != hipErrorNotReady) RSQ_HIP(e);
                if (e == hipSuccess || nowMs() > deadline) { waitForStream(ctx); sawSequence = false; break; }
            }
            __builtin_ia32_pause();
        }
        if (*flag != selectSeq) failRuntime("internal error: the candidate selection finished without publishing its sequence number");
        std::atomic_thread_fence(std::memory_order_acquire);
    } else waitForStream(ctx);
    if (!sawSequence) resolveKernelTime(q);      // (the stream is done: reading the events costs nothing now)
    const double tSeen = execTraceOn ? nowMs() : 0;
    auto execTrace = [&]() {
        if (!execTraceOn) return;
        static double acc[4] = {0, 0, 0, 0}; static int n = 0;
        const double tEnd = nowMs();
        resolveKernelTime(q);
        acc[0] += tEnqueued - t0; acc[1] += tSeen - tEnqueued; acc[2] += tEnd - tSeen; acc[3] += q.report.kernel_time_ms;
        if (++n == 8) {
            fprintf(stderr, "[rsq exec] enqueue %.1f us, enqueued -> stream done %.1f us, checks + tail %.1f us; first to last event on the stream %.1f us, %llu launches\n",
                    acc[0] / 8 * 1e3, acc[1] / 8 * 1e3, acc[2] / 8 * 1e3, acc[3] / 8 * 1e3, (unsigned long long)q.report.num_kernels);
            acc[0] = acc[1] = acc[2] = acc[3] = 0; n = 0;
        }
    };
    if (anyCompaction)      // a build pipeline that also ran its counting pass reports both passes: only ever an over-estimate
        for (size_t i = 0; i < q.pipelines.size(); i++) {
            Pipeline& p = q.pipelines[i];           // the first 64 workgroups report: scale to the grid
            if (p.compact) p.stage2Rows = (int64_t)((double)q.hPinned[words + 8 + i] * (double)std::max(1u, p.lastGrid) / (double)std::min(64u, std::max(1u, p.lastGrid)));
        }
    ctx.errWordClean = (uint32_t)q.hPinned[words] == 0;
    if (((uint32_t)q.hPinned[words] & 256u) && !async && !q.fusedSelectOff) {
        // a workgroup of the one-launch candidate selection gave up at a meeting point (aot_kernels.hip k_topk_range_select): its
        // candidates are void; this query takes the separate launches from now on and starts over
        q.fusedSelectOff = true;
        executeQuery(q, partialOnly, async);
        return;
    }
    if (((uint32_t)q.hPinned[words] & 128u) && !async) {
        // a workgroup of the one-launch rank index gave up waiting for its predecessor (aot_kernels.hip k_rank_blocks_chained): the
        // index is wrong (in range, never out of bounds); this query takes the two-launch index from now on and starts over
        q.chainedIndexOff = true;
        executeQuery(q, partialOnly, async);
        return;
    }
    if (((uint32_t)q.hPinned[words] & 64u) && !async) {
        // a rank dictionary met build rows it was not sized for (two rows with one key, or more rows than the sizing pass saw:
        // the build side's data changed): its tables go back to the hash form and the execution starts over
        bool any = false;
        for (auto& hp : q.hashTables) {
            HashTable& h = *hp;
            if (!h.rank) continue;
            any = true;
            ctx.free(h.dWords); ctx.free(h.dTemp); ctx.free(h.dTempUsed); ctx.free(h.dChunkTotal); ctx.free(h.dChunkBase);
            if (h.dAcc) ctx.free(h.dAcc);
            h.dWords = h.dTemp = nullptr; h.dAcc = nullptr; h.dTempUsed = h.dChunkTotal = h.dChunkBase = nullptr;
            h.rank = false; h.rankCapable = false; h.identity = false; h.capacity = 0; h.lastCount = 0;
            h.dupKeys = true;                      // (what the plan memo tells the next query of this shape)
        }
        if (any) { executeQuery(q, partialOnly, async); return; }
    }
    if (q.matWarmRun) {             // the write pass ran with the remembered total: it must be this execution's
        q.matWarmRun = false;
        if ((int64_t)q.hPinned[words + 3] != q.matLastTotal) { q.matLastTotal = -1; executeQuery(q, partialOnly, async); return; }
    }
    if (hashWarm) {
        HashTable& h = *q.hashTables[(size_t)q.aggTable];
        const uint32_t e = (uint32_t)q.hPinned[words];
        if (e & 2u) {               // the table overflowed (more groups than last time): four times the slots, on the careful path
            ctx.free(h.dState); ctx.free(h.dWords); ctx.free(h.dAcc);
            h.dState = nullptr; h.dWords = nullptr; h.dAcc = nullptr;
            if (h.capacity >= ((int64_t)1 << 31)) failRuntime("Hash table full");
            h.capacity *= 4; h.lastCount = 0;
            executeQuery(q, partialOnly, async);
            return;
        }
        if (((e & 32u) != 0) != q.charGroupsNeedMerge) { h.lastCount = 0; executeQuery(q, partialOnly, async); return; }      // (the careful path reads the note before it plans the tail)
    }
    checkDeviceError((uint32_t)q.hPinned[words]);
    // the short candidate selection overflowed its buffer (many rows share the leading bin): the exact radix select, now
    bool candOnHost = selectPublished;      // the candidates are in hCandRows (else: in hGroupRows, copied)
    auto exactCandidates = [&](uint32_t rowsBound) -> int64_t {
        candOnHost = false;
        if (q.topkImageRows < rowsBound) {
            if (q.dTopkImages) ctx.free(q.dTopkImages);
            q.dTopkImages = (uint64_t*)ctx.alloc((size_t)rowsBound * 8);
            q.topkImageRows = rowsBound;
        }
        selectTopCandidates(ctx, q.dGroupRows, q.groupRowWords, q.topkWord, q.topkIs32, q.topkDesc, q.dGroupCount, rowsBound, q.topkWant, q.dTopkImages,
                            q.dTopkHists, q.dCandRows, topkCapacity);
        uint32_t n = 0;
        RSQ_HIP(hipMemcpyAsync(&n, q.dCandCount, 4, hipMemcpyDeviceToHost, ctx.stream));
        waitForStream(ctx);
        if (n <= topkCapacity) {
            ensureHostGroupRows(q, (size_t)topkCapacity * (size_t)q.groupRowWords);
            RSQ_HIP(hipMemcpy(q.hGroupRows, q.dCandRows, (size_t)n * (size_t)q.groupRowWords * 8, hipMemcpyDeviceToHost));
        }
        return (int64_t)n;
    };
    if (!partialOnly) {
        double t1 = nowMs();
        if (denseTopk) {
            const int64_t nGroups = (int64_t)(uint32_t)q.hPinned[words + 1];
            if (nGroups == 0xffffffffll) failRuntime("internal error: the group-row compaction's look-back timed out");
            int64_t nCand = (int64_t)(uint32_t)q.hPinned[words + 2];
            if (topkRange && nCand > (int64_t)topkCapacity) { nCand = exactCandidates(groupRowsAllocated); topkSpec = topkCapacity; }
            const size_t rowBytes = (size_t)q.groupRowWords * 8;
            bool done = false;
            if (nCand <= (int64_t)topkCapacity && nCand < nGroups) {
                if (nCand > (int64_t)topkSpec)
                    RSQ_HIP(hipMemcpy((char*)q.hGroupRows + (size_t)topkSpec * rowBytes, (char*)q.dCandRows + (size_t)topkSpec * rowBytes,
                                      (size_t)(nCand - topkSpec) * rowBytes, hipMemcpyDeviceToHost));
                q.candidateRun = true;
                q.nGroupRows = nCand; q.totalGroups = nGroups;
                q.hRowsView = candOnHost ? q.hCandRows : nullptr;
                tailUnlessHeld(q);
                q.hRowsView = nullptr;
                q.candidateRun = false;
                done = !q.tailNeedsAllGroups;
            }
            if (!done) {            // the candidates do not decide it: the whole table after all
                enqueueTableReadback(q);
                waitForStream(ctx);
                tableFromPinned(q);
                tailUnlessHeld(q);
            }
            q.report.finalize_time_ms = nowMs() - t1;
            q.report.execution_time_ms = nowMs() - t0;
            return;
        }
        if (devTail) {
            q.report.finalize_time_ms = runDenseDeviceTail(q);
            checkDeviceError((uint32_t)q.hPinned[words]);
            q.report.execution_time_ms = nowMs() - t0;
            return;
        }
        if (denseMode(q)) tableFromPinned(q);
        else if (q.matOp && !q.agg) {
            q.hMatCols.resize(q.matSchema.size());
            for (size_t c = 0; c < q.matSchema.size(); c++) {
                size_t bytes = (size_t)q.matRows * (size_t)columnWidth(q.matSchema[c].type);
                q.hMatCols[c].resize(bytes);
                if (bytes && c < q.hMatMapped.size() && q.hMatMapped[c]) memcpy(q.hMatCols[c].data(), q.hMatMapped[c], bytes);      // (written by the device, complete with the stream)
                else if (bytes) RSQ_HIP(hipMemcpy(q.hMatCols[c].data(), q.dMatCols[c], bytes, hipMemcpyDeviceToHost));
            }
        } else {
            const int64_t nGroups = (int64_t)(uint32_t)q.hPinned[words + 1];
            if (nGroups == 0xffffffffll) failRuntime("internal error: the group-row compaction's look-back timed out");
            if (nGroups > (int64_t)groupRowsAllocated) {
                // more groups than the remembered entry count provided for (the build side changed under us): start over
                q.hashTables[(size_t)q.aggTable]->lastCount = 0;
                if (q.aggMode != AggMode::AT_JOIN_ENTRY && !hashWarm) failRuntime("internal error: more groups than hash-table entries");
                executeQuery(q, partialOnly, async);
                return;
            }
            if (q.aggMode == AggMode::HASH) q.hashTables[(size_t)q.aggTable]->lastCount = (uint32_t)nGroups;      // (what the next execution provides for)
            int64_t nCand = topkCapacity ? (int64_t)(uint32_t)q.hPinned[words + 2] : 0;
            if (narrowRows && nCand > (int64_t)topkCapacity) { q.narrowRowsOff = true; executeQuery(q, partialOnly, async); return; }      // (the exact selection reads full rows)
            if (topkRange && nCand > (int64_t)topkCapacity) { nCand = exactCandidates(groupRowsAllocated); topkSpec = topkCapacity; }
            const size_t rowBytes = (size_t)q.groupRowWords * 8;
            q.candidateRun = false;
            if (topkCapacity && nCand <= (int64_t)topkCapacity && nCand < nGroups) {
                const bool postClear = true;
                if (postClear && candOnHost && !trace && !interp && !q.holdTail) {
                    // the candidates are on the host and decide the answer (or the whole table is read below from buffers the clears
                    // leave alone): ready the next execution now, while the host does its tail
                    fillBatchAsync(ctx, f.data(), (int)f.size());
                    q.report.num_kernels += (f.size() + 23) / 24;
                    q.readied = true; q.readiedEpoch = ctx.execEpoch; q.readiedFill = f;
                }
                if (nCand > (int64_t)topkSpec)
                    RSQ_HIP(hipMemcpy((char*)q.hGroupRows + (size_t)topkSpec * rowBytes, (char*)q.dCandRows + (size_t)topkSpec * rowBytes,
                                      (size_t)(nCand - topkSpec) * rowBytes, hipMemcpyDeviceToHost));
                q.candidateRun = true;
                q.nGroupRows = nCand; q.totalGroups = nGroups;
                q.hRowsView = candOnHost ? q.hCandRows : nullptr;
                tailUnlessHeld(q);
                q.hRowsView = nullptr;
                q.candidateRun = false;
            }
            if (!topkCapacity || q.tailNeedsAllGroups || !(nCand <= (int64_t)topkCapacity && nCand < nGroups)) {
                if (narrowRows) { q.narrowRowsOff = true; executeQuery(q, partialOnly, async); return; }      // (every row is needed: none was written)
                q.nGroupRows = nGroups;
                if (rowsDeviceTailWanted(q, nGroups)) runRowsDeviceTail(q, nGroups);      // many groups: the rows are made on the device
                else if (inlineRows && nGroups <= (int64_t)std::min<uint32_t>(kInlineRows, groupRowsAllocated)) {      // (the rows came with the status words)
                    q.hRowsView = q.hInlineRows;
                    tailUnlessHeld(q);
                    q.hRowsView = nullptr;
                }
                else {
                    if (q.nGroupRows) {
                        ensureHostGroupRows(q, (size_t)q.nGroupRows * (size_t)q.groupRowWords);
                        RSQ_HIP(hipMemcpy(q.hGroupRows, q.dGroupRows, (size_t)q.nGroupRows * rowBytes, hipMemcpyDeviceToHost));
                    }
                    tailUnlessHeld(q);
                }
            }
            q.report.finalize_time_ms = nowMs() - t1;
            q.report.execution_time_ms = nowMs() - t0;
            execTrace();
            return;
        }
        tailUnlessHeld(q);
        q.report.finalize_time_ms = nowMs() - t1;
    }
    q.report.execution_time_ms = nowMs() - t0;
    execTrace();
}

void finalizeQuery(Query& q) {
    Context& ctx = q.ctx;
    if (!denseMode(q)) failUnsupported("partial execution / finalize is available for dense aggregations only");
    RSQ_HIP(hipSetDevice(ctx.device));
    double t1 = nowMs();
    const bool devTail = !q.mergePublishedSeq && denseDeviceTailWanted(q);
    double devTailMs = 0;
    if (q.mergePublishedSeq) {
        // the merge kernel stored the merged table into host-mapped memory: watch for its sequence number (see the fused step)
        const uint64_t seq = q.mergePublishedSeq;
        q.mergePublishedSeq = 0;
        volatile uint64_t* flag = q.hPinned + q.pinnedWords + 3;
        const double deadline = nowMs() + 5.0;
        unsigned spins = 0;
        while (*flag != seq) {
            if ((++spins & 1023u) == 0) {
                hipError_t e = hipStreamQuery(ctx.stream);
                if (e != hipSuccess && e != hipErrorNotReady) RSQ_HIP(e);
                if (e == hipSuccess || nowMs() > deadline) { waitForStream(ctx); break; }
            }
            __builtin_ia32_pause();
        }
        if (*flag != seq) failRuntime("internal error: the merge kernel finished without publishing its table");
        std::atomic_thread_fence(std::memory_order_acquire);
    } else if (devTail) {
        devTailMs = runDenseDeviceTail(q);          // (waits for the step's kernels and its merge first)
    } else {
        RSQ_HIP(hipMemcpyAsync(q.hPinned, q.dAgg, q.tableWords * 8, hipMemcpyDeviceToHost, ctx.stream));
        waitForStream(ctx);
    }
    if (q.pendingAsync) {         // the step was enqueued by rsq_query_execute_partial_async: account for it now
        q.pendingAsync = false;
        float ms = 0;
        RSQ_HIP(hipEventElapsedTime(&ms, q.gev0, q.gev1)); q.kernelTimePending = false;      // (one-launch step or not: the query's own event pair)
        if (q.pendingFused) q.fusedReady = true;
        q.report.kernel_time_ms = ms; q.kernelTimeSumMs += ms; q.kernelTimeLaunches++;
        q.report.hbm_gbps = ms > 0 ? (double)q.report.bytes_read / (ms * 1e-3) / 1e9 : 0;
        if (!q.pendingFused) ctx.errWordClean = (uint32_t)q.hPinned[q.pinnedWords] == 0;
        checkAsyncDeviceError((uint32_t)q.hPinned[q.pinnedWords]);
    }
    if (devTail) { checkDeviceError((uint32_t)q.hPinned[q.pinnedWords]); q.report.finalize_time_ms = devTailMs; return; }
    t1 = nowMs();          // (the wait for the step's kernels and its merge is not the host tail)
    if (q.tableWords >= (1u << 16)) q.hAggView = q.hPinned;      // (see tableFromPinned)
    else { q.hAggView = nullptr; memcpy(q.hAgg.data(), q.hPinned, q.tableWords * 8); }
    runTail(q);
    q.report.finalize_time_ms = nowMs() - t1;
}

// a rank of a multi-GPU step that does not finalise (only the root reads the merged table back): wait for its enqueued
// step, account for the kernel time, check its device error word
void settleAsync(Query& q) {
    Context& ctx = q.ctx;
    if (!q.pendingAsync) return;
    RSQ_HIP(hipSetDevice(ctx.device));
    waitForStream(ctx);
    q.pendingAsync = false;
    float ms = 0;
    RSQ_HIP(hipEventElapsedTime(&ms, q.gev0, q.gev1)); q.kernelTimePending = false;
    if (q.pendingFused) q.fusedReady = true;
    q.report.kernel_time_ms = ms; q.kernelTimeSumMs += ms; q.kernelTimeLaunches++;
    q.report.hbm_gbps = ms > 0 ? (double)q.report.bytes_read / (ms * 1e-3) / 1e9 : 0;
    if (!q.pendingFused) ctx.errWordClean = (uint32_t)q.hPinned[q.pinnedWords] == 0;
    checkAsyncDeviceError((uint32_t)q.hPinned[q.pinnedWords]);
}

// Results of the same plan over disjoint shards (every group lives in exactly one shard: the caller shards on the group key)
// become one result in `into`: the rows of all parts in part order, then the root's ORDER BY (multi-key, typed compare as
// the reference's Quicksorter, types.h:264-353; ties keep part order) and its LIMIT (orderby.h:87-93; materialize.h:197-206).
void mergeShardResults(Query& into, const std::vector<Query*>& parts) {
    const Schema& sch = into.resultSchema;
    const size_t ts = (size_t)schemaTupleSize(sch);
    std::vector<uint8_t> all;
    int64_t rows = 0;
    for (Query* p : parts) {
        if ((size_t)schemaTupleSize(p->resultSchema) != ts) failInvalid("shard results have different schemas");
        all.insert(all.end(), p->resultTuples.begin(), p->resultTuples.begin() + (size_t)p->resultRows * ts);
        rows += p->resultRows;
    }
    OpNode* root = into.root;
    if (root && root->tag == RSQ_OP_ORDERBY) {
        std::vector<OrderRequest> reqs;
        for (Expr* e : root->exprs) {
            const std::string& nm = e->child->symbol;
            bool found = false;
            for (auto& a : sch) if (a.name == nm) { reqs.push_back({schemaOffset(sch, nm), a.type, e->tag != RSQ_E_DESC}); found = true; break; }
            if (!found) failType("Order By attribute not found.");
        }
        std::vector<int64_t> idx((size_t)rows);
        for (int64_t i = 0; i < rows; i++) idx[(size_t)i] = i;
        std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) {
            const uint8_t* l = all.data() + (size_t)a * ts; const uint8_t* r = all.data() + (size_t)b * ts;
            for (const auto& o : reqs) {
                int c = compareTyped(o.type, l + o.offset, r + o.offset);
                if (c) return o.asc ? c < 0 : c > 0;
            }
            return false;
        });
        if (root->hasLimit && rows > root->limit) rows = std::max<int64_t>(root->limit, 0);
        into.resultTuples.resize((size_t)rows * ts);
        for (int64_t i = 0; i < rows; i++) memcpy(&into.resultTuples[(size_t)i * ts], all.data() + (size_t)idx[(size_t)i] * ts, ts);
    } else {
        if (root && root->hasLimit) rows = std::min<int64_t>(rows, std::max<int64_t>(root->limit, 1));
        all.resize((size_t)rows * ts);
        into.resultTuples.swap(all);
    }
    into.resultRows = rows;
}

void setHoldTail(Query& q, bool hold) { q.holdTail = hold; }
bool queryOrderedWithLimit(const Query& q) { return q.root && q.root->tag == RSQ_OP_ORDERBY && q.root->hasLimit; }
bool queryAsyncCapable(const Query& q) {
    if (!denseMode(q)) return false;
    for (auto& p : q.pipelines) if (p.sink != SinkKind::AGGREGATE) return false;
    return true;
}

// Can a group of this plan occur in two shards?  Not if one of the group-by values is a plain attribute of a scanned table whose
// [min, max] ranges (column statistics of every shard's instance of that table) do not overlap between the shards: the caller
// sharded on a boundary of that key.  `why` names the attribute, or says what was missing.
bool shardGroupsDisjoint(const std::vector<Query*>& parts, std::string& why) {
    if (parts.empty() || !parts[0]->agg) { why = "no aggregation"; return false; }
    const Query& q0 = *parts[0];
    for (Expr* g : q0.agg->exprs2) {
        if (g->tag != RSQ_E_ATTRIBUTE) continue;
        std::vector<std::pair<int64_t, int64_t>> ranges;
        bool usable = true;
        for (size_t i = 0; i < parts.size() && usable; i++) {
            const Table* tab = nullptr; int col = -1;
            for (Table* t : parts[i]->tables) { int c = t->findCol(g->symbol); if (c >= 0) { tab = t; col = c; break; } }
            if (!tab) { usable = false; break; }
            if (tab->nRows == 0) continue;
            const ColumnStats& st = tab->shardStats((size_t)col);      // what THIS shard's rows hold (stats may be the union over all shards)
            if (!st.valid || !st.distinctBytes.empty() || tab->cols[(size_t)col].type.isString()) { usable = false; break; }
            ranges.emplace_back(st.min, st.max);
        }
        if (!usable) continue;
        std::sort(ranges.begin(), ranges.end());
        bool disjoint = true;
        for (size_t i = 1; i < ranges.size(); i++) if (ranges[i].first <= ranges[i - 1].second) disjoint = false;
        if (disjoint) { why = "group key " + g->symbol + " has disjoint value ranges on the shards"; return true; }
        why = "the value ranges of group key " + g->symbol + " overlap between shards";
    }
    if (why.empty()) why = "no group key is a plain integer attribute with column statistics";
    return false;
}

void finalizeQueryHost(Query& q, const int64_t* words, size_t nWords) {
    if (!denseMode(q)) failUnsupported("host finalize is available for dense aggregations only");
    size_t need = q.accums.size() * (size_t)q.denseGroups;
    if (nWords != need) failInvalid("partial table has " + std::to_string(nWords) + " words, expected " + std::to_string(need));
    q.hAgg.assign((const uint64_t*)words, (const uint64_t*)words + nWords);
    q.hAggView = nullptr;
    runTail(q);
}

void bindPartial(Query& q, void* dptr, size_t bytes) {
    if (!denseMode(q)) failUnsupported("partial tables exist for dense aggregations only");
    size_t need = q.accums.size() * (size_t)q.denseGroups * 8;
    if (!dptr || bytes < need) failInvalid("partial buffer too small: need " + std::to_string(need) + " bytes");
    if (q.dAgg && q.dAggOwned) q.ctx.free(q.dAgg);
    q.dAgg = (uint64_t*)dptr;
    q.dAggOwned = false;
    for (auto& p : q.pipelines) if (p.sink == SinkKind::AGGREGATE && p.src->nRowsTotal < 0) q.firstRowsForeign = true;      // (the caller's collectives write into it)
}

// the kernel behind the merge collective: `gathered` = every rank's partial table back to back (one all-gather), reduced by
// segment into this query's own partial table; enqueued on the context's stream, no synchronisation
void mergeGathered(Query& q, const void* gathered, int nRanks) {
    if (!denseMode(q)) failUnsupported("partial tables exist for dense aggregations only");
    if (!gathered || nRanks < 1) failInvalid("merge needs the gathered tables of at least one rank");
    if (q.ctx.device < 0) throw Error(RSQ_ERR_DEVICE, "this context has no device (compile-only)");
    RSQ_HIP(hipSetDevice(q.ctx.device));
    const int64_t G = q.denseGroups;
    if (nRanks > 1) for (auto& p : q.pipelines) if (p.sink == SinkKind::AGGREGATE && p.src->nRowsTotal < 0) q.firstRowsForeign = true;
    // small tables: the merge kernel also publishes the result to host-mapped memory (finalizeQuery polls for it)
    const bool pollOk = !(getenv("RSQ_POLL") && atoi(getenv("RSQ_POLL")) == 0);
    const bool publish = pollOk && q.dFinHost && q.tableWords <= 2048;
    q.mergePublishedSeq = publish ? ++q.finSeqCounter : 0;
    mergePartialsAsync(q.ctx, (const int64_t*)gathered, nRanks, (int64_t)q.tableWords, q.nMinBlocks * G, q.nMaxBlocks * G, q.nSumBlocks * G, (int64_t*)q.dAgg,
                       publish ? (int64_t*)q.dFinHost : nullptr, publish ? q.dFinHost + q.pinnedWords + 3 : nullptr, q.mergePublishedSeq);
}

void partialBuffer(Query& q, void** dptr, int64_t* nMin, int64_t* nMax, int64_t* nSum) {
    if (!denseMode(q)) failUnsupported("partial tables exist for dense aggregations only");
    *dptr = q.dAgg;
    *nMin = q.nMinBlocks * q.denseGroups;
    *nMax = q.nMaxBlocks * q.denseGroups;
    *nSum = q.nSumBlocks * q.denseGroups;
}

void queryResult(Query& q, rsq_result_view* out) {
    size_t n = q.resultSchema.size();
    q.rvNames.assign(n * RSQ_SYMBOL_MAX + 1, 0);
    q.rvTypes.resize(n); q.rvOffsets.resize(n);
    int off = 0;
    for (size_t i = 0; i < n; i++) {
        snprintf(&q.rvNames[i * RSQ_SYMBOL_MAX], RSQ_SYMBOL_MAX, "%s", q.resultSchema[i].name.c_str());
        q.rvTypes[i] = q.resultSchema[i].type.toC();
        q.rvOffsets[i] = off;
        off += sizeInTuple(q.resultSchema[i].type, true);
    }
    out->n_cols = (int32_t)n;
    out->names = reinterpret_cast<const char(*)[RSQ_SYMBOL_MAX]>(q.rvNames.data());
    out->types = q.rvTypes.data();
    out->offsets = q.rvOffsets.data();
    out->tuple_size = off;
    out->n_rows = q.resultRows;
    out->tuples = q.resultInPinned ? q.resultPinned : q.resultTuples.data();
}

void queryReport(const Query& q, rsq_report* out) {
    resolveKernelTime(const_cast<Query&>(q));
    *out = q.report;
}
void queryKernelTimeStats(Query& q, double* sumMs, uint64_t* executions, bool reset) {
    resolveKernelTime(q);
    if (sumMs) *sumMs = q.kernelTimeSumMs;
    if (executions) *executions = q.kernelTimeLaunches;
    if (reset) { q.kernelTimeSumMs = 0; q.kernelTimeLaunches = 0; }
}
bool queryIsDense(const Query& q) { return denseMode(q); }
void queryDenseLayout(const Query& q, int64_t* nMin, int64_t* nMax, int64_t* nSum, void** dptr) {
    if (!denseMode(q)) failUnsupported("partial tables exist for dense aggregations only");
    *dptr = q.dAgg;
    *nMin = q.nMinBlocks * q.denseGroups; *nMax = q.nMaxBlocks * q.denseGroups; *nSum = q.nSumBlocks * q.denseGroups;
}
std::string queryPartialLayoutText(const Query& q) {
    size_t at = q.explainText.find("partial table:");
    if (at == std::string::npos) return "";
    size_t end = q.explainText.find('\n', at);
    return q.explainText.substr(at, end == std::string::npos ? std::string::npos : end - at);
}
const char* querySource(const Query& q) { return q.allSource.c_str(); }
const char* queryExplain(const Query& q) { return q.explainText.c_str(); }
void destroyQuery(Query* q) { delete q; }

std::string serializeResultView(const rsq_result_view& v) {
    std::string out;
    for (int64_t r = 0; r < v.n_rows; r++) {
        const uint8_t* t = v.tuples + (size_t)r * (size_t)v.tuple_size;
        for (int c = 0; c < v.n_cols; c++) {
            Type ty = Type::fromC(v.types[c]);
            out += serializeSqlValue(loadValue(t + v.offsets[c], ty), ty) + "|";
        }
        out += "\n";
    }
    return out;
}

}  // namespace rsq
