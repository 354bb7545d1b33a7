// codegen.cpp — operator tree -> device pipelines (HIP source per pipeline).
//
// Mirrors the reference's produce/consume code generation (reference src/operators/*.h driven from
// src/execute.h:228): produce() walks down to the scans; every scan opens a pipeline (scan.h:227-263)
// and the operators above it consume() into the body of that pipeline's row function until a
// pipeline breaker ends it — hash-join build (hashjoin.h:226-256) or aggregation
// (aggregation.h:240-295).  The hand-written skeleton around the row function (tile loads,
// reductions, hash-table access) comes from kernels/rsq_device.h.
#include "codegen_internal.h"

namespace rsq {
namespace cg {

void Walker::addArg(const std::string& name, const std::string& ctype, uint64_t v) {
    for (auto& a : pipe.args) if (a.name == name) return;
    pipe.args.push_back({name, ctype, v});
}

// -------------------------------------------------------------------------------------------
void Walker::produce(OpNode* o, std::vector<std::string> request) {
    switch (o->tag) {
        case RSQ_OP_SCAN: produceScan(o, request); break;
        case RSQ_OP_SELECTION: {                    // selection.h:39-49
            o->schema.clear();
            std::vector<std::string> r = request;
            requiredAttributes(o->exprs[0], r);
            requestOf[o] = request;
            produce(o->child[0], r);
            break;
        }
        case RSQ_OP_PROJECTION: {                   // projection.h:40-59
            std::vector<std::string> r;
            for (Expr* e : o->exprs) requiredAttributes(e, r);
            produce(o->child[0], r);
            break;
        }
        case RSQ_OP_HASHJOIN: {                     // hashjoin.h:98-116
            requestOf[o] = request;
            std::vector<std::string> all = request;
            for (Expr* e : o->exprs) requiredAttributes(e, all);
            joinPhase[o] = 1;
            produce(o->child[0], all);              // build pipeline(s)
            joinPhase[o] = 2;
            produce(o->child[1], all);              // probe pipeline(s)
            break;
        }
        case RSQ_OP_AGGREGATION: {                  // aggregation.h:155-164
            if (q.agg) failUnsupported("more than one aggregation in a plan");
            q.agg = o;
            std::vector<std::string> r;
            for (Expr* e : o->exprs) requiredAttributes(e, r);
            for (Expr* e : o->exprs2) requiredAttributes(e, r);
            produce(o->child[0], r);
            break;                                   // everything above runs on the host (tail.cpp)
        }
        case RSQ_OP_MATERIALIZE: case RSQ_OP_ORDERBY:
            produce(o->child[0], request);
            break;
        default: failUnsupported("operator not supported by the GPU engine");
    }
}

Schema Walker::prune(const Schema& s, const std::vector<std::string>& req) {
    Schema r;
    for (auto& a : s) if (has(req, a.name)) r.push_back(a);
    return r;
}

// -------------------------------------------------------------------------------------------
void Walker::produceScan(OpNode* o, std::vector<std::string> request) {     // scan.h:221-263
    Table* t = o->table;
    if (q.requestAll) request.clear();
    pipe = Pipeline();
    pipe.src = t;
    // tuned on MI355X with TPC-H Q1 SF10 (profiles/): 1 tile in flight per wave + non-temporal loads
    // 0.395 ms; 2 tiles 0.42 ms; 4 tiles 0.47 ms (fewer resident waves); without nt loads 0.45-0.47 ms
    pipe.unroll = 1;
    pipe.blockThreads = 256;
    // workgroups (of 256 threads) per launch; 0 = 2 per CU.  Measured on MI355X (Q1 SF10): 512 workgroups 0.348 ms,
    // 768: 0.367, 1024: 0.374, 2048: 0.395, 4096: 0.448 - a streaming kernel wants exactly 2 resident workgroups per CU
    pipe.maxGrid = (unsigned)0;
    colTypes.clear(); colIsString.clear(); rowParams.clear(); rowArgsTail.clear(); rowArgsTailGuarded.clear(); bitmapPrefetch.clear();
    body.clear(); stateDecl.clear(); stateInit.clear(); prologue.clear(); epilogue.clear(); fileScope.clear(); helperFns.clear();
    explainSteps.clear(); indent = 1; matchSlotTable = -1; slotVar.clear(); symbolOrigin.clear(); symbolWord.clear();
    multiMatchAbove = false;
    selective = false; compacted = false; stage2Body.clear(); cqLive.clear();
    leadCond.clear(); leadCols.clear(); leadPass = 1.0; leadPassComplete = true;
    pairSplit = std::string::npos; pairCond.clear();
    strPrefetch.clear(); strPrefetchWidth.clear(); strStaged.clear(); strStagedBytes = 0; postTile.clear(); eg.strWordVars.clear();
    eg.symbols.clear();
    o->schema.clear();
    for (size_t ci = 0; ci < t->cols.size(); ci++) {
        const TableColumn& c = t->cols[ci];
        // Values::dematerialize(..., required): an empty request set means all attributes
        if (!request.empty() && !has(request, c.name)) continue;
        if (!c.dptr) {
            if (request.empty()) continue;   // declared without data: cannot be part of `select *`
            failInvalid("column " + c.name + " is needed by the plan but was declared without data");
        }
        int k = (int)pipe.cols.size();
        pipe.cols.push_back((int)ci);
        std::string var = "v_" + std::to_string(k);
        eg.symbols[c.name] = Sym{var, c.type};
        symbolOrigin[c.name] = -1;
        o->schema.push_back({c.name, c.type});
        pipe.bytesPerRow += columnWidth(c.type);
        if (c.type.isString()) {
            addArg("c" + std::to_string(k), "const char*", (uint64_t)(uintptr_t)c.dptr);
            colIsString.push_back(1); colTypes.push_back("");
            line("const rsq::Str " + var + " = rsq::str(a.c" + std::to_string(k) + " + lr * " + std::to_string(c.type.len) + ", " +
                 std::to_string(c.type.len) + ");");
        } else {
            std::string ct = ExprGen::ctype(c.type);
            addArg("c" + std::to_string(k), "const " + ct + "*", (uint64_t)(uintptr_t)c.dptr);
            colIsString.push_back(0); colTypes.push_back(ct);
            rowParams += ", " + ct + " " + var;
            rowArgsTail += ", a.c" + std::to_string(k) + "[r]";
            rowArgsTailGuarded += ", (valid ? a.c" + std::to_string(k) + "[r] : (" + ct + ")0)";
        }
    }
    explainSteps.push_back("scan " + t->name + " [" + std::to_string((long long)t->nRows) + " rows, " +
                           std::to_string((long long)pipe.bytesPerRow) + " B/row]");
    consume(o->parent, o);
    finishPipeline();
}

// ---- the selection directly above the scan ---------------------------------------------------------------------------------
// value range of one side of a comparison: a column of the scanned table (its statistics), a constant, or either under a cast
bool Walker::sideRange(const Expr* e, double& lo, double& hi, int& col) {
    if (e->tag == RSQ_E_TYPECAST && e->child) {
        if (!sideRange(e->child, lo, hi, col)) return false;
        const int ds = (e->type.tag == RSQ_DECIMAL ? e->type.scale : 0) - (e->child->type.tag == RSQ_DECIMAL ? e->child->type.scale : 0);
        const double f = std::pow(10.0, (double)ds);
        lo *= f; hi *= f;
        return true;
    }
    // DATE values are yyyymmdd integers: spread them evenly (12 x 31 days a year) before they are taken as uniform
    auto linear = [&](double v) -> double {
        if (e->type.tag != RSQ_DATE) return v;
        const int64_t d = (int64_t)v;
        return (double)((d / 10000) * 372 + ((d / 100) % 100 - 1) * 31 + (d % 100 - 1));
    };
    if (e->tag == RSQ_E_CONSTANT && !e->type.isString()) { lo = hi = linear((double)e->ival); col = -1; return true; }
    if ((e->tag == RSQ_E_ADD || e->tag == RSQ_E_SUB) && e->child && e->child->next && e->type.tag != RSQ_DATE) {      // constant arithmetic (0.06 - 0.01)
        double alo, ahi, blo, bhi; int ac = -1, bc = -1;
        if (!sideRange(e->child, alo, ahi, ac) || !sideRange(e->child->next, blo, bhi, bc) || ac >= 0 || bc >= 0) return false;
        const int s0 = e->type.tag == RSQ_DECIMAL ? e->type.scale : 0;
        const double fa = std::pow(10.0, (double)(s0 - (e->child->type.tag == RSQ_DECIMAL ? e->child->type.scale : 0)));
        const double fb = std::pow(10.0, (double)(s0 - (e->child->next->type.tag == RSQ_DECIMAL ? e->child->next->type.scale : 0)));
        lo = hi = e->tag == RSQ_E_ADD ? alo * fa + blo * fb : alo * fa - blo * fb; col = -1;
        return true;
    }
    if (e->tag == RSQ_E_ATTRIBUTE) {
        auto so = symbolOrigin.find(e->symbol);
        if (so == symbolOrigin.end() || so->second != -1) return false;
        const int ci = pipe.src->findCol(e->symbol);
        if (ci < 0 || !pipe.src->cols[(size_t)ci].stats.valid || pipe.src->cols[(size_t)ci].type.isString()) return false;
        lo = linear((double)pipe.src->cols[(size_t)ci].stats.min); hi = linear((double)pipe.src->cols[(size_t)ci].stats.max); col = ci;
        return true;
    }
    return false;
}

// expected fraction of rows a predicate passes; 1 (no claim) for whatever it does not understand
double Walker::passFraction(const Expr* e) {
    if (e->tag == RSQ_E_AND || e->tag == RSQ_E_OR) {
        double all = 1.0, none = 1.0;
        for (Expr* c : e->children()) { const double f = passFraction(c); all *= f; none *= 1.0 - f; }
        return e->tag == RSQ_E_AND ? all : 1.0 - none;
    }
    if (e->tag < RSQ_E_LT || e->tag > RSQ_E_NEQ || !e->child || !e->child->next) { leadPassComplete = false; return 1.0; }
    double alo, ahi, blo, bhi; int ac = -1, bc = -1;
    if (!sideRange(e->child, alo, ahi, ac) || !sideRange(e->child->next, blo, bhi, bc)) { leadPassComplete = false; return 1.0; }
    if ((ac >= 0) == (bc >= 0)) { leadPassComplete = false; return 1.0; }                      // column against constant only
    int tag = e->tag;
    if (ac < 0) {                                                // constant OP column -> column OP' constant
        std::swap(alo, blo); std::swap(ahi, bhi);
        tag = tag == RSQ_E_LT ? RSQ_E_GT : tag == RSQ_E_LE ? RSQ_E_GE : tag == RSQ_E_GT ? RSQ_E_LT : tag == RSQ_E_GE ? RSQ_E_LE : tag;
    }
    const double width = ahi - alo + 1.0, c = blo;
    double below = (c - alo) / width;                            // fraction of values < c
    below = std::min(1.0, std::max(0.0, below));
    const double at = (c >= alo && c <= ahi) ? 1.0 / width : 0.0;
    switch (tag) {
        case RSQ_E_LT: return below;
        case RSQ_E_LE: return std::min(1.0, below + at);
        case RSQ_E_GT: return std::max(0.0, 1.0 - below - at);
        case RSQ_E_GE: return 1.0 - below;
        case RSQ_E_EQ: return at;
        default: return 1.0 - at;
    }
}

void Walker::leadColumnsOf(const Expr* e, std::vector<int>& out, bool& ok) {
    if (e->tag == RSQ_E_ATTRIBUTE) {
        auto sy = eg.symbols.find(e->symbol);
        auto so = symbolOrigin.find(e->symbol);
        if (sy == eg.symbols.end() || so == symbolOrigin.end() || so->second != -1 || sy->second.var.compare(0, 2, "v_") != 0) { ok = false; return; }
        const int k = atoi(sy->second.var.c_str() + 2);
        if (k < 0 || k >= (int)colIsString.size() || colIsString[(size_t)k]) { ok = false; return; }
        if (std::find(out.begin(), out.end(), k) == out.end()) out.push_back(k);
        return;
    }
    for (Expr* c : e->children()) leadColumnsOf(c, out, ok);
}

void Walker::prefetchComparedStrings(const Expr* e) {
    if (1 == 0 || 1 == 0) return;
    if ((e->tag == RSQ_E_EQ || e->tag == RSQ_E_NEQ) && e->child && e->child->next) {
        const Expr* l = e->child; const Expr* r = e->child->next;
        const Expr* col = l->tag == RSQ_E_ATTRIBUTE && r->tag == RSQ_E_CONSTANT ? l : r->tag == RSQ_E_ATTRIBUTE && l->tag == RSQ_E_CONSTANT ? r : nullptr;
        if (!col || !l->type.isString() || !r->type.isString()) return;
        auto sy = eg.symbols.find(col->symbol);
        auto so = symbolOrigin.find(col->symbol);
        if (sy == eg.symbols.end() || so == symbolOrigin.end() || so->second != -1 || sy->second.var.compare(0, 2, "v_") != 0) return;
        const int k = atoi(sy->second.var.c_str() + 2);
        const int W = sy->second.type.len;
        if (k < 0 || k >= (int)colIsString.size() || !colIsString[(size_t)k] || W < 2) return;
        for (auto& sp : strPrefetch) if (sp.first == k) return;
        // (up to 16 bytes: the whole value; longer: its first word - most values differ there, and the line it sits in is on its
        // way when the row function asks for the rest)
        const bool stage = W <= 32 && strStagedBytes + 128 * W <= 128 * 40;
        if (stage) { strStaged[k] = strStagedBytes; strStagedBytes += 128 * W; }
        const int PW = stage || W <= 16 ? W : 8;
        strPrefetch.push_back({k, PW});
        strPrefetchWidth[k] = W;
        eg.strWordVars[sy->second.var] = (PW + 7) / 8;
        for (int w = 0; w * 8 < PW; w++) {
            const std::string ld = "rsq::ld_bytes<" + std::to_string(std::min(8, PW - w * 8)) + ">(a.c" + std::to_string(k) + " + r * " + std::to_string(W) + " + " + std::to_string(w * 8) + ")";
            rowParams += ", u64 " + sy->second.var + "_w" + std::to_string(w);
            rowArgsTail += ", " + ld;
            rowArgsTailGuarded += ", (valid ? " + ld + " : 0ull)";
        }
        return;
    }
    if (e->tag == RSQ_E_AND || e->tag == RSQ_E_OR) for (Expr* c : e->children()) prefetchComparedStrings(c);
}

void Walker::noteLeadingSelection(const Expr* e, const std::string& cond) {
    bool ok = true;
    std::vector<int> cols;
    leadColumnsOf(e, cols, ok);
    if (!ok || cols.empty()) return;
    leadCond = cond; leadCols = cols; leadPassComplete = true; leadPass = passFraction(e);
}

// -------------------------------------------------------------------------------------------
void Walker::consume(OpNode* o, OpNode* from) {
    if (!o) failInvalid("plan root must be a materializing operator");
    switch (o->tag) {
        case RSQ_OP_SELECTION: {                    // selection.h:52-70
            o->schema = from->schema;
            if (!q.requestAll) o->schema = prune(o->schema, requestOf[o]);
            q.pool.addId(o->exprs[0]);
            const bool wasSelective = selective;
            selective = true;
            {
                // (RSQ_STRING_PREFETCH=2, measurement: also behind probes, as long as no selection came before)
                if (from->tag == RSQ_OP_SCAN || (1 == 2 && !wasSelective)) prefetchComparedStrings(o->exprs[0]);
                const std::string cond = eg.emit(o->exprs[0]);
                if (from->tag == RSQ_OP_SCAN && leadCond.empty()) noteLeadingSelection(o->exprs[0], cond);
                if (from->tag == RSQ_OP_SCAN && indent == 1 && pairCond.empty() && !compacted) { pairSplit = body.size(); pairCond = cond; }
                openScope("if (" + cond + ") {");
            }
            explainSteps.push_back("selection " + serializeExpr(o->exprs[0]));
            consume(o->parent, o);
            closeScope();
            break;
        }
        case RSQ_OP_PROJECTION: {                   // projection.h:62-72
            Schema s;
            std::vector<std::pair<std::string, Sym>> defs;
            openScope("{");
            int k = 0;
            for (Expr* e : o->exprs) {
                q.pool.addId(e);
                std::string var = "p" + std::to_string((int)(size_t)o->exprs.size()) + "_" + std::to_string(k++) + "_" + std::to_string(indent);
                line("const " + ExprGen::ctype(e->type) + " " + var + " = " + eg.emit(e) + ";");
                defs.push_back({expressionName(e), Sym{var, e->type}});
                s.push_back({expressionName(e), e->type});
            }
            for (auto& d : defs) { eg.symbols[d.first] = d.second; symbolOrigin[d.first] = -2; }
            o->schema = s;
            consume(o->parent, o);
            closeScope();
            break;
        }
        case RSQ_OP_HASHJOIN:
            if (joinPhase[o] == 1) consumeBuild(o, from); else consumeProbe(o, from);
            break;
        case RSQ_OP_AGGREGATION: consumeAggregation(o, from); break;
        case RSQ_OP_MATERIALIZE: consumeMaterialize(o, from); break;
        default: failUnsupported("operator not supported by the GPU engine");
    }
}

// ---- wave-level selection compaction ---------------------------------------------------------
// A selective predicate (or a join's key bitmap) leaves few lanes of a wave alive, and everything after it — hash
// probes, inserts, HBM atomics — is a chain of dependent random accesses whose latency the wave pays for no matter
// how few lanes take part.  So the row function is cut at that point: stage 1 (scan, predicates, bitmap test) pushes
// the values the rest needs into a per-wave LDS queue (ballot + prefix popcount, no atomics), and stage 2 runs only
// when 64 rows are queued, with every lane busy.  TPC-H Q3's orders pipeline (9.7 % of the rows reach the probe) then
// pays for one probe / insert pass per ~5 tiles instead of two per tile.
// Not for pipelines that materialise (output positions depend on the scan order of each lane).
bool Walker::downstreamMaterializes(OpNode* o) {
    for (OpNode* p = o; p; p = p->parent) {
        if (p->tag == RSQ_OP_AGGREGATION) return false;
        if (p->tag == RSQ_OP_HASHJOIN && joinPhase[p] == 1) return false;
        if (p->tag == RSQ_OP_MATERIALIZE) return true;
    }
    return true;
}

bool Walker::compactThen(OpNode* o, const std::function<void()>& downstream) {
    if (compacted || !selective || !envInt("RSQ_COMPACT", 1, 0, 1) || downstreamMaterializes(o)) return false;
    // not inside the match loop of a join probed for all matches: the queue takes ONE entry per row function call, and a row
    // with several matches would keep only its last (found with a constant build key: every build row the same key)
    if (multiMatchAbove) return false;
    compacted = true;
    pipe.compact = true;
    cqLive.assign(eg.symbols.begin(), eg.symbols.end());
    // Late column loads: a scanned (non-string) column that stage 1 never looked at is needed only by the rows that reach
    // stage 2.  The kernel exists in two forms from one source: RSQ_LAZY 0 loads it with the tile and carries it in the
    // queue; RSQ_LAZY 1 leaves it out of the tile loads and stage 2 reads it by row index.  The engine picks the lazy
    // form when the previous execution sent fewer than 1/32 of the rows to stage 2 (TPC-H Q3: 1.6 % of lineitem need
    // l_extendedprice and l_discount, 16 of the 24 bytes per row) — gathers for a few rows beat streaming for all, but
    // only then: a row gathered costs a 64-byte request per column.
    std::vector<int> lazyOf(cqLive.size(), -1);
    if (1) {
        for (size_t k = 0; k < cqLive.size(); k++) {
            const std::string& var = cqLive[k].second.var;
            auto org = symbolOrigin.find(cqLive[k].first);
            if (org == symbolOrigin.end() || org->second != -1 || cqLive[k].second.type.isString()) continue;
            if (var.compare(0, 2, "v_") != 0) continue;
            bool used = false;                      // does the stage-1 text mention the variable?
            for (size_t pos = body.find(var); pos != std::string::npos && !used; pos = body.find(var, pos + 1)) {
                const size_t end = pos + var.size();
                const bool left = pos == 0 || !(isalnum((unsigned char)body[pos - 1]) || body[pos - 1] == '_');
                const bool right = end >= body.size() || !(isalnum((unsigned char)body[end]) || body[end] == '_');
                if (left && right) used = true;
            }
            if (!used) { lazyOf[k] = atoi(var.c_str() + 2); pipe.lazyCols.push_back(lazyOf[k]); }
        }
    }
    // ---- everything downstream goes into stage 2, which sees the carried values under the names q_<k> ----
    const std::string stage1 = body; const int stage1Indent = indent;
    body.clear(); indent = 1;
    stage2Prefix.clear(); compFilters.clear(); inStage2 = true;
    for (size_t k = 0; k < cqLive.size(); k++) eg.symbols[cqLive[k].first] = Sym{"q_" + std::to_string(k), cqLive[k].second.type};
    explainSteps.push_back("wave compaction");
    downstream();
    while (indent > 1) closeScope();
    inStage2 = false;
    const std::string down = body;
    // Component bitmaps of later joins (consumeProbe): the value is a column of this scan, so the test belongs in STAGE 1 - the column
    // then streams with the tiles instead of being gathered row by row by every survivor, and the rows it rejects never enter the
    // queue (TPC-H Q5: 9.1 M of 60 M lineitem rows find their order, 1.8 M of those a supplier in ASIA).  A column the test reads
    // is not loaded late.  Where the stage-1 name of the value is not at hand the test stands at the top of stage 2.
    std::string stage1Cond;
    for (auto& cf : compFilters) {
        std::string v1;
        for (size_t k = 0; k < cqLive.size(); k++)
            if (cqLive[k].first == cf.symbol && cqLive[k].second.var.compare(0, 2, "v_") == 0) {
                v1 = cqLive[k].second.var;
                if (lazyOf[k] >= 0) { pipe.lazyCols.erase(std::remove(pipe.lazyCols.begin(), pipe.lazyCols.end(), lazyOf[k]), pipe.lazyCols.end()); lazyOf[k] = -1; }
            }
        const std::string C = cf.table + "_c";
        std::string test = "rsq::bit_in(a." + C + "_bm, (u64)((i64)(" + (v1.empty() ? cf.stage2Var : v1) + ") - a." + C + "_bmmin), a." + C + "_bmbits)";
        // A small bitmap (the supplier keys of TPC-H Q5: 12 KB) is read for BOTH rows of the lane and every tile in flight with the
        // tile loads, like the first probe's key bitmap: tested inside the row function, the eight rows a lane handles per iteration
        // each wait for their own load (measured: the pipeline 323 us; the loads hit the L1, their latency does not overlap).
        bool already = false;
        for (auto& pf : bitmapPrefetch) already = already || pf.first == C;
        if (!v1.empty() && cf.bits <= (1 << 20) && !already) {
            const int col = atoi(v1.c_str() + 2);
            bitmapPrefetch.push_back({C, col, false});
            const std::string call = "rsq::bm_word(a." + C + "_bm, a." + C + "_bmmin, a." + C + "_bmbits, (i64)";
            rowParams += ", const u32 pf_" + C;
            rowArgsTail += ", " + call + "a.c" + std::to_string(col) + "[r])";
            rowArgsTailGuarded += ", (valid ? " + call + "a.c" + std::to_string(col) + "[r]) : 0u)";
            test = "rsq::bit_of_word(pf_" + C + ", (u64)((i64)(" + v1 + ") - a." + C + "_bmmin), a." + C + "_bmbits)";
        }
        if (!v1.empty()) stage1Cond += (stage1Cond.empty() ? "" : " && ") + test;
        else stage2Prefix += "    if (!" + test + ") return;      // no build row of " + cf.table + " has this key component: the row cannot reach the sink\n";
    }
    // only the values stage 2 really reads travel through the queue (a date that was only filtered on does not); the
    // late-loaded ones take the LAST slots, which exist in the RSQ_LAZY 0 form only: the lazy form's queues are
    // smaller, more workgroups fit a CU, and a latency-bound pipeline (tile load, then the key bitmap's L2 load) gets
    // twice the waves
    auto mentions = [](const std::string& text, const std::string& var) {
        for (size_t pos = text.find(var); pos != std::string::npos; pos = text.find(var, pos + 1)) {
            const size_t end = pos + var.size();
            const bool left = pos == 0 || !(isalnum((unsigned char)text[pos - 1]) || text[pos - 1] == '_');
            const bool right = end >= text.size() || !(isalnum((unsigned char)text[end]) || text[end] == '_');
            if (left && right) return true;
        }
        return false;
    };
    std::vector<int> slot(cqLive.size(), -1);
    int nSlots = 0;
    for (int pass = 0; pass < 2; pass++)
        for (size_t k = 0; k < cqLive.size(); k++)
            if ((lazyOf[k] >= 0) == (pass == 1) && mentions(down, "q_" + std::to_string(k))) slot[k] = nSlots++;
    int nLazySlots = 0;
    for (size_t k = 0; k < cqLive.size(); k++) if (slot[k] >= 0 && lazyOf[k] < 0) nLazySlots++;
    // drop lazy columns nobody reads from the list of late loads (their tile loads can go in both forms... keep it simple:
    // they stay eager in the RSQ_LAZY 0 form and are simply not loaded in the lazy one)
    body.clear(); indent = 1;
    for (size_t k = 0; k < cqLive.size(); k++) {
        if (slot[k] < 0) continue;
        const Type& t = cqLive[k].second.type;
        const std::string v = "q_" + std::to_string(k);
        const std::string carried = "const " + ExprGen::ctype(t) + " " + v + " = " + fromWord("qw_" + std::to_string(slot[k]), t) + ";";
        if (lazyOf[k] < 0) line(carried);
        else {
            body += "#if RSQ_LAZY\n";
            line("const " + ExprGen::ctype(t) + " " + v + " = a.c" + std::to_string(lazyOf[k]) + "[row - a.row0];");
            body += "#else\n";
            line(carried);
            body += "#endif\n";
        }
    }
    stage2Body = body + stage2Prefix + down;
    body = stage1; indent = stage1Indent;
    if (!stage1Cond.empty()) openScope("if (" + stage1Cond + ") {");
    line("cq_pass = true;");
    for (size_t k = 0; k < cqLive.size(); k++) {
        if (slot[k] < 0) continue;
        const std::string push = "cq_" + std::to_string(slot[k]) + " = " + toWord(cqLive[k].second.var, cqLive[k].second.type) + ";";
        if (lazyOf[k] < 0) line(push);
        else { body += "#if !RSQ_LAZY\n"; line(push); body += "#endif\n"; }
    }
    if (!stage1Cond.empty()) closeScope();
    pipe.compactWords = nSlots;
    pipe.compactWordsLazy = pipe.lazyCols.empty() ? nSlots : nLazySlots;
    return true;
}

// ---- materialisation of a pipeline without aggregation (materialize.h:78-220) ----------------
// The reference appends tuples in scan order.  On the device the same order is kept with two passes of the
// same pipeline: pass 1 counts the tuples every lane emits per 128-row tile, an exclusive scan turns the counts
// into output offsets, pass 2 writes each tuple to its final position (struct of arrays; the host packs
// ReSQL tuples from them).
void Walker::consumeMaterialize(OpNode* o, OpNode* from) {
    if (q.agg) failUnsupported("materialize inside an aggregation input");
    if (q.matOp) failUnsupported("more than one materialisation on the device");
    o->schema = from->schema;
    q.matOp = o;
    q.matSchema = o->schema;
    openScope("{");
    line("#if RSQ_PASS == 1");
    line("st.cnt++;");
    line("#else");
    line("const u64 pos = st.pos++;");
    openScope("if (pos < a.out_limit) {");
    int k = 0;
    for (auto& a : o->schema) {
        auto it = eg.symbols.find(a.name);
        if (it == eg.symbols.end()) failType("materialize: symbol " + a.name + " not found");
        const Type& t = it->second.type;
        std::string on = "o" + std::to_string(k++);
        if (t.isString()) {
            addArg(on, "char*", 0);
            line("for (int i = 0; i < " + std::to_string(t.len) + "; i++) a." + on + "[pos * " + std::to_string(t.len) + " + i] = rsq::str_at(" + it->second.var + ", i);");
        } else {
            addArg(on, ExprGen::ctype(t) + "*", 0);
            line("a." + on + "[pos] = " + it->second.var + ";");
        }
    }
    closeScope();
    line("#endif");
    closeScope();
    // cnt[tile * 64 + lane]: tuples the lane's two rows emit; tcnt[tile]: their sum, by the wave; toffs = exclusive scan of tcnt.  The
    // write pass finds a lane's first position as toffs[tile] + the wave's exclusive prefix over cnt: the scan runs over one count
    // per 128 ROWS, not per lane (TPC-H Q19 at SF10: 30 M lane counts, 0.27 ms of scan kernels behind a 1.13 ms count pass).
    addArg("cnt", "u32*", 0); addArg("tcnt", "u32*", 0); addArg("toffs", "const u64*", 0); addArg("out_limit", "u64", 0);
    stateDecl += "    u32 cnt = 0;\n    u64 pos = 0;\n";
    pipe.sink = SinkKind::MATERIALIZE;
    explainSteps.push_back("materialize " + std::to_string(o->schema.size()) + " column(s) in scan order (count / scan / write)");
}

}  // namespace cg


void buildPipelines(Query& q) {
    cg::Walker w(q);
    w.produce(q.root, {});
    if (!q.agg && !q.matOp) failInvalid("plan has neither an aggregation nor a materialisation");
}

}  // namespace rsq
