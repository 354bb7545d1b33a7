// codegen_agg.cpp - the aggregation sinks: dense group ids in registers / LDS / HBM (plain, staged, partitioned), aggregation at the join
// entry, hash aggregation with its LDS front table.  Reference: src/operators/aggregation.h.
#include "codegen_internal.h"

namespace rsq {
namespace cg {

// ---- aggregation (aggregation.h:240-295) ----------------------------------------------------
void Walker::collectAccumulators(OpNode* o) {
    for (Expr* g : o->exprs2) q.pool.addId(g);
    for (Expr* s : o->splitAgg) q.pool.addId(s);
    q.accums.clear(); q.splitToAccum.clear();
    // accumulator 0: first input row of the group (drives the reference's emission order)
    q.accums.push_back(Accum{RSQ_E_MIN, "#firstrow", "row", Type(RSQ_BIGINT), 2});
    for (Expr* s : o->splitAgg) {
        Accum ac; ac.kind = s->tag; ac.type = s->type;
        switch (s->tag) {
            case RSQ_E_COUNT: ac.key = "COUNT"; ac.input = eg.emit(s); ac.merge = 0; ac.inputExpr = nullptr; break;
            case RSQ_E_SUM:
                if (s->type.tag != RSQ_DECIMAL && s->type.tag != RSQ_BIGINT) failType("ADD code generation not implemented for datatype");
                ac.key = "SUM" + structuralKey(s->child); ac.input = eg.emit(s); ac.merge = 0; ac.inputExpr = s->child; break;
            case RSQ_E_MIN: case RSQ_E_MAX:
                if (s->type.tag != RSQ_DECIMAL && s->type.tag != RSQ_BIGINT && s->type.tag != RSQ_DATE)
                    failType("LESS_THAN code generation not implemented for datatype");
                ac.key = std::string(s->tag == RSQ_E_MIN ? "MIN" : "MAX") + structuralKey(s->child);
                ac.input = "((i64)(" + eg.emit(s) + "))"; ac.merge = s->tag == RSQ_E_MIN ? 2 : 3; ac.inputExpr = s->child; break;
            default: failType("Aggregation type not implemented in updateAggregates(..).");
        }
        int found = -1;
        for (size_t i = 1; i < q.accums.size(); i++) if (q.accums[i].key == ac.key) found = (int)i;
        if (found < 0) { q.accums.push_back(ac); found = (int)q.accums.size() - 1; }
        q.splitToAccum.push_back(found);
    }
    // word blocks ordered [min | max | sum] so that each segment reduces with ONE collective across GPUs
    q.accumSlot.assign(q.accums.size(), 0);
    int slot = 0;
    q.nMinBlocks = q.nMaxBlocks = 0;
    for (int m : {2, 3, 0}) {
        for (size_t i = 0; i < q.accums.size(); i++) if (q.accums[i].merge == m) q.accumSlot[i] = slot++;
        if (m == 2) q.nMinBlocks = slot; else if (m == 3) q.nMaxBlocks = slot - q.nMinBlocks;
    }
    q.nSumBlocks = (int64_t)q.accums.size() - q.nMinBlocks - q.nMaxBlocks;
}

bool Walker::tryDenseKeys(OpNode* o) {
    Table* t = pipe.src;
    q.denseKeys.clear();
    int64_t total = 1;
    for (Expr* g : o->exprs2) {
        if (g->tag != RSQ_E_ATTRIBUTE) return false;
        auto org = symbolOrigin.find(g->symbol);
        if (org == symbolOrigin.end() || org->second != -1) return false;      // not a column of this pipeline's scan
        int ci = t->findCol(g->symbol);
        if (ci < 0 || !t->cols[(size_t)ci].dptr) return false;
        const TableColumn& c = t->cols[(size_t)ci];
        if (c.type.isString()) return false;               // string keys: generic hash aggregation (bytes as key words)
        DenseKey k; k.expr = g; k.type = c.type;
        // (an empty SHARD of a table plans with the statistics of the whole table, like every other shard: Table::nRowsTotal)
        if (t->nRows == 0 && !c.stats.valid) { k.card = 1; k.min = 0; }      // empty input: no row reaches the aggregation
        else if (!c.stats.valid) return false;
        else if (!c.stats.distinctBytes.empty()) { k.byteSet = true; k.values = c.stats.distinctBytes; k.card = (int64_t)k.values.size(); }
        else {
            if (c.type.isString()) return false;
            k.min = c.stats.min;
            unsigned __int128 range = (unsigned __int128)((__int128)c.stats.max - (__int128)c.stats.min) + 1;
            if (range > (unsigned __int128)(1u << 24)) return false;
            k.card = (int64_t)range;
        }
        if (total > (int64_t)(1 << 24) / k.card) return false;
        total *= k.card;
        q.denseKeys.push_back(k);
    }
    int64_t stride = 1;
    for (size_t i = q.denseKeys.size(); i-- > 0;) { q.denseKeys[i].stride = stride; stride *= q.denseKeys[i].card; }
    q.denseGroups = total;
    return true;
}

bool Walker::tryJoinEntry(OpNode* o) {
    // Every group-by value is determined by the entry matched by a single-match probe of this pipeline
    // (a build-side payload value, or the probe key that equals the build key): the group IS the entry,
    // and the aggregates can live beside it.  (TPC-H Q3: l_orderkey = o_orderkey, o_orderdate,
    // o_shippriority all hang off the matched orders entry.)
    if (matchSlotTable < 0 || o->exprs2.empty()) return false;
    q.groupSource.clear();
    for (Expr* g : o->exprs2) {
        if (g->tag != RSQ_E_ATTRIBUTE) return false;
        if (g->type.isString()) return false;      // string keys go through the generic table (bytes as key words)
        auto org = symbolOrigin.find(g->symbol);
        if (org != symbolOrigin.end() && org->second == matchSlotTable) { q.groupSource.push_back(symbolWord[g->symbol]); continue; }
        auto pk = probeKeyOf.find(g->symbol);
        if (pk != probeKeyOf.end() && pk->second.first == matchSlotTable) { q.groupSource.push_back(pk->second.second); continue; }
        return false;
    }
    // a probe key counts only if ALL keys of the table are covered, otherwise two entries could share the group
    HashTable& ht = *q.hashTables[(size_t)matchSlotTable];
    for (size_t kw = 0; kw < ht.keys.size(); kw++)
        if (std::find(q.groupSource.begin(), q.groupSource.end(), (int)kw) == q.groupSource.end()) return false;
    q.aggTable = matchSlotTable;
    return true;
}

void Walker::consumeAggregation(OpNode* o, OpNode* from) {
    (void)from;
    collectAccumulators(o);
    const int W = (int)q.accums.size();
    std::string mode;
    const int forced = envInt("RSQ_AGG_MODE", 0, 0, 5);     // 5 = generic hash aggregation even where a dense id exists (tests)
    if (!(forced == 5 && !o->exprs2.empty()) && tryDenseKeys(o)) {
        const int64_t D = q.denseGroups, cells = D * W;
        // measured on MI355X (Q1 SF10, 42 cells): registers 0.47 ms, lane-private LDS 0.71 ms
        if ((cells <= 64 && forced == 0) || forced == 1) { q.aggMode = AggMode::DENSE_REG; if (cells > 64) failUnsupported("too many groups for register accumulators"); }
        else if ((cells <= 56 && forced == 0) || forced == 2) { q.aggMode = AggMode::DENSE_LDS_PRIVATE; if (cells > 56) failUnsupported("too many groups for lane-private LDS accumulators"); }
        else if ((cells <= 6144 && forced == 0) || forced == 3) { q.aggMode = AggMode::DENSE_LDS_SHARED; if (cells > 6144) failUnsupported("too many groups for an LDS table"); }
        else q.aggMode = AggMode::DENSE_GLOBAL;
        // The HBM-table forms could sit behind the compaction too; measured (200 M rows, 2^20 groups): 3 % faster at 1 %
        // selectivity, 15-25 % SLOWER at 10 / 50 % (the count / scatter passes pay for the queue without needing it),
        // so it stays off unless asked for.
        if (!(q.aggMode == AggMode::DENSE_GLOBAL && 0 &&
              compactThen(o, [&] { collectAccumulators(o); emitDenseAggregation(o); })))
            emitDenseAggregation(o);
    } else if (forced != 5 && tryJoinEntry(o)) {
        q.aggMode = AggMode::AT_JOIN_ENTRY;
        emitJoinEntryAggregation(o);
    } else {
        q.aggMode = AggMode::HASH;
        // (behind the compaction the accumulator inputs must be emitted again: they name stage-2 values now)
        if (!compactThen(o, [&] { collectAccumulators(o); emitHashAggregation(o); })) emitHashAggregation(o);
    }
    pipe.sink = SinkKind::AGGREGATE;
}

// Generic hash aggregation (computed keys, wide domains): an open-addressing table in HBM keyed by the group
// values, insert-or-find inside the kernel, aggregates beside the entries.  Slot protocol: state 0 empty ->
// CAS to 1 (being written) -> keys stored -> fence -> 2 (ready); a lane that loses the CAS or meets state 1
// looks at the slot again (see the note at the loop about keeping this safe inside one wave).
void Walker::emitHashAggregation(OpNode* o) {
    pipe.gridPerCU = 8;
    std::unique_ptr<HashTable> ht(new HashTable());
    ht->id = (int)q.hashTables.size();
    const std::string T = "ht" + std::to_string(ht->id);
    const int W = (int)q.accums.size();
    ht->nAccBlocks = W;
    std::vector<std::string> keyVars;
    std::vector<std::pair<size_t, size_t>> charKeyWords;      // per CHAR(n) group value: [first, last] key word
    openScope("{");
    int k = 0;
    // ---- group values that are functions of other group values -------------------------------------------------------
    // A single-match probe hands every row with the same probe key the same entry, so the payload values of that entry are
    // functions of the key.  If the group-by list holds the key of such a table (its build-side key attributes, or the probe-
    // side attributes equal to them) — or the table is probed with values that are themselves determined this way — the other
    // group values taken from its entry cannot tell two groups apart: they are CARRIED (stored once, when the group is
    // created) instead of hashed and compared.  A probe for all matches does the same whenever its table is a bitmap-rank
    // dictionary in this execution (a.htN_rank: the build keys proved unique — the planner's list of unique attributes,
    // planner.h:218-241, misses c_custkey): the kernel then takes the short comparison, and the full one when the table fell
    // back to the hash form.  TPC-H Q10 groups by c_custkey and six more values hanging off the customer and nation entries,
    // 32 key words of which 31 are carried.  The group rows, and so the result, are the same.  RSQ_GROUP_FD=0 compares every
    // value as before.
    std::vector<bool> carried(o->exprs2.size(), false);
    std::string fdCond;                      // run-time condition of the dependencies ("" = they always hold)
    std::vector<int> condTablesOfFd;         // ... the tables that must be rank dictionaries for it
    if (1) {
        std::set<std::string> groupSyms, determined;
        for (Expr* g : o->exprs2) if (g->tag == RSQ_E_ATTRIBUTE) groupSyms.insert(g->symbol);
        determined = groupSyms;
        std::set<int> detTables, condTables; std::set<std::string> covers;
        for (bool changed = true; changed;) {
            changed = false;
            for (auto& ps : probesInScope) {
                if ((!ps.single && !ps.rankCapable) || detTables.count(ps.table)) continue;
                HashTable& bt = *q.hashTables[(size_t)ps.table];
                bool viaProbeKeys = !ps.keySymbols.empty();
                for (auto& ks : ps.keySymbols) viaProbeKeys = viaProbeKeys && !ks.empty() && determined.count(ks);
                bool viaCover = !bt.keys.empty();
                std::vector<std::string> cv;
                for (size_t kw = 0; kw < bt.keys.size() && viaCover; kw++) {
                    std::string hit;
                    for (auto& gs : groupSyms) {
                        auto org = symbolOrigin.find(gs);
                        if (org != symbolOrigin.end() && org->second == ps.table && symbolWord[gs] == (int)kw) { hit = gs; break; }
                        auto pk = probeKeyOf.find(gs);
                        if (pk != probeKeyOf.end() && pk->second.first == ps.table && pk->second.second == (int)kw) { hit = gs; break; }
                    }
                    if (hit.empty()) viaCover = false; else cv.push_back(hit);
                }
                if (!viaProbeKeys && !viaCover) continue;
                detTables.insert(ps.table);
                if (!ps.single) condTables.insert(ps.table);
                if (!viaProbeKeys) covers.insert(cv.begin(), cv.end());
                for (auto& so : symbolOrigin) if (so.second == ps.table) determined.insert(so.first);
                changed = true;
            }
        }
        size_t kept = 0;
        for (size_t i = 0; i < o->exprs2.size(); i++) {
            Expr* g = o->exprs2[i];
            if (g->tag != RSQ_E_ATTRIBUTE) { kept++; continue; }
            auto org = symbolOrigin.find(g->symbol);
            carried[i] = org != symbolOrigin.end() && org->second >= 0 && detTables.count(org->second) && !covers.count(g->symbol);
            if (!carried[i]) kept++;
        }
        if (kept == 0) std::fill(carried.begin(), carried.end(), false);      // (cannot happen: a chain of dependencies ends in a kept value)
        for (int t : condTables) fdCond += (fdCond.empty() ? "" : " && ") + std::string("a.ht") + std::to_string(t) + "_rank != 0";
        condTablesOfFd.assign(condTables.begin(), condTables.end());
    }
    bool anyCarried = false;
    for (bool c : carried) anyCarried = anyCarried || c;
    q.groupSource.assign(o->exprs2.size(), 0);
    for (size_t gi = 0; gi < o->exprs2.size(); gi++) {
        if (carried[gi]) continue;
        Expr* g = o->exprs2[gi];
        const size_t w0 = keyVars.size();
        q.groupSource[gi] = (int)w0;                // first table word of this group value
        for (auto& kv : keyWords(g, T + "_g" + std::to_string(k++), false)) keyVars.push_back(kv);
        if (g->type.tag == RSQ_CHAR && g->type.len > 1) charKeyWords.push_back({w0, keyVars.size() - 1});
        for (size_t w = w0; w < keyVars.size(); w++)
            ht->keys.push_back({w == w0 ? expressionName(g) : expressionName(g) + "#" + std::to_string(w - w0), w == w0 && !g->type.isString() ? g->type : Type(RSQ_BIGINT)});
    }
    if (keyVars.empty()) failUnsupported("hash aggregation without group keys");
    const int K = (int)keyVars.size();
    // carried values: their words follow the key words in the table (HashTable::payload), written by the lane that creates the group
    struct Carried { Expr* g; std::string var, ctype; int firstWord, nWords; };
    std::vector<Carried> carriedVals;
    {
        int cw = K;
        for (size_t gi = 0; gi < o->exprs2.size(); gi++) {
            if (!carried[gi]) continue;
            Expr* g = o->exprs2[gi];
            const int nw = g->type.isString() ? (g->type.len + 7) / 8 : 1;
            q.groupSource[gi] = cw;
            carriedVals.push_back({g, eg.symbols[g->symbol].var, ExprGen::ctype(g->type), cw, nw});
            for (int w = 0; w < nw; w++)
                ht->payload.push_back({w == 0 ? expressionName(g) : expressionName(g) + "#" + std::to_string(w), w == 0 && !g->type.isString() ? g->type : Type(RSQ_BIGINT)});
            cw += nw;
        }
    }
    for (int w = 1; w < W; w++) line("const i64 in" + std::to_string(w) + " = " + q.accums[(size_t)w].input + ";");
    addArg(T + "_state", "u32*", 0); addArg(T + "_words", "i64*", 0); addArg(T + "_cap", "u64", 0); addArg(T + "_count", "u32*", 0);
    addArg(T + "_acc", "u64*", 0);

    // A group's words next to each other (words[slot][w]) when it has several: creating a group with 32 words is then a few
    // cache lines instead of 32 stores a table-length apart (TPC-H Q10 at SF10: 380 K new groups per execution).
    int NWtab = K;
    for (auto& c : carriedVals) NWtab += c.nWords;
    ht->aos = NWtab > 1 && 1 != 0;
    const bool aggAos = ht->aos;
    // ... and while the group values that depend on the key stand in the entries by address (below), an entry is its key words and ONE word
    // per carried value: the kernel then addresses the table with that stride (T_nw; TPC-H Q10: 7 words instead of 32 - the million slots
    // the groups spread over are 56 MB, not 268), and with the full stride when a table of the dependency chain fell back to its hash form
    const int NWc = K + (int)carriedVals.size();
    const bool compactLayout = aggAos && NWc < NWtab && !carriedVals.empty() && envInt("RSQ_GROUP_VALUES_BY_ADDRESS", 1, 0, 1) != 0;
    auto aggWord = [&, NWtab, aggAos, compactLayout](int w) {
        if (aggAos && compactLayout) return "a." + T + "_words[" + T + "_s * " + T + "_nw + " + std::to_string(w) + "]";
        return aggAos ? "a." + T + "_words[" + T + "_s * " + std::to_string(NWtab) + " + " + std::to_string(w) + "]"
                      : "a." + T + "_words[" + std::to_string(w) + " * a." + T + "_cap + " + T + "_s]";
    };

    // ---- LDS front table (per workgroup) ------------------------------------------------------------------------
    // Direct-mapped slots {state, key words, accumulators} in LDS (256 .. 1024, by their size) in front of the HBM table: a row whose group
    // already owns its slot is aggregated with LDS atomics and never leaves the CU; a row that finds the slot taken by
    // another group, or still being written, goes to the HBM table as before (no waiting, so no wave can block
    // itself).  At the end of the kernel every occupied slot is merged into the HBM table by the same upsert.  With
    // few groups (TPC-H Q12: 2, Q5: 5) nearly every row stays in LDS; with many, nearly every row pays one LDS probe.
    const int slotBytes = 8 * (K + W) + 4;
    int LS = 0;
    if (LS == 0) LS = slotBytes * 1024 <= 48 * 1024 ? 1024 : slotBytes * 512 <= 48 * 1024 ? 512 : 256;
    while (LS & (LS - 1)) LS &= LS - 1;           // power of two
    const bool lds = 1 && LS >= 64 && slotBytes * LS <= 48 * 1024 && !anyCarried;     // (a front-table slot holds no carried values to create its group with)
    if (lds) {
        stateDecl += "    u32* lc_state;\n    i64* lc_key;\n    u64* lc_acc;\n";
        prologue += "    __shared__ u32 s_lc_state[" + std::string("RSQ_LC_SLOTS") + "];\n    __shared__ i64 s_lc_key[" + std::to_string(K) + " * RSQ_LC_SLOTS" +
                    "];\n    __shared__ u64 s_lc_acc[" + std::to_string(W) + " * RSQ_LC_SLOTS];\n";
        prologue += "    st.lc_state = s_lc_state; st.lc_key = s_lc_key; st.lc_acc = s_lc_acc;\n";
        prologue += "    for (int i = threadIdx.x; i < RSQ_LC_SLOTS; i += blockDim.x) s_lc_state[i] = 0u;\n    __syncthreads();\n";
        pipe.extraLdsBytes += (8 * (K + W) + 4) * LS;
        pipe.ldsSlots = LS; pipe.ldsSlotBytes = 8 * (K + W) + 4;
        // (the slot count is a macro: the engine compiles the same text with a handful of slots once it knows that the query has a handful of
        // groups - TPC-H Q12: 2, Q5: 5 -, and the table no longer costs the scan its occupancy: engine.cpp launchPipeline)
        fileScope += "#ifndef RSQ_LC_SLOTS\n#define RSQ_LC_SLOTS " + std::to_string(LS) + "\n#endif\n";
        // flush (before the entry counter's flush below: the upserts count new entries)
        std::string f = "    __syncthreads();\n    for (int i = threadIdx.x; i < RSQ_LC_SLOTS; i += blockDim.x) {\n";
        f += "        if (st.lc_state[i] == 2u) " + T + "_upsert(a, st, (i64)st.lc_acc[i]";
        for (int i = 0; i < K; i++) f += ", st.lc_key[" + std::to_string(i) + " * RSQ_LC_SLOTS + i]";
        for (int w = 1; w < W; w++) f += ", (i64)st.lc_acc[" + std::to_string(w) + " * RSQ_LC_SLOTS + i]";
        f += ");\n    }\n";
        epilogue += f;
    }
    countPerThread(T);

    // ---- the HBM table's insert-or-find + update, as a function of (first row, key words, accumulator inputs) ---
    // Slot protocol: state 0 empty -> CAS to 1 (being written) -> keys stored -> fence -> 2 (ready); a lane that
    // loses the CAS or meets state 1 looks at the slot again.
    {
        std::string savedBody = body; const int savedIndent = indent;
        body.clear(); indent = 1;
        std::vector<std::string> kp;
        for (int i = 0; i < K; i++) kp.push_back("k" + std::to_string(i));
        line("const u64 " + T + "_mask = a." + T + "_cap - 1;");
        line("u64 " + T + "_s = " + hashOf(kp) + " & " + T + "_mask;");
        // Insert-or-find, written so that it cannot deadlock inside a wave: the lane that wins the CAS writes the keys and
        // publishes state 2 in a plain if-block that is followed by code every lane runs (the reload), so the publish
        // stays inside the loop body.  (With `if (won) {publish; hit} if (!hit) continue; ...; break;` the compiler threads
        // the winner straight to the loop exit, the structurizer parks it there until the whole wave has left the loop,
        // and the losers of the same wave spin on a slot that is never published.)
        // carried group values: their words are needed by the lane that creates a group — and, while the dependencies are not
        // certain (a table of the chain is in its hash form), by every lane for the full comparison
        int nCarriedWords = 0;
        for (auto& c : carriedVals) nCarriedWords += c.nWords;
        // (strings by address: RSQ_GROUP_VALUES_BY_ADDRESS=0 copies them into the entries as before)
        bool byAddress = nCarriedWords > 0 && envInt("RSQ_GROUP_VALUES_BY_ADDRESS", 1, 0, 1) != 0;
        for (auto& c : carriedVals) if (c.g->type.isString() && (NWtab > 255 || c.g->type.len >= 4096)) byAddress = false;      // (what entryDerefCode can say)
        if (byAddress) {
            // row word w of the table part of a group row <- where it stands in the entry (engine.h entryDerefCode / entryPlainCode; 0: word w itself)
            ht->derefCodes.assign((size_t)NWtab, 0);
            for (size_t ci = 0; ci < carriedVals.size(); ci++) {
                const Carried& c = carriedVals[ci];
                const int src = compactLayout ? K + (int)ci : c.firstWord;
                for (int w = 0; w < c.nWords; w++)
                    ht->derefCodes[(size_t)(c.firstWord + w)] = c.g->type.isString() ? entryDerefCode(src, 8 * w, std::min(8, c.g->type.len - 8 * w))
                                                                                       : (src != c.firstWord + w ? entryPlainCode(src) : 0);
            }
            ht->derefCondTables = condTablesOfFd;
            ht->compactStride = compactLayout ? NWc : NWtab;
        }
        if (nCarriedWords) {
            line("const bool " + T + "_fd = " + (fdCond.empty() ? std::string("true") : fdCond) + ";");
            if (compactLayout) line("const u64 " + T + "_nw = " + (byAddress ? T + "_fd ? " + std::to_string(NWc) + "ull : " : std::string()) + std::to_string(NWtab) + "ull;");

        }
        line("u64 " + T + "_adv = 0; u32 " + T + "_spin = 0; bool " + T + "_found = false;");
        line("if (rsq::ld_agent(a.err) & (u32)rsq::ERR_HT_FULL) return;      // another lane found the table too small: this run is void");
        openScope("for (;;) {");
        line("u32 stt = rsq::ld_agent(&a." + T + "_state[" + T + "_s]);");
        openScope("if (stt == 0u) {");
        openScope("if (atomicCAS(&a." + T + "_state[" + T + "_s], 0u, 1u) == 0u) {");
        for (int i = 0; i < K; i++)
            line("rsq::st_agent(&" + aggWord(i) + ", " + kp[(size_t)i] + ");");
        // the carried group values of the new group (written by the lane that creates it; compared only in the full form)
        if (nCarriedWords) {
            // (with the dependencies certain the words go from their loads straight into the table, value by value: staged in
            // the array first, 31 words of TPC-H Q10's group values were 62 more live VGPRs - the kernel held 163 and ran three
            // waves per SIMD)
            // While the dependencies hold (T_fd) nobody READS the carried words inside this kernel - they are not compared, and the group
            // rows are gathered by the next kernel -, so they are PLAIN stores: the compiler merges neighbouring words into 16-byte
            // stores and nothing waits for a write-through to be acknowledged word by word.  Agent-scope stores only in the full form,
            // where other lanes compare them.  TPC-H Q10 at SF10 (380 K new groups of 31 carried words): the pipeline 559 -> 419 us.
            // ... and a STRING among them is not copied at all then: its address goes into the value's first word (the bytes stay in the
            // column they came from, immutable while the query runs), and the kernels that make group rows for the host rebuild the words
            // for the rows they deliver (aot_kernels.hip table_word, HashTable::derefCodes) - TPC-H Q10 at SF10 creates 380 K groups of which
            // the statement wants 20: 31 words gathered from five strings and stored per new group were 100 of the pipeline's 320 us.
            auto storeCarried = [&](bool plain, const std::string& tag) {
                for (size_t ci = 0; ci < carriedVals.size(); ci++) {
                    const Carried& c = carriedVals[ci];
                    if (plain && byAddress) {
                        const int dst = compactLayout ? K + (int)ci : c.firstWord;
                        if (c.g->type.isString()) { line(aggWord(dst) + " = (i64)(u64)(" + c.var + ").p;"); continue; }
                        if (compactLayout) {
                            openScope("{");
                            std::vector<std::string> one = keyWords(c.g, T + "_n" + tag + std::to_string(ci), false);
                            line(aggWord(dst) + " = " + one[0] + ";");
                            closeScope();
                            continue;
                        }
                    }
                    openScope("{");
                    std::vector<std::string> words = keyWords(c.g, T + "_n" + tag + std::to_string(ci), false);
                    for (int w = 0; w < c.nWords; w++)
                        line(plain ? aggWord(c.firstWord + w) + " = " + words[(size_t)w] + ";" : "rsq::st_agent(&" + aggWord(c.firstWord + w) + ", " + words[(size_t)w] + ");");
                    if (!plain && c.g->type.tag == RSQ_CHAR && c.g->type.len > 1) {
                        // (in the full form a carried CHAR(n) value tells groups apart like a key word does: one that ends with a space may
                        // equal another group's value in the reference's sense, and the host has to be told - see the key words' note below.
                        // Found by a random plan, seed 30102: 'MAIL' and 'MAIL  ' from two build rows of one key came out as two groups.)
                        std::string last = words[0];
                        for (int w = 1; w < c.nWords; w++) last = "(" + words[(size_t)w] + " != 0 ? " + words[(size_t)w] + " : " + last + ")";
                        line("if (rsq::top_byte_is_space(" + last + ")) atomicOr(a.err, (u32)rsq::NOTE_CHAR_GROUP_ENDS_WITH_SPACE);");
                    }
                    closeScope();
                }
            };
            if (fdCond.empty()) storeCarried(true, "");
            else {
                openScope("if (" + T + "_fd) {");
                storeCarried(true, "p");
                closeScope();
                openScope("else {");
                storeCarried(false, "");
                closeScope();
            }
        }
        // The key (and carried) words must be visible before the state says "ready".  They are agent-scope stores (write-through
        // to the level all XCDs see); once the stores have been ACKNOWLEDGED (s_waitcnt vmcnt(0)) a reader that sees state 2
        // with its own agent-scope loads finds them.  A __threadfence() here instead — buffer_wbl2 + buffer_inv, tens of
        // microseconds under load — made every NEW group cost a cache write-back: TPC-H Q10 at SF10 creates 380 K groups and
        // spent 2.5 of its 2.9 ms there (device timestamps; round 3).  RSQ_HASH_FENCE=1 restores the fence.
        if (0) line("__threadfence();");
        else line("asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");");
        line("rsq::st_agent(&a." + T + "_state[" + T + "_s], 2u);");
        line("st.n_" + T + "++;");
        if (!charKeyWords.empty()) {
            // Groups are keyed by the exact bytes; the reference's CHAR equality ignores trailing spaces, so the host merges
            // such groups — which is only ever needed when some group value ends with a space.  The lane that creates a
            // group tells the host (once per group, nearly never): without the flag the host skips the merge and may take
            // the candidate path of ORDER BY ... LIMIT.  The last character is the top non-zero byte of the value's last
            // non-zero key word.
            std::string any;
            for (auto& r : charKeyWords) {
                std::string last = kp[r.first];
                for (size_t w = r.first + 1; w <= r.second; w++) last = "(" + kp[w] + " != 0 ? " + kp[w] + " : " + last + ")";
                any += (any.empty() ? "" : " || ") + std::string("rsq::top_byte_is_space(") + last + ")";
            }
            line("if (" + any + ") atomicOr(a.err, (u32)rsq::NOTE_CHAR_GROUP_ENDS_WITH_SPACE);");
        }
        closeScope();
        line("stt = rsq::ld_agent(&a." + T + "_state[" + T + "_s]);      // our own publish, or whoever won the slot");
        closeScope();
        openScope("if (stt == 2u) {");
        std::string cond;
        for (int i = 0; i < K; i++)
            cond += (i ? " && " : "") + std::string("rsq::ld_agent(&") + aggWord(i) + ") == " + kp[(size_t)i];
        if (nCarriedWords && !fdCond.empty()) {
            // (the full comparison, while a table of the dependency chain is in its hash form: the values' words are made here, where
            // they are compared - kept in an array across the loop they were 62 live VGPRs for TPC-H Q10's 31 words)
            line("bool " + T + "_eq = " + cond + ";");
            openScope("if (" + T + "_eq && !" + T + "_fd) {");
            for (size_t ci = 0; ci < carriedVals.size(); ci++) {
                const Carried& c = carriedVals[ci];
                openScope("{");
                std::vector<std::string> words = keyWords(c.g, T + "_m" + std::to_string(ci), false);
                for (int w = 0; w < c.nWords; w++)
                    line(T + "_eq = " + T + "_eq && rsq::ld_agent(&" + aggWord(c.firstWord + w) + ") == " + words[(size_t)w] + ";");
                closeScope();
            }
            closeScope();
            cond = T + "_eq";
        }
        openScope("if (" + cond + ") {");
        line(T + "_found = true;");
        line("break;");
        closeScope();
        line(T + "_s = (" + T + "_s + 1) & " + T + "_mask;");
        // A probe sequence of thousands of slots means the table is (nearly) full: linear probing degrades to a scan of
        // the table per row long before every slot is taken (1 M groups in 1 M slots: 90 ns per row, 2.2 s per 25 M rows).
        // Report "full" early; the host re-runs with a four times larger table and keeps the load below one half.
        line("if (++" + T + "_adv > (" + T + "_mask < 4096 ? " + T + "_mask : 4096)) { atomicOr(a.err, (u32)rsq::ERR_HT_FULL); break; }");
        closeScope();
        line("else if (++" + T + "_spin > (1u << 22)) { atomicOr(a.err, (u32)rsq::ERR_STUCK); break; }   // a slot another wave is writing");
        closeScope();
        // The updates, after the loop (the wave has reconverged): lanes of this wave that found the SAME slot are folded
        // into one update by their first lane while such sets are large (a few groups in the whole input); as soon as the
        // first set is small the group domain is wide and every lane updates its own slot.
        auto updates = [&](const std::string& slot, const std::string& members) {
            for (int w = 0; w < W; w++) {
                const std::string in = "x" + std::to_string(w);
                const std::string op = std::to_string(q.accums[(size_t)w].merge);
                const std::string dst = "a." + T + "_acc + " + std::to_string(q.accumSlot[(size_t)w]) + " * a." + T + "_cap + " + slot;
                if (members.empty()) line("rsq::global_merge_always<" + op + ">(" + dst + ", (u64)" + in + ");");
                else line("{ const u64 r = rsq::subset_reduce<" + op + ">((u64)" + in + ", " + members + "); if (wl_lane == wl_leader) rsq::global_merge_always<" +
                          op + ">(" + dst + ", r); }");
            }
        };
        if (1) {
            line("const int wl_lane = (int)(threadIdx.x & 63);");
            line("bool wl_mine = " + T + "_found;");
            line("u64 wl_todo = __ballot(wl_mine);");
            openScope("while (wl_todo) {");
            line("const int wl_leader = __ffsll((long long)wl_todo) - 1;");
            line("const u64 wl_slot = rsq::readlane_u64(" + T + "_s, wl_leader);");
            line("const u64 wl_set = __ballot(wl_mine && " + T + "_s == wl_slot);");
            line("if (__popcll(wl_set) < 4) break;");
            updates("wl_slot", "wl_set");
            line("if (" + T + "_s == wl_slot) wl_mine = false;");
            line("wl_todo &= ~wl_set;");
            closeScope();
            openScope("if (wl_mine) {");
            updates(T + "_s", "");
            closeScope();
        } else {
            openScope("if (" + T + "_found) {");
            updates(T + "_s", "");
            closeScope();
        }
        std::string fn = "static RSQ_DEV void " + T + "_upsert(const Args& a, State& st, const i64 x0";
        for (int i = 0; i < K; i++) fn += ", const i64 k" + std::to_string(i);
        for (int w = 1; w < W; w++) fn += ", const i64 x" + std::to_string(w);
        for (auto& c : carriedVals) fn += ", const " + c.ctype + " " + c.var;          // (named like the row function's symbol: keyWords above refers to it)
        fn += ") {\n" + body + "}\n";
        helperFns += fn;
        body = savedBody; indent = savedIndent;
    }

    // ---- the row: LDS front table first, the HBM table otherwise ---------------------------------------------------
    std::string call = T + "_upsert(a, st, row";
    for (int i = 0; i < K; i++) call += ", " + keyVars[(size_t)i];
    for (int w = 1; w < W; w++) call += ", in" + std::to_string(w);
    for (auto& c : carriedVals) call += ", " + c.var;
    call += ");";
    if (lds) {
        line("bool " + T + "_done = false;");
        openScope("{");
        // up to four consecutive slots: two groups that map to the same slot would otherwise send one of them to the
        // HBM table for good — with few groups that is a handful of HBM words taking every update of a hot group
        // (64 groups, 1024 slots: 5.1 ms per 100 M rows against 2.9 ms for 1024 groups, before the probing)
        line("u32 ls = (u32)(" + hashOf(keyVars) + " >> 44) & (u32)(RSQ_LC_SLOTS - 1);");
        openScope("for (int lt = 0; lt < 4; lt++, ls = (ls + 1u) & (u32)(RSQ_LC_SLOTS - 1)) {");
        line("u32 lst = __hip_atomic_load(&st.lc_state[ls], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);");
        openScope("if (lst == 0u && atomicCAS(&st.lc_state[ls], 0u, 1u) == 0u) {");
        for (int i = 0; i < K; i++) line("st.lc_key[" + std::to_string(i) + " * RSQ_LC_SLOTS + ls] = " + keyVars[(size_t)i] + ";");
        for (int w = 0; w < W; w++) {
            const int m = q.accums[(size_t)w].merge;
            line("st.lc_acc[" + std::to_string(w) + " * RSQ_LC_SLOTS + ls] = " + (m == 0 ? "0ull" : m == 2 ? "0x7fffffffffffffffull" : m == 3 ? "0x8000000000000000ull" : "~0ull") + ";");
        }
        line("__hip_atomic_store(&st.lc_state[ls], 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);");
        line("lst = 2u;");
        closeScope();
        line("if (lst != 2u) break;           // another lane is writing this slot: do not wait, take the HBM table");
        std::string eq;
        for (int i = 0; i < K; i++) eq += std::string(i ? " && " : "") + "st.lc_key[" + std::to_string(i) + " * RSQ_LC_SLOTS + ls] == " + keyVars[(size_t)i];
        openScope("if (" + eq + ") {");
        for (int w = 0; w < W; w++)
            line("rsq::lds_merge<" + std::to_string(q.accums[(size_t)w].merge) + ">(&st.lc_acc[" + std::to_string(w) + " * RSQ_LC_SLOTS + ls], (u64)(" +
                 (w == 0 ? std::string("row") : "in" + std::to_string(w)) + "));");
        line(T + "_done = true;");
        line("break;");
        closeScope();
        closeScope();
        closeScope();
        line("if (!" + T + "_done) " + call);
    } else line(call);
    closeScope();
    q.aggTable = ht->id;
    explainSteps.push_back("hash aggregation in " + T + " (" + std::to_string(ht->keys.size()) + " key word(s)" +
                           (anyCarried ? " + " + std::to_string(ht->payload.size()) + " carried word(s) of group values that depend on them" : "") + (lds ? ", LDS front table" : "") +
                           ") accumulators=" + std::to_string(W - 1) + " (of " + std::to_string(o->splitAgg.size()) + " in the reference)");
    q.hashTables.push_back(std::move(ht));
}

// Dense group id from the column statistics the table was created with.  The statistics are a promise about the data,
// not a guarantee (rsq_table_create_device adopts caller-owned memory): every rank is checked, a value outside its
// column's recorded domain raises ERR_GROUP_OVERFLOW and is counted into group 0 — no access leaves the table, and the
// host fails the execution.
std::string Walker::groupIdExpr() {
    std::string gid = "0";
    for (size_t ki = 0; ki < q.denseKeys.size(); ki++) {
        DenseKey& k = q.denseKeys[ki];
        std::string v = eg.emit(k.expr), rank;
        const std::string rv = "gk" + std::to_string(ki);
        // (columns the engine owns — uploaded, generated, loaded from '.tbl' — cannot change after their statistics were
        // taken: only adopted columns pay for the checks; TPC-H Q1's kernel is 6 % slower with them)
        bool check = true;
        if (k.expr->tag == RSQ_E_ATTRIBUTE) { const int ci = pipe.src->findCol(k.expr->symbol); if (ci >= 0 && pipe.src->cols[(size_t)ci].owned) check = false; }
        if (envInt("RSQ_CHECK_STATS", 0, 0, 1)) check = true;
        if (k.byteSet && !check) {
            rank = "0";
            for (size_t d = 1; d < k.values.size(); d++) {
                std::string an = "k" + std::to_string(ki) + "_" + std::to_string(d);
                addArg(an, "u64", k.values[d]);
                rank += " + (int)((u8)(" + v + ") >= (u8)a." + an + ")";
            }
            line("const int " + rv + " = " + rank + ";");
        } else if (!k.byteSet && !check) {
            std::string an = "k" + std::to_string(ki) + "_min";
            addArg(an, "i64", (uint64_t)k.min);
            line("const int " + rv + " = (int)((i64)(" + v + ") - a." + an + ");");
        } else
        if (k.byteSet) {
            rank = "0";
            std::string member;
            for (size_t d = 0; d < k.values.size(); d++) {
                std::string an = "k" + std::to_string(ki) + "_" + std::to_string(d);
                addArg(an, "u64", k.values[d]);
                if (d) rank += " + (int)((u8)(" + v + ") >= (u8)a." + an + ")";
                member += std::string(d ? " | " : "") + "(int)((u8)(" + v + ") == (u8)a." + an + ")";
            }
            line("int " + rv + " = " + rank + ";");
            line("if (!(" + (member.empty() ? std::string("1") : member) + ")) { atomicOr(a.err, (u32)rsq::ERR_GROUP_OVERFLOW); " + rv + " = 0; }");
        } else {
            std::string an = "k" + std::to_string(ki) + "_min";
            addArg(an, "i64", (uint64_t)k.min);
            line("int " + rv + " = (int)((i64)(" + v + ") - a." + an + ");");
            line("if ((u64)((i64)(" + v + ") - a." + an + ") >= " + std::to_string((long long)k.card) + "ull) { atomicOr(a.err, (u32)rsq::ERR_GROUP_OVERFLOW); " + rv + " = 0; }");
        }
        gid += " + " + rv + " * " + std::to_string((long long)k.stride);
    }
    return gid;
}

std::string Walker::blockIdentityExpr(const std::string& blk) {
    return blk + " < " + std::to_string((long long)q.nMinBlocks) + " ? 0x7fffffffffffffffull : " + blk + " < " +
           std::to_string((long long)(q.nMinBlocks + q.nMaxBlocks)) + " ? 0x8000000000000000ull : 0ull";
}

// `stride` words between the cells of the table the kernel flushes into (1: the [block][group] table itself)
void Walker::emitGlobalFlush(std::ostringstream& s, const std::string& count, const std::string& srcExpr, int64_t D, int stride) {
    // padded flush: the stride is a macro, so that the same source also gives the unpadded kernel partial executions use
    const std::string at = stride == 1 ? "a.out + i" : "a.out + i * RSQ_OUT_STRIDE";
    s << "    for (int i = threadIdx.x; i < " << count << "; i += blockDim.x) {\n";
    s << "        const int blk = i / " << D << ";\n        const u64 v = " << srcExpr << ";\n";
    s << "        if (blk < " << q.nMinBlocks << ") rsq::global_merge<2>(" << at << ", v);\n";
    s << "        else if (blk < " << (q.nMinBlocks + q.nMaxBlocks) << ") rsq::global_merge<3>(" << at << ", v);\n";
    s << "        else rsq::global_merge<0>(" << at << ", v);\n    }\n";
}

// Form 3 of a large dense aggregation (rsq_device.h "staged partitioning"): the passing row becomes a PACKED record — the
// group's index inside its partition and the accumulator inputs, each in as many bits as its column's statistics need —
// handed to the workgroup's LDS rings; the first-row tracker is kept beside it (stage_track).  Available when the
// record fits 128 bits and the partitions fit the rings (<= 256); the wider cases stay with form 2.
void Walker::emitStagedScatter(int64_t D, int W, int gpp, int shift, int P) {
    if (P > 256 || !envInt("RSQ_STAGED", 1, 0, 1) || q.accums[0].merge != 2) return;
    struct Field { int w; int bits; int64_t min; bool check; int word, off; };
    std::vector<Field> fields;
    fields.push_back({-1, shift, 0, false, 0, 0});
    for (int w : pipe.partRecordInputs) {
        Field f{w, 64, 0, false, 0, 0};
        const Expr* e = q.accums[(size_t)w].inputExpr;
        if (e && e->tag == RSQ_E_ATTRIBUTE) {
            const int ci = pipe.src->findCol(e->symbol);
            if (ci >= 0 && pipe.src->cols[(size_t)ci].stats.valid) {
                const TableColumn& c = pipe.src->cols[(size_t)ci];
                const uint64_t range = (uint64_t)c.stats.max - (uint64_t)c.stats.min;
                int bits = 1; while (bits < 64 && (range >> bits) != 0) bits++;
                if (bits < 64) { f.bits = bits; f.min = c.stats.min; f.check = !c.owned || envInt("RSQ_CHECK_STATS", 0, 0, 1); }
            }
        }
        fields.push_back(f);
    }
    int used[2] = {0, 0};
    for (auto& f : fields) {
        int wd = 0;
        while (wd < 2 && used[wd] + f.bits > 64) wd++;
        if (wd == 2) return;                                  // wider than two words: form 2
        f.word = wd; f.off = used[wd]; used[wd] += f.bits;
    }
    const int RECW = used[1] ? 2 : 1;
    const int ncolsNow = (int)colTypes.size();
    // rows per thread and round: 4 (2 for wide rows: registers); 8 when few rows are expected to pass - the rounds' barriers then
    // weigh more than the records (1.25 B rows, 2^20 groups: 10 % 7.54 -> 7.15 ms; at 50 % 8 rows cost 11.1 instead of 9.6 ms)
    const int RPT = ncolsNow <= 6 ? (!leadCond.empty() && leadPass <= 0.15 ? 8 : 4) : 2;
    pipe.staged = true; pipe.stagedRecWords = RECW; pipe.stagedRows = RPT;
    const std::string Ps = std::to_string(P), Rs = std::to_string(RECW), Ts = std::to_string(RPT);
    const std::string LDS = "rsq::StageLds<" + Rs + ", " + Ps + ">";
    line("#elif RSQ_AGG_VARIANT == 3");
    openScope("{");
    line("const u32 sp_p = (u32)(gid >> " + std::to_string(shift) + ");");
    line("if (a.sp_mode) atomicAdd(&st.sp->tail[sp_p], 1u);          // counting only: exact region sizes after an overflow");
    openScope("else {");
    line("if ((u64)row < st.sp_wm) rsq::stage_track(*st.sp, a.out + " + std::to_string((long long)(q.accumSlot[0] * D)) + " + gid, row);");
    for (int wd = 0; wd < RECW; wd++) {
        std::string ex;
        for (auto& f : fields) {
            if (f.word != wd) continue;
            std::string v;
            if (f.w < 0) v = "(u64)(gid & " + std::to_string(gpp - 1) + ")";
            else {
                const std::string in = "in" + std::to_string(f.w);
                if (f.bits == 64) v = "(u64)" + in;
                else {
                    const std::string an = "sp_min" + std::to_string(f.w);
                    addArg(an, "i64", (uint64_t)f.min);
                    const std::string mask = std::to_string((unsigned long long)((1ull << f.bits) - 1ull)) + "ull";
                    if (f.check) line("if ((u64)(" + in + " - a." + an + ") > " + mask + ") atomicOr(a.err, (u32)rsq::ERR_GROUP_OVERFLOW);");
                    v = "((u64)(" + in + " - a." + an + ") & " + mask + ")";
                }
            }
            if (f.off) v = "(" + v + " << " + std::to_string(f.off) + ")";
            ex += (ex.empty() ? "" : " | ") + v;
        }
        line("st.sp_rec[SP_SLOT * " + Rs + " + " + std::to_string(wd) + "] = " + (ex.empty() ? std::string("0ull") : ex) + ";");
    }
    line("st.sp_p[SP_SLOT] = sp_p;");
    line("st.sp_k[SP_SLOT] = atomicAdd(&st.sp->tail[sp_p], 1u);");
    line("st.sp_pending |= 1u << SP_SLOT;");
    closeScope();
    closeScope();
    addArg("sp_base", "const u64*", 0); addArg("sp_cap", "const u32*", 0); addArg("sp_ctl", "rsq::StageCtl*", 0);
    addArg("sp_counts", "u32*", 0); addArg("sp_rec", "u64*", 0); addArg("sp_mode", "u32", 0);
    stateDecl += "#if RSQ_AGG_VARIANT == 3\n    " + LDS + "* sp;\n    u64 sp_wm;\n    u64 sp_rec[" + std::to_string(RPT * RECW) + "];\n    u32 sp_k[" + Ts +
                 "];\n    u32 sp_p[" + Ts + "];\n    u32 sp_pending;\n#endif\n";
    prologue += "#if RSQ_AGG_VARIANT == 3\n    __shared__ " + LDS + " s_stage;\n    rsq::stage_init(s_stage, a.sp_base, a.sp_cap, a.sp_ctl);\n";
    prologue += "    st.sp = &s_stage; st.sp_pending = 0u; st.sp_wm = ~0ull;\n#endif\n";
    // ---- aggregation of one partition's records (all workgroups' regions of it) in an LDS table ----
    std::ostringstream k;
    auto& A3 = pipe.argsStagedAgg;
    A3.push_back({"sp_rec", "const u64*", 0});
    A3.push_back({"sp_base", "const u64*", 0});
    A3.push_back({"sp_cap", "const u32*", 0});
    A3.push_back({"sp_counts", "const u32*", 0});
    A3.push_back({"sp_nwg", "u32", 0});
    A3.push_back({"out", "u64*", 0});
    for (auto& f : fields) if (f.w >= 0 && f.bits < 64) A3.push_back({"sp_min" + std::to_string(f.w), "i64", (uint64_t)f.min});
    k << "// generated by resql_amd/csrc/codegen.cpp: aggregation of one partition of packed records in an LDS table\n";
    k << "#include \"rsq_device.h\"\nstruct Args {\n";
    for (auto& a : A3) k << "    " << a.ctype << " " << a.name << ";\n";
    k << "};\n";
    // accumulator blocks 1 .. W-1 of the table (block 0, the first row, is the tracker's)
    k << "static RSQ_DEV void merge_record(const Args& a, u64* tab";
    for (int wd = 0; wd < RECW; wd++) k << ", const u64 w" << wd;
    k << ") {\n";
    for (auto& f : fields) {
        std::string v = "w" + std::to_string(f.word);
        if (f.off) v = "(" + v + " >> " + std::to_string(f.off) + ")";
        if (f.bits < 64) v = "(" + v + " & " + std::to_string((unsigned long long)((1ull << f.bits) - 1ull)) + "ull)";
        if (f.w < 0) k << "    const int g = (int)" << v << ";\n";
        else if (f.bits < 64) k << "    const u64 in" << f.w << " = (u64)((i64)" << v << " + a.sp_min" << f.w << ");\n";
        else k << "    const u64 in" << f.w << " = " << v << ";\n";
    }
    for (int w = 1; w < W; w++) {
        std::string in = "(u64)" + q.accums[(size_t)w].input;             // a constant (COUNT's 1) unless it travels
        for (auto& f : fields) if (f.w == w) in = "in" + std::to_string(w);
        k << "    rsq::lds_merge<" << q.accums[(size_t)w].merge << ">(&tab[" << (int64_t)(w - 1) * gpp << " + g], " << in << ");\n";
    }
    k << "}\n";
    k << "extern \"C\" __global__ void __launch_bounds__(1024) rsq_staged_agg(Args a) {\n";
    k << "    __shared__ u64 s_tab[" << (int64_t)(W - 1) * gpp << "];\n";
    k << "    for (int i = threadIdx.x; i < " << (int64_t)(W - 1) * gpp << "; i += blockDim.x) { const int w = 1 + (i >> " << shift << "); s_tab[i] = ";
    for (int w = 1; w < W; w++) k << (w > 1 ? " : " : "") << (w < W - 1 ? "w == " + std::to_string(w) + " ? " : "") << identityOf(q.accums[(size_t)w].merge);
    k << "; }\n    __syncthreads();\n";
    k << "    const int p = blockIdx.x, lane = threadIdx.x & 63;\n";
    k << "    const u64 base = a.sp_base[p]; const u32 cap = a.sp_cap[p];\n";
    k << "    for (u32 wg = threadIdx.x >> 6; wg < a.sp_nwg; wg += blockDim.x >> 6) {\n";
    k << "        const u64 st = base + (u64)wg * cap;\n        const u32 cnt = min(a.sp_counts[(u64)wg * " << P << " + p], cap);\n";
    const int AU = 4;
    // AU 16-byte loads per lane in flight (one per lane leaves a CU with 16 KB outstanding: 4.8 TB/s; four: see DESIGN §4)
    const int step = RECW == 1 ? 128 : 64;              // records one wave-load covers
    k << "        for (u32 i0 = 0; i0 < cnt; i0 += " << AU * step << ") {\n";
    k << "            rsq::u64x2 v[" << AU << "];\n";
    k << "#pragma unroll\n            for (int u = 0; u < " << AU << "; u++) {\n";
    k << "                const u32 i = i0 + u * " << step << " + lane * " << (RECW == 1 ? 2 : 1) << ";\n";
    k << "                if (i < cnt) v[u] = *reinterpret_cast<const rsq::u64x2*>(a.sp_rec + (st + i) * " << RECW << ");\n            }\n";
    k << "#pragma unroll\n            for (int u = 0; u < " << AU << "; u++) {\n";
    k << "                const u32 i = i0 + u * " << step << " + lane * " << (RECW == 1 ? 2 : 1) << ";\n";
    if (RECW == 1) k << "                if (i < cnt) merge_record(a, s_tab, v[u].x);\n                if (i + 1 < cnt) merge_record(a, s_tab, v[u].y);\n";
    else k << "                if (i < cnt) merge_record(a, s_tab, v[u].x, v[u].y);\n";
    k << "            }\n        }\n";
    k << "    }\n    __syncthreads();\n";
    k << "    for (int i = threadIdx.x; i < " << (int64_t)(W - 1) * gpp << "; i += blockDim.x) {\n";
    k << "        const int w = 1 + (i >> " << shift << ");\n";
    k << "        const i64 g = (i64)p * " << gpp << " + (i & " << (gpp - 1) << ");\n";
    // accumulator w lives in block accumSlot[w] of the [block][group] table
    k << "        const i64 blk = ";
    for (int w = 1; w < W; w++) k << (w > 1 ? " : " : "") << (w < W - 1 ? "w == " + std::to_string(w) + " ? " : "") << "(i64)" << q.accumSlot[(size_t)w];
    k << ";\n        if (g < " << D << ") a.out[blk * " << D << " + g] = s_tab[i];\n    }\n}\n";
    pipe.sourceStagedAgg = k.str();
}

void Walker::emitDenseAggregation(OpNode* o) {
    const int64_t D = q.denseGroups;
    const int W = (int)q.accums.size();
    line("const int gid = " + groupIdExpr() + ";");
    for (int w = 1; w < W; w++) line("const i64 in" + std::to_string(w) + " = " + q.accums[(size_t)w].input + ";");
    auto inOf = [&](int w) { return w == 0 ? std::string("row") : "in" + std::to_string(w); };
    addArg("out", "u64*", 0);
    std::ostringstream ep;
    if (q.aggMode == AggMode::DENSE_REG) {
        // accumulators in VGPRs, branch-free per-group update.  (An `if (gid == g) acc_g += x` chain gets its
        // common tail sunk by the compiler into one store through a selected pointer, which forces every
        // accumulator into scratch.)
        for (int w = 0; w < W; w++)
            for (int64_t g = 0; g < D; g++)
                stateDecl += "    i64 acc_" + std::to_string(w) + "_" + std::to_string((long long)g) + " = (i64)" + identityOf(q.accums[(size_t)w].merge) + ";\n";
        const bool branchy = (D > 1 ? 1 : 0) == 1;
        for (int64_t g = 0; g < D; g++) {
            if (branchy) {
                // EXEC-masked update of one group's accumulators (2 VALU per 64-bit add instead of the
                // select form's 4).  The distinct asm comment at the end of every block is load-bearing: without
                // it the compiler sinks the identical tails of the blocks into one store through a selected
                // pointer, which forces all accumulators into scratch memory.
                openScope("if (gid == " + std::to_string((long long)g) + ") {");
                for (int w = 0; w < W; w++) {
                    std::string acc = "st.acc_" + std::to_string(w) + "_" + std::to_string((long long)g), in = inOf(w);
                    int m = q.accums[(size_t)w].merge;
                    if (m == 0) line(acc + " = rsq::add(" + acc + ", " + in + ");");
                    else if (m == 2) line(acc + " = " + in + " < " + acc + " ? " + in + " : " + acc + ";");
                    else line(acc + " = " + in + " > " + acc + " ? " + in + " : " + acc + ";");
                }
                line("asm volatile(\"; rsq group " + std::to_string((long long)g) + "\");");
                closeScope();
                continue;
            }
            openScope("{");
            line("const bool m = gid == " + std::to_string((long long)g) + ";");
            for (int w = 0; w < W; w++) {
                std::string acc = "st.acc_" + std::to_string(w) + "_" + std::to_string((long long)g), in = inOf(w);
                int m = q.accums[(size_t)w].merge;
                if (m == 0) line(acc + " = rsq::add(" + acc + ", m ? " + in + " : (i64)0);");
                else if (m == 2) line(acc + " = (m && " + in + " < " + acc + ") ? " + in + " : " + acc + ";");
                else line(acc + " = (m && " + in + " > " + acc + ") ? " + in + " : " + acc + ";");
            }
            closeScope();
        }
        const bool dbgTail = envInt("RSQ_DEBUG_TAIL", 0, 0, 1) != 0;      // (measurement only: device timestamps of the epilogue's stages)
        auto stamp = [&](int k) { if (dbgTail) ep << "    if (a.dbg && threadIdx.x == 0) a.dbg[(u64)blockIdx.x * 8 + " << k << "] = (u64)wall_clock64();\n"; };
        if (dbgTail) { addArg("dbg", "u64*", 0); prologue += "    if (a.dbg && threadIdx.x == 0) a.dbg[(u64)blockIdx.x * 8 + 0] = (u64)wall_clock64();\n"; }
        stamp(1);
        ep << "    __shared__ u64 s_acc[" << W * D << "];\n";
        if (1) {
            // The workgroup's accumulators meet LANE BY LANE first: every wave merges its cells into s_lane[cell][lane] (LDS
            // atomics, no two lanes on one word), then each wave folds a share of the cells across the 64 lanes (DPP, rsq_device.h
            // wave_reduce_to_lane63).  One cross-lane reduction per cell and workgroup instead of one per cell and WAVE: the
            // reductions of TPC-H Q1's 42 cells in all 8 waves took 14-20 us of every launch as ds_bpermute butterflies and
            // still 9-11 us as DPP (device timestamps, RSQ_DEBUG_TAIL).
            const int64_t cells = W * D;
            ep << "    __shared__ u64 s_lane[" << cells * 64 << "];\n";
            ep << "    for (int i = threadIdx.x; i < " << cells * 64 << "; i += blockDim.x) { const int blk = (i >> 6) / " << D << "; s_lane[i] = " << blockIdentityExpr("blk") << "; }\n";
            ep << "    __syncthreads();\n";
            for (int w = 0; w < W; w++)
                for (int64_t g = 0; g < D; g++)
                    ep << "    rsq::lds_merge<" << q.accums[(size_t)w].merge << ">(&s_lane[" << (q.accumSlot[(size_t)w] * D + g) * 64 << " + (threadIdx.x & 63)], (u64)st.acc_" << w << "_" << g << ");\n";
            ep << "    __syncthreads();\n";
            ep << "    for (int c = threadIdx.x >> 6; c < " << cells << "; c += blockDim.x >> 6) {\n";
            ep << "        const int blk = c / " << D << ";\n        const u64 v = s_lane[c * 64 + (threadIdx.x & 63)];\n";
            ep << "        const u64 r = blk < " << q.nMinBlocks << " ? rsq::wave_reduce_to_lane63<2>(v) : blk < " << (q.nMinBlocks + q.nMaxBlocks)
               << " ? rsq::wave_reduce_to_lane63<3>(v) : rsq::wave_reduce_to_lane63<0>(v);\n";
            ep << "        if ((threadIdx.x & 63) == 63) s_acc[c] = r;\n    }\n";
            ep << "    __syncthreads();\n";
        } else {
        ep << "    for (int i = threadIdx.x; i < " << W * D << "; i += blockDim.x) { const int blk = i / " << D << "; s_acc[i] = " << blockIdentityExpr("blk") << "; }\n";
        ep << "    __syncthreads();\n";
        for (int w = 0; w < W; w++)
            for (int64_t g = 0; g < D; g++)
                ep << "    rsq::wave_to_lds<" << q.accums[(size_t)w].merge << ">(&s_acc[" << (q.accumSlot[(size_t)w] * D + g) << "], (u64)st.acc_" << w << "_" << g << ");\n";
        ep << "    __syncthreads();\n";
        }
        // The workgroups flush into a PADDED copy of the table, one cell per 64-byte line (engine.cpp unpads it):
        // memory-side atomics serialise per line, and the 42 cells of TPC-H Q1 otherwise share six lines.
        q.aggPad = 8;
        stamp(2);
        emitGlobalFlush(ep, std::to_string((long long)(W * D)), "s_acc[i]", D, q.aggPad);
        stamp(3);
        // The step in ONE launch (engine.cpp runFusedStep): the workgroup that flushes last hands the finished table to
        // the host — plain stores into host-mapped pinned memory (a full execution) or into the partial table the
        // group-by merge reads (a multi-GPU step) — together with the device error word, and puts the working table,
        // the error word and the ticket back to their identities for the next execution.  That replaces the D2D
        // copy that readied the table, the error-word memset and the two read-back copies of every step.
        // Order: a thread waits until its flush atomics have been performed (s_waitcnt vmcnt(0): device-scope atomics are
        // coherent across the XCDs once performed) before the workgroup takes its ticket, so the holder of the last ticket
        // finds every cell final; it reads with agent-scope exchanges, which execute where the flush atomics did.  A
        // release fence instead (__threadfence: buffer_wbl2 + buffer_inv in every wave) cost 30 us per launch — more than
        // the copies it was meant to save.
        addArg("fin_out", "u64*", 0);
        addArg("fin_err", "u64*", 0);
        addArg("fin_ticket", "u32*", 0);
        addArg("fin_seq", "u64", 0);
        ep << "    if (a.fin_out) {\n        __shared__ u32 s_last;\n        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n        __syncthreads();\n";
        ep << "        if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(a.fin_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1u : 0u;\n        __syncthreads();\n";
        if (dbgTail) ep << "        if (a.dbg && threadIdx.x == 0) a.dbg[(u64)blockIdx.x * 8 + 4] = (u64)wall_clock64();\n";
        ep << "        if (s_last) {\n";
        ep << "            for (int i = threadIdx.x; i < " << W * D << "; i += blockDim.x) {\n                const int blk = i / " << D << ";\n";
        ep << "                const u64 idv = " << blockIdentityExpr("blk") << ";\n";
        ep << "                a.fin_out[i] = __hip_atomic_exchange(a.out + i" << (q.aggPad > 1 ? " * RSQ_OUT_STRIDE" : "") << ", idv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n            }\n";
        ep << "            if (threadIdx.x == 0) {\n                a.fin_err[0] = (u64)atomicExch(a.err, 0u);\n";
        ep << "                __hip_atomic_store(a.fin_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n            }\n";
        // a full execution is announced to the polling host by a sequence number behind the table: written after every
        // thread's table stores have been acknowledged, with a system-scope release (one wave, once per launch)
        if (dbgTail) ep << "            if (a.dbg && threadIdx.x == 0) a.dbg[(u64)blockIdx.x * 8 + 5] = (u64)wall_clock64();\n";
        ep << "            if (a.fin_seq) {\n                asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n                __syncthreads();\n";
        if (dbgTail) ep << "                if (a.dbg && threadIdx.x == 0) a.dbg[(u64)blockIdx.x * 8 + 6] = (u64)wall_clock64();\n";
        ep << "                if (threadIdx.x == 0) __hip_atomic_store(a.fin_err + 1, a.fin_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);\n";
        if (dbgTail) ep << "                if (a.dbg && threadIdx.x == 0) a.dbg[(u64)blockIdx.x * 8 + 7] = (u64)wall_clock64();\n";
        ep << "            }\n";
        ep << "        }\n    }\n";
        // One 512-thread workgroup per CU: the same 8 waves per CU as 2 x 256, but half as many workgroups flush.
        // The flush is 42 atomics per workgroup (TPC-H Q1) onto six 64-byte lines, where they serialise: going from
        // 512 to 256 workgroups took 8 us off the 352 us SF10 kernel and 9 off the 67 us SF1 kernel (1024 and 2048
        // workgroups: +25 / +75 us).  A slab-per-workgroup + ticket + last-workgroup reduction was tried instead of
        // the atomics and measured 28 us SLOWER (write-through slab stores, a serial reducer), so it is not here.
        pipe.blockThreads = 512;
    } else if (q.aggMode == AggMode::DENSE_LDS_PRIVATE) {
        // one private copy of the [block][group] table per LANE in LDS, laid out [cell][thread] so that a
        // wave's 64 accesses to one cell are 64 consecutive 8-byte words: conflict-free, no contention,
        // one ds_add_u64 / ds_min_i64 per accumulator and row instead of a select+add per group.
        const int64_t cells = W * D;
        pipe.blockThreads = cells <= 28 ? 256 : 128;
        const int B = pipe.blockThreads;
        prologue += "    __shared__ u64 s_priv[" + std::to_string((long long)(cells * B)) + "];\n";
        prologue += "    for (int i = threadIdx.x; i < " + std::to_string((long long)(cells * B)) + "; i += blockDim.x) { const int blk = (i / " +
                    std::to_string(B) + ") / " + std::to_string((long long)D) + "; s_priv[i] = " + blockIdentityExpr("blk") + "; }\n";
        prologue += "    __syncthreads();\n    st.priv = s_priv + threadIdx.x;\n";
        stateDecl += "    u64* priv;\n";
        for (int w = 0; w < W; w++) {
            std::string cell = "st.priv + (" + std::to_string((long long)(q.accumSlot[(size_t)w] * D)) + " + gid) * " + std::to_string(B);
            int m = q.accums[(size_t)w].merge;
            if (m == 0) line("rsq::lds_merge<0>(" + cell + ", (u64)" + inOf(w) + ");");
            else line("rsq::lds_merge<" + std::to_string(m) + ">(" + cell + ", (u64)" + inOf(w) + ");");
        }
        // flush: every wave folds whole cells: lanes stride over the B private copies, butterfly, one atomic
        ep << "    __syncthreads();\n";
        ep << "    for (int c = (threadIdx.x >> 6); c < " << cells << "; c += (blockDim.x >> 6)) {\n";
        ep << "        const int blk = c / " << D << ";\n        const int lane = threadIdx.x & 63;\n";
        ep << "        u64 v = s_priv[c * " << B << " + lane];\n";
        ep << "        for (int j = lane + 64; j < " << B << "; j += 64) {\n            const u64 o = s_priv[c * " << B << " + j];\n";
        ep << "            if (blk < " << q.nMinBlocks << ") v = (i64)o < (i64)v ? o : v; else if (blk < " << (q.nMinBlocks + q.nMaxBlocks)
           << ") v = (i64)o > (i64)v ? o : v; else v += o;\n        }\n";
        ep << "        if (blk < " << q.nMinBlocks << ") { v = (u64)rsq::wave_min_i64((i64)v); if (lane == 0) rsq::global_merge<2>(a.out + c, v); }\n";
        ep << "        else if (blk < " << (q.nMinBlocks + q.nMaxBlocks) << ") { v = (u64)rsq::wave_max_i64((i64)v); if (lane == 0) rsq::global_merge<3>(a.out + c, v); }\n";
        ep << "        else { v = rsq::wave_sum(v); if (lane == 0) rsq::global_merge<0>(a.out + c, v); }\n    }\n";
    } else if (q.aggMode == AggMode::DENSE_LDS_SHARED) {
        // one [block][group] table per workgroup in LDS, LDS atomics (many groups => little contention),
        // flushed once per workgroup with global atomics
        const int64_t cells = W * D;
        prologue += "    __shared__ u64 s_tab[" + std::to_string((long long)cells) + "];\n";
        prologue += "    for (int i = threadIdx.x; i < " + std::to_string((long long)cells) + "; i += blockDim.x) { const int blk = i / " +
                    std::to_string((long long)D) + "; s_tab[i] = " + blockIdentityExpr("blk") + "; }\n    __syncthreads();\n    st.tab = s_tab;\n";
        stateDecl += "    u64* tab;\n";
        for (int w = 0; w < W; w++)
            line("rsq::lds_merge<" + std::to_string(q.accums[(size_t)w].merge) + ">(st.tab + " + std::to_string((long long)(q.accumSlot[(size_t)w] * D)) +
                 " + gid, (u64)" + inOf(w) + ");");
        ep << "    __syncthreads();\n";
        emitGlobalFlush(ep, std::to_string((long long)cells), "s_tab[i]", D);
    } else {   // DENSE_GLOBAL: the table lives in HBM
        // Three forms of the same pipeline, one source (RSQ_AGG_VARIANT):
        //  0 direct     every passing row merges into the table with HBM atomics.  Those execute at the memory side,
        //               ≈25 G requests/s chip-wide, so this form is atomic-bound once many rows pass the filter.
        //  1 count      per (workgroup, partition) row counts in LDS (partition = group id / groups-per-partition);
        //               with a.tile_step > 1 it samples every n-th tile: the engine's selectivity estimate.
        //  2 scatter    each passing row becomes a record (group-in-partition, row, accumulator inputs) written to
        //               its partition's region at a position taken from a workgroup-local LDS cursor that starts
        //               at the exclusive prefix of the counts: no HBM atomics at all.
        // A fourth kernel (rsq_part_agg, emitted below) aggregates each partition in an LDS table and stores the
        // finished groups with plain stores.  The engine picks direct or partitioned per execution from the counts.
        pipe.gridPerCU = 8;
        int gpp = 1;
        while ((int64_t)gpp * 2 * W * 8 <= 128 * 1024 && gpp * 2 <= (1 << 20)) gpp *= 2;
        const int64_t P = (D + gpp - 1) / gpp;
        int shift = 0; while ((1 << shift) < gpp) shift++;
        const bool part = P >= 2 && P <= 4096 && envInt("RSQ_PARTITION", 1, 0, 2) != 0;
        if (part) {
            pipe.partitioned = true; pipe.partCount = (int)P; pipe.partGroups = gpp;
            line("#if RSQ_AGG_VARIANT == 1");
            line("atomicAdd(&st.part[gid >> " + std::to_string(shift) + "], 1u);");
            line("#elif RSQ_AGG_VARIANT == 2");
            openScope("{");
            // records are arrays of R words, stored whole (array of structures): a workgroup then streams into ONE
            // address range per partition, and with one 1024-thread workgroup per CU the partially written lines of
            // all its partitions stay in the XCD's L2 until they are full.  (Struct of arrays with 8 workgroups per
            // CU measured 4.4 ms for 100 M records — every 8-byte store left L2 as its own partial write.)
            for (int w = 1; w < W; w++)
                if (q.accums[(size_t)w].input != "((i64)1)") pipe.partRecordInputs.push_back(w);   // COUNT's input is the constant 1
            const std::string R = std::to_string(1 + pipe.partRecordInputs.size());
            line("const u32 pos = atomicAdd(&st.part[gid >> " + std::to_string(shift) + "], 1u);");
            line("u64* rec = a.rec + (u64)pos * " + R + ";");
            line("rec[0] = ((u64)(gid & " + std::to_string(gpp - 1) + ") << 40) | (u64)(row - a.row0);");
            addArg("rec", "u64*", 0);
            for (size_t j = 0; j < pipe.partRecordInputs.size(); j++)
                line("rec[" + std::to_string(j + 1) + "] = (u64)in" + std::to_string(pipe.partRecordInputs[j]) + ";");
            closeScope();
            emitStagedScatter(D, W, gpp, shift, (int)P);
            line("#else");
        }
        pipe.partAtomicsPerRow = 0;
        for (int w = 0; w < W; w++) {
            if (q.accums[(size_t)w].merge == 0) pipe.partAtomicsPerRow++;
            line("rsq::global_merge<" + std::to_string(q.accums[(size_t)w].merge) + ">(a.out + " + std::to_string((long long)(q.accumSlot[(size_t)w] * D)) +
                 " + gid, (u64)" + inOf(w) + ");");
        }
        if (part) {
            line("#endif");
            addArg("part_counts", "u32*", 0); addArg("part_start", "const u32*", 0); addArg("tile_step", "i64", 1);
            stateDecl += "    u32* part;\n";
            const std::string Ps = std::to_string((long long)P);
            prologue += "#if RSQ_AGG_VARIANT == 1 || RSQ_AGG_VARIANT == 2\n    __shared__ u32 s_part[" + Ps + "];\n";
            prologue += "    for (int i = threadIdx.x; i < " + Ps + "; i += blockDim.x)\n";
            prologue += "        s_part[i] = RSQ_AGG_VARIANT == 2 ? a.part_start[i] + a.part_counts[(u64)blockIdx.x * " + Ps + " + i] : 0u;\n";
            prologue += "    __syncthreads();\n    st.part = s_part;\n#endif\n";
            ep << "#if RSQ_AGG_VARIANT == 1\n    __syncthreads();\n";
            ep << "    for (int i = threadIdx.x; i < " << P << "; i += blockDim.x) a.part_counts[(u64)blockIdx.x * " << P << " + i] = s_part[i];\n#endif\n";
            // ---- the per-partition aggregation kernel ----
            std::ostringstream k;
            auto& A2 = pipe.argsPartAgg;
            A2.push_back({"rec", "const u64*", 0});
            A2.push_back({"part_start", "const u32*", 0});
            A2.push_back({"out", "u64*", 0});
            A2.push_back({"row0", "i64", (uint64_t)pipe.src->row0});
            k << "// generated by resql_amd/csrc/codegen.cpp: aggregation of one partition of records in an LDS table\n";
            k << "#include \"rsq_device.h\"\nstruct Args {\n";
            for (auto& a : A2) k << "    " << a.ctype << " " << a.name << ";\n";
            k << "};\nextern \"C\" __global__ void __launch_bounds__(1024) rsq_part_agg(Args a) {\n";
            k << "    __shared__ u64 s_tab[" << (int64_t)W * gpp << "];\n";
            k << "    for (int i = threadIdx.x; i < " << (int64_t)W * gpp << "; i += blockDim.x) { const int blk = i >> " << shift << "; s_tab[i] = " << blockIdentityExpr("blk") << "; }\n";
            k << "    __syncthreads();\n";
            k << "    const u32 b = a.part_start[blockIdx.x], e = a.part_start[blockIdx.x + 1];\n";
            k << "    for (u32 i = b + threadIdx.x; i < e; i += blockDim.x) {\n";
            const size_t RW = 1 + pipe.partRecordInputs.size();
            k << "        const u64* rec = a.rec + (u64)i * " << RW << ";\n";
            k << "        const u64 key = rec[0];\n        const int g = (int)(key >> 40);\n";
            k << "        const i64 row = a.row0 + (i64)(key & ((1ull << 40) - 1));\n";
            for (int w = 0; w < W; w++) {
                std::string in = "row";
                if (w > 0) {
                    in = "(i64)1";
                    for (size_t j = 0; j < pipe.partRecordInputs.size(); j++)
                        if (pipe.partRecordInputs[j] == w) in = "rec[" + std::to_string(j + 1) + "]";
                }
                k << "        rsq::lds_merge<" << q.accums[(size_t)w].merge << ">(&s_tab[" << (int64_t)q.accumSlot[(size_t)w] * gpp << " + g], (u64)(" << in << "));\n";
            }
            k << "    }\n    __syncthreads();\n";
            k << "    for (int i = threadIdx.x; i < " << (int64_t)W * gpp << "; i += blockDim.x) {\n";
            k << "        const i64 g = (i64)blockIdx.x * " << gpp << " + (i & " << (gpp - 1) << ");\n";
            k << "        if (g < " << D << ") a.out[(i64)(i >> " << shift << ") * " << D << " + g] = s_tab[i];\n    }\n}\n";
            pipe.sourcePartAgg = k.str();
        }
    }
    epilogue += ep.str();
    static const char* names[] = {"none", "registers", "lane-private LDS", "workgroup LDS table", "HBM table", "join entry", "hash"};
    explainSteps.push_back("aggregation dense groups=" + std::to_string((long long)D) + " accumulators=" + std::to_string(W - 1) +
                           " (of " + std::to_string(o->splitAgg.size()) + " in the reference) in " + names[(int)q.aggMode] +
                           (pipe.partitioned ? " (atomics, or " + std::to_string(pipe.partCount) + " partitions x " + std::to_string(pipe.partGroups) +
                                               " groups aggregated in LDS when many rows pass)" : ""));
}

void Walker::emitJoinEntryAggregation(OpNode* o) {
    HashTable& ht = *q.hashTables[(size_t)q.aggTable];
    const std::string T = "ht" + std::to_string(ht.id);
    const int W = (int)q.accums.size();
    ht.nAccBlocks = W;
    addArg(T + "_acc", "u64*", 0);
    // Entries of a rank dictionary are in key order, and rows clustered by the key update neighbouring entries: their
    // atomics would queue on a handful of cache lines.  The accumulators of entry r therefore live at rsq::rank_mix(r), a
    // bijection of [0, capacity) (rsq_device.h; the capacity of a dictionary that carries aggregates is a power of two).
    std::string accIdx = slotVar[ht.id];
    if (ht.rankCapable && 1) {
        line("const u64 " + T + "_ai = a." + T + "_rank ? rsq::rank_mix(" + slotVar[ht.id] + ", a." + T + "_cap) : " + slotVar[ht.id] + ";");
        accIdx = T + "_ai";
    }
    const int dbgAcc = 0;      // (measurement only: 1 no first-row tracker, 2 no aggregates, 3 neither)
    for (int w = 0; w < W; w++) {
        if ((w == 0 && (dbgAcc & 1)) || (w > 0 && (dbgAcc & 2))) continue;
        std::string in = w == 0 ? "row" : q.accums[(size_t)w].input;
        line("rsq::global_merge_always<" + std::to_string(q.accums[(size_t)w].merge) + ">(a." + T + "_acc + " + std::to_string(q.accumSlot[(size_t)w]) +
             " * a." + T + "_cap + " + accIdx + ", (u64)(" + in + "));");
    }
    explainSteps.push_back("aggregation at the matched entry of " + T + " accumulators=" + std::to_string(W - 1) + " (of " +
                           std::to_string(o->splitAgg.size()) + " in the reference)");
}

}  // namespace cg
}  // namespace rsq
