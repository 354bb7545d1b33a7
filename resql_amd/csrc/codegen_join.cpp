// codegen_join.cpp - hash-table builds and probes: key words, the build sink (sizing pass, rank dictionary or hash form, key and component
// bitmaps), the probe (bitmap test, rank or hash lookup, matches).  Reference: src/operators/hashjoin.h.
#include "codegen_internal.h"

namespace rsq {
namespace cg {

// Entries are counted per thread in a register and added to the table's counter once per wave at the end of
// the kernel.  (One atomic per inserted entry on a single word serialises: 1.45 M of them cost 4.6 ms on
// MI355X, more than the rest of TPC-H Q3 together; inside a divergent probe loop neither the compiler nor a
// ballot folds them, the matching lanes arrive one at a time.)
void Walker::countPerThread(const std::string& T, bool identityCapable) {
    stateDecl += "    u32 n_" + T + " = 0;\n";
    // (a dictionary whose entry numbers are the row numbers holds n_rows entries: one store says so.  The workgroups of a build finish
    // together, and their 1 792 adds to the one counter word were the last ~20 us of TPC-H Q10's 48 us customer build.)
    if (identityCapable)
        epilogue += "    if (a." + T + "_ident && a." + T + "_rank && !a." + T + "_countonly) { if (blockIdx.x == 0 && threadIdx.x == 0) *a." + T + "_count = (u32)a.n_rows; }\n    else\n";
    // wave sum -> LDS -> ONE global atomic per workgroup: atomics on a single word serialise (~11 ns each), and a
    // random-access pipeline launches 8 workgroups per CU
    epilogue += "    {\n        __shared__ u32 s_n_" + T + ";\n        if (threadIdx.x == 0) s_n_" + T + " = 0;\n        __syncthreads();\n";
    epilogue += "        const u64 v = rsq::wave_sum((u64)st.n_" + T + ");\n        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_n_" + T + ", (u32)v);\n";
    epilogue += "        __syncthreads();\n        if (threadIdx.x == 0 && s_n_" + T + ") atomicAdd(a." + T + "_count, s_n_" + T + ");\n    }\n";
}

// Key value(s) of one expression as table words: one word for numbers, ceil(len / 8) words of bytes for strings
// (see rsq_device.h: str_word).  stripChar: CHAR(n) equality ignores trailing spaces (joins); group keys keep the
// exact bytes and the host merges space-equivalent groups, because the group shows the FIRST row's spelling.
// stripMode: -1 by the expression's own type (CHAR keys ignore trailing spaces when stripChar), 0 exact bytes, 1 ignore trailing spaces
std::vector<std::string> Walker::keyWords(Expr* e, const std::string& prefix, bool stripChar, std::vector<std::string>* endsWithSpace, int stripMode) {
    std::vector<std::string> out;
    const std::string v = eg.emit(e);
    if (!e->type.isString()) {
        line("const i64 " + prefix + " = " + toWord(v, e->type) + ";");
        out.push_back(prefix);
        return out;
    }
    const bool strip = stripMode >= 0 ? stripMode == 1 : (stripChar && e->type.tag == RSQ_CHAR);
    if (e->tag == RSQ_E_CONSTANT) {
        // a string constant as a key: its words are literals (the text, NUL padded to the constant's declared length;
        // without its trailing spaces where the comparison ignores them)
        std::string text = e->symbol.substr(0, (size_t)std::max(0, e->type.len));
        const size_t nul = text.find('\0');
        if (nul != std::string::npos) text.resize(nul);
        if (endsWithSpace && e->type.tag == RSQ_CHAR && !strip) endsWithSpace->push_back(!text.empty() && text.back() == ' ' ? "true" : "false");
        if (strip) while (!text.empty() && text.back() == ' ') text.pop_back();
        for (int w = 0; w < (e->type.len + 7) / 8; w++) {
            uint64_t word = 0;
            for (int b = 0; b < 8; b++) { const size_t i = (size_t)w * 8 + (size_t)b; if (i < text.size()) word |= (uint64_t)(unsigned char)text[i] << (8 * b); }
            std::string kv = prefix + "_" + std::to_string(w);
            line("const i64 " + kv + " = (i64)" + std::to_string((unsigned long long)word) + "ull;");
            out.push_back(kv);
        }
        return out;
    }
    if (!strip) {
        // exact bytes: the column is NUL padded to its width (resql_plan.h), so the key words ARE the stored bytes —
        // one unaligned load per word instead of a byte loop per word (32 key words for TPC-H Q10's group-by)
        if (endsWithSpace && e->type.tag == RSQ_CHAR) endsWithSpace->push_back("rsq::ends_with_space(" + v + ")");
        for (int w = 0; w < (e->type.len + 7) / 8; w++) {
            std::string kv = prefix + "_" + std::to_string(w);
            const int rbytes = std::min(8, e->type.len - w * 8);
            line("const i64 " + kv + " = (i64)rsq::ld_bytes<" + std::to_string(rbytes) + ">((" + v + ").p + " + std::to_string(w * 8) + ");");
            out.push_back(kv);
        }
        return out;
    }
    line("const int " + prefix + "_n = rsq::str_len_char(" + v + ");");
    for (int w = 0; w < (e->type.len + 7) / 8; w++) {
        std::string kv = prefix + "_" + std::to_string(w);
        line("const i64 " + kv + " = rsq::str_word(" + v + ", " + prefix + "_n, " + std::to_string(w) + ");");
        out.push_back(kv);
    }
    return out;
}

int Walker::joinKeyStripMode(const Expr* side, const Expr* probeSide, const Expr* buildSide) {
    (void)side;
    return mixedStringKinds(probeSide, buildSide) ? (probeSide->type.tag == RSQ_CHAR ? 1 : 0) : -1;
}

void Walker::padKeyWords(Expr* mine, Expr* other, size_t w0, std::vector<std::string>& keyVars, bool buildSide) {
    if (!mine->type.isString() || !other->type.isString()) return;
    if (mine->type.tag != other->type.tag) {
        const size_t want = (size_t)(std::max(mine->type.len, other->type.len) + 7) / 8;
        while (keyVars.size() - w0 < want) keyVars.push_back("((i64)0)");
        keyVars.push_back(mine->type.tag == RSQ_CHAR ? "((i64)" + std::to_string(mine->type.len) + ")" : "((i64)rsq::str_len_exact(" + eg.emit(mine) + "))");
        return;
    }
    if (mine->type.len == other->type.len) return;
    const size_t want = (size_t)(std::max(mine->type.len, other->type.len) + 7) / 8 + (mine->type.tag == RSQ_CHAR ? 1 : 0);
    const std::string pad = mine->type.tag == RSQ_CHAR && !buildSide ? "((i64)-1)" : "((i64)0)";
    while (keyVars.size() - w0 < want) keyVars.push_back(pad);
}

// ---- hash join build (hashjoin.h:226-256) ---------------------------------------------------
std::string Walker::hashOf(const std::vector<std::string>& keyVars) {
    std::string h = "rsq::hash64((u64)" + keyVars[0] + ")";
    for (size_t i = 1; i < keyVars.size(); i++) h = "rsq::hash64(" + h + " ^ ((u64)" + keyVars[i] + " * 0x9E3779B97F4A7C15ull))";
    return h;
}

// Word w of the slot in `T_s` of a join table.  Join tables keep a slot's words next to each other (array of
// structures): the CAS on the key and the payload stores of an insert fall into one cache line, which the memory side
// then writes back once instead of read-modify-writing three lines; a probe that matches finds the payload in the line
// it already fetched for the key.  (The generic aggregation's tables stay structure-of-arrays: their key words are
// compared one array at a time and their accumulators live in separate blocks anyway.)
std::string Walker::wordAt(const HashTable& ht, const std::string& T, int w) {
    const int nw = std::max<int>(1, (int)(ht.keys.size() + ht.payload.size()));
    if (ht.aos) return "a." + T + "_words[" + T + "_s * " + std::to_string(nw) + " + " + std::to_string(w) + "]";
    return "a." + T + "_words[" + std::to_string(w) + " * a." + T + "_cap + " + T + "_s]";
}

// Home slot of a join key.  RSQ_BLOCKED_HASH=1 (off by default — measured and rejected) makes the hash of one integer key
// of known range BLOCKED: 128 consecutive key values share a hashed base slot and spread, in key order, over the slots
// behind it, so that tables clustered by the key insert and probe neighbouring slots from neighbouring rows.  On MI355X
// that is 2-13x SLOWER (Q3 SF10 0.51 -> 1.1 ms, Q14 SF1 0.09 -> 1.2 ms): the 64 CAS of a wave then land in a handful of
// cache lines, and atomics on one line serialise at the memory side just like atomics on one word.  Scattering the
// inserts over the table is what keeps them fast.
std::string Walker::slotOf(const HashTable& ht, const std::string& T, const std::vector<std::string>& keyVars) {
    (void)ht;
    return hashOf(keyVars) + " & " + T + "_mask";
}

void Walker::consumeBuild(OpNode* o, OpNode* from) {
    if (compactThen(o, [&] { consumeBuildBody(o, from); })) return;
    consumeBuildBody(o, from);
}

void Walker::consumeBuildBody(OpNode* o, OpNode* from) {
    pipe.gridPerCU = 8;
    std::unique_ptr<HashTable> ht(new HashTable());
    ht->id = (int)q.hashTables.size();
    ht->unique = o->singleMatch;
    const std::string T = "ht" + std::to_string(ht->id);
    std::vector<std::string> keyVars;
    std::vector<int> keyFirstWord;           // per key expression: its table word (-1 for multi-word string keys)
    openScope("{");
    int k = 0;
    for (Expr* eq : o->exprs) {
        if (eq->tag != RSQ_E_EQ) failType("The elements of the expression list passed to equalitiesLeftSide(..) need the tag Expr::EQ");
        Expr* l = eq->child;
        q.pool.addId(l);
        size_t w0 = keyVars.size();
        keyFirstWord.push_back(l->type.isString() ? -1 : (int)w0);
        for (auto& kv : keyWords(l, T + "_k" + std::to_string(k++), true, nullptr, joinKeyStripMode(l, eq->child->next, l))) keyVars.push_back(kv);
        padKeyWords(l, eq->child->next, w0, keyVars, true);
        for (size_t w = w0; w < keyVars.size(); w++)
            ht->keys.push_back({w == w0 ? expressionName(l) : expressionName(l) + "#" + std::to_string(w - w0), w == w0 && !l->type.isString() ? l->type : Type(RSQ_BIGINT)});
    }
    // build payload = the attributes of the left child's schema (Values::get(_lChild->_schema)); an attribute that is
    // itself a (one-word) join key is not stored again: a matching probe already holds its value
    for (auto& a : from->schema) {
        auto it = eg.symbols.find(a.name);
        if (it == eg.symbols.end()) failType("hash join build value " + a.name + " has no symbol");
        int alias = -1;
        for (size_t ki = 0; ki < o->exprs.size(); ki++) {
            Expr* l = o->exprs[ki]->child;
            if (l->tag == RSQ_E_ATTRIBUTE && l->symbol == a.name && !l->type.isString() && keyFirstWord[ki] >= 0) alias = keyFirstWord[ki];
        }
        if (alias >= 0) ht->keyAlias.push_back({{a.name, it->second.type}, alias});
        else ht->payload.push_back({a.name, it->second.type});
    }
    // key-domain bitmap (see HashTable): one integer key that is a column of this pipeline's scan with usable statistics
    if (o->exprs.size() == 1 && keyVars.size() == 1 && envInt("RSQ_JOIN_BITMAP", 1, 0, 1)) {
        Expr* l = o->exprs[0]->child;
        auto org = symbolOrigin.find(l->symbol);
        if (l->tag == RSQ_E_ATTRIBUTE && !l->type.isString() && org != symbolOrigin.end() && org->second == -1) {
            int ci = pipe.src->findCol(l->symbol);
            if (ci >= 0 && pipe.src->cols[(size_t)ci].stats.valid && pipe.src->nRows > 0) {
                const ColumnStats& st = pipe.src->cols[(size_t)ci].stats;
                unsigned __int128 range = (unsigned __int128)((__int128)st.max - (__int128)st.min) + 1;
                if (range <= ((unsigned __int128)1 << 28)) { ht->hasBitmap = true; ht->bmMin = st.min; ht->bmBits = (int64_t)range; }
            }
        }
    }
    // ... or, for a table with several key words, a bitmap over ONE integer component (HashTable::hasCompBitmap)
    if (!ht->hasBitmap && keyVars.size() > 1 && envInt("RSQ_JOIN_BITMAP", 1, 0, 1))
        for (size_t ki = 0; ki < o->exprs.size() && !ht->hasCompBitmap; ki++) {
            Expr* l = o->exprs[ki]->child;
            auto org = symbolOrigin.find(l->symbol);
            if (l->tag != RSQ_E_ATTRIBUTE || l->type.isString() || org == symbolOrigin.end() || org->second != -1 || keyFirstWord[ki] < 0) continue;
            const int ci = pipe.src->findCol(l->symbol);
            if (ci < 0 || !pipe.src->cols[(size_t)ci].stats.valid || pipe.src->nRows == 0) continue;
            const ColumnStats& st = pipe.src->cols[(size_t)ci].stats;
            const unsigned __int128 range = (unsigned __int128)((__int128)st.max - (__int128)st.min) + 1;
            if (range > ((unsigned __int128)1 << 26)) continue;
            ht->hasCompBitmap = true; ht->compWord = keyFirstWord[ki]; ht->cbMin = st.min; ht->cbBits = (int64_t)range;
        }
    // capacity: the reference sizes its table lChild.getSize() * 5 / 3 and grows it; ours cannot grow
    // inside a kernel, so it is sized for twice the rows the build pipeline can deliver and re-run
    // at double size if it still overflows (engine.cpp).
    ht->aos = 1 != 0;
    ht->capacity = 0;     // decided by the sizing pass at execute time (engine.cpp)
    // One integer key word whose values can never be INT64_MIN: the key word itself is the slot's state.  A 64-bit CAS
    // from the EMPTY sentinel claims the slot and publishes the key in one memory request (instead of a CAS on a state
    // word plus a key store), a probe step reads one word instead of two dependent ones.  Scattered HBM requests are what
    // a build costs.
    if (keyVars.size() == 1 && o->exprs.size() == 1 && 1) {
        Expr* l = o->exprs[0]->child;
        const int tg = l->type.tag;
        if (tg == RSQ_INT || tg == RSQ_DATE || tg == RSQ_BOOL || (tg == RSQ_CHAR && l->type.len == 1)) ht->keyCas = true;   // widened 32-bit / 8-bit values
        else if ((tg == RSQ_BIGINT || tg == RSQ_DECIMAL) && l->tag == RSQ_E_ATTRIBUTE) {
            auto org = symbolOrigin.find(l->symbol);
            int ci = pipe.src->findCol(l->symbol);
            if (org != symbolOrigin.end() && org->second == -1 && ci >= 0 && pipe.src->cols[(size_t)ci].stats.valid &&
                pipe.src->cols[(size_t)ci].stats.min > INT64_MIN) ht->keyCas = true;
        }
    }
    // Bitmap-rank dictionary (HashTable::rankCapable, kernels/rsq_device.h rank_of): a table that is probed single-match over
    // one integer key with a key bitmap needs no hashing when its build keys prove unique.  The same kernel carries both
    // forms behind a uniform branch on a.<T>_rank; the host decides once, from the sizing pass.
    // (A join probed for ALL matches qualifies too: with unique build keys every probe has at most one.  If such a table carries
    // nothing but its key - TPC-H Q3's customer side - the bitmap IS the table in the rank form: a KEY SET, no entries at all.)
    ht->rankCapable = (ht->unique || 1) && ht->hasBitmap && ht->keyCas && ht->aos && keyVars.size() == 1 &&
                      envInt("RSQ_JOIN_RANK", 1, 0, 1) != 0;
    ht->setOnly = ht->rankCapable && !ht->unique && ht->payload.empty();
    // IDENTITY: the build pipeline is the bare scan of a table in the order of its (engine-owned, hence immutable) key column.  If the
    // sizing pass then finds the keys unique and every row inserted, entry number rank(key) IS the row's number: the build writes
    // its record straight to words[row] - coalesced, streaming - and the arrival buffer and the placement kernel are not needed
    // (TPC-H Q12 builds on all 15 M orders: 240 MB appended, read again and scattered to entries 16 bytes at a time).
    {
        const int ci = o->exprs[0]->child->tag == RSQ_E_ATTRIBUTE ? pipe.src->findCol(o->exprs[0]->child->symbol) : -1;
        ht->identityCapable = ht->rankCapable && !ht->setOnly && from->tag == RSQ_OP_SCAN && ci >= 0 && pipe.src->cols[(size_t)ci].owned &&
                              pipe.src->cols[(size_t)ci].stats.valid && pipe.src->cols[(size_t)ci].stats.ascending;
        ht->uniqueKnown = ht->identityCapable && pipe.src->cols[(size_t)ci].stats.strictlyAscending && pipe.src->nRows < 0xffffffffll;
    }
    // DIRECT (HashTable::directCapable): every payload value is a column of this scan - it is, over a bare scan - and the statistics promise
    // unique keys: if the keys also fill their range (known when the form is decided), the probes read the table's columns themselves
    if (ht->uniqueKnown) {
        ht->directCapable = true;
        for (auto& p : ht->payload) {
            const int pc = pipe.src->findCol(p.name);
            if (pc < 0 || pipe.src->cols[(size_t)pc].type.tag != p.type.tag || pipe.src->cols[(size_t)pc].type.len != p.type.len) { ht->directCapable = false; break; }
            ht->directCols.push_back(pc);
        }
        if (!ht->directCapable) ht->directCols.clear();
        ht->directSrc = pipe.src;
        ht->directKeyCol = pipe.src->findCol(o->exprs[0]->child->symbol);
    }
    // (a table that may become a rank dictionary keeps its bitmap in the interleaved layout, rsq_device.h bmi_word)
    ht->bmInterleaved = ht->rankCapable;
    const std::string bmw = ht->bmInterleaved ? "rsq::bmi_word(d)" : "d >> 5";
    // (a key outside the range the statistics promised sets no bit and raises ERR_GROUP_OVERFLOW: the host fails the execution)
    bool checkKey = true, combineBits = false;
    if (ht->hasBitmap && o->exprs[0]->child->tag == RSQ_E_ATTRIBUTE) {
        const int ci = pipe.src->findCol(o->exprs[0]->child->symbol);
        if (ci >= 0 && pipe.src->cols[(size_t)ci].owned && !envInt("RSQ_CHECK_STATS", 0, 0, 1)) checkKey = false;      // (engine-owned columns cannot change)
        // a table scanned in the order of its build key (column statistics): the rows of a wave fall into a few bitmap words, and
        // the lanes that meet in one word set their bits with ONE atomic (rsq_device.h bm_set_combined).  Memory-side atomics
        // run at ~25 G/s chip-wide: a build over all 15 M orders (TPC-H Q12) spent 0.6 of its 0.73 ms on them.
        if (ci >= 0 && pipe.src->cols[(size_t)ci].stats.valid && pipe.src->cols[(size_t)ci].stats.ascending && 1)
            combineBits = true;
    }
    const std::string setBit = combineBits ? "rsq::bm_set_combined(a." + T + "_bm, (u32)(" + bmw + "), 1u << (d & 31));"
                                           : "atomicOr(&a." + T + "_bm[" + bmw + "], 1u << (d & 31));";
    const std::string bitSet = !ht->hasBitmap ? std::string() : !checkKey ? "const u64 d = (u64)(" + keyVars[0] + " - a." + T + "_bmmin); " :
                               "const u64 d0 = (u64)(" + keyVars[0] + " - a." + T + "_bmmin); if (d0 >= a." + T +
                               "_bmbits) atomicOr(a.err, (u32)rsq::ERR_GROUP_OVERFLOW); const u64 d = d0 < a." + T + "_bmbits ? d0 : 0; ";
    addArg(T + "_state", "u32*", 0); addArg(T + "_words", "i64*", 0); addArg(T + "_cap", "u64", 0); addArg(T + "_count", "u32*", 0);
    addArg(T + "_countonly", "u64", 0);
    if (ht->hasBitmap) { addArg(T + "_bm", "u32*", 0); addArg(T + "_bmmin", "i64", (uint64_t)ht->bmMin); addArg(T + "_bmbits", "u64", (uint64_t)ht->bmBits); }
    if (ht->hasCompBitmap) {
        // every build row sets its component's bit, in the sizing pass and in the build alike (a value outside the range the statistics promised raises ERR_GROUP_OVERFLOW)
        addArg(T + "_c_bm", "u32*", 0); addArg(T + "_c_bmmin", "i64", (uint64_t)ht->cbMin); addArg(T + "_c_bmbits", "u64", (uint64_t)ht->cbBits);
        line("{ const u64 cd = (u64)(" + keyVars[(size_t)ht->compWord] + " - a." + T + "_c_bmmin); if (cd < a." + T + "_c_bmbits) { const u32 cb = 1u << (cd & 31); if (!(a." + T +
             "_c_bm[cd >> 5] & cb)) atomicOr(&a." + T + "_c_bm[cd >> 5], cb); } else atomicOr(a.err, (u32)rsq::ERR_GROUP_OVERFLOW); }");
    }
    // sizing pass: the same pipeline run once with countonly = 1 tells the host how many entries to expect — and, for a
    // table that could be a rank dictionary, whether two build rows share a key (a bit that is already set)
    countPerThread(T, ht->identityCapable);
    if (ht->rankCapable) {
        addArg(T + "_rank", "u64", 0); addArg(T + "_temp", "i64*", 0);
        openScope("if (a." + T + "_countonly) {");
        line("st.n_" + T + "++;");
        line("{ " + bitSet + "const u32 b = 1u << (d & 31); if (atomicOr(&a." + T + "_bm[" + bmw + "], b) & b) atomicOr(a.err, (u32)rsq::NOTE_BUILD_KEYS_NOT_UNIQUE); }");
        closeScope();
        if (ht->setOnly) {
            // key set: the bit is everything; a bit that is already set means the build side changed since the sizing pass
            openScope("else if (a." + T + "_rank) {");
            line("{ " + bitSet + "const u32 b = 1u << (d & 31); if (atomicOr(&a." + T + "_bm[" + bmw + "], b) & b) atomicOr(a.err, (u32)rsq::NOTE_BUILD_KEYS_NOT_UNIQUE); }");
            line("st.n_" + T + "++;");
            closeScope();
            openScope("else {");
        } else {
        openScope("else if (a." + T + "_rank) {");
        // The record goes to the arrival-order buffer, into the region of the wave that produced it: a.<T>_treg records
        // per wave (the host sizes the regions at four times the mean from the sizing pass; tiles are dealt to the waves
        // round-robin, so every wave sees an even sample of the table).  No atomics: a returning atomic on ONE counter word
        // serialises at ~11 ns, and even one reservation per wave and 256 records made this pipeline 2x slower.  A wave's
        // fill count lives in LDS, because the lanes of a wave reach this point in diverged groups; it is written to
        // a.<T>_tused[wave] at the end, where the placement kernel finds it.  A wave that overflows its region says so
        // (the host then keeps the hash form).
        line("{ " + bitSet + setBit + " }");
        if (ht->identityCapable) {
            addArg(T + "_ident", "u64", 0);
            // Records at the row's number: the 128 rows of a tile are 128 x NW consecutive words of the table.  A lane storing ITS rows' words
            // writes 8 bytes at a stride of 16 x NW per instruction (TPC-H Q12's orders table: 240 MB of records in 94 us, Q10's customers:
            // seven words per row in 48 us); the records go through the wave's LDS region instead and leave as the wave's 16-byte stores,
            // every line whole (the flush behind the tile's rows, finishPipeline).  The rows behind the last tile store directly.
            const std::string NWI = std::to_string(1 + (int)ht->payload.size());
            stateDecl += "    i64* rec_" + T + ";\n    bool in_tile = true;\n";
            prologue += "    __shared__ __attribute__((aligned(16))) i64 s_rec_" + T + "[(RSQ_BLOCK_THREADS / 64) * 128 * " + NWI + "];\n    st.rec_" + T + " = s_rec_" + T +
                        " + (threadIdx.x >> 6) * 128 * " + NWI + ";\n";
            pipe.extraLdsBytes += (pipe.blockThreads / 64) * 128 * 8 * (1 + (int)ht->payload.size());
            // (a DIRECT table keeps no records at all: the probes read the columns - only the key bits are set, for the index)
            const std::string keepsRecords = ht->directCapable ? " && !a." + T + "_direct" : "";
            if (ht->directCapable) addArg(T + "_direct", "u64", 0);
            postTile += "            if (a." + T + "_ident && a." + T + "_rank && !a." + T + "_countonly" + keepsRecords + ") rsq::flush_tile_records<" + NWI + ">(st.rec_" + T + ", a." + T + "_words + (u64)(($TILE) << 7) * " + NWI +
                        "ull, lane);\n";
            openScope("if (a." + T + "_ident) {");
            if (ht->directCapable) openScope("if (!a." + T + "_direct) {");
            line("i64* rec = st.in_tile ? st.rec_" + T + " + (u32)(lr & 127) * " + NWI + "u : a." + T + "_words + (u64)(row - a.row0) * " + NWI + "ull;");
            line("rec[0] = " + keyVars[0] + ";");
            int iw = 1;
            for (auto& p : ht->payload) line("rec[" + std::to_string(iw++) + "] = " + toWord(eg.symbols[p.name].var, p.type) + ";");
            if (ht->directCapable) closeScope();
            closeScope();
            openScope("else {");
        }
        addArg(T + "_treg", "u64", 0); addArg(T + "_tused", "u32*", 0);
        stateDecl += "    u32* tch_" + T + ";\n";
        prologue += "    __shared__ u32 s_tch_" + T + "[RSQ_BLOCK_THREADS / 64];\n    st.tch_" + T + " = s_tch_" + T + " + (threadIdx.x >> 6);\n" +
                    "    if ((threadIdx.x & 63) == 0) st.tch_" + T + "[0] = 0u;\n";
        const int nw = 1 + (int)ht->payload.size();
        const std::string NW = std::to_string(nw);
        line("u64 " + T + "_t;");
        openScope("{");
        line("const u64 act = __ballot(1);");
        line("const int ln = (int)(threadIdx.x & 63), leader = __ffsll((long long)act) - 1;");
        line("const u32 pos = st.tch_" + T + "[0];");
        line(T + "_t = (u64)pos + (u64)__popcll(act & ((1ull << ln) - 1ull));");
        line("if (ln == leader) st.tch_" + T + "[0] = pos + (u32)__popcll(act);");
        closeScope();
        {
            openScope("if (" + T + "_t < a." + T + "_treg) {");
            line("i64* rec = a." + T + "_temp + (((u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * a." + T + "_treg + " + T + "_t) * " + NW + ";");
            line("rec[0] = " + keyVars[0] + ";");
            int tw = 1;
            for (auto& p : ht->payload)
                line("rec[" + std::to_string(tw++) + "] = " + toWord(eg.symbols[p.name].var, p.type) + ";");
            closeScope();
            line("else atomicOr(a.err, (u32)rsq::NOTE_BUILD_KEYS_NOT_UNIQUE);      // the region is full: this table is not for the dictionary");
        }
        if (ht->identityCapable) closeScope();
        epilogue += "    if (a." + T + "_rank && !a." + T + "_countonly && (threadIdx.x & 63) == 0) {\n        const u32 used = st.tch_" + T + "[0];\n" +
                    "        a." + T + "_tused[(u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = used < a." + T + "_treg ? used : (u32)a." + T + "_treg;\n    }\n";
        line("st.n_" + T + "++;");
        closeScope();
        openScope("else {");
        }
    } else
    openScope("if (a." + T + "_countonly) { st.n_" + T + "++; } else {");
    line("const u64 " + T + "_mask = a." + T + "_cap - 1;");
    line("u64 " + T + "_s = " + slotOf(*ht, T, keyVars) + ";");
    line("u64 " + T + "_n = 0;");
    // (measurement only, wrong results: 1 no payload stores, 2 a plain store into the home slot instead of the CAS loop,
    // 4 no insert at all — to see what each part of an insert costs)
    const int dbgBuild = 0;
    openScope("for (;; " + T + "_n++) {");
    line("if (" + T + "_n > " + T + "_mask) { atomicOr(a.err, (u32)rsq::ERR_HT_FULL); break; }");
    if (dbgBuild & 4) line("break;");
    else if ((dbgBuild & 2) && ht->keyCas) line(wordAt(*ht, T, 0) + " = " + keyVars[0] + "; break;");
    if (ht->keyCas)
        line("if (atomicCAS(reinterpret_cast<unsigned long long*>(&" + wordAt(*ht, T, 0) + "), 0x8000000000000000ull, (unsigned long long)" +
             keyVars[0] + ") == 0x8000000000000000ull) break;");
    else line("if (atomicCAS(&a." + T + "_state[" + T + "_s], 0u, 1u) == 0u) break;");
    line(T + "_s = (" + T + "_s + 1) & " + T + "_mask;");
    closeScope();
    openScope("if (" + T + "_n <= " + T + "_mask) {");
    int w = 0;
    for (auto& kv : keyVars) {
        if (!ht->keyCas) line(wordAt(*ht, T, w) + " = " + kv + ";");
        w++;
    }
    for (auto& p : ht->payload) {
        if (dbgBuild & 5) { w++; continue; }
        line(wordAt(*ht, T, w++) + " = " + toWord(eg.symbols[p.name].var, p.type) + ";");
    }
    line("st.n_" + T + "++;");
    if (ht->hasBitmap) {
        line("{ " + bitSet + setBit + " }");
    }
    closeScope();
    closeScope();
    closeScope();
    pipe.sink = SinkKind::BUILD;
    pipe.buildTable = ht->id;
    o->hashTable = ht->id;
    explainSteps.push_back("build hash table " + T + " (" + std::to_string(ht->keys.size()) + " key(s), " +
                           std::to_string(ht->payload.size()) + " payload word(s), sized by a counting pass" +
                           (ht->hasBitmap ? ", key bitmap of " + std::to_string((long long)ht->bmBits) + " bits" : "") +
                           (ht->keyCas ? ", key word is the slot state" : "") +
                           (ht->setOnly ? "; nothing but the bitmap when the build keys prove unique" :
                            ht->rankCapable ? "; a bitmap-rank dictionary instead when the build keys prove unique" : "") + ")");
    q.hashTables.push_back(std::move(ht));
}

// ---- hash join probe (hashjoin.h:118-214) ---------------------------------------------------
// key words of the probe side of join `o` (emits their computation at the current position)
void Walker::probeKeys(OpNode* o, const std::string& T, std::vector<std::string>& keyVars, std::vector<std::string>& probeKeyNames) {
    int k = 0;
    for (Expr* eq : o->exprs) {
        Expr* r = eq->child->next;
        q.pool.addId(r);
        const size_t w0 = keyVars.size();
        for (auto& kv : keyWords(r, T + "_p" + std::to_string(k++), true, nullptr, joinKeyStripMode(r, r, eq->child))) keyVars.push_back(kv);
        padKeyWords(r, eq->child, w0, keyVars, false);
        // (only a one-word key can stand in for the build key of the matched entry, see tryJoinEntry)
        probeKeyNames.push_back(keyVars.size() - w0 == 1 ? expressionName(r) : std::string());
        for (size_t w = w0 + 1; w < keyVars.size(); w++) probeKeyNames.push_back(std::string());
    }
}

void Walker::consumeProbe(OpNode* o, OpNode* from) {
    pipe.gridPerCU = 8;
    HashTable& ht = *q.hashTables[(size_t)o->hashTable];
    const std::string T = "ht" + std::to_string(ht.id);
    o->schema = o->child[0]->schema;
    for (auto& a : from->schema) o->schema.push_back(a);
    if (!q.requestAll) o->schema = prune(o->schema, requestOf[o]);
    std::vector<std::string> keyVars, probeKeyNames;
    openScope("{");
    probeKeys(o, T, keyVars, probeKeyNames);
    if (keyVars.size() != ht.keys.size()) failUnsupported("string join keys of different declared lengths");
    addArg(T + "_state", "const u32*", 0); addArg(T + "_words", "const i64*", 0); addArg(T + "_cap", "u64", 0);
    bool compScope = false;
    if (ht.hasCompBitmap && !compacted) {
        // the probe in front of which the compaction is cut (or a pipeline without one): the component's bit is tested right here
        int ki = -1, w = 0;
        for (size_t k = 0; k < o->exprs.size(); k++) {
            if (w == ht.compWord) { ki = (int)k; break; }
            Expr* r = o->exprs[k]->child->next;
            w += r->type.isString() ? (r->type.len + 7) / 8 : 1;
        }
        if (ki >= 0 && !o->exprs[(size_t)ki]->child->next->type.isString()) {
            Expr* r = o->exprs[(size_t)ki]->child->next;
            const std::string C = T + "_c";
            addArg(C + "_bm", "const u32*", 0); addArg(C + "_bmmin", "i64", (uint64_t)ht.cbMin); addArg(C + "_bmbits", "u64", (uint64_t)ht.cbBits);
            auto org = r->tag == RSQ_E_ATTRIBUTE ? symbolOrigin.find(r->symbol) : symbolOrigin.end();
            auto sym = r->tag == RSQ_E_ATTRIBUTE ? eg.symbols.find(r->symbol) : eg.symbols.end();
            bool already = false;
            for (auto& pf : bitmapPrefetch) already = already || pf.first == C;
            const std::string d = "(u64)(" + keyVars[(size_t)ht.compWord] + " - a." + C + "_bmmin)";
            if (!already && ht.cbBits <= (1 << 20) && org != symbolOrigin.end() && org->second == -1 && sym != eg.symbols.end() && sym->second.var.compare(0, 2, "v_") == 0) {
                const int col = atoi(sym->second.var.c_str() + 2);      // (a small bitmap: the word arrives with the tile, for both rows of the lane)
                bitmapPrefetch.push_back({C, col, false});
                const std::string call = "rsq::bm_word(a." + C + "_bm, a." + C + "_bmmin, a." + C + "_bmbits, (i64)";
                rowParams += ", const u32 pf_" + C;
                rowArgsTail += ", " + call + "a.c" + std::to_string(col) + "[r])";
                rowArgsTailGuarded += ", (valid ? " + call + "a.c" + std::to_string(col) + "[r]) : 0u)";
                openScope("if (rsq::bit_of_word(pf_" + C + ", " + d + ", a." + C + "_bmbits)) {");
            } else openScope("if (rsq::bit_in(a." + C + "_bm, " + d + ", a." + C + "_bmbits)) {");
            compScope = true;
            selective = true;
            explainSteps.push_back("component bitmap of " + T + " tested in front of the probe");
        }
    }
    if (ht.hasCompBitmap && compacted && inStage2) {
        // the component's value on the probe side, if it is a column of this pipeline's scan: known at the top of stage 2, where the test goes
        int ki = -1, w = 0;
        for (size_t k = 0; k < o->exprs.size(); k++) {
            if (w == ht.compWord) { ki = (int)k; break; }
            Expr* r = o->exprs[k]->child->next;
            w += r->type.isString() ? (r->type.len + 7) / 8 : 1;
        }
        if (ki >= 0) {
            Expr* r = o->exprs[(size_t)ki]->child->next;
            auto org = r->tag == RSQ_E_ATTRIBUTE ? symbolOrigin.find(r->symbol) : symbolOrigin.end();
            auto sym = r->tag == RSQ_E_ATTRIBUTE ? eg.symbols.find(r->symbol) : eg.symbols.end();
            if (org != symbolOrigin.end() && org->second == -1 && sym != eg.symbols.end() && sym->second.var.compare(0, 2, "q_") == 0 && !r->type.isString()) {
                addArg(T + "_c_bm", "const u32*", 0); addArg(T + "_c_bmmin", "i64", (uint64_t)ht.cbMin); addArg(T + "_c_bmbits", "u64", (uint64_t)ht.cbBits);
                compFilters.push_back({T, r->symbol, sym->second.var, ht.cbBits});      // (compactThen places the test: in stage 1 if it can, else at the top of stage 2)
                explainSteps.push_back("component bitmap of " + T + " tested in front of the compaction");
            }
        }
    }
    if (ht.hasBitmap) {
        // keys outside the build side's [min, max] or with a clear bit cannot match: skip the table altogether
        addArg(T + "_bm", "const u32*", 0); addArg(T + "_bmmin", "i64", (uint64_t)ht.bmMin); addArg(T + "_bmbits", "u64", (uint64_t)ht.bmBits);
        line("const u64 " + T + "_d = (u64)(" + keyVars[0] + " - a." + T + "_bmmin);");
        // When the probe key is a column of this pipeline's scan, its bitmap word is fetched by the scan skeleton for BOTH
        // rows of the lane (and every tile in flight) before the first row is processed, and handed to the row function:
        // the two row functions of a lane otherwise run one after the other, each with its own dependent load — a cache
        // round trip per row that nothing overlaps (TPC-H Q3's lineitem pipeline spent a quarter of its time there).
        int pfCol = -1;
        {
            Expr* r = o->exprs[0]->child->next;
            auto org = r->tag == RSQ_E_ATTRIBUTE ? symbolOrigin.find(r->symbol) : symbolOrigin.end();
            auto sym = r->tag == RSQ_E_ATTRIBUTE ? eg.symbols.find(r->symbol) : eg.symbols.end();
            // ... worth it when the table is clustered by the key (column statistics): the 64 lanes of a wave then read one or
            // two cache lines.  For keys in random order (orders.o_custkey) a wave's load touches 64 lines, and fetching for
            // the rows the filter in front would have dropped made TPC-H Q3's orders pipeline 30 % slower.
            if (!compacted && o->exprs.size() == 1 && org != symbolOrigin.end() && org->second == -1 && sym != eg.symbols.end() &&
                sym->second.var.compare(0, 2, "v_") == 0 && !r->type.isString()) {
                const int ci = pipe.src->findCol(r->symbol);
                const int mode = 1;       // 0 never, 1 clustered keys (or gated, below), 2 always
                if (ci >= 0 && (mode == 2 || (mode == 1 && pipe.src->cols[(size_t)ci].stats.valid && pipe.src->cols[(size_t)ci].stats.ascending)))
                    pfCol = atoi(sym->second.var.c_str() + 2);
                // (Keys in random order behind a selection: fetching their bitmap words with the tile for the rows the selection passes
                // was tried and measured no gain - TPC-H Q3's orders pipeline 0.306-0.308 against 0.305-0.315 ms for the query: what
                // its probes cost, 34 of its 88 us, is the cache lines they move from the L2 - 7 M probes of a 187 KB bitmap, one
                // 128-byte line each - not their latency.)
            }
            for (auto& pf : bitmapPrefetch) if (pf.first == T) pfCol = -1;        // (one probe per table and pipeline)
        }
        if (pfCol >= 0) {
            bitmapPrefetch.push_back({T, pfCol, ht.bmInterleaved});
            const std::string call = std::string(ht.bmInterleaved ? "rsq::bmi_load(a." : "rsq::bm_word(a.") + T + "_bm, a." + T + "_bmmin, a." + T + "_bmbits, (i64)";
            rowParams += ", const u32 pf_" + T;
            rowArgsTail += ", " + call + "a.c" + std::to_string(pfCol) + "[r])";
            rowArgsTailGuarded += ", (valid ? " + call + "a.c" + std::to_string(pfCol) + "[r]) : 0u)";
            openScope("if (" + T + "_d < a." + T + "_bmbits && ((pf_" + T + " >> (" + T + "_d & 31)) & 1u)) {");
        } else
        {
            // DENSE: a rank dictionary whose build keys fill their whole range (TPC-H's c_custkey 1..n: every bit is set and rank(key) is the key's
            // offset) needs neither the bit nor the rank block - one random 32-byte access less per probe (engine_pipelines.cpp sizeJoinTable)
            const std::string bit = "((a." + T + "_bm[" + (ht.bmInterleaved ? "rsq::bmi_word(" + T + "_d)" : T + "_d >> 5") + "] >> (" + T + "_d & 31)) & 1u)";
            if (ht.rankCapable) { addArg(T + "_dense", "u64", 0); openScope("if (" + T + "_d < a." + T + "_bmbits && (a." + T + "_dense || " + bit + ")) {"); }
            else openScope("if (" + T + "_d < a." + T + "_bmbits && " + bit + ") {");
        }
        selective = true;
    }
    // the table walk (dependent random accesses) runs behind the wave compaction when the pipeline is selective
    const bool cut = compactThen(o, [&] {
        std::vector<std::string> kv2, names2;
        openScope("{");
        probeKeys(o, T, kv2, names2);
        probeTable(o, ht, T, kv2, names2);
        closeScope();
    });
    if (!cut) probeTable(o, ht, T, keyVars, probeKeyNames);
    if (ht.hasBitmap) closeScope();
    if (compScope) closeScope();
    closeScope();
}

// what a match exposes: the build side's values become symbols (hashjoin.h:146-147 / 204-205), then the parent consumes
void Walker::consumeMatch(OpNode* o, HashTable& ht, const std::string& T, const std::vector<std::string>& keyVars,
                  const std::vector<std::string>& probeKeyNames) {
    int w = (int)ht.keys.size();
    for (auto& p : ht.payload) {
        std::string var = T + "_v" + std::to_string(w);
        std::string word = wordAt(ht, T, w);
        if (ht.directCapable) {
            // (the direct form: the value of the build table's column at row T_s - a string's address is base + T_s * width, no load at all)
            const std::string src = "a." + T + "_src" + std::to_string(w);
            addArg(T + "_direct", "u64", 0); addArg(T + "_src" + std::to_string(w), "const char*", 0);
            const std::string direct = p.type.isString() ? "(i64)(u64)(" + src + " + " + T + "_s * " + std::to_string(p.type.len) + "ull)"
                                                         : "(i64)(reinterpret_cast<const " + ExprGen::ctype(p.type) + "*>(" + src + ")[" + T + "_s])";
            line("const i64 " + T + "_w" + std::to_string(w) + " = a." + T + "_direct ? " + direct + " : " + word + ";");
            word = T + "_w" + std::to_string(w);
        }
        line("const " + ExprGen::ctype(p.type) + " " + var + " = " + fromWord(word, p.type) + ";");
        eg.symbols[p.name] = Sym{var, p.type};
        symbolOrigin[p.name] = ht.id; symbolWord[p.name] = w;
        w++;
    }
    // build-side attributes that are key values: equal to this row's probe key, nothing to load
    for (auto& al : ht.keyAlias) {
        std::string var = T + "_a" + std::to_string(al.second) + "_" + std::to_string(w);
        line("const " + ExprGen::ctype(al.first.type) + " " + var + " = " + fromWord(keyVars[(size_t)al.second], al.first.type) + ";");
        eg.symbols[al.first.name] = Sym{var, al.first.type};
        symbolOrigin[al.first.name] = ht.id; symbolWord[al.first.name] = al.second;
        w++;
    }
    // probe-side key attributes are equal to the build keys of the matched entry
    for (size_t i = 0; i < probeKeyNames.size(); i++)
        if (!probeKeyNames[i].empty() && symbolOrigin.count(probeKeyNames[i]) && symbolOrigin[probeKeyNames[i]] == -1) probeKeyOf[probeKeyNames[i]] = {ht.id, (int)i};
    int prevMatch = matchSlotTable; bool prevMulti = multiMatchAbove;
    slotVar[ht.id] = T + "_s";
    if (o->singleMatch) matchSlotTable = ht.id; else { multiMatchAbove = true; }
    explainSteps.push_back(std::string("probe ") + T + (o->singleMatch ? " (single match)" : " (all matches)"));
    selective = true;                       // whatever follows a join probe sees only the matching rows
    {
        ProbeInScope ps{ht.id, o->singleMatch, ht.rankCapable, {}};
        for (Expr* eq : o->exprs) { Expr* r = eq->child->next; ps.keySymbols.push_back(r->tag == RSQ_E_ATTRIBUTE && !r->type.isString() ? r->symbol : std::string()); }
        probesInScope.push_back(ps);
    }
    consume(o->parent, o);
    probesInScope.pop_back();
    matchSlotTable = prevMatch; multiMatchAbove = prevMulti;
}

void Walker::probeTable(OpNode* o, HashTable& ht, const std::string& T, const std::vector<std::string>& keyVars,
                const std::vector<std::string>& probeKeyNames) {
    if (ht.rankCapable && o->singleMatch) {
        // both forms of the table behind a uniform branch: the entry of a key whose bit is set (tested above) is entry
        // number rank(key) of the dictionary — or the first key-equal slot of the hash walk when the host kept the hash form
        addArg(T + "_rank", "u64", 0);
        line("u64 " + T + "_s = 0; bool " + T + "_hit = false;");
        openScope("if (a." + T + "_rank) {");
        {
            const int dbgRank = 0;      // (measurement only, wrong results: 1 the key offset, 2 its hash instead of the rank)
            const std::string dd = "(u64)(" + keyVars[0] + " - a." + T + "_bmmin)";
            if (dbgRank == 1) line(T + "_s = " + dd + " & (a." + T + "_cap - 1);");
            else if (dbgRank == 2) line(T + "_s = rsq::hash64(" + dd + ") & (a." + T + "_cap - 1);");
            else
            { addArg(T + "_dense", "u64", 0); line(T + "_s = a." + T + "_dense ? " + dd + " : rsq::rank_of(a." + T + "_bm, " + dd + ");"); }
        }
        line(T + "_hit = true;");
        closeScope();
        openScope("else {");
        line("const u64 " + T + "_mask = a." + T + "_cap - 1;");
        line(T + "_s = " + slotOf(ht, T, keyVars) + ";");
        openScope("for (u64 " + T + "_n = 0; " + T + "_n <= " + T + "_mask; " + T + "_n++, " + T + "_s = (" + T + "_s + 1) & " + T + "_mask) {");
        line("const i64 " + T + "_kk = " + wordAt(ht, T, 0) + ";");
        line("if (" + T + "_kk == (i64)0x8000000000000000ull) break;");
        line("if (" + T + "_kk == " + keyVars[0] + ") { " + T + "_hit = true; break; }");
        closeScope();
        closeScope();
        openScope("if (" + T + "_hit) {");
        consumeMatch(o, ht, T, keyVars, probeKeyNames);
        closeScope();
        return;
    }
    if (ht.rankCapable) {
        // all matches of a table that may be a rank dictionary (unique build keys): the walk below in the hash form; in the rank
        // form the one entry of a key whose bit is set (tested above) - one pass through the same loop body
        addArg(T + "_rank", "u64", 0);
        line("const u64 " + T + "_mask = a." + T + "_cap - 1;");
        if (ht.setOnly) line("u64 " + T + "_s = a." + T + "_rank ? 0ull : " + slotOf(ht, T, keyVars) + ";");
        else {
            addArg(T + "_dense", "u64", 0);
            const std::string dd = "(u64)(" + keyVars[0] + " - a." + T + "_bmmin)";
            line("u64 " + T + "_s = a." + T + "_rank ? (a." + T + "_dense ? " + dd + " : rsq::rank_of(a." + T + "_bm, " + dd + ")) : " + slotOf(ht, T, keyVars) + ";");
        }
        openScope("for (u64 " + T + "_n = 0; " + T + "_n <= " + T + "_mask; " + T + "_n++, " + T + "_s = (" + T + "_s + 1) & " + T + "_mask) {");
        line("bool " + T + "_eq = true;");
        openScope("if (!a." + T + "_rank) {");
        line("const i64 " + T + "_kk = " + wordAt(ht, T, 0) + ";");
        line("if (" + T + "_kk == (i64)0x8000000000000000ull) break;");
        line(T + "_eq = " + T + "_kk == " + keyVars[0] + ";");
        closeScope();
        openScope("if (" + T + "_eq) {");
        consumeMatch(o, ht, T, keyVars, probeKeyNames);
        closeScope();
        line("if (a." + T + "_rank) break;");
        closeScope();
        return;
    }
    line("const u64 " + T + "_mask = a." + T + "_cap - 1;");
    line("u64 " + T + "_s = " + slotOf(ht, T, keyVars) + ";");
    openScope("for (u64 " + T + "_n = 0; " + T + "_n <= " + T + "_mask; " + T + "_n++, " + T + "_s = (" + T + "_s + 1) & " + T + "_mask) {");
    std::string cond;
    if (ht.keyCas) {
        line("const i64 " + T + "_kk = " + wordAt(ht, T, 0) + ";");
        line("if (" + T + "_kk == (i64)0x8000000000000000ull) break;");
        cond = T + "_kk == " + keyVars[0];
    } else {
        line("if (a." + T + "_state[" + T + "_s] == 0u) break;");
        for (size_t i = 0; i < keyVars.size(); i++)
            cond += (i ? " && " : "") + wordAt(ht, T, (int)i) + " == " + keyVars[i];
    }
    openScope("if (" + cond + ") {");
    consumeMatch(o, ht, T, keyVars, probeKeyNames);
    if (o->singleMatch) line("break;");
    closeScope();
    closeScope();
}

}  // namespace cg
}  // namespace rsq
