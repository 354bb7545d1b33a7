// tail.cpp — from the device's aggregate table to ReSQL's result relation, on the host.
//
// What the reference does after its aggregation hash table is complete (reference
// src/operators/aggregation.h:298-343 consumeAggregateFlounder: scan the table in slot order, AVG =
// (sum * 100) / count; projection.h:62-72; materialize.h:78-220; orderby.h:87-136 + qlib/sort.h) runs
// here over the #groups rows the device produced — same values, same types, same order.
#include <algorithm>
#include <cstring>
#include <functional>
#include <map>
#include <thread>

#include "engine_internal.h"
#include "hostpar.h"

namespace rsq {

// fn(begin, end, part) over [0, n) on the host's worker pool (hostpar.h); small ranges stay on the calling thread.
// An exception in any part (e.g. a division by zero in a projection) is re-thrown on the calling thread.
static int tailThreads(size_t n) { return partsFor(n); }
static void parallelFor(size_t n, int parts, const std::function<void(size_t, size_t, int)>& fn) { parallelRanges(n, parts, fn); }

int schemaTupleSize(const Schema& s) { int n = 0; for (auto& a : s) n += sizeInTuple(a.type, true); return n; }
int schemaOffset(const Schema& s, const std::string& name) {
    int off = 0;
    for (auto& a : s) { if (a.name == name) return off; off += sizeInTuple(a.type, true); }
    failType("The attribute " + name + " was not found in the schema");
}

namespace {

// ---- scalar evaluation with the device code's semantics, symbols resolved to slots once ----------
struct HostExpr {
    int kind = 0;              // 0 symbol slot, 1 constant, 2 operator
    int slot = -1;
    int tag = 0;
    Type type, opType;
    Val constant{};
    bool int16Cast = false;    // TYPECAST INT -> BIGINT as the reference's JIT executes it (RSQ_COMPAT_JIT_INT16_CAST)
    std::vector<HostExpr> kids;
};

struct HostCompiler {
    std::vector<std::string> names;      // symbol table: slot -> name
    std::vector<Type> types;
    bool int16Cast = false;

    int slotOf(const std::string& n) const {
        for (size_t i = 0; i < names.size(); i++) if (names[i] == n) return (int)i;
        return -1;
    }
    int define(const std::string& n, const Type& t) {
        int s = slotOf(n);
        if (s < 0) { names.push_back(n); types.push_back(t); s = (int)names.size() - 1; }
        else types[(size_t)s] = t;
        return s;
    }

    HostExpr compile(Expr* e) {
        HostExpr h; h.type = e->type; h.int16Cast = int16Cast;
        int s = slotOf(expressionName(e));
        if (s >= 0) { h.kind = 0; h.slot = s; return h; }
        switch (e->structure) {
            case LITERAL:
                if (e->tag == RSQ_E_CONSTANT) {
                    h.kind = 1;
                    if (e->type.isString()) h.constant.s = e->symbol.c_str(); else h.constant.i = e->ival;
                    return h;
                }
                if (e->tag == RSQ_E_STAR) { h.kind = 1; h.constant.i = 0; return h; }
                failType("attribute " + e->symbol + " is not available after the aggregation");
            case UNARY:
                h.kind = 2; h.tag = e->tag;
                if (e->tag == RSQ_E_COUNT) return h;
                h.opType = e->child->type;
                h.kids.push_back(compile(e->child));
                return h;
            case BINARY:
                h.kind = 2; h.tag = e->tag; h.opType = e->child->type;
                if (e->tag == RSQ_E_LIKE && (!e->child->type.isString() || !e->child->next->type.isString()))
                    failType("LIKE on a CHAR(1) operand is undefined in the reference");
                h.kids.push_back(compile(e->child));
                h.kids.push_back(compile(e->child->next));
                return h;
            case OTHER:
                h.kind = 2; h.tag = RSQ_E_CASE;
                for (Expr* c = e->child; c; c = c->next) {
                    if (c->tag == RSQ_E_WHENTHEN) {
                        HostExpr wt; wt.kind = 2; wt.tag = RSQ_E_WHENTHEN; wt.type = c->type;
                        wt.kids.push_back(compile(c->child)); wt.kids.push_back(compile(c->child->next));
                        h.kids.push_back(wt);
                    } else h.kids.push_back(compile(c));
                }
                return h;
            default: failType("emitExpression(..)");
        }
    }
};

int cmpVal(Val a, Val b, const Type& t) {
    if (t.tag == RSQ_DATE || t.tag == RSQ_INT) {
        int32_t x = (int32_t)(uint32_t)a.i, y = (int32_t)(uint32_t)b.i;
        return x < y ? -1 : x > y;
    }
    return a.i < b.i ? -1 : a.i > b.i;
}
bool strEqChar(const char* a, const char* b) {
    while (*a && *b) { if (*a != *b) return false; a++; b++; }
    while (*a) { if (*a != ' ') return false; a++; }
    while (*b) { if (*b != ' ') return false; b++; }
    return true;
}
bool strEqVarchar(const char* a, const char* b) {
    while (*a && *b) { if (*a != *b) return false; a++; b++; }
    return *a == *b;
}
bool equalsVal(Val a, Val b, const Type& t) {
    switch (t.tag) {
        case RSQ_DECIMAL: case RSQ_BIGINT: return a.i == b.i;
        case RSQ_INT: case RSQ_DATE: return (uint32_t)a.i == (uint32_t)b.i;
        case RSQ_BOOL: return (uint8_t)a.i == (uint8_t)b.i;
        case RSQ_CHAR: return t.len > 1 ? strEqChar(a.s, b.s) : (uint8_t)a.i == (uint8_t)b.i;
        case RSQ_VARCHAR: return strEqVarchar(a.s, b.s);
        default: failType("EQUALS code generation not implemented for datatype");
    }
}
int64_t sdiv(int64_t a, int64_t b) {
    if (b == 0) failRuntime("Division by zero");
    if (a == INT64_MIN && b == -1) failRuntime("Division overflow");
    return a / b;
}
int64_t pow10i(int n) { int64_t v = 1; while (n-- > 0) v *= 10; return v; }

Val evalHost(const HostExpr& h, const std::vector<Val>& sym) {
    if (h.kind == 0) return sym[(size_t)h.slot];
    if (h.kind == 1) return h.constant;
    Val r; r.i = 0;
    switch (h.tag) {
        case RSQ_E_COUNT: r.i = 1; return r;
        case RSQ_E_SUM: case RSQ_E_AVG: case RSQ_E_MIN: case RSQ_E_MAX: case RSQ_E_AS: case RSQ_E_ASC: case RSQ_E_DESC:
            return evalHost(h.kids[0], sym);
        case RSQ_E_TYPECAST: {
            Val c = evalHost(h.kids[0], sym);
            const Type &from = h.opType, &to = h.type;
            if (to.tag == RSQ_DECIMAL) {
                if (from.tag == RSQ_DECIMAL) {
                    if (to.scale == from.scale) return c;
                    if (to.scale > from.scale) r.i = (int64_t)((uint64_t)c.i * (uint64_t)pow10i(to.scale - from.scale));
                    else r.i = sdiv(c.i, pow10i(from.scale - to.scale));
                } else if (from.tag == RSQ_BIGINT) r.i = (int64_t)((uint64_t)c.i * (uint64_t)pow10i(to.scale));
                else failType("emitTypecastToDECIMAL(..) code generation not implemented for datatype");
            } else if (to.tag == RSQ_BIGINT) {
                if (from.tag == RSQ_INT) {
                    r.i = h.int16Cast ? (int64_t)(int16_t)c.i : (int64_t)(int32_t)c.i;      // see codegen.cpp emitUnary
                }
                else if (from.tag == RSQ_DECIMAL) r.i = sdiv(c.i, pow10i(from.scale));
                else r = c;
            } else failType("emitTypecast(..) code generation not implemented for datatype");
            return r;
        }
        case RSQ_E_CASE:
            for (const HostExpr& k : h.kids) {
                if (k.kind == 2 && k.tag == RSQ_E_WHENTHEN) { if ((uint8_t)evalHost(k.kids[0], sym).i) return evalHost(k.kids[1], sym); }
                else return evalHost(k, sym);
            }
            return r;
        default: break;
    }
    Val a = evalHost(h.kids[0], sym), b = evalHost(h.kids[1], sym);
    switch (h.tag) {
        case RSQ_E_ADD: r.i = (int64_t)((uint64_t)a.i + (uint64_t)b.i); break;
        case RSQ_E_SUB: r.i = (int64_t)((uint64_t)a.i - (uint64_t)b.i); break;
        case RSQ_E_MUL: r.i = (int64_t)((uint64_t)a.i * (uint64_t)b.i); break;
        case RSQ_E_DIV: r.i = sdiv(a.i, b.i); break;
        case RSQ_E_AND: r.i = (uint8_t)a.i & (uint8_t)b.i; break;
        case RSQ_E_OR: r.i = (uint8_t)a.i | (uint8_t)b.i; break;
        case RSQ_E_LT: r.i = cmpVal(a, b, h.opType) < 0; break;
        case RSQ_E_LE: r.i = cmpVal(a, b, h.opType) <= 0; break;
        case RSQ_E_GT: r.i = cmpVal(a, b, h.opType) > 0; break;
        case RSQ_E_GE: r.i = cmpVal(a, b, h.opType) >= 0; break;
        case RSQ_E_EQ: r.i = equalsVal(a, b, h.opType); break;
        case RSQ_E_NEQ: r.i = 1 - (int)equalsVal(a, b, h.opType); break;
        case RSQ_E_LIKE: r.i = refLike(a.s, b.s) ? 1 : 0; break;
        default: failUnsupported(std::string("host evaluation of ") + exprTagNames[h.tag]);
    }
    return r;
}

// ---- the groups the device produced ---------------------------------------------------------------
struct Groups {
    size_t n = 0, nKeys = 0, nAcc = 0;
    std::vector<int64_t> firstRow;                 // [n]
    std::vector<Val> keyData;                      // [n][nKeys], flat
    std::vector<int64_t> accData;                  // [n][nAcc], flat (index = accums index)
    std::vector<char> strings;                     // NUL-terminated bytes of string key values (Val::s points in here)
    const Val* keys(size_t i) const { return keyData.data() + i * nKeys; }
    const int64_t* acc(size_t i) const { return accData.data() + i * nAcc; }
};

}  // namespace

// What a query's tail keeps between executions: the group arrays and the scratch of the emission order.  A million groups are
// tens of MB; fresh vectors every execution cost more in page faults than the work done in them.
struct TailState {
    Groups groups;
    SortScratch sort;
    ReplayScratch replay;
    std::vector<uint64_t> keys, hashes;
    std::vector<uint32_t> byFirst, slotOrder, order;
};
void destroyTailState(TailState* t) { delete t; }
static TailState& tailState(Query& q) { if (!q.tailState) q.tailState = new TailState(); return *q.tailState; }
std::vector<uint32_t>& tailOrderBuffer(Query& q) { return tailState(q).slotOrder; }
ReplayScratch& tailReplayScratch(Query& q) { return tailState(q).replay; }

namespace {

void groupsFromDense(Query& q, Groups& G) {
    const int64_t D = q.denseGroups;
    const size_t W = q.accums.size();
    const uint64_t* table = q.hAggView ? q.hAggView : q.hAgg.data();
    auto word = [&](size_t w, int64_t g) { return (int64_t)table[(size_t)(q.accumSlot[w] * D + g)]; };
    G.nKeys = q.denseKeys.size(); G.nAcc = W;
    // two passes over the dense table, both split over the host threads: count the groups present per part, then
    // fill each part's slice (group order = dense id order, as before)
    const int parts = tailThreads((size_t)D);
    std::vector<size_t> cnt((size_t)parts + 1, 0);
    parallelFor((size_t)D, parts, [&](size_t b, size_t e, int p) {
        size_t c = 0;
        for (size_t g = b; g < e; g++) if (word(0, (int64_t)g) != INT64_MAX) c++;
        cnt[(size_t)p + 1] = c;
    });
    for (int p = 0; p < parts; p++) cnt[(size_t)p + 1] += cnt[(size_t)p];
    const size_t present = cnt[(size_t)parts];
    G.n = present;
    G.firstRow.resize(present); G.keyData.resize(present * G.nKeys); G.accData.resize(present * W);
    parallelFor((size_t)D, parts, [&](size_t b, size_t e, int p) {
        size_t o = cnt[(size_t)p];
        for (size_t gi = b; gi < e; gi++) {
            const int64_t g = (int64_t)gi;
            if (word(0, g) == INT64_MAX) continue;
            G.firstRow[o] = word(0, g);
            size_t k = 0;
            for (auto& dk : q.denseKeys) {
                int64_t rank = (g / dk.stride) % dk.card;
                Val v; v.i = dk.byteSet ? (int64_t)dk.values[(size_t)rank] : dk.min + rank;
                G.keyData[o * G.nKeys + k++] = v;
            }
            for (size_t w = 0; w < W; w++) G.accData[o * W + w] = word(w, g);
            o++;
        }
    });
}

// candidate rows of a dense aggregate table (engine.cpp: ORDER BY ... LIMIT over a large dense table):
// [first row | group id | accumulator blocks]
Groups groupsFromDenseRows(Query& q) {
    const size_t W = q.accums.size();
    const size_t stride = (size_t)q.groupRowWords;
    Groups G;
    G.n = (size_t)q.nGroupRows; G.nKeys = q.denseKeys.size(); G.nAcc = W;
    G.firstRow.resize(G.n); G.keyData.resize(G.n * G.nKeys); G.accData.resize(G.n * W);
    for (size_t i = 0; i < G.n; i++) {
        const int64_t* r = &(q.hRowsView ? q.hRowsView : q.hGroupRows)[i * stride];
        const int64_t g = r[1];
        G.firstRow[i] = r[0];
        size_t k = 0;
        for (auto& dk : q.denseKeys) {
            const int64_t rank = (g / dk.stride) % dk.card;
            Val v; v.i = dk.byteSet ? (int64_t)dk.values[(size_t)rank] : dk.min + rank;
            G.keyData[i * G.nKeys + k++] = v;
        }
        for (size_t w = 0; w < W; w++) G.accData[i * W + w] = r[2 + (size_t)q.accumSlot[w]];
    }
    return G;
}

Groups groupsFromJoinEntries(Query& q) {
    // rows compacted on the device: [firstrow | table words (keys, payload) | accumulator blocks]
    HashTable& ht = *q.hashTables[(size_t)q.aggTable];
    const size_t W = q.accums.size();
    const size_t nTabWords = ht.keys.size() + ht.payload.size();
    const size_t stride = (size_t)q.groupRowWords;
    Groups G;
    G.n = (size_t)q.nGroupRows; G.nKeys = q.groupSource.size(); G.nAcc = W;
    G.firstRow.resize(G.n); G.keyData.resize(G.n * G.nKeys); G.accData.resize(G.n * W);
    // string group values arrive as their bytes in consecutive key words (rsq_device.h: str_word)
    size_t strBytes = 0;
    for (Expr* g : q.agg->exprs2) if (g->type.isString()) strBytes += (size_t)g->type.len + 1;
    G.strings.assign(G.n * strBytes, 0);
    parallelFor(G.n, tailThreads(G.n), [&](size_t lo, size_t hi, int) {
        for (size_t i = lo; i < hi; i++) {
            const int64_t* r = &(q.hRowsView ? q.hRowsView : q.hGroupRows)[i * stride];
            size_t sp = i * strBytes;
            G.firstRow[i] = r[0];
            for (size_t k = 0; k < G.nKeys; k++) {
                const Type& t = q.agg->exprs2[k]->type;
                if (t.isString()) {
                    memcpy(&G.strings[sp], &r[1 + (size_t)q.groupSource[k]], (size_t)t.len);      // little-endian words = the bytes in order
                    G.keyData[i * G.nKeys + k].s = &G.strings[sp];
                    sp += (size_t)t.len + 1;
                } else G.keyData[i * G.nKeys + k].i = r[1 + (size_t)q.groupSource[k]];
            }
            for (size_t w = 0; w < W; w++) G.accData[i * W + w] = r[1 + nTabWords + (size_t)q.accumSlot[w]];
            G.accData[i * W] = r[0];
        }
    });
    return G;
}

// The device groups CHAR(n) keys by their exact bytes; the reference's group equality for CHAR ignores trailing spaces
// (Values::checkEqualityBool -> compareChar, qlib/scalar.h:27-46) and shows the spelling of the group's FIRST row.
// Merge the groups that are equal in that sense: accumulators by their merge kind, keys from the member with the
// smallest first row.  The same merge joins the shards of a multi-GPU plan (runTailMerged): a group that occurs in several
// shards comes in several rows (reference: ONE hash table all workers reach, aggregation.h:240-295).
// Keys are normalised into fixed-width blobs (8 bytes per value, strings NUL padded to their width, CHAR without its trailing
// spaces) and matched through an open-addressing table of group indices.
void mergeEqualGroups(Query& q, Groups& G) {
    if (G.n < 2) return;
    const size_t K = G.nKeys, W = G.nAcc;
    std::vector<size_t> off(K + 1, 0);
    for (size_t k = 0; k < K; k++) {
        const Type& t = q.agg->exprs2[k]->type;
        off[k + 1] = off[k] + (t.isString() ? (((size_t)t.len + 8) & ~(size_t)7) : 8);
    }
    const size_t kb = std::max<size_t>(off[K], 8);
    std::vector<uint8_t> blob(G.n * kb, 0);
    std::vector<uint64_t> hash(G.n);
    parallelFor(G.n, tailThreads(G.n), [&](size_t lo, size_t hi, int) {
        for (size_t i = lo; i < hi; i++) {
            uint8_t* b = &blob[i * kb];
            for (size_t k = 0; k < K; k++) {
                const Type& t = q.agg->exprs2[k]->type;
                const Val& v = G.keys(i)[k];
                if (t.isString()) {
                    size_t n = strnlen(v.s, (size_t)t.len);
                    if (t.tag == RSQ_CHAR) while (n > 0 && v.s[n - 1] == ' ') n--;
                    memcpy(b + off[k], v.s, n);
                } else {
                    int64_t x = v.i;
                    if (t.tag == RSQ_INT || t.tag == RSQ_DATE) x = (int64_t)(uint32_t)x;
                    else if (t.tag == RSQ_BOOL || t.tag == RSQ_CHAR) x = (int64_t)(uint8_t)x;
                    memcpy(b + off[k], &x, 8);
                }
            }
            uint64_t h = 0x9E3779B97F4A7C15ull;
            for (size_t w = 0; w < kb; w += 8) { uint64_t x; memcpy(&x, b + w, 8); h = (h ^ x) * 0xBF58476D1CE4E5B9ull; h ^= h >> 29; }
            hash[i] = h;
        }
    });
    size_t cap = 16; while (cap < G.n * 2) cap <<= 1;
    std::vector<uint32_t> slots(cap, 0xffffffffu);
    if (G.n >= 0xffffffffull) failUnsupported("more than 4 G group rows in a merge");
    std::vector<uint32_t> target(G.n);
    bool merged = false;
    for (size_t i = 0; i < G.n; i++) {
        size_t s0 = (size_t)hash[i] & (cap - 1);
        for (;;) {
            const uint32_t j = slots[s0];
            if (j == 0xffffffffu) { slots[s0] = (uint32_t)i; target[i] = (uint32_t)i; break; }
            if (hash[j] == hash[i] && memcmp(&blob[(size_t)j * kb], &blob[i * kb], kb) == 0) { target[i] = j; merged = true; break; }
            s0 = (s0 + 1) & (cap - 1);
        }
    }
    if (!merged) return;
    std::vector<uint32_t> bestMember(G.n);
    for (size_t i = 0; i < G.n; i++) bestMember[i] = (uint32_t)i;
    for (size_t i = 0; i < G.n; i++) {
        const size_t t = target[i];
        if (t == i) continue;
        int64_t* dst = &G.accData[t * W];
        const int64_t* src = &G.accData[i * W];
        for (size_t w = 0; w < W; w++) {
            const int m = q.accums[w].merge;
            if (m == 0) dst[w] = (int64_t)((uint64_t)dst[w] + (uint64_t)src[w]);
            else if (m == 2) dst[w] = std::min(dst[w], src[w]);
            else dst[w] = std::max(dst[w], src[w]);
        }
        if (G.firstRow[i] < G.firstRow[bestMember[t]]) bestMember[t] = (uint32_t)i;
    }
    Groups R;
    R.nKeys = K; R.nAcc = W;
    R.strings.swap(G.strings);                     // Val::s pointers stay valid
    for (size_t i = 0; i < G.n; i++) {
        if (target[i] != i) continue;
        const size_t b = bestMember[i];
        R.firstRow.push_back(G.firstRow[b]);
        for (size_t k = 0; k < K; k++) R.keyData.push_back(G.keys(b)[k]);
        for (size_t w = 0; w < W; w++) R.accData.push_back(G.accData[i * W + w]);
        R.accData[R.firstRow.size() * W - W] = G.firstRow[b];
    }
    R.n = R.firstRow.size();
    G = std::move(R);
}

void mergeSpaceEquivalentGroups(Query& q, Groups& G) {
    bool any = false;
    for (Expr* g : q.agg->exprs2) if (g->type.tag == RSQ_CHAR && g->type.len > 1) any = true;
    if (!any || G.n < 2 || !q.charGroupsNeedMerge) return;
    mergeEqualGroups(q, G);
}

}  // namespace

// plans without aggregation: the device delivered the materialised columns in scan order; pack ReSQL tuples and
// apply ORDER BY / LIMIT of an OrderByOp above (orderby.h:87-136)
static void runMaterializeTail(Query& q) {
    const Schema& cur = q.matSchema;
    q.resultSchema = cur;
    const size_t ts = (size_t)schemaTupleSize(cur);
    const size_t n = (size_t)q.matRows;
    q.resultTuples.assign(n * ts, 0);
    q.resultRows = (int64_t)n;
    std::vector<int> colOff(cur.size());
    { int off = 0; for (size_t c = 0; c < cur.size(); c++) { colOff[c] = off; off += sizeInTuple(cur[c].type, true); } }
    // rows are independent: split them over the host threads (20 M tuples took 150 ms on one)
    parallelFor(n, tailThreads(n), [&](size_t lo, size_t hi, int) {
        std::vector<char> strbuf;
        for (size_t c = 0; c < cur.size(); c++) {
            const Type& t = cur[c].type;
            const size_t w = (size_t)columnWidth(t);
            const uint8_t* src = q.hMatCols[c].data();
            if (t.isString()) strbuf.assign(w + 1, 0);
            for (size_t r = lo; r < hi; r++) {
                uint8_t* dst = &q.resultTuples[r * ts + (size_t)colOff[c]];
                if (t.isString()) {
                    memcpy(strbuf.data(), src + r * w, w); strbuf[w] = 0;
                    Val v; v.s = strbuf.data();
                    storeValue(dst, v, t);
                } else memcpy(dst, src + r * w, w);      // same little-endian value widths as the packed tuple
            }
        }
    });
    OpNode* orderBy = nullptr;
    for (OpNode* o = q.matOp->parent; o; o = o->parent) {
        if (o->tag == RSQ_OP_ORDERBY) orderBy = o;
        else failUnsupported("operator above a materialisation other than order by");
    }
    if (orderBy) {
        std::vector<OrderRequest> reqs;
        for (Expr* e : orderBy->exprs) {
            const std::string& nm = e->child->symbol;
            bool found = false;
            for (auto& a : cur) if (a.name == nm) { reqs.push_back({schemaOffset(cur, nm), a.type, e->tag != RSQ_E_DESC}); found = true; break; }
            if (!found) failType("Order By attribute not found.");
        }
        refQuicksort(q.resultTuples.data(), q.resultRows, ts, reqs);
        if (orderBy->hasLimit && q.resultRows > orderBy->limit) {
            q.resultRows = orderBy->limit;
            q.resultTuples.resize((size_t)q.resultRows * ts);
        }
    }
}

// Device-side pre-selection for `ORDER BY ... LIMIT k` (orderby.h:87-93 sorts everything, then applyLimit): possible when
// the first sort key is, through AS / pass-through projections only, a non-string group value or a non-AVG aggregate,
// i.e. one word of the group rows the device compacts.  Symbols are resolved the way the tail's HostCompiler does.
void planDeviceTopK(Query& q) {
    q.topkWord = -1;
    const char* env = getenv("RSQ_DEVICE_TOPK");
    if (env && atoi(env) == 0) return;
    OpNode* agg = q.agg;
    const bool dense = q.aggMode == AggMode::DENSE_GLOBAL;         // candidate rows [first row | group id | accumulator blocks]
    if (!agg || (q.aggMode != AggMode::AT_JOIN_ENTRY && q.aggMode != AggMode::HASH && !dense)) return;
    OpNode* mat = nullptr; OpNode* orderBy = nullptr;
    std::vector<OpNode*> projections;
    for (OpNode* o = agg->parent; o; o = o->parent) {
        if (o->tag == RSQ_OP_PROJECTION) { if (mat) return; projections.push_back(o); }
        else if (o->tag == RSQ_OP_MATERIALIZE) { if (mat) return; mat = o; }
        else if (o->tag == RSQ_OP_ORDERBY) orderBy = o;
        else return;
    }
    if (!mat || mat->hasLimit || !orderBy || !orderBy->hasLimit || orderBy->exprs.empty()) return;
    if (orderBy->limit < 0 || orderBy->limit > (1 << 20)) return;
    // CHAR(n) group values of a hash aggregation may need the host's merge of space-equivalent groups; the kernel
    // reports whether any does (charGroupsNeedMerge), the candidate path is taken only when none does
    q.topkNeedsNoMerge = false;
    if (q.aggMode == AggMode::HASH)
        for (Expr* g : agg->exprs2) if (g->type.tag == RSQ_CHAR && g->type.len > 1) q.topkNeedsNoMerge = true;
    const int nTab = dense ? 1 : (int)(q.hashTables[(size_t)q.aggTable]->keys.size() + q.hashTables[(size_t)q.aggTable]->payload.size());
    typedef std::pair<int, Type> Src;
    std::map<std::string, Src> src;
    if (!dense)      // (the group values of a dense table are functions of the group id, not words of the row)
        for (size_t k = 0; k < agg->exprs2.size(); k++) src[expressionName(agg->exprs2[k])] = Src(1 + q.groupSource[k], agg->exprs2[k]->type);
    {
        size_t si = 0;
        for (Expr* a : agg->exprs) {
            if (a->tag == RSQ_E_AVG) { q.pool.addId(a); src.erase(expressionName(a)); si += 2; }
            else {
                Expr* s = agg->splitAgg[si];
                src[expressionName(s)] = Src(1 + nTab + q.accumSlot[(size_t)q.splitToAccum[si]], s->type);
                si += 1;
            }
        }
    }
    std::function<const Src*(Expr*)> resolve = [&](Expr* e) -> const Src* {
        auto it = src.find(expressionName(e));
        if (it != src.end()) return &it->second;
        if (e->structure == UNARY && e->tag == RSQ_E_AS) return resolve(e->child);
        return nullptr;
    };
    for (OpNode* p : projections) {
        for (Expr* e : p->exprs) q.pool.addId(e);
        std::vector<std::pair<std::string, const Src*>> defs;
        std::vector<Src> keep;
        keep.reserve(p->exprs.size());
        for (Expr* e : p->exprs) {
            const Src* r = resolve(e);
            if (r) { keep.push_back(*r); defs.emplace_back(expressionName(e), &keep.back()); }
            else defs.emplace_back(expressionName(e), nullptr);
        }
        for (auto& d : defs) { if (d.second) src[d.first] = *d.second; else src.erase(d.first); }
    }
    Expr* first = orderBy->exprs[0];
    auto it = src.find(first->child->symbol);
    if (it == src.end()) return;
    const Type& t = it->second.second;
    if (t.tag == RSQ_BIGINT || t.tag == RSQ_DECIMAL) q.topkIs32 = false;
    else if (t.tag == RSQ_INT || t.tag == RSQ_DATE) q.topkIs32 = true;
    else return;
    q.topkDesc = first->tag == RSQ_E_DESC;
    q.topkWant = (uint32_t)orderBy->limit + 1;
    q.topkWord = it->second.first;
}

static void runTailOn(Query& q, Groups& G);

void runTail(Query& q) {
    q.tailNeedsAllGroups = false;
    q.resultInPinned = false;
    if (!q.agg) { runMaterializeTail(q); return; }
    const double t0 = nowMs();
    Groups& G = tailState(q).groups;
    if (q.aggMode == AggMode::AT_JOIN_ENTRY || q.aggMode == AggMode::HASH) G = groupsFromJoinEntries(q);
    else if (q.candidateRun) G = groupsFromDenseRows(q);
    else groupsFromDense(q, G);
    if (q.aggMode == AggMode::HASH) mergeSpaceEquivalentGroups(q, G);
    if (getenv("RSQ_TRACE")) fprintf(stderr, "[rsq trace]     tail: %.3f ms  groups from the device tables\n", nowMs() - t0);
    runTailOn(q, G);
}

// The shards of a multi-GPU plan that does not end in a dense partial table (multi.cpp): every part has run its pipelines with
// the tail held back (engine.cpp setHoldTail) and holds its group rows — or, without an aggregation, its materialised columns in
// scan order — in host memory.  Groups that occur in several parts are merged by key (sum / min / max by accumulator kind, the
// first row = the smallest one: rows are numbered over the whole table, Table::row0), then `root`'s tail runs over all of them:
// AVG, projections, the replay of the reference's hash table for the emission order, ORDER BY, LIMIT.  The result is the one the
// unsharded plan gives on one GPU, whatever the sharding (reference: one hash table all workers reach, aggregation.h:240-343).
void runTailMerged(Query& root, const std::vector<Query*>& parts) {
    root.tailNeedsAllGroups = false;
    root.resultInPinned = false;
    root.candidateRun = false;
    if (!root.agg) {
        // materialize.h:78-220 appends in scan order: the parts' columns back to back, parts in shard order
        const size_t nCols = root.matSchema.size();
        std::vector<std::vector<uint8_t>> cols(nCols);
        int64_t rows = 0;
        for (Query* p : parts) {
            if (p->matSchema.size() != nCols) failInvalid("shard results have different schemas");
            for (size_t c = 0; c < nCols; c++) {
                const size_t bytes = (size_t)p->matRows * (size_t)columnWidth(root.matSchema[c].type);
                if (p->hMatCols.size() != nCols || p->hMatCols[c].size() < bytes) failRuntime("internal error: a shard holds no materialised rows");
                cols[c].insert(cols[c].end(), p->hMatCols[c].begin(), p->hMatCols[c].begin() + (long)bytes);
            }
            rows += p->matRows;
        }
        if (root.matOp->hasLimit) rows = std::min<int64_t>(rows, std::max<int64_t>(root.matOp->limit, 1));      // materialize.h:197-206
        root.hMatCols.swap(cols);
        root.matRows = rows;
        runMaterializeTail(root);
        return;
    }
    if (root.aggMode != AggMode::AT_JOIN_ENTRY && root.aggMode != AggMode::HASH) failUnsupported("group-level merge of a dense aggregation (its partial tables merge on the device)");
    Groups G;
    G.nKeys = root.agg->exprs2.size(); G.nAcc = root.accums.size();
    std::vector<Groups> each;
    size_t total = 0, strBytes = 0;
    for (Query* p : parts) {
        if (p->aggMode != root.aggMode || p->accums.size() != root.accums.size()) failInvalid("shards disagree on the aggregation strategy");
        each.push_back(groupsFromJoinEntries(*p));
        total += each.back().n; strBytes += each.back().strings.size();
    }
    G.firstRow.reserve(total); G.keyData.reserve(total * G.nKeys); G.accData.reserve(total * G.nAcc);
    G.strings.reserve(strBytes + 1);               // (reserved once: Val::s pointers into it stay valid while the parts are appended)
    for (Groups& e : each) {
        const char* oldBase = e.strings.data();
        const size_t at = G.strings.size();
        G.strings.insert(G.strings.end(), e.strings.begin(), e.strings.end());
        const char* newBase = G.strings.data() + at;
        G.firstRow.insert(G.firstRow.end(), e.firstRow.begin(), e.firstRow.end());
        G.accData.insert(G.accData.end(), e.accData.begin(), e.accData.end());
        for (size_t i = 0; i < e.n; i++)
            for (size_t k = 0; k < G.nKeys; k++) {
                Val v = e.keys(i)[k];
                if (root.agg->exprs2[k]->type.isString()) v.s = newBase + (v.s - oldBase);
                G.keyData.push_back(v);
            }
    }
    G.n = total;
    const double t0 = nowMs();
    mergeEqualGroups(root, G);
    if (getenv("RSQ_TRACE")) fprintf(stderr, "[rsq trace]     tail: %.3f ms  %zu group rows of %zu shards merged into %zu groups\n", nowMs() - t0, total, parts.size(), G.n);
    runTailOn(root, G);
}

// ---- what sits above the aggregation, analysed once per tail run ------------------------------------------------------
struct AggOut { bool avg; int sumAcc, cntAcc, slot; };
struct Proj { std::vector<HostExpr> exprs; std::vector<int> slots; };
// where an output column comes from when the projections above the aggregation only pass values through (AS, plain symbols)
struct Src { int kind = 3; int a = 0, b = 0; };     // 0 group value, 1 accumulator, 2 AVG(sum accumulator a, count accumulator b), 3 computed
struct TailShape {
    OpNode* mat = nullptr; OpNode* orderBy = nullptr;
    HostCompiler hc;
    std::vector<int> keySlots;
    std::vector<AggOut> outs;
    std::vector<Proj> projs;
    Schema cur;
    size_t ts = 0;
    std::vector<int> offs, matSlots;
    std::vector<OrderRequest> reqs;
    std::vector<Src> colSrc;
    bool directRows = true;          // every output column is a group value, an accumulator or an AVG, none a string
};

static TailShape buildTailShape(Query& q) {
    TailShape sh;
    OpNode* agg = q.agg;
    OpNode*& mat = sh.mat; OpNode*& orderBy = sh.orderBy;
    std::vector<OpNode*> projections;
    for (OpNode* o = agg->parent; o; o = o->parent) {
        if (o->tag == RSQ_OP_PROJECTION) { if (mat) failUnsupported("projection above materialize"); projections.push_back(o); }
        else if (o->tag == RSQ_OP_MATERIALIZE) { if (mat) failUnsupported("two materializations"); mat = o; }
        else if (o->tag == RSQ_OP_ORDERBY) orderBy = o;
        else failUnsupported("operator above an aggregation other than projection / materialize / order by");
    }
    if (!mat) failInvalid("plan has no materialization");

    HostCompiler& hc = sh.hc;
    hc.int16Cast = jitInt16Cast(q.ctx);
    std::vector<int>& keySlots = sh.keySlots;
    Schema aggSchema;
    for (Expr* g : agg->exprs2) { keySlots.push_back(hc.define(expressionName(g), g->type)); aggSchema.push_back({expressionName(g), g->type}); }
    std::vector<AggOut>& outs = sh.outs;
    {
        size_t si = 0;
        for (Expr* a : agg->exprs) {
            if (a->tag == RSQ_E_AVG) {      // mergeAverages (aggregation.h:207-238)
                q.pool.addId(a);
                Type st = agg->splitAgg[si]->type;
                if (st.tag != RSQ_BIGINT && st.tag != RSQ_DECIMAL) failType("getAvgFromSumAndCount(..) not supported for datatype");
                outs.push_back({true, q.splitToAccum[si], q.splitToAccum[si + 1], hc.define(expressionName(a), a->type)});
                aggSchema.push_back({expressionName(a), a->type});
                si += 2;
            } else {
                Expr* s = agg->splitAgg[si];
                outs.push_back({false, q.splitToAccum[si], -1, hc.define(expressionName(s), s->type)});
                aggSchema.push_back({expressionName(s), s->type});
                si += 1;
            }
        }
    }
    agg->schema = aggSchema;
    Schema& cur = sh.cur;
    cur = aggSchema;
    std::vector<Proj>& projs = sh.projs;
    for (OpNode* p : projections) {
        Proj pr; Schema s;
        for (Expr* e : p->exprs) { q.pool.addId(e); pr.exprs.push_back(hc.compile(e)); }
        for (Expr* e : p->exprs) { pr.slots.push_back(hc.define(expressionName(e), e->type)); s.push_back({expressionName(e), e->type}); }
        p->schema = s; cur = s;
        projs.push_back(std::move(pr));
    }
    mat->schema = cur;
    sh.ts = (size_t)schemaTupleSize(cur);
    for (auto& a : cur) {
        sh.offs.push_back(schemaOffset(cur, a.name));
        int s = hc.slotOf(a.name);
        if (s < 0) failType("materialize: symbol " + a.name + " not found");
        sh.matSlots.push_back(s);
    }
    if (orderBy)
        for (Expr* e : orderBy->exprs) {
            const std::string& n = e->child->symbol;
            bool found = false;
            for (auto& a : cur) if (a.name == n) { sh.reqs.push_back({schemaOffset(cur, n), a.type, e->tag != RSQ_E_DESC}); found = true; break; }
            if (!found) failType("Order By attribute not found.");
        }
    // column sources
    sh.colSrc.resize(cur.size());
    std::vector<Src> src(hc.names.size());
    for (size_t k = 0; k < keySlots.size(); k++) src[(size_t)keySlots[k]] = Src{0, (int)k, 0};
    for (auto& o : outs) src[(size_t)o.slot] = o.avg ? Src{2, o.sumAcc, o.cntAcc} : Src{1, o.sumAcc, 0};
    std::function<Src(const HostExpr&)> resolve = [&](const HostExpr& h) -> Src {
        if (h.kind == 0) return src[(size_t)h.slot];
        if (h.kind == 2 && (h.tag == RSQ_E_AS || h.tag == RSQ_E_SUM || h.tag == RSQ_E_AVG || h.tag == RSQ_E_MIN || h.tag == RSQ_E_MAX) && h.kids.size() == 1)
            return resolve(h.kids[0]);
        return Src{};
    };
    for (auto& pr : projs) {
        std::vector<Src> now(pr.exprs.size());
        for (size_t i = 0; i < pr.exprs.size(); i++) now[i] = resolve(pr.exprs[i]);
        for (size_t i = 0; i < pr.exprs.size(); i++) src[(size_t)pr.slots[i]] = now[i];
    }
    for (size_t c = 0; c < cur.size(); c++) {
        sh.colSrc[c] = src[(size_t)sh.matSlots[c]];
        if (sh.colSrc[c].kind == 3 || cur[c].type.isString()) sh.directRows = false;
    }
    return sh;
}

// Can the rows of this dense aggregation be produced on the device (devtail.hip)?  Yes when every output column is a group
// value, an accumulator or an AVG (no computed projection, no strings) and nothing sorts the rows afterwards.  Fills the column
// and key descriptions the kernels take.
bool planDenseDeviceTail(Query& q, DenseTailKeys& keys, DenseTailCols& cols, int& tupleSize, int64_t& limitRows) {
    if (!q.agg || q.denseKeys.empty() || q.denseKeys.size() > 4) return false;
    TailShape sh = buildTailShape(q);
    if (!sh.directRows || sh.orderBy || sh.cur.size() > 24) return false;
    keys.n = (int32_t)q.denseKeys.size();
    for (size_t k = 0; k < q.denseKeys.size(); k++) {
        const DenseKey& dk = q.denseKeys[k];
        DenseTailKey& o = keys.k[k];
        o.min = dk.min; o.card = dk.card; o.stride = dk.stride; o.byteSet = dk.byteSet ? 1 : 0; o.typeTag = dk.type.tag;
        if (dk.type.isString()) return false;
        if (dk.byteSet) { if (dk.values.size() > sizeof o.values) return false; memset(o.values, 0, sizeof o.values); memcpy(o.values, dk.values.data(), dk.values.size()); }
    }
    cols.n = (int32_t)sh.cur.size();
    for (size_t c = 0; c < sh.cur.size(); c++) {
        const Src& s = sh.colSrc[c];
        DenseTailCol& o = cols.c[c];
        o.kind = s.kind; o.offset = sh.offs[c]; o.width = sizeInTuple(sh.cur[c].type, true);
        if (s.kind == 0) { o.a = s.a; o.b = 0; }
        else { o.a = q.accumSlot[(size_t)s.a]; o.b = s.kind == 2 ? q.accumSlot[(size_t)s.b] : 0; }      // accumulator index -> block of the [block][group] table
        if (o.width != 8 && o.width != 4 && o.width != 2 && o.width != 1) return false;      // (CHAR(1) takes two bytes of a tuple: the character and a NUL)
    }
    tupleSize = (int)sh.ts;
    limitRows = sh.mat->hasLimit ? std::max<int64_t>(sh.mat->limit, 1) : -1;      // materialize.h:197-206
    q.resultSchema = sh.cur;
    return true;
}

// The same question for the group rows of a hash / join-entry aggregation ([first row | table words | accumulator blocks]): yes when every
// output column is a group value (strings too: their bytes sit in consecutive words of the row), an accumulator or an AVG.  An ORDER BY
// above does not prevent it: the device then delivers the tuples in the reference's EMISSION order and the host only runs the
// reference's quicksort over them (runRowsTailSort) - group decoding, hashing, the replay and the row building stay on the device.
bool planRowsDeviceTail(Query& q, RowTailKeys& keys, RowTailCols& cols, int& tupleSize, int64_t& limitRows, bool& sorts) {
    if (!q.agg || (q.aggMode != AggMode::HASH && q.aggMode != AggMode::AT_JOIN_ENTRY) || q.agg->exprs2.empty() || q.agg->exprs2.size() > 16) return false;
    TailShape sh = buildTailShape(q);
    if (sh.cur.size() > 24) return false;
    const HashTable& ht = *q.hashTables[(size_t)q.aggTable];
    const int nTab = (int)(ht.keys.size() + ht.payload.size());
    keys.n = (int32_t)q.agg->exprs2.size();
    for (size_t k = 0; k < q.agg->exprs2.size(); k++) {
        const Type& t = q.agg->exprs2[k]->type;
        keys.k[k] = RowTailKey{1 + q.groupSource[k], (int32_t)t.tag, t.isString() ? t.len : (t.tag == RSQ_CHAR ? 1 : 0), 0};
    }
    cols.n = (int32_t)sh.cur.size();
    for (size_t c = 0; c < sh.cur.size(); c++) {
        const Src& s = sh.colSrc[c];
        RowTailCol& o = cols.c[c];
        const Type& t = sh.cur[c].type;
        o.kind = s.kind; o.offset = sh.offs[c]; o.width = sizeInTuple(t, true); o.len = 0; o.b = 0;
        if (s.kind == 3) return false;                                    // a computed projection: the host's expression interpreter
        if (s.kind == 0) {
            if (s.a < 0 || (size_t)s.a >= q.agg->exprs2.size()) return false;
            o.a = 1 + q.groupSource[(size_t)s.a];
            if (t.isString()) { o.len = t.len; if (o.width != t.len + 1) return false; }
        } else {
            if (t.isString()) return false;
            o.a = 1 + nTab + q.accumSlot[(size_t)s.a];
            if (s.kind == 2) o.b = 1 + nTab + q.accumSlot[(size_t)s.b];
        }
        if (!t.isString() && o.width != 8 && o.width != 4 && o.width != 2 && o.width != 1) return false;
    }
    tupleSize = (int)sh.ts;
    limitRows = sh.mat->hasLimit ? std::max<int64_t>(sh.mat->limit, 1) : -1;      // materialize.h:197-206
    sorts = sh.orderBy != nullptr;
    q.resultSchema = sh.cur;
    return true;
}

// ... and the ORDER BY over the tuples the device delivered in emission order: the reference's quicksort, then its LIMIT (orderby.h:87-93)
void runRowsTailSort(Query& q, uint8_t* tuples, int64_t& rows) {
    TailShape sh = buildTailShape(q);
    if (!sh.orderBy) return;
    refQuicksort(tuples, rows, sh.ts, sh.reqs);
    if (sh.orderBy->hasLimit && rows > sh.orderBy->limit) rows = std::max<int64_t>(sh.orderBy->limit, 0);
}

static void runTailOn(Query& q, Groups& G) {
    OpNode* agg = q.agg;
    const bool trace = getenv("RSQ_TRACE") != nullptr;
    double tPhase = nowMs();
    auto phase = [&](const char* what) {
        if (!trace) return;
        double t = nowMs();
        fprintf(stderr, "[rsq trace]     tail: %.3f ms  %s\n", t - tPhase, what);
        tPhase = t;
    };
    TailShape sh = buildTailShape(q);
    OpNode* mat = sh.mat; OpNode* orderBy = sh.orderBy;
    HostCompiler& hc = sh.hc;
    std::vector<int>& keySlots = sh.keySlots;
    std::vector<AggOut>& outs = sh.outs;
    std::vector<Proj>& projs = sh.projs;
    const Schema& cur = sh.cur;
    q.resultSchema = cur;
    const size_t ts = sh.ts;
    const std::vector<int>& offs = sh.offs; const std::vector<int>& matSlots = sh.matSlots;
    const std::vector<OrderRequest>& reqs = sh.reqs;
    const std::vector<Src>& colSrc = sh.colSrc;
    const bool directRows = sh.directRows;

    // ---- per group: dematerialize, AVG, projections, materialize ----
    auto materializeGroupWith = [&](size_t gi, uint8_t* dst, std::vector<Val>& sym) {
        const Val* gk = G.keys(gi);
        const int64_t* ga = G.acc(gi);
        for (size_t k = 0; k < keySlots.size(); k++) sym[(size_t)keySlots[k]] = gk[k];
        for (auto& o : outs) {
            Val v;
            if (o.avg) v.i = sdiv((int64_t)((uint64_t)ga[(size_t)o.sumAcc] * 100ull), ga[(size_t)o.cntAcc]);
            else v.i = ga[(size_t)o.sumAcc];
            sym[(size_t)o.slot] = v;
        }
        for (auto& pr : projs) {
            Val tmp[RSQ_MAX_OP_EXPRS];
            for (size_t i = 0; i < pr.exprs.size(); i++) tmp[i] = evalHost(pr.exprs[i], sym);
            for (size_t i = 0; i < pr.exprs.size(); i++) sym[(size_t)pr.slots[i]] = tmp[i];
        }
        memset(dst, 0, ts);
        for (size_t c = 0; c < cur.size(); c++) storeValue(dst + offs[c], sym[(size_t)matSlots[c]], cur[c].type);
    };
    // (sources of pass-through columns: TailShape::colSrc.  A row is then a handful of fixed-width copies and the expression
    // interpreter stays out of the per-row loop: a million rows 20 ms -> 2 ms on the GPU box's host.)
    auto materializeDirect = [&](size_t gi, uint8_t* dst) {
        const Val* gk = G.keys(gi);
        const int64_t* ga = G.acc(gi);
        for (size_t c = 0; c < cur.size(); c++) {
            const Src& sc = colSrc[c];
            int64_t v;
            if (sc.kind == 0) v = gk[sc.a].i;
            else if (sc.kind == 1) v = ga[sc.a];
            else v = sdiv((int64_t)((uint64_t)ga[sc.a] * 100ull), ga[sc.b]);
            switch (cur[c].type.tag) {       // (storeValue's widths; no strings on this path)
                case RSQ_BIGINT: case RSQ_DECIMAL: memcpy(dst + offs[c], &v, 8); break;
                case RSQ_INT: case RSQ_DATE: { const uint32_t x = (uint32_t)v; memcpy(dst + offs[c], &x, 4); break; }
                default: dst[offs[c]] = (uint8_t)v; break;
            }
        }
    };
    // many groups: every output row is independent, so the rows are split over the host threads
    // (groupOf == nullptr: row i is group i)
    auto materializeMany = [&](size_t count, const uint32_t* groupOf, uint8_t* base) {
        parallelFor(count, tailThreads(count), [&](size_t lo, size_t hi, int) {
            if (directRows) { for (size_t i = lo; i < hi; i++) materializeDirect(groupOf ? groupOf[i] : i, base + i * ts); return; }
            std::vector<Val> sym(hc.names.size());
            for (size_t i = lo; i < hi; i++) materializeGroupWith(groupOf ? groupOf[i] : i, base + i * ts, sym);
        });
    };

    // The reference's order of the materialized rows: groups enter its hash table in the order of their first
    // input row and leave it in slot order.  Only needed when that order is observable.  rsq_config.emission_order =
    // RSQ_EMIT_ANY skips it: the rows come in the order the device tables hold them (the reference's own tests compare
    // un-ordered results as multisets, test/test_common.h:152-190).
    TailState& T = tailState(q);
    auto emissionOrder = [&]() -> const uint32_t* {
        if (G.n >= 0xffffffffull) failUnsupported("more than 4 G groups");
        if (q.ctx.cfg.emission_order == RSQ_EMIT_ANY) return nullptr;
        T.keys.resize(G.n);
        parallelFor(G.n, tailThreads(G.n), [&](size_t lo, size_t hi, int) { for (size_t i = lo; i < hi; i++) T.keys[i] = (uint64_t)G.firstRow[i]; });
        // (stable: groups that share a first row — the matches of one probe row — keep their order)
        parallelSortIndex(T.keys.data(), G.n, T.byFirst, T.sort);
        phase("sort groups by first row");
        T.hashes.resize(G.n);
        parallelFor(G.n, tailThreads(G.n), [&](size_t lo, size_t hi, int) {
            for (size_t i = lo; i < hi; i++) {
                uint64_t h = 0;
                const Val* gk = G.keys(T.byFirst[i]);
                for (size_t k = 0; k < agg->exprs2.size(); k++) h = refHashValue(h, gk[k], agg->exprs2[k]->type);
                T.hashes[i] = h;
            }
        });
        phase("reference hashes");
        refEmissionOrderParallel(T.hashes.data(), G.n, opSize(agg), T.slotOrder, T.replay);
        T.order.resize(G.n);
        parallelFor(G.n, tailThreads(G.n), [&](size_t lo, size_t hi, int) { for (size_t i = lo; i < hi; i++) T.order[i] = T.byFirst[T.slotOrder[i]]; });
        phase("replay of the reference's hash table (slot order)");
        return T.order.data();
    };

    q.resultRows = 0;                      // (resultTuples keeps its pages: every path below sets its size)

    // ---- ORDER BY ... LIMIT k over many groups: the k first rows of the sorted order are determined by the
    // sort keys alone unless rows tie on ALL of them; select them without sorting (or replaying) everything ----
    const size_t allGroups = q.candidateRun ? (size_t)q.totalGroups : G.n;
    if (q.candidateRun && !(orderBy && orderBy->hasLimit && !mat->hasLimit && orderBy->limit >= 0 && (size_t)orderBy->limit * 4 < allGroups)) {
        q.tailNeedsAllGroups = true;
        return;
    }
    if (orderBy && orderBy->hasLimit && !mat->hasLimit && orderBy->limit >= 0 && (size_t)orderBy->limit * 4 < allGroups) {
        const size_t k = (size_t)orderBy->limit;
        std::vector<uint8_t> all(G.n * ts);
        materializeMany(G.n, nullptr, all.data());
        auto before = [&](const uint8_t* a, const uint8_t* b) {
            for (const auto& o : reqs) {
                int c = compareTyped(o.type, a + o.offset, b + o.offset);
                if (o.asc) { if (c < 0) return true; if (c > 0) return false; }
                else { if (c > 0) return true; if (c < 0) return false; }
            }
            return false;
        };
        std::vector<size_t> idx(G.n);
        for (size_t i = 0; i < G.n; i++) idx[i] = i;
        std::partial_sort(idx.begin(), idx.begin() + (long)std::min(k + 1, G.n), idx.end(),
                          [&](size_t a, size_t b) { return before(&all[a * ts], &all[b * ts]); });
        bool tie = false;
        for (size_t i = 0; i + 1 < std::min(k + 1, G.n) && !tie; i++)
            if (!before(&all[idx[i] * ts], &all[idx[i + 1] * ts])) tie = true;     // sorted, so "not before" means equal keys
        phase("top-k selection");
        if (!tie) {
            q.resultTuples.resize(std::min(k, G.n) * ts);
            for (size_t i = 0; i < std::min(k, G.n); i++) memcpy(&q.resultTuples[i * ts], &all[idx[i] * ts], ts);
            q.resultRows = (int64_t)std::min(k, G.n);
            return;
        }
        // ties among the leading rows: fall through to the faithful path, which needs every group
        if (q.candidateRun) { q.tailNeedsAllGroups = true; return; }
    }

    const uint32_t* order = emissionOrder();
    // materialize.h:197-206: with a LIMIT the pipeline is left once count >= limit, i.e. after max(limit, 1) tuples
    size_t emit = G.n;
    if (mat->hasLimit) emit = std::min(emit, (size_t)std::max<int64_t>(mat->limit, 1));
    q.resultTuples.resize(emit * ts);
    materializeMany(emit, order, q.resultTuples.data());
    q.resultRows = (int64_t)emit;
    phase("AVG / projections / materialize");
    if (orderBy) {
        refQuicksort(q.resultTuples.data(), q.resultRows, ts, reqs);
        phase("order by (the reference's quicksort)");
        if (orderBy->hasLimit && q.resultRows > orderBy->limit) {      // applyLimit after the sort (orderby.h:87-93)
            q.resultRows = orderBy->limit;
            q.resultTuples.resize((size_t)q.resultRows * ts);
        }
    }
}

}  // namespace rsq
