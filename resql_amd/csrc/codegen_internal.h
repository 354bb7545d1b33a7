// codegen_internal.h - the walker that turns an operator tree into device pipelines: its state and the declarations of its parts.
// The parts, by what they generate: codegen.cpp (the walk: scans, selections, wave compaction, materialisation), codegen_join.cpp (hash-table
// builds and probes), codegen_agg.cpp (the aggregation sinks), codegen_loop.cpp (the tile loop and the kernel around the row function).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <set>
#include <sstream>

#include "engine_internal.h"


namespace rsq {
namespace cg {


struct Sym { std::string var; Type type; };

inline int envInt(const char* name, int def, int lo, int hi) {
    const char* e = getenv(name);
    int v = e ? atoi(e) : def;
    return v < lo ? lo : v > hi ? hi : v;
}


// ================================================================================================
// expressions -> device code (emitExpression, reference src/ExpressionsJitFlounder.h:1080-1114)
// ================================================================================================
struct ExprGen {
    std::map<std::string, Sym> symbols;     // JitContextFlounder::symbolTable of the current pipeline
    std::map<std::string, int> strWordVars; // string columns whose first words arrive as row-function parameters <var>_w0, _w1 (loaded with the tile): how many
    bool int16Cast = false;                 // rsq_config.compat_flags & RSQ_COMPAT_JIT_INT16_CAST

    static std::string ctype(const Type& t) {
        switch (t.tag) {
            case RSQ_BIGINT: case RSQ_DECIMAL: return "i64";
            case RSQ_INT: case RSQ_DATE: return "i32";
            case RSQ_BOOL: return "u8";
            case RSQ_CHAR: return t.len == 1 ? "u8" : "rsq::Str";
            case RSQ_VARCHAR: return "rsq::Str";
            default: failType("no device type for " + serializeType(t));
        }
    }
    static std::string lit64(int64_t v) {
        if (v == INT64_MIN) return "((i64)0x8000000000000000ull)";
        return "((i64)" + std::to_string((long long)v) + "ll)";
    }
    static std::string cstring(const std::string& s) {
        std::string o = "\"";
        for (unsigned char c : s) {
            char buf[8];
            if (c == '"' || c == '\\') { o += '\\'; o += (char)c; }
            else if (c < 32 || c > 126) { snprintf(buf, sizeof buf, "\\%03o", c); o += buf; }
            else o += (char)c;
        }
        return o + "\"";
    }
    static int64_t pow10(int n) { int64_t v = 1; while (n-- > 0) v *= 10; return v; }

    std::string constant(const Expr* e) {
        switch (e->type.tag) {
            case RSQ_DECIMAL: case RSQ_BIGINT: return lit64(e->ival);
            case RSQ_INT: case RSQ_DATE: return "((i32)" + std::to_string((long long)(int32_t)e->ival) + ")";
            case RSQ_BOOL: return "((u8)" + std::to_string((int)(uint8_t)e->ival) + ")";
            case RSQ_CHAR:
                if (e->type.len == 1) return "((u8)" + std::to_string((int)(uint8_t)e->ival) + ")";
                [[fallthrough]];
            case RSQ_VARCHAR: return "rsq::str(" + cstring(e->symbol) + ", " + std::to_string(e->type.len) + ")";
            default: failType("Constant code generation not implemented for datatype");
        }
    }

    std::string emit(Expr* e) {
        if (e->type.tag == RSQ_NT) failType("Expression type undefined in emitExpression(..). Have you derived the expression types?");
        auto it = symbols.find(expressionName(e));
        if (it != symbols.end()) return it->second.var;          // value already available under this name
        switch (e->structure) {
            case LITERAL:
                if (e->tag == RSQ_E_ATTRIBUTE) failType("attribute " + e->symbol + " is not available in this pipeline");
                if (e->tag == RSQ_E_CONSTANT) return constant(e);
                if (e->tag == RSQ_E_STAR) return "((i64)0)";
                failType(std::string("emitExpressionLiteral(..) not implemented for expression type") + exprTagNames[e->tag]);
            case UNARY: return emitUnary(e);
            case BINARY: return emitBinary(e);
            case OTHER: return emitCase(e);
            default: failType("emitExpression(..)");
        }
    }

    std::string emitUnary(Expr* e) {
        if (e->tag == RSQ_E_COUNT) {
            // emitCount: every row counts.  The reference still emits the argument first (emitExpressionUnary,
            // ExpressionsJitFlounder.h:925-928), so an argument it cannot compile refuses the whole plan: emit it for
            // its checks and drop the text.
            if (e->child && e->child->tag != RSQ_E_STAR) (void)emit(e->child);
            return "((i64)1)";
        }
        std::string c = emit(e->child);
        const Type from = e->child->type, to = e->type;
        switch (e->tag) {
            case RSQ_E_SUM: case RSQ_E_AVG: case RSQ_E_MIN: case RSQ_E_MAX: case RSQ_E_AS: return c;
            case RSQ_E_TYPECAST:
                if (to.tag == RSQ_DECIMAL) {
                    if (from.tag == RSQ_DECIMAL) {
                        if (to.scale == from.scale) return c;
                        int d = to.scale - from.scale;
                        if (d > 8 || d < -8) failType("typecast beyond the supported scale difference");
                        if (d > 0) return "rsq::mul(" + c + ", " + lit64(pow10(d)) + ")";
                        return "((i64)((" + c + ") / " + lit64(pow10(-d)) + "))";
                    }
                    if (from.tag == RSQ_BIGINT) {
                        if (to.scale > 8) failType("typecast beyond the supported scale");
                        return "rsq::mul(" + c + ", " + lit64(pow10(to.scale)) + ")";
                    }
                    failType("emitTypecastToDECIMAL(..) code generation not implemented for datatype");
                }
                if (to.tag == RSQ_BIGINT) {
                    // INT -> BIGINT is a 32 -> 64 sign extension (ExpressionsJitFlounder.h:818-824 `movsx`).  The reference's
                    // asmjit back end encodes the 16-bit movsx for it, so its JIT extends the low 16
                    // bits; rsq_config.compat_flags & RSQ_COMPAT_JIT_INT16_CAST reproduces exactly that for hosts that need the JIT's answers.
                    if (from.tag == RSQ_INT) return int16Cast ? "((i64)(short)(" + c + "))" : "((i64)(" + c + "))";
                    if (from.tag == RSQ_DECIMAL) {
                        if (from.scale > 8) failType("typecast beyond the supported scale");
                        return "((i64)((" + c + ") / " + lit64(pow10(from.scale)) + "))";
                    }
                    if (from.tag == RSQ_BIGINT) return c;
                    failType("emitTypecastToBIGINT(..) code generation not implemented for datatype");
                }
                failType("emitTypecast(..) code generation not implemented for datatype");
            default:
                failType(std::string("emitExpression(..) not implemented for expression type") + exprTagNames[e->tag]);
        }
    }

    std::string emitBinary(Expr* e) {
        std::string l = emit(e->child), r = emit(e->child->next);
        const Type res = e->type, op = e->child->type;
        auto arithOk = [&]() {
            if (res.tag != RSQ_DECIMAL && res.tag != RSQ_BIGINT)
                failType(std::string(exprTagNames[e->tag]) + " code generation not implemented for datatype");
        };
        auto orderedOk = [&]() {
            if (op.tag != RSQ_DECIMAL && op.tag != RSQ_DATE && op.tag != RSQ_BIGINT)
                failType(std::string(exprTagNames[e->tag]) + " code generation not implemented for datatype");
        };
        // string = constant: word-wise against the constant's bytes (rsq_device.h ld_bytes) instead of the byte loop
        auto equalsConstant = [&](bool charSemantics) -> std::string {
            Expr* lc = e->child; Expr* rc = e->child->next;
            const bool lConst = lc->tag == RSQ_E_CONSTANT, rConst = rc->tag == RSQ_E_CONSTANT;
            if (lConst == rConst || !lc->type.isString() || !rc->type.isString() || 1 == 0) return "";
            const Expr* cst = lConst ? lc : rc;
            const std::string& x = lConst ? r : l;
            const int cap = (lConst ? rc : lc)->type.len;
            std::string text = cst->symbol;
            if (text.find('\0') != std::string::npos) return "";
            if (charSemantics) while (!text.empty() && text.back() == ' ') text.pop_back();
            if ((int)text.size() > cap) return "((u8)0)";            // longer than any value of the column
            std::string cond, condRest;          // (condRest: the words behind the prefetched ones - fetched only if those match)
            const int nPre = strWordVars.count(x) ? strWordVars[x] : 0;
            for (int w = 0; w * 8 < cap; w++) {
                const int rbytes = std::min(8, cap - w * 8);
                uint64_t cw = 0, mask = 0;
                for (int i = 0; i < rbytes; i++) {
                    const size_t k = (size_t)(w * 8 + i);
                    if (k < text.size()) { cw |= (uint64_t)(uint8_t)text[k] << (8 * i); mask |= 0xFFull << (8 * i); }
                    else mask |= (charSemantics ? 0xDFull : 0xFFull) << (8 * i);
                }
                char buf[200];
                if (w < nPre) snprintf(buf, sizeof buf, "((%s_w%d ^ 0x%llxull) & 0x%llxull)", x.c_str(), w, (unsigned long long)cw, (unsigned long long)mask);
                else
                snprintf(buf, sizeof buf, "((rsq::ld_bytes<%d>((%s).p + %d) ^ 0x%llxull) & 0x%llxull)", rbytes, x.c_str(), w * 8,
                         (unsigned long long)cw, (unsigned long long)mask);
                std::string& into = nPre > 0 && w >= nPre ? condRest : cond;
                into += (into.empty() ? "" : " | ") + std::string(buf);
            }
            if (!condRest.empty()) return "((u8)(((" + cond + ") == 0ull) && ((" + condRest + ") == 0ull)))";
            return "((u8)((" + cond + ") == 0ull))";
        };
        auto equals = [&]() -> std::string {
            if (op.tag == RSQ_VARCHAR || (op.tag == RSQ_CHAR && op.len > 1)) {
                const std::string fast = equalsConstant(op.tag == RSQ_CHAR);
                if (!fast.empty()) return fast;
            }
            switch (op.tag) {
                case RSQ_DECIMAL: case RSQ_INT: case RSQ_BIGINT: case RSQ_BOOL: case RSQ_DATE:
                    return "((u8)((" + l + ") == (" + r + ")))";
                case RSQ_CHAR:
                    if (op.len > 1) return "rsq::compare_char(" + l + ", " + r + ")";
                    return "((u8)((" + l + ") == (" + r + ")))";
                case RSQ_VARCHAR: return "rsq::compare_varchar(" + l + ", " + r + ")";
                default: failType("EQUALS code generation not implemented for datatype");
            }
        };
        switch (e->tag) {
            case RSQ_E_ADD: arithOk(); return "rsq::add(" + l + ", " + r + ")";
            case RSQ_E_SUB: arithOk(); return "rsq::sub(" + l + ", " + r + ")";
            case RSQ_E_MUL: arithOk(); return "rsq::mul(" + l + ", " + r + ")";
            case RSQ_E_DIV: arithOk(); return "rsq::div(" + l + ", " + r + ", a.err)";
            case RSQ_E_AND: return "((u8)((" + l + ") & (" + r + ")))";      // no short circuit, as in the reference
            case RSQ_E_OR: return "((u8)((" + l + ") | (" + r + ")))";
            case RSQ_E_LT: orderedOk(); return "((u8)((" + l + ") < (" + r + ")))";
            case RSQ_E_LE: orderedOk(); return "((u8)((" + l + ") <= (" + r + ")))";
            case RSQ_E_GT: orderedOk(); return "((u8)((" + l + ") > (" + r + ")))";
            case RSQ_E_GE: orderedOk(); return "((u8)((" + l + ") >= (" + r + ")))";
            case RSQ_E_EQ: return equals();
            case RSQ_E_NEQ: return "((u8)(1 - " + equals() + "))";
            case RSQ_E_LIKE: {
                // emitLike passes both operands to stringLikeCheck as char* (ExpressionsJitFlounder.h:695-705): a CHAR(1)
                // operand is a byte there, not a pointer — undefined in the reference, refused here
                const Type rt = e->child->next->type;
                if (!op.isString() || !rt.isString()) failType("LIKE on a CHAR(1) operand is undefined in the reference");
                return "rsq::like(" + l + ", " + r + ")";
            }
            default: failType(std::string("emitExpressionBinary(..) not implemented for expression type") + exprTagNames[e->tag]);
        }
    }

    std::string emitCase(Expr* e) {   // ExpressionsJitFlounder.h:720-754
        std::string out, close;
        Expr* c = e->child;
        for (; c && c->tag == RSQ_E_WHENTHEN; c = c->next) {
            out += "((" + emit(c->child) + ") ? (" + emit(c->child->next) + ") : ";
            close += ")";
        }
        if (c) out += "(" + emit(c) + ")";
        else out += (e->type.isString() ? std::string("rsq::str(\"\", 0)") : "((" + ctype(e->type) + ")0)");
        return out + close;
    }
};

// value of a 64-bit table word as a typed device value, and back
inline std::string fromWord(const std::string& w, const Type& t) {
    switch (t.tag) {
        case RSQ_BIGINT: case RSQ_DECIMAL: return w;
        case RSQ_INT: case RSQ_DATE: return "((i32)(" + w + "))";
        case RSQ_BOOL: return "((u8)(" + w + "))";
        case RSQ_CHAR: if (t.len == 1) return "((u8)(" + w + "))"; [[fallthrough]];
        case RSQ_VARCHAR: return "rsq::str_from_addr(" + w + ", " + std::to_string(t.len) + ")";     // payload strings travel by address
        default: failUnsupported("value type cannot be carried in a hash table word");
    }
}
inline std::string toWord(const std::string& v, const Type& t) {
    // a string is carried as the device address of its bytes in the (immutable, device-resident) column it comes from
    if (t.isString()) return "rsq::str_addr(" + v + ")";
    return "((i64)(" + v + "))";
}

// ================================================================================================
// the walk
// ================================================================================================
struct Walker {
    Query& q;
    ExprGen eg;

    // state of the pipeline under construction
    Pipeline pipe;
    std::vector<std::string> colTypes;        // device type per scanned (vector-loadable) column
    std::vector<int> colIsString;
    std::string rowParams, rowArgsTail, rowArgsTailGuarded;
    // key-bitmap words fetched for both rows of a lane (and all tiles in flight) before the first row is processed:
    // (table name, scanned column index) — see consumeProbe
    struct BitmapPrefetch { std::string first; int second; bool interleaved; };
    std::vector<BitmapPrefetch> bitmapPrefetch;
    std::string body;                          // row function body
    std::string closers;                       // closing braces of the open scopes
    std::string stateDecl, stateInit, prologue, epilogue, fileScope;
    std::string helperFns;                     // device functions behind Args / State, in front of the row function
    std::vector<std::string> explainSteps;
    int indent = 1;
    int matchSlotTable = -1;                   // innermost single-match probe whose slot variable is in scope
    std::map<int, std::string> slotVar;        // hash table id -> device variable holding the matched slot
    std::map<std::string, int> symbolOrigin;   // symbol -> hash table id it was read from (or -1: scan column)
    std::map<std::string, int> symbolWord;     // symbol -> word index in that table
    bool multiMatchAbove = false;
    // wave-level compaction (see compactThen)
    bool selective = false, compacted = false;
    // the selection directly above the scan: its text over the row's column variables, the columns it reads and the fraction of
    // rows it is expected to pass (column statistics, values taken as uniform) - the late-load form of the tile loop (below)
    std::string leadCond; std::vector<int> leadCols; double leadPass = 1.0;
    std::string stage2Prefix;               // necessary conditions of later joins, tested at the top of stage 2 (consumeProbe: component bitmaps)
    struct CompFilter { std::string table, symbol, stage2Var; int64_t bits; };
    std::vector<CompFilter> compFilters;    // ... as consumeProbe found them while stage 2 was generated
    bool inStage2 = false;                  // the walk is generating the code behind the wave compaction
    bool leadPassComplete = true;       // every part of the predicate was understood (else the estimate is an upper bound only)
    // the selection directly above the scan, as text: where its scope starts in `body` and its condition (finishPipeline: the count pass of a
    // materialisation evaluates it for both rows of a lane first, "pairCond")
    size_t pairSplit = std::string::npos; std::string pairCond;
    std::string stage2Body;
    std::vector<std::pair<std::string, Sym>> cqLive;     // carried symbols: name -> stage-1 variable and type

    explicit Walker(Query& q_) : q(q_) { eg.int16Cast = jitInt16Cast(q_.ctx); }

    void line(const std::string& s) { body += std::string((size_t)indent * 4, ' ') + s + "\n"; }
    void openScope(const std::string& head) { line(head); indent++; }
    void closeScope() { indent--; line("}"); }
    void addArg(const std::string& name, const std::string& ctype, uint64_t v);

    void produce(OpNode* o, std::vector<std::string> request);

    std::map<OpNode*, std::vector<std::string>> requestOf;
    std::map<OpNode*, int> joinPhase;

    static bool has(const std::vector<std::string>& v, const std::string& s) { return std::find(v.begin(), v.end(), s) != v.end(); }

    Schema prune(const Schema& s, const std::vector<std::string>& req);

    void produceScan(OpNode* o, std::vector<std::string> request);

    bool sideRange(const Expr* e, double& lo, double& hi, int& col);
    double passFraction(const Expr* e);
    void leadColumnsOf(const Expr* e, std::vector<int>& out, bool& ok);
    // Short string columns the selection right above the scan compares with constants: their bytes are loaded WITH the tile (one or
    // two 8-byte words per row, in flight together with the numeric columns) and reach the row function as parameters, instead of
    // being fetched inside it row by row - eight dependent round trips per lane and iteration (TPC-H Q3's customer pipeline:
    // c_mktsegment = 'BUILDING').
    std::vector<std::pair<int, int>> strPrefetch;         // (scanned column, bytes of it that arrive with the tile), in the order of the row function's parameters
    std::map<int, int> strPrefetchWidth;                   // scanned column -> its width (the row stride)
    // Staged string tiles.  A lane that fetches ITS two rows of a CHAR(25) column asks for 8 bytes at a stride of 50: the wave's one load
    // instruction touches 25 memory lines, the next word's the same 25 again, and the texture unit, not the memory, bounds the kernel (TPC-H
    // Q19 at SF10: 63 B rows at 3.5 TB/s, against 6.8 for Q1's plain columns).  A column of at most 32 bytes is therefore fetched as what it
    // is - 128 rows x W contiguous bytes per tile, 16 bytes per lane and load, every line once - and passed through the wave's own LDS
    // region, from which each lane reads its rows' words (ds_read_b64 takes any address on gfx950).  All words of the value then arrive
    // as row-function parameters.
    std::string postTile;                                  // code behind the two row_fn calls of a tile in the tile loops ($TILE = the tile's number; wave-uniform)
    std::map<int, int> strStaged;                          // scanned column -> byte offset of its tile in the wave's LDS region
    int strStagedBytes = 0;                                // bytes of that region (128 x the staged widths)
    void prefetchComparedStrings(const Expr* e);
    void noteLeadingSelection(const Expr* e, const std::string& cond);

    void consume(OpNode* o, OpNode* from);

    bool downstreamMaterializes(OpNode* o);
    bool compactThen(OpNode* o, const std::function<void()>& downstream);

    void countPerThread(const std::string& T, bool identityCapable = false);

    std::vector<std::string> keyWords(Expr* e, const std::string& prefix, bool stripChar, std::vector<std::string>* endsWithSpace = nullptr, int stripMode = -1);

    // String join keys of different declared lengths.  The reference hashes each side with its own type: hashVarchar stops
    // at the NUL, so VARCHAR(a) = VARCHAR(b) matches equal strings — both sides take the word count of the wider one, the
    // narrower side's missing words are zero.  hashChar pads with spaces to the DECLARED length (qlib/hash.h:131-147), so
    // CHAR(a) = CHAR(b), a != b, never has equal hashes and never matches: the two sides get pad words that differ.
    // CHAR against VARCHAR (any lengths): equal hashes need the VARCHAR value to be exactly as long as the CHAR column is wide
    // (hashChar counts the pad spaces, hashVarchar only the characters), and the key comparison is the PROBE side's
    // (checkEquality(probeKeys, entryKeys), hashjoin.h:142/191: compareChar ignores trailing spaces, compareVarchar does not).
    // So both sides form their words the probe side's way (joinKeyStripMode), padded to the wider side's word count, plus one
    // word that holds the hashed length: the declared width of a CHAR key, the actual length of a VARCHAR key.
    static bool mixedStringKinds(const Expr* a, const Expr* b) { return a->type.isString() && b->type.isString() && a->type.tag != b->type.tag; }
    static int joinKeyStripMode(const Expr* side, const Expr* probeSide, const Expr* buildSide);
    void padKeyWords(Expr* mine, Expr* other, size_t w0, std::vector<std::string>& keyVars, bool buildSide);

    std::string hashOf(const std::vector<std::string>& keyVars);

    static std::string wordAt(const HashTable& ht, const std::string& T, int w);

    std::string slotOf(const HashTable& ht, const std::string& T, const std::vector<std::string>& keyVars);

    void consumeBuild(OpNode* o, OpNode* from);
    void consumeBuildBody(OpNode* o, OpNode* from);

    void probeKeys(OpNode* o, const std::string& T, std::vector<std::string>& keyVars, std::vector<std::string>& probeKeyNames);

    void consumeProbe(OpNode* o, OpNode* from);

    void consumeMatch(OpNode* o, HashTable& ht, const std::string& T, const std::vector<std::string>& keyVars,
                      const std::vector<std::string>& probeKeyNames);

    void probeTable(OpNode* o, HashTable& ht, const std::string& T, const std::vector<std::string>& keyVars,
                    const std::vector<std::string>& probeKeyNames);
    std::map<std::string, std::pair<int, int>> probeKeyOf;   // probe-side key symbol -> (table, key word)
    // the join probes whose match is in scope (innermost last): table, single match?, the probe side's key symbols ("" where a key
    // is not a one-word attribute) — emitHashAggregation's functional dependencies
    struct ProbeInScope { int table; bool single; bool rankCapable; std::vector<std::string> keySymbols; };
    std::vector<ProbeInScope> probesInScope;

    void collectAccumulators(OpNode* o);

    bool tryDenseKeys(OpNode* o);

    bool tryJoinEntry(OpNode* o);

    void consumeAggregation(OpNode* o, OpNode* from);

    void emitHashAggregation(OpNode* o);

    std::string groupIdExpr();

    static const char* identityOf(int merge) { return merge == 0 ? "0ull" : merge == 2 ? "0x7fffffffffffffffull" : "0x8000000000000000ull"; }

    std::string blockIdentityExpr(const std::string& blk);

    void emitGlobalFlush(std::ostringstream& s, const std::string& count, const std::string& srcExpr, int64_t D, int stride = 1);

    void emitStagedScatter(int64_t D, int W, int gpp, int shift, int P);

    void emitDenseAggregation(OpNode* o);

    void emitJoinEntryAggregation(OpNode* o);

    void consumeMaterialize(OpNode* o, OpNode* from);

    std::string postTileFor(const std::string& tile);
    int stagedRounds(int col) { return (8 * strPrefetchWidth[col] + 63) / 64; }
    void stagedChunkDecls(std::ostringstream& s, const std::string& ind, const char* pre, int col, int u) {
        for (int r = 0; r < stagedRounds(col); r++) s << ind << "rsq::u32v4 " << pre << col << "_" << u << "_" << r << " = {0u, 0u, 0u, 0u};\n";
    }
    void stagedChunkLoads(std::ostringstream& s, const std::string& ind, const char* pre, int col, int u);
    void stagedUnstage(std::ostringstream& s, const std::string& ind, int u);

    void emitLateLoads(std::ostringstream& s, const std::string& tile, int u, const std::vector<char>& lateCol, const std::string& tileEnd = "ntiles");

    void finishPipeline();
};

}  // namespace cg
}  // namespace rsq
